// Micro-benchmark: what does a kernel boundary cost on gfx950 as a function of HOW the producer stored its output?
// A chain of dependent streaming kernels (y = f(x): read n bytes, write n bytes) with the output stored (0) plainly - dirty lines
// pile up in the XCD's write-back L2 and are written back by the end-of-kernel release -, (1) write-through at agent scope (sc1),
// (2) non-temporal (nt), (3) sc0 sc1 (system scope).  Prints us per kernel for several tensor sizes.
//   hipcc --offload-arch=gfx950 -O3 -o store_policy store_policy.hip && ./store_policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int POLICY>
__device__ __forceinline__ void store16(uint4* p, uint4 v4) {
  const u32x4 v = {v4.x, v4.y, v4.z, v4.w};
  if constexpr (POLICY == 0) *p = v4;
  else if constexpr (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

template <int POLICY>
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, long long n, unsigned add) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    uint4 v = x[i];
    v.x += add; v.y ^= add; v.z += 1; v.w += 3;
    store16<POLICY>(y + i, v);
  }
}

template <int POLICY>
float run(uint4* a, uint4* b, long long n16, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = (int)std::min<long long>((n16 + 255) / 256, 2048);
  for (int i = 0; i < 4; ++i) { stream_kernel<POLICY><<<grid, 256>>>(a, b, n16, i); stream_kernel<POLICY><<<grid, 256>>>(b, a, n16, i); }
  hipDeviceSynchronize();
  std::vector<float> ts;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) { stream_kernel<POLICY><<<grid, 256>>>(a, b, n16, i); stream_kernel<POLICY><<<grid, 256>>>(b, a, n16, i); }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ts.push_back(ms * 1e3f / (2 * iters));
  }
  std::sort(ts.begin(), ts.end());
  return ts[2];
}

int main() {
  const long long sizes_mb[] = {1, 4, 15, 30, 59, 118, 236};
  uint4 *a, *b;
  hipMalloc(&a, 256ll << 20); hipMalloc(&b, 256ll << 20);
  hipMemset(a, 1, 256ll << 20); hipMemset(b, 2, 256ll << 20);
  printf("%8s %10s %10s %10s %10s   (us per kernel: read n + write n)\n", "MB", "plain", "sc1", "nt", "sc0sc1");
  for (long long mb : sizes_mb) {
    const long long n16 = (mb << 20) / 16;
    const float t0 = run<0>(a, b, n16, 20), t1 = run<1>(a, b, n16, 20), t2 = run<2>(a, b, n16, 20), t3 = run<3>(a, b, n16, 20);
    printf("%8lld %10.2f %10.2f %10.2f %10.2f\n", mb, t0, t1, t2, t3);
  }
  return 0;
}
