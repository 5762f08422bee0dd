// Micro-benchmark (tuning tool, not product): do the MFMAs of one wave and the VALU instructions of ANOTHER wave on the
// same SIMD overlap on gfx950?  One 512-thread workgroup on one CU; waves w and w + 4 share a SIMD.  Wave 0 runs a chain of
// v_mfma_f32_32x32x16_bf16 (two accumulators), wave 4 a chain of independent v_fma_f32 (8 chains); each is timed alone,
// then both together (released by one s_barrier), then against the same partner on a DIFFERENT SIMD (wave 1).
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// role per wave: 0 idle, 1 MFMA chain, 2 VALU chain, 3 mixed (1 MFMA + NV VALU per step)
struct Roles { int r[8]; };

__global__ __launch_bounds__(512) void k(unsigned long long* out, Roles roles, int n_mfma, int n_valu, float seed) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int role = roles.r[wave];
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = seed + threadIdx.x * 1e-3f + i;
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (role == 1) {
    for (int it = 0; it < n_mfma / 8; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
      }
    }
  } else if (role == 2) {
    for (int it = 0; it < n_valu / 32; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[u & 7]));
    }
  }
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += x[i];
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  if (threadIdx.x % 64 == 0) { out[wave * 2] = t1 - t0; out[wave * 2 + 1] = (unsigned long long)(s != 12345.f); }
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 16 * sizeof(unsigned long long));
  unsigned long long h[16];
  const int NM = 1024, NV = 8192;
  struct Case { const char* name; Roles r; } cases[] = {
      {"MFMA chain alone (wave 0)", {{1, 0, 0, 0, 0, 0, 0, 0}}},
      {"VALU chain alone (wave 4)", {{0, 0, 0, 0, 2, 0, 0, 0}}},
      {"MFMA wave 0 + VALU wave 4 (same SIMD)", {{1, 0, 0, 0, 2, 0, 0, 0}}},
      {"MFMA wave 0 + VALU wave 1 (other SIMD)", {{1, 2, 0, 0, 0, 0, 0, 0}}},
      {"MFMA wave 0 + MFMA wave 4 (same SIMD)", {{1, 0, 0, 0, 1, 0, 0, 0}}},
      {"VALU wave 0 + VALU wave 4 (same SIMD)", {{2, 0, 0, 0, 2, 0, 0, 0}}},
  };
  printf("n_mfma %d (32 cycles each = %d), n_valu %d\n", NM, NM * 32, NV);
  for (auto& c : cases) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, d, c.r, NM, NV, 1.0f);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-42s:", c.name);
    for (int w = 0; w < 8; ++w)
      if (c.r.r[w]) printf("  wave %d (%s) %8llu cycles", w, c.r.r[w] == 1 ? "MFMA" : "VALU", h[w * 2]);
    printf("\n");
  }
  return 0;
}
