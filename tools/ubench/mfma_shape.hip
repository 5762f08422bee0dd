// Tuning tool: does the matrix pipe draw less power - the power-limited chip clock higher - on 16x16x32 MFMAs than on 32x32x16 ones?
// (hipBLASLt's 256 x 256 x 64 kernel is built from 16x16 MFMAs and holds a 5 % higher clock than the hand-placed 32x32 tile at the same
// duty, docs/history/rounds_1_5_measured.md.)  Dependent MFMA chains on random / zero operands, ~30 ms per launch, s_memtime against
// s_memrealtime.   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_shape.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>   // 0: 32x32x16, two accumulators; 1: 16x16x32, four accumulators; 2: 16x16x32, eight accumulators, two operand pairs
__global__ __launch_bounds__(512) void k(unsigned long long* out, const unsigned* data, int iters) {
  bf16x8 a, b, a2, b2;
  {
    const uint4 u = ((const uint4*)data)[threadIdx.x], w = ((const uint4*)data)[threadIdx.x + 512];
    a = __builtin_bit_cast(bf16x8, u);
    b = __builtin_bit_cast(bf16x8, w);
    const uint4 u2 = ((const uint4*)data)[(threadIdx.x + 77) & 511], w2 = ((const uint4*)data)[((threadIdx.x + 191) & 511) + 512];
    a2 = __builtin_bit_cast(bf16x8, u2);
    b2 = __builtin_bit_cast(bf16x8, w2);
  }
  f32x16 acc0, acc1;
  f32x4 c[8];
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) c[j][r] = 0.f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a2), "v"(b2));
      }
    } else {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int j = MODE == 1 ? (u & 3) : (u & 7);
        if (u & 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a2), "v"(b2));
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sink = 0.f;
  for (int r = 0; r < 16; ++r) sink += acc0[r] + acc1[r];
  for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) sink += c[j][r];
  if (sink == 12345.678f) out[1 << 16] = 1;
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, int threads, const unsigned* d_data, unsigned long long* d, int iters) {
  std::vector<unsigned long long> h(512);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, d, d_data, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
  }
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(h.data(), d, 512 * 8, hipMemcpyDeviceToHost);
  double sc = 0, rt = 0;
  for (int i = 0; i < 256; ++i) { sc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
  const double mhz = sc / rt * 100.0;
  const double mf = 256.0 * (threads / 64) * iters * 8.0 * 2.0 * 32 * 32 * 16;   // both shapes: 8 x 32768 FLOP x 2 per iteration... (16 x 16384)
  printf("%-44s %d waves/SIMD: %.0f MHz, kernel %.2f ms, %.0f TFLOP/s, %.2f cycles per 32768 MAC-pairs\n", name, threads / 256, mhz, ms,
         mf / (ms * 1e-3) / 1e12, sc / 256.0 / iters / 8.0);
}

int main() {
  unsigned long long* d;
  unsigned *d_rand, *d_zero;
  (void)hipMalloc(&d, (1 << 16) * 8 + 64);
  (void)hipMalloc(&d_rand, 1024 * 16);
  (void)hipMalloc(&d_zero, 1024 * 16);
  std::vector<unsigned> hr(4096);
  unsigned s = 12345;
  for (auto& v : hr) {   // random bf16 pairs in (-2, 2)
    s = s * 1664525u + 1013904223u;
    const unsigned lo = 0x3f00u | ((s >> 8) & 0x80ffu), hi = 0x3f00u | ((s >> 20) & 0x80ffu);
    v = lo | (hi << 16);
  }
  (void)hipMemcpy(d_rand, hr.data(), 4096 * 4, hipMemcpyHostToDevice);
  (void)hipMemset(d_zero, 0, 1024 * 16);
  const int iters = 200000;   // x 8 x 32 cycles = 51 M cycles ~ 25-30 ms per wave
  for (int round = 0; round < 2; ++round) {
    run<0>("32x32x16 bf16, random operands", 256, d_rand, d, iters);
    run<1>("16x16x32 bf16 (4 acc), random operands", 256, d_rand, d, iters);
    run<2>("16x16x32 bf16 (8 acc), random operands", 256, d_rand, d, iters);
    run<0>("32x32x16 bf16, random operands", 512, d_rand, d, iters / 2);
    run<2>("16x16x32 bf16 (8 acc), random operands", 512, d_rand, d, iters / 2);
    run<0>("32x32x16 bf16, zero operands", 256, d_zero, d, iters);
    run<2>("16x16x32 bf16 (8 acc), zero operands", 256, d_zero, d, iters);
  }
  return 0;
}
