// Shader clock under load (tuning tool): s_memtime (shader cycles) against s_memrealtime (100 MHz) around ~2 ms of
// (a) dependent MFMAs on 1 / 2 waves per SIMD, random / zero operands, (b) MFMA + VALU mix, (c) VALU only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>   // 0 mfma only, 1 mfma + 6 valu, 2 valu only, 3 mfma + 4 exp
__global__ __launch_bounds__(512) void k(unsigned long long* out, const unsigned* data, int iters) {
  bf16x8 a, b;
  {
    const uint4 u = ((const uint4*)data)[threadIdx.x], w = ((const uint4*)data)[threadIdx.x + 512];
    a = __builtin_bit_cast(bf16x8, u);
    b = __builtin_bit_cast(bf16x8, w);
  }
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = 1.0f + i + threadIdx.x * 1e-3f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE != 2) {
        if (u & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
      }
      if (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int j = 0; j < 6; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[j]) : "v"(x[7]));
      }
      if (MODE == 3) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j]));
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sink = 0.f;
  for (int i = 0; i < 8; ++i) sink += x[i];
  for (int r = 0; r < 16; ++r) sink += acc0[r] + acc1[r];
  if (sink == 12345.678f) out[1 << 16] = 1;
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, int threads, const unsigned* d_data, unsigned long long* d, int iters) {
  std::vector<unsigned long long> h(512);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, d, d_data, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
  }
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(h.data(), d, 512 * 8, hipMemcpyDeviceToHost);
  double sc = 0, rt = 0;
  for (int i = 0; i < 256; ++i) { sc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
  const double mhz = sc / rt * 100.0;
  const double mf = (MODE == 2) ? 0.0 : 256.0 * (threads / 64) * iters * 8.0 * 2.0 * 32 * 32 * 16;
  printf("%-40s %d waves/SIMD: %.0f MHz shader clock (s_memtime / s_memrealtime x 100 MHz), kernel %.2f ms, %.0f TFLOP/s\n", name,
         threads / 256, mhz, ms, mf / (ms * 1e-3) / 1e12);
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
  const int iters = 20000;   // x 8 MFMAs x 32 cycles = 5.1 M cycles ~ 2-3 ms per wave
  run<0>("MFMA only, random operands", 256, d_rand, d, iters);
  run<0>("MFMA only, random operands", 512, d_rand, d, iters / 2);
  run<0>("MFMA only, zero operands", 256, d_zero, d, iters);
  run<1>("MFMA + 6 v_fma, random", 256, d_rand, d, iters);
  run<1>("MFMA + 6 v_fma, random", 512, d_rand, d, iters / 2);
  run<3>("MFMA + 4 v_exp, random", 512, d_rand, d, iters / 2);
  run<2>("6 v_fma only", 512, d_rand, d, iters / 2);
  return 0;
}
