// Micro-benchmark (tuning tool, not product): what one wave pays per VALU / transcendental / MFMA instruction on gfx950,
// alone and with 2 / 4 waves per SIMD, and how many VALU instructions hide behind one v_mfma_f32_32x32x16_bf16.
// Every test is a loop of 64 x UNROLL independent instructions (8 chains) timed with s_memtime on the wave itself.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

enum { OP_FMA, OP_ADD, OP_EXP, OP_MAX3, OP_CVTPK, OP_PERM, OP_PKMUL, OP_RCP, OP_MOV, OP_MFMA, OP_MFMA_EXP, OP_MFMA_ADD, OP_MFMA_MIX, NOPS };
static const char* NAMES[NOPS] = {"v_fma_f32", "v_add_f32", "v_exp_f32", "v_max3_f32", "v_cvt_pk_bf16_f32", "v_permlane32_swap", "v_pk_mul_f32",
                                  "v_rcp_f32", "v_mov_b32", "mfma32x32x16 (2 acc)", "mfma + N v_exp", "mfma + N v_add", "mfma + N (exp,add,max3,cvt mix)"};

template <int OP, int NV>   // NV: VALU instructions per MFMA for the mixed tests
__global__ __launch_bounds__(256) void k(unsigned long long* out, float seed) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = seed + threadIdx.x * 1e-3f + i;
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
  constexpr int IT = 64;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < IT; ++it) {
    if constexpr (OP < OP_MFMA) {
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        float& v = x[u & 7];
        if constexpr (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v));
        if constexpr (OP == OP_ADD) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v));
        if constexpr (OP == OP_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
        if constexpr (OP == OP_MAX3) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(v));
        if constexpr (OP == OP_CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v));
        if constexpr (OP == OP_PERM) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v), "+v"(x[(u + 4) & 7]));
        if constexpr (OP == OP_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*(double*)&x[(u & 3) * 2]));
        if constexpr (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(v));
        if constexpr (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "v"(x[(u + 1) & 7]));
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          float& v = x[j & 7];
          if constexpr (OP == OP_MFMA_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
          if constexpr (OP == OP_MFMA_ADD) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v));
          if constexpr (OP == OP_MFMA_MIX) {
            if ((j & 3) == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
            if ((j & 3) == 1) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v));
            if ((j & 3) == 2) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(v));
            if ((j & 3) == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v));
          }
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sink = 0.f;
  for (int i = 0; i < 8; ++i) sink += x[i];
  for (int r = 0; r < 16; ++r) sink += acc0[r] + acc1[r];
  if (sink == 12345.678f) out[1 << 20] = 1;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP, int NV>
void run(const char* name, int per_iter, unsigned long long* d, std::vector<unsigned long long>& h) {
  printf("%-34s NV=%2d:", name, NV);
  for (int wps : {1, 2, 4}) {   // waves per SIMD: blocks of 256 threads = 4 waves = one per SIMD
    const int blocks = 256 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, NV>), dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, NV>), dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks * 4; ++i) s += (double)h[i];
    const double ticks = s / (blocks * 4) / (64.0 * per_iter);
    printf("  %dw/SIMD %7.2f ticks/unit (kernel %6.1f us)", wps, ticks, ms * 1e3);
  }
  printf("\n");
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, (1 << 20) * 8 + 64);
  std::vector<unsigned long long> h(256 * 4 * 4 + 16);
  printf("unit = one instruction (plain tests) or one MFMA + its NV VALU (mixed tests); ticks = s_memtime\n");
  run<OP_FMA, 0>(NAMES[OP_FMA], 32, d, h);
  run<OP_ADD, 0>(NAMES[OP_ADD], 32, d, h);
  run<OP_MOV, 0>(NAMES[OP_MOV], 32, d, h);
  run<OP_EXP, 0>(NAMES[OP_EXP], 32, d, h);
  run<OP_RCP, 0>(NAMES[OP_RCP], 32, d, h);
  run<OP_MAX3, 0>(NAMES[OP_MAX3], 32, d, h);
  run<OP_CVTPK, 0>(NAMES[OP_CVTPK], 32, d, h);
  run<OP_PERM, 0>(NAMES[OP_PERM], 32, d, h);
  run<OP_PKMUL, 0>(NAMES[OP_PKMUL], 32, d, h);
  run<OP_MFMA, 0>(NAMES[OP_MFMA], 8, d, h);
  run<OP_MFMA_ADD, 2>(NAMES[OP_MFMA_ADD], 8, d, h);
  run<OP_MFMA_ADD, 4>(NAMES[OP_MFMA_ADD], 8, d, h);
  run<OP_MFMA_ADD, 6>(NAMES[OP_MFMA_ADD], 8, d, h);
  run<OP_MFMA_ADD, 8>(NAMES[OP_MFMA_ADD], 8, d, h);
  run<OP_MFMA_ADD, 12>(NAMES[OP_MFMA_ADD], 8, d, h);
  run<OP_MFMA_EXP, 2>(NAMES[OP_MFMA_EXP], 8, d, h);
  run<OP_MFMA_EXP, 4>(NAMES[OP_MFMA_EXP], 8, d, h);
  run<OP_MFMA_EXP, 8>(NAMES[OP_MFMA_EXP], 8, d, h);
  run<OP_MFMA_MIX, 4>(NAMES[OP_MFMA_MIX], 8, d, h);
  run<OP_MFMA_MIX, 8>(NAMES[OP_MFMA_MIX], 8, d, h);
  run<OP_MFMA_MIX, 12>(NAMES[OP_MFMA_MIX], 8, d, h);
  // s_memtime tick rate: time a long kernel both ways
  return 0;
}
