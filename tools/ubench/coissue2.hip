// Micro-benchmark (tuning tool, not product): how do flash-attention-like waves share ONE SIMD on gfx950?  One 1024-thread
// workgroup on one CU; waves w, w + 4, w + 8, w + 12 share a SIMD.  Roles:
//   1  MFMA chain, accumulators in VGPRs          5  MFMA chain, accumulators in AGPRs
//   2  v_fma_f32 chains                           3  v_exp_f32 chains
//   4  the softmax mix per two scores: 2 v_exp_f32, 1 v_max3_f32, 2 v_add_f32, 1 v_cvt_pk_bf16_f32
//   6  flash-like: 8 MFMAs (VGPR accumulators), then 48 instructions of the mix, repeated (one step = half a key tile)
//   7  the same with AGPR accumulators            8  flash-like, MFMAs and mix interleaved (1 MFMA, 6 mix)
// Each configuration is timed with s_memtime on the waves themselves.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct Roles { int r[16]; };

#define MFMA_V(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA_A(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
// six instructions on independent registers (x0..x7 carry the chains)
#define MIX6(i)                                                                                                  \
  asm volatile("v_exp_f32 %0, %0\n\tv_max3_f32 %2, %2, %0, %1\n\tv_exp_f32 %1, %1\n\tv_add_f32 %3, %3, %0\n\t"   \
               "v_add_f32 %4, %4, %1\n\tv_cvt_pk_bf16_f32 %5, %0, %1"                                             \
               : "+v"(x[(2 * i) & 7]), "+v"(x[(2 * i + 1) & 7]), "+v"(mx), "+v"(l0), "+v"(l1), "=v"(pk))

__global__ __launch_bounds__(1024) void k(unsigned long long* out, Roles roles, int steps, float seed) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int role = roles.r[wave];
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = -seed - threadIdx.x * 1e-3f - i;
  float mx = -1e30f, l0 = 0.f, l1 = 0.f;
  unsigned pk = 0;
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (role == 1) {
    for (int it = 0; it < steps; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { MFMA_V(acc0); MFMA_V(acc1); }
    }
  } else if (role == 5) {
    for (int it = 0; it < steps; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { MFMA_A(acc0); MFMA_A(acc1); }
    }
  } else if (role == 2) {
    for (int it = 0; it < steps; ++it) {
#pragma unroll
      for (int u = 0; u < 48; ++u) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[u & 7]));
    }
  } else if (role == 3) {
    for (int it = 0; it < steps; ++it) {
#pragma unroll
      for (int u = 0; u < 48; ++u) asm volatile("v_exp_f32 %0, %0" : "+v"(x[u & 7]));
    }
  } else if (role >= 9 && role <= 14) {
    float2 pa = make_float2(x[0], x[1]), pb = make_float2(x[2], x[3]), pc = make_float2(x[4], x[5]), pd = make_float2(x[6], x[7]);
#define CHAIN(R, BODY)                              \
  if (role == R)                                    \
    for (int it = 0; it < steps; ++it) {            \
      _Pragma("unroll") for (int u = 0; u < 48; ++u) { BODY; } \
    }
    CHAIN(9, asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x[u & 7]) : "v"(pk), "v"(0x3f803f80u)))
    CHAIN(10, {
      if ((u & 3) == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pa) : "v"(pb));
      if ((u & 3) == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pb) : "v"(pa));
      if ((u & 3) == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pc) : "v"(pd));
      if ((u & 3) == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pd) : "v"(pc));
    })
    CHAIN(11, asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(x[u & 7]), "v"(x[(u + 1) & 7])))
    CHAIN(12, asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[u & 7]) : "v"(x[(u + 1) & 7]), "v"(x[(u + 2) & 7])))
    CHAIN(13, asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[u & 7]) : "v"(x[(u + 1) & 7])))
    CHAIN(14, asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[u & 7]), "+v"(x[(u + 4) & 7])))
    x[0] += pa.x + pb.y + pc.x + pd.y;
  } else if (role == 4) {
    for (int it = 0; it < steps; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) MIX6(u);
    }
  } else if (role == 6) {
    for (int it = 0; it < steps; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { MFMA_V(acc0); MFMA_V(acc1); }
#pragma unroll
      for (int u = 0; u < 8; ++u) MIX6(u);
    }
  } else if (role == 7) {
    for (int it = 0; it < steps; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { MFMA_A(acc0); MFMA_A(acc1); }
#pragma unroll
      for (int u = 0; u < 8; ++u) MIX6(u);
    }
  } else if (role == 8) {
    for (int it = 0; it < steps; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { MFMA_V(acc0); MIX6(2 * u); MFMA_V(acc1); MIX6(2 * u + 1); }
    }
  }
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = mx + l0 + l1 + __uint_as_float(pk);
  for (int i = 0; i < 8; ++i) s += x[i];
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  if (threadIdx.x % 64 == 0) { out[wave * 2] = t1 - t0; out[wave * 2 + 1] = (unsigned long long)(s != 12345.f); }
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 32 * sizeof(unsigned long long));
  unsigned long long h[32];
  const int STEPS = 512;   // one step = 8 MFMAs (256 MFMA-pipe cycles) and / or 48 VALU
  struct Case { const char* name; Roles r; } cases[] = {
      {"MFMA (VGPR acc) alone", {{1}}},
      {"MFMA (AGPR acc) alone", {{5}}},
      {"v_fma alone", {{2}}},
      {"v_exp alone", {{3}}},
      {"softmax mix alone", {{4}}},
      {"v_dot2c_f32_bf16 alone", {{9}}},
      {"v_dot2c_f32_bf16 x2 same SIMD", {{9, 0, 0, 0, 9}}},
      {"v_pk_add_f32 alone", {{10}}},
      {"v_pk_add_f32 x2 same SIMD", {{10, 0, 0, 0, 10}}},
      {"v_cvt_pk_bf16_f32 alone", {{11}}},
      {"v_cvt_pk_bf16_f32 x2 same SIMD", {{11, 0, 0, 0, 11}}},
      {"v_max3_f32 alone", {{12}}},
      {"v_max3_f32 x2 same SIMD", {{12, 0, 0, 0, 12}}},
      {"v_add_f32 alone", {{13}}},
      {"v_add_f32 x2 same SIMD", {{13, 0, 0, 0, 13}}},
      {"v_add_f32 x3 same SIMD", {{13, 0, 0, 0, 13, 0, 0, 0, 13}}},
      {"v_exp_f32 x2 same SIMD", {{3, 0, 0, 0, 3}}},
      {"v_permlane32_swap alone", {{14}}},
      {"MFMA(V) + v_dot2c same SIMD", {{1, 0, 0, 0, 9}}},
      {"MFMA(V) + v_pk_add same SIMD", {{1, 0, 0, 0, 10}}},
      {"mix x2 same SIMD", {{4, 0, 0, 0, 4}}},
      {"mix x3 same SIMD", {{4, 0, 0, 0, 4, 0, 0, 0, 4}}},
      {"MFMA(V) + v_fma same SIMD", {{1, 0, 0, 0, 2}}},
      {"MFMA(V) + v_exp same SIMD", {{1, 0, 0, 0, 3}}},
      {"MFMA(V) + mix same SIMD", {{1, 0, 0, 0, 4}}},
      {"MFMA(A) + mix same SIMD", {{5, 0, 0, 0, 4}}},
      {"MFMA(V) + mix x2 same SIMD", {{1, 0, 0, 0, 4, 0, 0, 0, 4}}},
      {"flash-like (V) alone", {{6}}},
      {"flash-like (A) alone", {{7}}},
      {"flash-like interleaved alone", {{8}}},
      {"flash-like (V) x2 same SIMD", {{6, 0, 0, 0, 6}}},
      {"flash-like (V) x3 same SIMD", {{6, 0, 0, 0, 6, 0, 0, 0, 6}}},
      {"flash-like (V) x4 same SIMD", {{6, 0, 0, 0, 6, 0, 0, 0, 6, 0, 0, 0, 6}}},
      {"flash-like (A) x3 same SIMD", {{7, 0, 0, 0, 7, 0, 0, 0, 7}}},
      {"flash-like interleaved x2 same SIMD", {{8, 0, 0, 0, 8}}},
      {"flash-like interleaved x3 same SIMD", {{8, 0, 0, 0, 8, 0, 0, 0, 8}}},
      {"flash-like (V) x3 on all four SIMDs", {{6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6}}},
  };
  printf("steps %d: per step 8 MFMAs (256 pipe cycles) and / or 48 VALU; MFMA-only floor %d cycles\n", STEPS, STEPS * 256);
  for (auto& c : cases) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, d, c.r, STEPS, 1.0f);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-40s:", c.name);
    for (int w = 0; w < 16; ++w)
      if (c.r.r[w]) printf(" w%d(r%d) %llu", w, c.r.r[w], h[w * 2]);
    printf("\n");
  }
  return 0;
}
