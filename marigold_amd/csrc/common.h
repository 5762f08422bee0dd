// Shared device helpers for libmarigold_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include "../../include/marigold_hip.h"

// The 16-bit operand type of the build: bf16 (the product library, libmarigold_hip.so) or IEEE fp16 (libmarigold_hip_f16.so, built
// from the same sources with -DMG_OPERAND_F16_BUILD=1: what the reference computes with `--fp16` / torch_dtype=torch.float16,
// script/depth/run.py:203-211).  Same MFMA rate (v_mfma_f32_32x32x16_f16), three more mantissa bits, five fewer exponent bits:
// every conversion, unpack and matrix instruction of the kernels goes through the helpers below, so the kernels are written once.
// The names keep "bf16" (the product type); in the fp16 build they hold fp16 bits.
constexpr bool MG_F16 = MG_OPERAND_F16_BUILD != 0;
typedef uint16_t bf16_t;  // raw operand bits
typedef std::conditional_t<MG_F16, _Float16, __bf16> mg_op16_t;
typedef __attribute__((ext_vector_type(8))) mg_op16_t bf16x8;
typedef __attribute__((ext_vector_type(2))) mg_op16_t mg_bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float mg_f32x2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// mnemonics for the hand-placed instruction streams (csrc/gen_*.py emit them as adjacent string literals)
#if MG_OPERAND_F16_BUILD
#define MG_MFMA32_ASM "v_mfma_f32_32x32x16_f16"
#define MG_MFMA16_ASM "v_mfma_f32_16x16x32_f16"
#define MG_CVT_PK_ASM "v_cvt_pk_f16_f32"
#else
#define MG_MFMA32_ASM "v_mfma_f32_32x32x16_bf16"
#define MG_MFMA16_ASM "v_mfma_f32_16x16x32_bf16"
#define MG_CVT_PK_ASM "v_cvt_pk_bf16_f32"
#endif
constexpr float MG_OP16_MAX = MG_F16 ? 65504.0f : 3.3895314e38f;   // largest finite operand value

typedef __attribute__((ext_vector_type(8))) _Float16 mg_f16x8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 mg_bf16x8_t;
__device__ __forceinline__ f32x16 mg_mfma32(bf16x8 a, bf16x8 b, f32x16 c) {   // D = A (32 x 16) B (16 x 32) + C
  if constexpr (MG_F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(mg_f16x8_t, a), __builtin_bit_cast(mg_f16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mg_bf16x8_t, a), __builtin_bit_cast(mg_bf16x8_t, b), c, 0, 0, 0);
}

__device__ __forceinline__ float bf2f(bf16_t v) {
  if constexpr (MG_F16) return (float)__builtin_bit_cast(_Float16, v);
  else return __uint_as_float(((uint32_t)v) << 16);
}
// two fp32 -> packed operands (round to nearest even) in one instruction: v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32; the fp16 form
// saturates at +-65504 first (a bf16 value of 1e5 is a number, an fp16 inf poisons every sum it enters)
__device__ __forceinline__ uint32_t cvt_pk_bf16_f32(float lo, float hi) {
  if constexpr (MG_F16) {
    lo = __builtin_amdgcn_fmed3f(lo, -MG_OP16_MAX, MG_OP16_MAX);
    hi = __builtin_amdgcn_fmed3f(hi, -MG_OP16_MAX, MG_OP16_MAX);
  }
  mg_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mg_bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) {  // round-to-nearest-even, NaN preserved
  if constexpr (MG_F16) return (bf16_t)(cvt_pk_bf16_f32(f, 0.f) & 0xffffu);
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  if constexpr (MG_F16) return cvt_pk_bf16_f32(lo, hi);
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
__device__ __forceinline__ float bflo(uint32_t w) {
  if constexpr (MG_F16) return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu));
  else return __uint_as_float(w << 16);
}
__device__ __forceinline__ float bfhi(uint32_t w) {
  if constexpr (MG_F16) return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
  else return __uint_as_float(w & 0xffff0000u);
}
constexpr uint32_t MG_OP16_ONE_X2 = MG_F16 ? 0x3c003c00u : 0x3f803f80u;   // (1.0, 1.0) as packed operands
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// GELU (diffusers GEGLU: F.gelu, exact-erf form) with the Gaussian CDF as an odd polynomial, Phi(x) = 1/2 + x P(x^2) on |x| <= 4 (degree 6 in x^2,
// |error| <= 1.1e-4 on Phi, <= 4e-4 on x Phi(x) inside the interval - an eighth of a bf16 rounding step of the result -
// and <= 1e-4 relative beyond it, where x is clamped): 11 full-rate VALU operations, no transcendental.  The GEGLU
// epilogue of the K = 320 projection issues more VALU than its K loop issues MFMAs; this is its cheapest form.
__device__ __forceinline__ float gelu_poly_f(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f);
  const float t = xc * xc;
  float p = __builtin_fmaf(2.816104822e-08f, t, -1.891892127e-06f);
  p = __builtin_fmaf(p, t, 5.419045219e-05f);
  p = __builtin_fmaf(p, t, -8.789814671e-04f);
  p = __builtin_fmaf(p, t, 9.112954036e-03f);
  p = __builtin_fmaf(p, t, -6.538836045e-02f);
  p = __builtin_fmaf(p, t, 3.985269148e-01f);
  const float phi = fmaxf(__builtin_fmaf(xc, p, 0.5f), 0.0f);
  return x * phi;
}

// lane^32 regroup used by the MFMA epilogues: every lane holds two values g0, g1 (accumulator
// groups g and g+1 of its 32-lane half); afterwards `lo` = what the lower-half lane of the pair owns
// first, i.e. for lanes 0-31: (own g0, partner's g0), for lanes 32-63: (partner's g1, own g1).
// One v_permlane32_swap_b32 (VALU, no LDS traffic).
__device__ __forceinline__ void half_swap(float g0, float g1, float& first, float& second) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(g0), __float_as_uint(g1), false, false);
  first = __uint_as_float(r[0]);
  second = __uint_as_float(r[1]);
}

__device__ __forceinline__ float silu_fast_f(float x) {  // x * sigmoid(x): one v_exp, one v_rcp
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// Sum over the 64 lanes of a wave with DPP moves (VALU speed; __shfl_xor goes through the LDS crossbar
// - six dependent ds_bpermute round trips per reduction).  Result is wave-uniform.
__device__ __forceinline__ float wave_sum_f(float v) {
  auto dpp = [](float x, auto ctrl_tag, auto row_mask_tag) {
    constexpr int ctrl = decltype(ctrl_tag)::value, rm = decltype(row_mask_tag)::value;
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), ctrl, rm, 0xF, false));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});   // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});   // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{});  // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{});  // row_mirror: 16-lane row sums
  v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});  // row_bcast15 -> rows 1, 3
  v += dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});  // row_bcast31 -> rows 2, 3
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 63));
}

// 16-byte async global -> LDS copy.  LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_wave_base, 16, 0,
                                   0);
}

// Bijective XCD-aware remap of a linear workgroup id: consecutive remapped ids live on the
// same XCD (hardware places block b on XCD b % 8), so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned NX = 8;
  const unsigned xcd = bid % NX, idx = bid / NX;
  const unsigned q = nwg / NX, r = nwg % NX;
  const unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return (int)(start + idx);
}

// Division of x < 2^31 by a launch-time constant d >= 1 in three integer instructions (the compiler's generic 32-bit
// division is ~25, and the GEMM prologue / epilogue run a dozen of them per thread - measured: half of the instructions a
// short-K tile executes).  L = ceil(log2 d), m = ceil(2^(31+L) / d) < 2^32: floor(x / d) = (x * m) >> (31 + L) exactly,
// since the error term x * (m d - 2^(31+L)) < 2^31 * d <= 2^(31+L).  Written as umulhi(2x, m) >> L so that d = 1 needs no case.
struct mg_fastdiv {
  uint32_t m, l;
};
inline mg_fastdiv mg_make_fastdiv(long long d) {
  mg_fastdiv f;
  uint32_t l = 0;
  while ((1ull << l) < (unsigned long long)d) ++l;
  f.l = l;
  f.m = (uint32_t)(((1ull << (31 + l)) + (unsigned long long)d - 1) / (unsigned long long)d);   // d < 2^31: l <= 31
  return f;
}
__device__ __forceinline__ int fdiv(int x, const mg_fastdiv f) { return (int)(__umulhi((uint32_t)x << 1, f.m) >> f.l); }

// Cross-workgroup hand-offs inside one launch (the row-statistics tickets of igemm2_body.h, GroupNorm's last-block finalize in
// norm.hip, the key pieces of flash4w.hip).  The product form is the write-through one of cdna_hip_programming.md Guideline 16 /
// MI355X_MICROARCH.md "valid forms": 16-byte / 8-byte sc1 stores of the payload -> every storing wave drains vmcnt ->
// __syncthreads -> ONE relaxed agent-scope ticket; the last arriver reads the payload with sc1 loads ("sc1 loads may replace the
// acquire only when the producer stored sc1").  MG_HANDOFF_FENCES (Makefile: HANDOFF_FENCES=1) adds the C++-memory-model form on
// top - an agent-scope release fence in front of the ticket, an acquire fence in the last arriver - for same-box A/B runs of what
// that costs (profiles/r6_ab_handoff_fences.log).
constexpr bool MG_HANDOFF_FENCES = MG_HANDOFF_FENCES_BUILD != 0;
__device__ __forceinline__ void mg_handoff_release() {   // lane 0, behind the workgroup barrier that follows the drained stores
  if constexpr (MG_HANDOFF_FENCES) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (ROCm 7.2 may drop the wait behind buffer_wbl2: restated where it cannot)
  }
}
__device__ __forceinline__ void mg_handoff_acquire() {   // the last arriver's lane 0, in front of the barrier that releases its readers
  if constexpr (MG_HANDOFF_FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

#define MG_ZERO_BYTES (128 * 1024)
extern void* g_zero_page;  // MG_ZERO_BYTES zero bytes in device memory (mg_init): padding source
// Dry run (mg_program_validate): every launcher checks its op's shape / alignment contract and
// returns before touching the device - the CPU test-suite validates full-size programs with it.
extern thread_local bool g_dry_run;
#define MG_SPLITK_WS_BYTES (64ll * 1024 * 1024)
#define MG_LN_COUNTERS 65536
extern unsigned* g_ln_counters;  // MG_LN_COUNTERS zeroed tickets (mg_init): one per row block of a GEMM launch that writes row
                                 // statistics (the last column tile of a row block reduces them); self-resetting, stream-ordered
extern void* g_splitk_ws;  // fp32 partial sums of split-K GEMM launches (mg_init); stream-ordered reuse
#define MG_LAUNCH(...)                                   \
  do {                                                   \
    if (!g_dry_run) hipLaunchKernelGGL(__VA_ARGS__);     \
  } while (0)
void mg_set_error(const char* fmt, ...);
// Tuning switch `name` (an environment variable) if MARIGOLD_TUNING=1, else `dflt`: the library's only access to the
// environment (runtime.hip).  Deployments run the compiled-in defaults.
int mg_tuning_int(const char* name, int dflt);
#define MG_CHECK_HIP(expr)                                                         \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      mg_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return 1;                                                                    \
    }                                                                              \
  } while (0)
#define MG_REQUIRE(cond, ...)      \
  do {                             \
    if (!(cond)) {                 \
      mg_set_error(__VA_ARGS__);   \
      return 2;                    \
    }                              \
  } while (0)

// launchers (one per .hip file)
int mg_launch_igemm(const mg_op* op, hipStream_t s);
int mg_launch_igemm2(const mg_op* op, hipStream_t s, int variant);  // -1: shape outside the kernel's contract
int mg_launch_conv_patch(const mg_op* op, hipStream_t s);
int mg_conv3x3_gn_slots_of(const mg_op* op);
int mg_launch_rowgemm(const mg_op* op, hipStream_t s);
int mg_launch_norm(const mg_op* op, hipStream_t s);
int mg_launch_attention(const mg_op* op, hipStream_t s);
int mg_launch_flash512(const mg_op* op, hipStream_t s);
int mg_launch_misc(const mg_op* op, hipStream_t s);
int mg_launch_head_conv(const mg_op* op, hipStream_t s);
int mg_launch_ensemble(const mg_op* op, hipStream_t s);
int mg_launch_resize(const mg_op* op, hipStream_t s);
