// MG_OP_ROWGEMM: row-resident GEMM for the token-local Linear layers of the two widest transformer levels (gfx950 only).
//
//   out[M][N] = epilogue( x[M][K] W[N][K]^T ),  K = 320 (the 96 x 96-token level of the SD2 UNet), M = B * tokens = 92 160
//
// Why a second GEMM kernel.  At K = 320 the tile GEMM (igemm2) is neither MFMA- nor HBM-bound: a 128 x 320 output tile
// re-stages the whole 200 KB weight matrix and its 80 KB of rows through LDS for five K steps of work, then stalls on its
// own epilogue - 46 us for 118 MB (profiles/r2_trans_section_probe.log), 176 us for the fused QKV projection in the
// pipeline, against ~50 us of HBM time.  Here the roles are turned round:
//   * a WAVE owns 32 whole rows of x for the entire launch: their K = 320 values are loaded ONCE, straight into the 80
//     VGPRs the MFMAs read them from (no LDS pass, no re-read per column tile);
//   * the WEIGHTS stream: all waves of the workgroup (12 = three per SIMD, 384 rows) walk the N columns in 64-column
//     stages; a stage is 40 KB of weights pre-packed on the host in MFMA FRAGMENT order (weights.pack_rowgemm), so one
//     `global_load_lds_dwordx4` moves one fragment (1 KB, contiguous in memory AND in LDS - conflict-free `ds_read_b128`
//     at immediate offsets from a single address register, no swizzle, no index arithmetic), three stages deep;
//   * a stage's 32 x 64 outputs are finished in registers (folded LayerNorm, bias, residual, GEGLU, V^T) and stored
//     while the other two waves of the SIMD keep its matrix pipe busy; row statistics for the next folded LayerNorm
//     fall out per wave (it holds whole rows) - no tickets, no second pass.
// 240 workgroups x 12 waves = one round over the chip for E = 10; the weights are read from L2 once per workgroup
// (614 KB for QKV), x from HBM exactly once.
//
// K order.  Lane (row r = lane % 32, half h = lane / 32) holds x[r][16 s + 8 h .. + 8] as the operand of K step s - the
// MFMA's own order, so a stage's packed bf16 output registers ARE the operand of K step 4 j + 2 t + q of a following GEMM
// (the fused cross-attention uses that for its probabilities) and a row's own channels are at hand as the residual of an
// in-place layer.  Channel order inside a 32-channel tile: MFMA row index mm holds channel chan(mm) = mm with bits 2 and 3 exchanged, so
// that a lane's accumulator registers 0-7 / 8-15 are channels 8h .. 8h+7 / 16+8h .. 16+8h+7 - 16-byte stores with no
// lane exchange.  In the V^T section of the QKV projection the operand roles are swapped (x fragment as MFMA A, weights
// as B - the register images are the same) and the accumulator then holds, per channel, tokens {4h..4h+3, 8+4h..} - the
// key order MG_OP_FLASH_ATTN64 i[7] consumes, again 16-byte stores.
//
// Synchronisation: one s_barrier per stage.  A wave waits for its OWN LDS-DMA pieces of stage j with a counted vmcnt
// before barrier j; vmcnt retires in order on gfx9 and counts stores, so the wait leaves exactly the younger operations in
// flight: this wave's pieces of stage j+1 (>= 3) and the S stores of the previous stage's epilogue.  M % 32 == 0 is
// required, so a wave either owns 32 real rows and issues every store unconditionally (S is a compile-time constant) or
// - the surplus waves of the last workgroup - owns none and only moves its share of the weight stream.
#include <stdlib.h>

#include "common.h"

namespace {

struct RgArgs {
  const bf16_t* x;      // [M][ldx]
  const char* wp;       // packed weights: N/64 stages x (2 tiles x K/16 fragments of 1 KB + 1 KB trailer: fp32 [64] per-channel
                        // constants (bias + folded-LayerNorm c), [64] folded-LayerNorm g) - weights.pack_rowgemm
  bf16_t* out;          // [M][ldo]
  bf16_t* vt;           // QKV: V^T [B][N - trans_from][ldt]
  const bf16_t* res;    // [M][ldr] | NULL
  const float2* ln_in;  // [M] (mean, rstd)
  float2* ln_out;       // [M] | NULL
  const float* gn_ss;   // [B][2][K] GroupNorm (scale, shift) applied to x while it is loaded | NULL
  int M, N, ldx, ldo, ldr, ldt, T, trans_stage;
  float ln_eps;
  double inv_n;
  int prio;
  int spl;              // stages per workgroup (N / 64 unless the columns are split over gridDim.y)
  // XA (round 6): the collapsed cross-attention as the PROLOGUE of the GEGLU form - the rows x = h1 are updated in registers
  // (h2 = P VO^T + bias + h1, stored once to xout for ff.out's residual) and the folded LayerNorm of the GEGLU projection takes its
  // (mean, rstd) from the wave's own sums: no cross-attention launch, no second read of the residual stream
  const char* xwp;      // weights.pack_rowgemm_xattn image (scores stage + VO^T fragments + bias) | NULL
  bf16_t* xout;         // [M][ldx] the updated rows (may alias x)
  int sm_cols;
  float sm_scale;
  double inv_k;
  unsigned long long* dbg;   // tuning only: per-wave phase cycles [wave-tiles][4] (wait+barrier, issue, MFMA, epilogue) | NULL
};

enum { RG_BF16 = 0, RG_GEGLU = 1, RG_QKV = 2, RG_XATTN = 3 };

// The stage trailer (per-channel constants) is read with hand-placed LDS instructions: hipcc's waitcnt pass makes every LDS
// load it can see wait for ALL outstanding LDS-DMA (vmcnt(0)) - in the middle of the stage that would drain the weight
// prefetch.  The fragment reads escape that only because they carry no memory operand after unrolling; these would not.
template <int OFF>
__device__ __forceinline__ f32x4 rg_lds16(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ float rg_lds4(uint32_t addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// all LDS reads of this wave have landed; the operands tie the consumers behind the wait
__device__ __forceinline__ void rg_lgk0(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void rg_lgk0(f32x4& a, f32x4& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void rg_lgk0(float& a, float& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }

template <int OFF>
__device__ __forceinline__ bf16x8 rg_ldsw(uint32_t addr) {   // one weight fragment
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// <= N of this wave's LDS reads still in flight (they return in order); the operands tie their consumers behind the wait
template <int N>
__device__ __forceinline__ void rg_lgk(bf16x8& a, bf16x8& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int I, int N, class F>
__device__ __forceinline__ void rg_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    rg_static_for<I + 1, N>(f);
  }
}

template <int N>
__device__ __forceinline__ void rg_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int K, int NW, int EPI, bool LN, bool GN, bool RES, bool LNO, bool XA = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4)))
void rowgemm_kernel(const RgArgs a) {
  static_assert(!XA || (K == 320 && EPI == RG_GEGLU && LN && !GN && !RES && !LNO), "the cross-attention prologue rides on the K = 320 GEGLU form");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // a stage = 2 tiles x KS fragments of 1 KB + one 1 KB trailer: fp32 [64] per-channel constants, [64] folded-LayerNorm g
  // K = 640: a SLOT of the ring holds ONE 32-channel tile (40 fragments) and a 64-channel stage is two consecutive slots
  // (the second one's trailer carries the stage's constants) - the slot is 41 KB either way
  constexpr int KS = K / 16, TS = K <= 320 ? 2 : 1, SPS = 2 / TS;   // tiles per slot, slots per stage
  constexpr int PIECES = TS * KS + 1, STAGE = PIECES * 1024, NSTAGE = 3, TRL = TS * KS * 1024;
  constexpr int NWMIN = PIECES / NW, NWREM = PIECES - NWMIN * NW;   // LDS-DMA pieces per wave and stage: NWMIN (+1 for waves < NWREM)
  constexpr bool GEGLU = EPI == RG_GEGLU;
  constexpr int S = GEGLU ? 2 : 4;                         // stores per stage epilogue
  constexpr int XW = NWMIN + S + (RES ? 4 : 0);            // operations younger than stage j's pieces that may stay in flight
  static_assert((K == 320 || K == 640) && NWMIN >= 1, "geometry");
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  int row0 = blockIdx.x * (NW * 32) + wave * 32;
  // a surplus wave of the last workgroup (M is a multiple of 32, not of the workgroup's rows) only moves its share of the
  // weight stream and keeps the barriers: it must not touch rows that their owner may already have updated in place
  const bool active = row0 < a.M;
  if (!active) row0 = a.M - 32;
  const int m = row0 + l31;
  // this workgroup's stages [j0, nst): all N / 64 of them, or - few rows, many columns - one of gridDim.y column ranges
  const int j0 = blockIdx.y * a.spl;
  const int nst = (a.N >> 6) < j0 + a.spl ? (a.N >> 6) : j0 + a.spl;
  // The three waves of a SIMD (wave, wave + 4, wave + 8) leave every stage barrier together; with equal priority their
  // MFMA phases interleave and end together, and then all three run their VALU epilogues with the matrix pipe idle.  Fixed,
  // distinct priorities serialise the MFMA phases instead: the first wave's epilogue runs under the second wave's MFMAs.
  if (a.prio) {
    if (wave < 4) __builtin_amdgcn_s_setprio(3);
    else if (wave < 8) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(1);
  }

  auto issue = [&](int j, int slot) {
    const char* src = a.wp + (long long)j * STAGE + lane * 16;
    char* dst = smem + slot * STAGE;
#pragma unroll
    for (int i = 0; i < NWMIN; ++i) glds16(src + (wave + i * NW) * 1024, dst + (wave + i * NW) * 1024);
    if (wave < NWREM) glds16(src + (wave + NWMIN * NW) * 1024, dst + (wave + NWMIN * NW) * 1024);
  };
  const int h0 = j0 * SPS, hend = nst * SPS;   // this workgroup's slots
  issue(h0, 0);
  if constexpr (!XA) { if (h0 + 1 < hend) issue(h0 + 1, 1); }

  // the wave's 32 rows of x: K step s -> x[m][16 s + 8 half .. + 8] (the MFMA's own K order)
  bf16x8 xf[KS];
  {
    const bf16_t* px = a.x + (long long)m * a.ldx + 8 * half;
#pragma unroll
    for (int s = 0; s < KS; ++s) xf[s] = *(const bf16x8*)(px + 16 * s);
    if constexpr (GN) {   // GroupNorm apply folded into the load: bf16(x * scale + shift), the rounding MG_OP_GN_APPLY has
      const float* ps = a.gn_ss + (long long)(row0 / a.T) * 2 * K + 8 * half;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float4 s0 = *(const float4*)(ps + 16 * s), s1 = *(const float4*)(ps + 16 * s + 4);
        const float4 h0 = *(const float4*)(ps + K + 16 * s), h1 = *(const float4*)(ps + K + 16 * s + 4);
        const uint4 v = __builtin_bit_cast(uint4, xf[s]);
        uint4 o;
        o.x = cvt_pk_bf16_f32(__builtin_fmaf(bflo(v.x), s0.x, h0.x), __builtin_fmaf(bfhi(v.x), s0.y, h0.y));
        o.y = cvt_pk_bf16_f32(__builtin_fmaf(bflo(v.y), s0.z, h0.z), __builtin_fmaf(bfhi(v.y), s0.w, h0.w));
        o.z = cvt_pk_bf16_f32(__builtin_fmaf(bflo(v.z), s1.x, h1.x), __builtin_fmaf(bfhi(v.z), s1.y, h1.y));
        o.w = cvt_pk_bf16_f32(__builtin_fmaf(bflo(v.w), s1.z, h1.z), __builtin_fmaf(bfhi(v.w), s1.w, h1.w));
        xf[s] = __builtin_bit_cast(bf16x8, o);
      }
    }
  }
  float l_sc = 1.f, l_mr = 0.f;
  if constexpr (LN) {
    const float2 st = a.ln_in[m];
    l_sc = st.y;
    l_mr = -st.y * st.x;
  }
  if constexpr (XA) {
    // ---- the collapsed cross-attention on the rows in registers (rowgemm_xattn_kernel's arithmetic, operation for operation:
    // the results are bit-identical to the separate launch).  Its two weight images sit behind ring slot 0, which already
    // receives the first GEGLU stage; slots 1 and 2 are primed once every wave has left the images.
    constexpr int NSUB = K / 64, P0 = 2 * KS + 1, P1 = NSUB * 8 + 2, S0 = P0 * 1024, XOFF = STAGE;
    auto issue_img = [&](auto pieces_tag, int src_off, int dst_off) {
      constexpr int P = decltype(pieces_tag)::value;
      const char* src = a.xwp + src_off + lane * 16;
      char* dst = smem + XOFF + dst_off;
#pragma unroll
      for (int i = 0; i < P / NW; ++i) glds16(src + (wave + i * NW) * 1024, dst + (wave + i * NW) * 1024);
      if (wave < P % NW) glds16(src + (wave + (P / NW) * NW) * 1024, dst + (wave + (P / NW) * NW) * 1024);
    };
    issue_img(std::integral_constant<int, P0>{}, 0, 0);
    issue_img(std::integral_constant<int, P1>{}, S0, S0);
    rg_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (active) {
      const char* const sb = smem + XOFF + lane * 16;
      const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_AS char*)smem + (uint32_t)XOFF;
      bf16x8 pf[4];
      {
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const bf16x8 w0 = __builtin_bit_cast(bf16x8, *(const uint4*)(sb + s * 1024));
          const bf16x8 w1 = __builtin_bit_cast(bf16x8, *(const uint4*)(sb + (KS + s) * 1024));
          acc0 = mg_mfma32(w0, xf[s], acc0);
          acc1 = mg_mfma32(w1, xf[s], acc1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int c = t * 32 + q * 16 + 8 * half;   // this lane's 8 score columns: 4 (key 0, key 1) pairs
            const uint32_t cbp = lds0 + (uint32_t)(2 * KS * 1024 + c * 4);
            f32x4 c0 = rg_lds16<0>(cbp), c1 = rg_lds16<16>(cbp), g0 = rg_lds16<256>(cbp), g1 = rg_lds16<272>(cbp);
            rg_lgk0(c0, c1, g0, g1);
            const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            uint32_t w4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float a0 = t == 0 ? acc0[8 * q + 2 * k] : acc1[8 * q + 2 * k], a1 = t == 0 ? acc0[8 * q + 2 * k + 1] : acc1[8 * q + 2 * k + 1];
              const float s0 = __builtin_fmaf(a0, l_sc, __builtin_fmaf(l_mr, gg[2 * k], cc[2 * k])) * a.sm_scale;
              const float s1 = __builtin_fmaf(a1, l_sc, __builtin_fmaf(l_mr, gg[2 * k + 1], cc[2 * k + 1])) * a.sm_scale;
              const float mx = fmaxf(s0, s1);
              const float e0 = __expf(s0 - mx), e1 = __expf(s1 - mx);
              const float inv = 1.0f / (e0 + e1);
              w4[k] = (c + 2 * k < a.sm_cols) ? pack2bf(e0 * inv, e1 * inv) : 0u;
            }
            pf[t * 2 + q] = __builtin_bit_cast(bf16x8, make_uint4(w4[0], w4[1], w4[2], w4[3]));
          }
      }
      bf16_t* const px2 = a.xout + (long long)m * a.ldx + 8 * half;
      double sd2 = 0.0, qd2 = 0.0;
#pragma unroll
      for (int jj = 0; jj < NSUB; ++jj) {
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bf16x8 w0 = __builtin_bit_cast(bf16x8, *(const uint4*)(sb + S0 + ((jj * 2 + 0) * 4 + s) * 1024));
          const bf16x8 w1 = __builtin_bit_cast(bf16x8, *(const uint4*)(sb + S0 + ((jj * 2 + 1) * 4 + s) * 1024));
          acc0 = mg_mfma32(w0, pf[s], acc0);
          acc1 = mg_mfma32(w1, pf[s], acc1);
        }
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int c = jj * 64 + t * 32 + q * 16;
            const uint32_t bp = lds0 + (uint32_t)(S0 + NSUB * 8 * 1024 + (c + 8 * half) * 4);
            f32x4 b0 = rg_lds16<0>(bp), b1 = rg_lds16<16>(bp);
            rg_lgk0(b0, b1);
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            const uint4 r4 = __builtin_bit_cast(uint4, xf[jj * 4 + t * 2 + q]);   // the row's own channels c + 8 half .. + 8
            const float rr[8] = {bflo(r4.x), bfhi(r4.x), bflo(r4.y), bfhi(r4.y), bflo(r4.z), bfhi(r4.z), bflo(r4.w), bfhi(r4.w)};
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              v[i] = (t == 0 ? acc0[8 * q + i] : acc1[8 * q + i]) + bb[i] + rr[i];
              ps += v[i];
              pq = __builtin_fmaf(v[i], v[i], pq);
            }
            uint4 pk;
            pk.x = cvt_pk_bf16_f32(v[0], v[1]); pk.y = cvt_pk_bf16_f32(v[2], v[3]);
            pk.z = cvt_pk_bf16_f32(v[4], v[5]); pk.w = cvt_pk_bf16_f32(v[6], v[7]);
            *(uint4*)(px2 + c) = pk;
            xf[jj * 4 + t * 2 + q] = __builtin_bit_cast(bf16x8, pk);   // ... and K steps 4 jj + 2 t + q of the GEGLU projection
          }
        asm volatile("" : "+v"(ps), "+v"(pq));
        sd2 += (double)ps;
        qd2 += (double)pq;
        __builtin_amdgcn_sched_barrier(0);
      }
      // (mean, rstd) of the new rows exactly as the separate launch stores and the GEGLU form reloads them
      sd2 += __shfl_xor(sd2, 32);
      qd2 += __shfl_xor(qd2, 32);
      const double mean = sd2 * a.inv_k;
      const float var = fmaxf((float)__builtin_fma(qd2, a.inv_k, -mean * mean), 0.f);
      const float mean_f = (float)mean, rstd = __builtin_amdgcn_rsqf(var + a.ln_eps);
      l_sc = rstd;
      l_mr = -rstd * mean_f;
    }
    __builtin_amdgcn_s_barrier();   // every wave has left the images: ring slots 1 and 2 are free
    if (h0 + 1 < hend) issue(h0 + 1, 1);
  }
  bf16_t* const po = a.out + (long long)m * a.ldo + 8 * half;
  const bf16_t* const pr = RES ? a.res + (long long)m * a.ldr + 8 * half : nullptr;
  uint4 rres[4] = {};   // the residual rows of the NEXT stage, requested before that stage's weights (vmcnt retires in order)
  auto load_res = [&](int j) {
#pragma unroll
    for (int g = 0; g < 4; ++g) rres[g] = *(const uint4*)(pr + j * 64 + g * 16);
  };
  if constexpr (RES) load_res(j0);
  // V^T: image b, first token of the wave's tile; lane = channel chan(l31) of a tile, registers = tokens
  bf16_t* pvt = nullptr;
  if constexpr (EPI == RG_QKV) {
    const int b = row0 / a.T, tok0 = row0 - b * a.T;
    const int ch = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);   // bits 2 and 3 exchanged
    pvt = a.vt + ((long long)b * (a.N - a.trans_stage * 64) + ch) * a.ldt + tok0 + 8 * half;
  }
  double sd = 0.0, qd = 0.0;

  int slot = 0, slot_i = 2;   // stage j lives in slot j % 3; stage j + 2 goes to slot (j + 2) % 3
  // One stage.  VSEC (the V^T section of the QKV form) is a compile-time tag and the two sections are two loops: as one
  // body with a branch, hipcc hoists the (identical) fragment reads of both arms above it - 160 live registers of fragments.
  auto advance = [&]() {
    slot = slot == NSTAGE - 1 ? 0 : slot + 1;
    slot_i = slot_i == NSTAGE - 1 ? 0 : slot_i + 1;
  };
  unsigned long long dwait = 0ull;
  // slot h has landed for every wave and slot h - 1 is free: request slot h + 2 into it
  auto head = [&](int h) {
    const unsigned long long d0 = a.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    if (active) {
      if (h + 1 < hend) rg_wait_vmcnt<XW>(); else rg_wait_vmcnt<XW - NWMIN>();
    } else {   // nothing but weight pieces in this wave's queue
      if (h + 1 < hend) rg_wait_vmcnt<NWMIN>(); else rg_wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (a.dbg) dwait += __builtin_amdgcn_s_memtime() - d0;
    if (h + 2 < hend) issue(h + 2, slot_i);
  };
  // The fragment reads run PD K steps ahead of their MFMAs, hand-placed with counted lgkmcnt waits: hipcc keeps one or two
  // in flight, which leaves the phase bound by LDS latency (2 700 cycles for 1 280 cycles of MFMAs: r3_rowgemm_small_batch.log)
  constexpr int PD = RES ? (LNO ? 1 : 2) : 3, PR = PD + 1;   // (the residual forms hold 16-22 more registers)
  auto mfma_slot = [&](auto vsec_tag, f32x16& accA, f32x16& accB) {   // TS == 2: both tiles; TS == 1: the slot's tile into accA
    constexpr bool vsec = decltype(vsec_tag)::value;
    const uint32_t fb = (uint32_t)(uintptr_t)(LDS_AS char*)smem + (uint32_t)(slot * STAGE + lane * 16);
    bf16x8 w0[PR], w1[PR];
    rg_static_for<0, PD>([&](auto it) {
      constexpr int s = decltype(it)::value;
      w0[s] = rg_ldsw<s * 1024>(fb);
      if constexpr (TS == 2) w1[s] = rg_ldsw<(KS + s) * 1024>(fb);
    });
    rg_static_for<0, KS>([&](auto it) {
      constexpr int s = decltype(it)::value;
      if constexpr (s + PD < KS) {
        w0[(s + PD) % PR] = rg_ldsw<(s + PD) * 1024>(fb);
        if constexpr (TS == 2) w1[(s + PD) % PR] = rg_ldsw<(KS + s + PD) * 1024>(fb);
      }
      constexpr int ahead = (KS - 1 - s) < PD ? (KS - 1 - s) : PD;
      if constexpr (TS == 2) rg_lgk<2 * ahead>(w0[s % PR], w1[s % PR]);
      else rg_lgk<ahead>(w0[s % PR], w0[s % PR == 0 ? 1 : 0]);
      if constexpr (!vsec) {
        accA = mg_mfma32(w0[s % PR], xf[s], accA);
        if constexpr (TS == 2) accB = mg_mfma32(w1[s % PR], xf[s], accB);
      } else {
        accA = mg_mfma32(xf[s], w0[s % PR], accA);
        if constexpr (TS == 2) accB = mg_mfma32(xf[s], w1[s % PR], accB);
      }
    });
  };
  // One 64-channel stage.  VSEC (the V^T section of the QKV form) is a compile-time tag and the two sections are two loops: as
  // one body with a branch, hipcc hoists the (identical) fragment reads of both arms above it - 160 live registers of fragments.
  auto stage = [&](int j, auto vsec_tag) {
    constexpr bool vsec = decltype(vsec_tag)::value;
    const unsigned long long d1 = a.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long w_before = dwait;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    if constexpr (TS == 2) {
      head(j);
      if (!active) { advance(); return; }
      mfma_slot(vsec_tag, acc0, acc1);
    } else {
      head(2 * j);
      if (active) mfma_slot(vsec_tag, acc0, acc1);
      advance();
      head(2 * j + 1);
      if (!active) { advance(); return; }
      mfma_slot(vsec_tag, acc1, acc0);
    }
    // fp32 [64] constants, [64] g of this stage (K = 640: in the trailer of its second slot); + this lane's first channel
    // (8 half; the V^T section: chan(l31))
    const uint32_t tcb = (uint32_t)(uintptr_t)(LDS_AS char*)smem + (uint32_t)(slot * STAGE + TRL);
    // ---- the stage's 32 rows x 64 channels, in registers ----
    __builtin_amdgcn_sched_barrier(0);   // fragment registers are dead before the epilogue's temporaries go live
    unsigned long long d2 = 0ull;
    if (a.dbg) {
      asm volatile("s_nop 0" : "+v"(acc0), "+v"(acc1));
      d2 = __builtin_amdgcn_s_memtime();
    }
    const unsigned long long d0 = d1;   // (stamps: stage start, end of the MFMA phase(s), end of the epilogue; the heads' waits apart)
    if constexpr (GEGLU) {
      // tile 0 = 32 value channels, tile 1 = their gates (weights.rowgemm_geglu_order); out column = 32 j + channel
      uint4 pk[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float o[8];
#pragma unroll
        for (int t = 1; t >= 0; --t) {   // the gate tile first: gelu(gate) is all that stays live across the value tile
          const uint32_t cbp = tcb + (uint32_t)(t * 128 + q * 64 + 32 * half);
          f32x4 c0 = rg_lds16<0>(cbp), c1 = rg_lds16<16>(cbp), g0 = c0, g1 = c0;
          if constexpr (LN) { g0 = rg_lds16<256>(cbp); g1 = rg_lds16<272>(cbp); }
          rg_lgk0(c0, c1, g0, g1);
          const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
          const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float av = t == 0 ? acc0[8 * q + i] : acc1[8 * q + i];
            const float v = __builtin_fmaf(av, l_sc, LN ? __builtin_fmaf(l_mr, gg[i], cc[i]) : cc[i]);
            if (t == 1) o[i] = gelu_poly_f(v);
            else o[i] = v * o[i];
          }
        }
        pk[q].x = cvt_pk_bf16_f32(o[0], o[1]); pk[q].y = cvt_pk_bf16_f32(o[2], o[3]);
        pk[q].z = cvt_pk_bf16_f32(o[4], o[5]); pk[q].w = cvt_pk_bf16_f32(o[6], o[7]);
      }
      *(uint4*)(po + j * 32) = pk[0];
      *(uint4*)(po + j * 32 + 16) = pk[1];
    } else if constexpr (vsec) {
      // registers 8q .. 8q+7 of tile t = tokens (key slots) 16 q + 8 half .. + 8 of channel 32 t + chan(l31)
      bf16_t* pv = pvt + (long long)((j - a.trans_stage) * 64) * a.ldt;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ch = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
        float cc = rg_lds4<0>(tcb + (uint32_t)((t * 32 + ch) * 4)), gg = cc;
        if constexpr (LN) gg = rg_lds4<256>(tcb + (uint32_t)((t * 32 + ch) * 4));
        rg_lgk0(cc, gg);
        // the folded LayerNorm's row terms belong to the TOKEN here (a register), not to the lane: fetched from the lane
        // that owns the row (ds_bpermute - not a vector load, whose in-order return would wait for the weight prefetch)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float av = t == 0 ? acc0[8 * q + i] : acc1[8 * q + i];
            float sc = 1.f, mr = 0.f;
            if constexpr (LN) {
              const int tokl = 16 * q + 8 * (i >> 2) + 4 * half + (i & 3);   // MFMA row of register 8q+i
              sc = __shfl(l_sc, tokl);
              mr = __shfl(l_mr, tokl);
            }
            v[i] = __builtin_fmaf(av, sc, LN ? __builtin_fmaf(mr, gg, cc) : cc);
          }
          uint4 pk;
          pk.x = cvt_pk_bf16_f32(v[0], v[1]); pk.y = cvt_pk_bf16_f32(v[2], v[3]);
          pk.z = cvt_pk_bf16_f32(v[4], v[5]); pk.w = cvt_pk_bf16_f32(v[6], v[7]);
          *(uint4*)(pv + (long long)(t * 32) * a.ldt + 16 * q) = pk;
        }
      }
    } else {
      float ps = 0.f, pq = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int c = j * 64 + t * 32 + q * 16;
          const uint32_t cbp = tcb + (uint32_t)(t * 128 + q * 64 + 32 * half);
          f32x4 c0 = rg_lds16<0>(cbp), c1 = rg_lds16<16>(cbp), g0 = c0, g1 = c0;
          if constexpr (LN) { g0 = rg_lds16<256>(cbp); g1 = rg_lds16<272>(cbp); }
          rg_lgk0(c0, c1, g0, g1);
          const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
          const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
          uint4 r4 = make_uint4(0, 0, 0, 0);
          if constexpr (RES) r4 = rres[t * 2 + q];
          const float rr[8] = {bflo(r4.x), bfhi(r4.x), bflo(r4.y), bfhi(r4.y), bflo(r4.z), bfhi(r4.z), bflo(r4.w), bfhi(r4.w)};
          float v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float av = t == 0 ? acc0[8 * q + i] : acc1[8 * q + i];
            v[i] = __builtin_fmaf(av, l_sc, LN ? __builtin_fmaf(l_mr, gg[i], cc[i]) : cc[i]);
            if constexpr (RES) v[i] += rr[i];
            if constexpr (LNO) { ps += v[i]; pq = __builtin_fmaf(v[i], v[i], pq); }
          }
          uint4 pk;
          pk.x = cvt_pk_bf16_f32(v[0], v[1]); pk.y = cvt_pk_bf16_f32(v[2], v[3]);
          pk.z = cvt_pk_bf16_f32(v[4], v[5]); pk.w = cvt_pk_bf16_f32(v[6], v[7]);
          *(uint4*)(po + c) = pk;
        }
      if constexpr (LNO) { sd += (double)ps; qd += (double)pq; }
      // the next stage's residual rows: behind this stage's stores, ahead of the next weight prefetch in the vmcnt queue
      if constexpr (RES) { if (j + 1 < nst) load_res(j + 1); }
    }
    if (a.dbg && lane == 0) {
      const unsigned long long d3 = __builtin_amdgcn_s_memtime();
      unsigned long long* o = a.dbg + (long long)(blockIdx.x * NW + wave) * 4;
      o[0] += dwait - w_before; o[1] += d2 - d0 - (dwait - w_before); o[2] += d3 - d2; o[3] += 1;
    }
    advance();
  };
  if constexpr (EPI == RG_QKV) {
    const int nq = a.trans_stage < nst ? a.trans_stage : nst;
    for (int j = j0; j < nq; ++j) stage(j, std::false_type{});
    for (int j = nq > j0 ? nq : j0; j < nst; ++j) stage(j, std::true_type{});
  } else {
    for (int j = j0; j < nst; ++j) stage(j, std::false_type{});
  }
  if constexpr (LNO) {   // (mean, rstd) of the new rows: the wave holds them whole (fp64 only for E[x^2] - mean^2)
    sd += __shfl_xor(sd, 32);
    qd += __shfl_xor(qd, 32);
    if (half == 0 && active) {
      const double mean = sd * a.inv_n;
      const float var = fmaxf((float)__builtin_fma(qd, a.inv_n, -mean * mean), 0.f);
      a.ln_out[m] = make_float2((float)mean, __builtin_amdgcn_rsqf(var + a.ln_eps));
    }
  }
}

// ---- the collapsed 2-token cross-attention (diffusers BasicTransformerBlock.attn2 against the constant empty-prompt
// context; MG_OP_IGEMM's MG_EPI_XATTN2) in the row-resident form, IN PLACE on the residual stream -------------------------
//   scores [32 rows][64] = LN(x) Wqk^T (folded LayerNorm, one ordinary stage) -> softmax over column pairs (2 keys per head)
//   -> the packed probabilities are K steps 0..3 of  out = P VO^T + bias + x  (c2 = K channels, K/64 sub-stages of 8
//   fragments) -> (mean, rstd) of the new rows.  x is read once (its registers are also the residual), the result written
//   once: 118 MB for the 92 160 x 320 level.  Two weight images, both requested up front: [41 KB scores stage][K/64 x 8 KB
//   of VO^T fragments + 2 KB of bias].
struct RgXArgs {
  const bf16_t* x;
  const char* wp;
  bf16_t* out;
  const float2* ln_in;
  float2* ln_out;
  int M, ldx, ldo, sm_cols;
  float sm_scale, ln_eps;
  double inv_n;
};

template <int K, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4)))
void rowgemm_xattn_kernel(const RgXArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = K / 16, NSUB = K / 64, P0 = 2 * KS + 1, P1 = NSUB * 8 + 2, S0 = P0 * 1024;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  int row0 = blockIdx.x * (NW * 32) + wave * 32;
  const bool active = row0 < a.M;   // (see rowgemm_kernel: a surplus wave moves weights and keeps the barriers, nothing else)
  if (!active) row0 = a.M - 32;
  const int m = row0 + l31;
  auto issue = [&](auto pieces_tag, int src_off, int dst_off) {
    constexpr int P = decltype(pieces_tag)::value;
    const char* src = a.wp + src_off + lane * 16;
    char* dst = smem + dst_off;
#pragma unroll
    for (int i = 0; i < P / NW; ++i) glds16(src + (wave + i * NW) * 1024, dst + (wave + i * NW) * 1024);
    if (wave < P % NW) glds16(src + (wave + (P / NW) * NW) * 1024, dst + (wave + (P / NW) * NW) * 1024);
  };
  issue(std::integral_constant<int, P0>{}, 0, 0);
  bf16x8 xf[KS];
  {
    const bf16_t* px = a.x + (long long)m * a.ldx + 8 * half;
#pragma unroll
    for (int s = 0; s < KS; ++s) xf[s] = *(const bf16x8*)(px + 16 * s);
  }
  const float2 st = a.ln_in[m];
  const float l_sc = st.y, l_mr = -st.y * st.x;
  issue(std::integral_constant<int, P1>{}, S0, S0);
  rg_wait_vmcnt<P1 / NW>();   // everything older than this wave's second-image pieces: the scores stage, x, the statistics
  __builtin_amdgcn_s_barrier();
  const char* const sb = smem + lane * 16;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_AS char*)smem;
  bf16x8 pf[4];
  {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bf16x8 w0 = __builtin_bit_cast(bf16x8, *(const uint4*)(sb + s * 1024));
      const bf16x8 w1 = __builtin_bit_cast(bf16x8, *(const uint4*)(sb + (KS + s) * 1024));
      acc0 = mg_mfma32(w0, xf[s], acc0);
      acc1 = mg_mfma32(w1, xf[s], acc1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int c = t * 32 + q * 16 + 8 * half;   // this lane's 8 score columns: 4 (key 0, key 1) pairs
        const uint32_t cbp = lds0 + (uint32_t)(2 * KS * 1024 + c * 4);
        f32x4 c0 = rg_lds16<0>(cbp), c1 = rg_lds16<16>(cbp), g0 = rg_lds16<256>(cbp), g1 = rg_lds16<272>(cbp);
        rg_lgk0(c0, c1, g0, g1);
        const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        uint32_t w4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a0 = t == 0 ? acc0[8 * q + 2 * k] : acc1[8 * q + 2 * k], a1 = t == 0 ? acc0[8 * q + 2 * k + 1] : acc1[8 * q + 2 * k + 1];
          const float s0 = __builtin_fmaf(a0, l_sc, __builtin_fmaf(l_mr, gg[2 * k], cc[2 * k])) * a.sm_scale;
          const float s1 = __builtin_fmaf(a1, l_sc, __builtin_fmaf(l_mr, gg[2 * k + 1], cc[2 * k + 1])) * a.sm_scale;
          const float mx = fmaxf(s0, s1);
          const float e0 = __expf(s0 - mx), e1 = __expf(s1 - mx);
          const float inv = 1.0f / (e0 + e1);
          w4[k] = (c + 2 * k < a.sm_cols) ? pack2bf(e0 * inv, e1 * inv) : 0u;
        }
        pf[t * 2 + q] = __builtin_bit_cast(bf16x8, make_uint4(w4[0], w4[1], w4[2], w4[3]));
      }
  }
  rg_wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (!active) return;
  bf16_t* const po = a.out + (long long)m * a.ldo + 8 * half;
  double sd = 0.0, qd = 0.0;
#pragma unroll
  for (int jj = 0; jj < NSUB; ++jj) {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 w0 = __builtin_bit_cast(bf16x8, *(const uint4*)(sb + S0 + ((jj * 2 + 0) * 4 + s) * 1024));
      const bf16x8 w1 = __builtin_bit_cast(bf16x8, *(const uint4*)(sb + S0 + ((jj * 2 + 1) * 4 + s) * 1024));
      acc0 = mg_mfma32(w0, pf[s], acc0);
      acc1 = mg_mfma32(w1, pf[s], acc1);
    }
    float ps = 0.f, pq = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int c = jj * 64 + t * 32 + q * 16;
        const uint32_t bp = lds0 + (uint32_t)(S0 + NSUB * 8 * 1024 + (c + 8 * half) * 4);
        f32x4 b0 = rg_lds16<0>(bp), b1 = rg_lds16<16>(bp);
        rg_lgk0(b0, b1);
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        const uint4 r4 = __builtin_bit_cast(uint4, xf[jj * 4 + t * 2 + q]);   // the row's own channels c + 8 half .. + 8
        const float rr[8] = {bflo(r4.x), bfhi(r4.x), bflo(r4.y), bfhi(r4.y), bflo(r4.z), bfhi(r4.z), bflo(r4.w), bfhi(r4.w)};
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          v[i] = (t == 0 ? acc0[8 * q + i] : acc1[8 * q + i]) + bb[i] + rr[i];
          ps += v[i];
          pq = __builtin_fmaf(v[i], v[i], pq);
        }
        uint4 pk;
        pk.x = cvt_pk_bf16_f32(v[0], v[1]); pk.y = cvt_pk_bf16_f32(v[2], v[3]);
        pk.z = cvt_pk_bf16_f32(v[4], v[5]); pk.w = cvt_pk_bf16_f32(v[6], v[7]);
        *(uint4*)(po + c) = pk;
      }
    asm volatile("" : "+v"(ps), "+v"(pq));   // taken here (hipcc otherwise sinks the sums into the `if (ln_out)` block and keeps all 320 values alive)
    sd += (double)ps;
    qd += (double)pq;
    __builtin_amdgcn_sched_barrier(0);   // (unrolled for the register-resident residual: keep the sub-stages' fragment reads apart)
  }
  if (a.ln_out) {
    sd += __shfl_xor(sd, 32);
    qd += __shfl_xor(qd, 32);
    if (half == 0) {
      const double mean = sd * a.inv_n;
      const float var = fmaxf((float)__builtin_fma(qd, a.inv_n, -mean * mean), 0.f);
      a.ln_out[m] = make_float2((float)mean, __builtin_amdgcn_rsqf(var + a.ln_eps));
    }
  }
}

template <int K, int NW>
int rg_launch_xattn(const RgXArgs& a, hipStream_t s) {
  constexpr int LDS = (2 * (K / 16) + 1 + (K / 64) * 8 + 2) * 1024;
  static bool attr_set = false;
  auto kern = rowgemm_xattn_kernel<K, NW>;
  if (!attr_set && !g_dry_run) {
    MG_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  const int rows = NW * 32;
  MG_LAUNCH(kern, dim3((a.M + rows - 1) / rows), dim3(NW * 64), LDS, s, a);
  return 0;
}

// ---- the collapsed cross-attention on the DEEP levels (K = c2 = 640 / 1 280 channels, a few thousand rows): K-split form ----
// The tile-GEMM form (MG_EPI_XATTN2) walks 20 K steps and 40 column blocks serially in 45-180 workgroups: 42-82 us for a few
// MFLOP.  Here a workgroup is 32 rows x 4 waves and BOTH reductions are cut four ways: wave w holds the K / 4 channels
// [K/4 w, K/4 (w+1)) of the rows in registers, computes the partial 32 x 64 scores over them (weight fragments straight from
// global / L2, pre-packed per wave: weights.pack_rowgemm_xattn_ksplit), the partials meet in LDS, every wave forms the
// probabilities, and wave w then produces the output channels of ITS slice - whose residual it already holds.  M / 32
// workgroups (180 at the 24 x 24 level), each a chain of ~100 loads and 100 MFMAs.
struct RgXkArgs {
  const bf16_t* x;
  const char* wp;      // [4 waves][2 tiles][K/64 steps] score fragments, [4][K/128 tiles][4 steps] VO^T fragments, fp32 cb[64] lg[64] bias[K]
  bf16_t* out;
  const float2* ln_in;
  float2* ln_out;
  int M, ldx, ldo, sm_cols;
  float sm_scale, ln_eps;
  double inv_n;
};

template <int K>
__global__ __launch_bounds__(256) void rowgemm_xattn_ksplit_kernel(const RgXkArgs a) {
  constexpr int KQ = K / 4, KSQ = KQ / 16, NT2 = KQ / 32;
  constexpr int WQK_BYTES = 4 * 2 * KSQ * 1024, VOT_BYTES = 4 * NT2 * 4 * 1024;
  __shared__ __attribute__((aligned(16))) float part[4][32][64];   // partial score accumulators [wave][register][lane]
  __shared__ double rstat[4][32][2];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int m = blockIdx.x * 32 + l31;
  bf16x8 xf[KSQ];
  {
    const bf16_t* px = a.x + (long long)m * a.ldx + KQ * wave + 8 * half;
#pragma unroll
    for (int s = 0; s < KSQ; ++s) xf[s] = *(const bf16x8*)(px + 16 * s);
  }
  const float2 st = a.ln_in[m];
  const float l_sc = st.y, l_mr = -st.y * st.x;
  // ---- partial scores over this wave's K quarter ----
  {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const char* wq = a.wp + (long long)wave * (2 * KSQ * 1024) + lane * 16;
#pragma unroll
    for (int s = 0; s < KSQ; ++s) {
      const bf16x8 w0 = *(const bf16x8*)(wq + s * 1024);
      const bf16x8 w1 = *(const bf16x8*)(wq + (KSQ + s) * 1024);
      acc0 = mg_mfma32(w0, xf[s], acc0);
      acc1 = mg_mfma32(w1, xf[s], acc1);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { part[wave][r][lane] = acc0[r]; part[wave][16 + r][lane] = acc1[r]; }
  }
  __syncthreads();
  bf16x8 pf[4];
  {
    const float* cbp = (const float*)(a.wp + WQK_BYTES + VOT_BYTES);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int c = t * 32 + q * 16 + 8 * half;
        const float4 c0 = *(const float4*)(cbp + c), c1 = *(const float4*)(cbp + c + 4);
        const float4 g0 = *(const float4*)(cbp + 64 + c), g1 = *(const float4*)(cbp + 64 + c + 4);
        const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        float sv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = t * 16 + q * 8 + i;
          const float av = (part[0][r][lane] + part[1][r][lane]) + (part[2][r][lane] + part[3][r][lane]);
          sv[i] = __builtin_fmaf(av, l_sc, __builtin_fmaf(l_mr, gg[i], cc[i])) * a.sm_scale;
        }
        uint32_t w4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float mx = fmaxf(sv[2 * k], sv[2 * k + 1]);
          const float e0 = __expf(sv[2 * k] - mx), e1 = __expf(sv[2 * k + 1] - mx);
          const float inv = 1.0f / (e0 + e1);
          w4[k] = (c + 2 * k < a.sm_cols) ? pack2bf(e0 * inv, e1 * inv) : 0u;
        }
        pf[t * 2 + q] = __builtin_bit_cast(bf16x8, make_uint4(w4[0], w4[1], w4[2], w4[3]));
      }
  }
  // ---- this wave's output channels [KQ wave, KQ (wave + 1)): P VO^T + bias + x ----
  const char* wv = a.wp + WQK_BYTES + (long long)wave * (NT2 * 4 * 1024) + lane * 16;
  const float* pb = (const float*)(a.wp + WQK_BYTES + VOT_BYTES) + 128 + KQ * wave + 8 * half;
  bf16_t* const po = a.out + (long long)m * a.ldo + KQ * wave + 8 * half;
  float ps = 0.f, pq = 0.f;
  double sd = 0.0, qd = 0.0;
#pragma unroll
  for (int tt = 0; tt < NT2; ++tt) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 w = *(const bf16x8*)(wv + (tt * 4 + s) * 1024);
      acc = mg_mfma32(w, pf[s], acc);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float4 b0 = *(const float4*)(pb + tt * 32 + q * 16), b1 = *(const float4*)(pb + tt * 32 + q * 16 + 4);
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      const uint4 r4 = __builtin_bit_cast(uint4, xf[tt * 2 + q]);
      const float rr[8] = {bflo(r4.x), bfhi(r4.x), bflo(r4.y), bfhi(r4.y), bflo(r4.z), bfhi(r4.z), bflo(r4.w), bfhi(r4.w)};
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[i] = acc[8 * q + i] + bb[i] + rr[i];
        ps += v[i];
        pq = __builtin_fmaf(v[i], v[i], pq);
      }
      uint4 pk;
      pk.x = cvt_pk_bf16_f32(v[0], v[1]); pk.y = cvt_pk_bf16_f32(v[2], v[3]);
      pk.z = cvt_pk_bf16_f32(v[4], v[5]); pk.w = cvt_pk_bf16_f32(v[6], v[7]);
      *(uint4*)(po + tt * 32 + q * 16) = pk;
    }
    if ((tt & 1) == 1 || tt == NT2 - 1) {   // fp32 partials of <= 64 values, then fp64 (as the tile GEMM's statistics)
      asm volatile("" : "+v"(ps), "+v"(pq));
      sd += (double)ps; qd += (double)pq;
      ps = 0.f; pq = 0.f;
    }
  }
  if (a.ln_out) {   // (mean, rstd) of the new rows: the four waves' channel quarters meet in LDS, summed in wave order
    sd += __shfl_xor(sd, 32);
    qd += __shfl_xor(qd, 32);
    if (half == 0) { rstat[wave][l31][0] = sd; rstat[wave][l31][1] = qd; }
    __syncthreads();
    if (wave == 0 && half == 0) {
      const double s4 = (rstat[0][l31][0] + rstat[1][l31][0]) + (rstat[2][l31][0] + rstat[3][l31][0]);
      const double q4 = (rstat[0][l31][1] + rstat[1][l31][1]) + (rstat[2][l31][1] + rstat[3][l31][1]);
      const double mean = s4 * a.inv_n;
      const float var = fmaxf((float)__builtin_fma(q4, a.inv_n, -mean * mean), 0.f);
      a.ln_out[m] = make_float2((float)mean, __builtin_amdgcn_rsqf(var + a.ln_eps));
    }
  }
}

template <int K>
int rg_launch_xattn_ksplit(const RgXkArgs& a, hipStream_t s) {
  MG_LAUNCH(rowgemm_xattn_ksplit_kernel<K>, dim3(a.M / 32), dim3(256), 0, s, a);
  return 0;
}

template <int K, int NW, int EPI, bool LN, bool GN, bool RES, bool LNO, bool XA = false>
int rg_launch(const RgArgs& a, hipStream_t s) {
  constexpr int RING = 3 * ((K <= 320 ? 2 : 1) * (K / 16) + 1) * 1024;
  // XA: the cross-attention images (scores stage + VO^T fragments + bias) sit behind ring slot 0
  constexpr int LDS = XA ? (RING / 3 + (2 * (K / 16) + 1 + (K / 64) * 8 + 2) * 1024 > RING ? RING / 3 + (2 * (K / 16) + 1 + (K / 64) * 8 + 2) * 1024 : RING) : RING;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  auto kern = rowgemm_kernel<K, NW, EPI, LN, GN, RES, LNO, XA>;
  if (!attr_set && !g_dry_run) {
    MG_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  const int rows = NW * 32;
  MG_LAUNCH(kern, dim3((a.M + rows - 1) / rows, ((a.N >> 6) + a.spl - 1) / a.spl), dim3(NW * 64), LDS, s, a);
  return 0;
}

template <int K, int NW>
int rg_dispatch(const RgArgs& a, int epi, hipStream_t s) {
  const bool ln = a.ln_in != nullptr, gn = a.gn_ss != nullptr, res = a.res != nullptr, lno = a.ln_out != nullptr;
  if (epi == RG_GEGLU) {
    MG_REQUIRE(!gn && !res && !lno, "rowgemm: the GEGLU form takes no GroupNorm input / residual / row statistics");
    if (a.xwp) {
      if constexpr (K == 320) {
        MG_REQUIRE(ln, "rowgemm: the cross-attention prologue needs the (mean, rstd) table of its input rows");
        return rg_launch<K, NW, RG_GEGLU, true, false, false, false, true>(a, s);
      } else MG_REQUIRE(false, "rowgemm: the cross-attention prologue is instantiated for K = 320");
    }
    return ln ? rg_launch<K, NW, RG_GEGLU, true, false, false, false>(a, s) : rg_launch<K, NW, RG_GEGLU, false, false, false, false>(a, s);
  }
  if (epi == RG_QKV) {
    MG_REQUIRE(!gn && !res && !lno, "rowgemm: the QKV form takes no GroupNorm input / residual / row statistics");
    return ln ? rg_launch<K, NW, RG_QKV, true, false, false, false>(a, s) : rg_launch<K, NW, RG_QKV, false, false, false, false>(a, s);
  }
  MG_REQUIRE(!(gn && (ln || res)), "rowgemm: GroupNorm input excludes the folded LayerNorm and the residual");
  if (gn) return lno ? rg_launch<K, NW, RG_BF16, false, true, false, true>(a, s) : rg_launch<K, NW, RG_BF16, false, true, false, false>(a, s);
  if (ln) {
    MG_REQUIRE(!lno, "rowgemm: folded LayerNorm input + row statistics output is not instantiated");
    return res ? rg_launch<K, NW, RG_BF16, true, false, true, false>(a, s) : rg_launch<K, NW, RG_BF16, true, false, false, false>(a, s);
  }
  if (res) return lno ? rg_launch<K, NW, RG_BF16, false, false, true, true>(a, s) : rg_launch<K, NW, RG_BF16, false, false, true, false>(a, s);
  return lno ? rg_launch<K, NW, RG_BF16, false, false, false, true>(a, s) : rg_launch<K, NW, RG_BF16, false, false, false, false>(a, s);
}

}  // namespace

int mg_launch_rowgemm(const mg_op* op, hipStream_t s) {
  RgArgs a;
  a.M = op->i[0];
  const int K = op->i[1];
  a.N = op->i[2];
  a.ldx = op->i[3] > 0 ? op->i[3] : K;
  const int epi = op->i[6];
  a.ldo = op->i[4] > 0 ? op->i[4] : (epi == RG_GEGLU ? a.N / 2 : a.N);
  a.ldr = op->i[5] > 0 ? op->i[5] : a.N;
  a.T = op->i[7];
  a.ldt = op->i[8];
  const int trans_from = op->i[9];
  const int nw = op->i[10] > 0 ? op->i[10] : 12;
  a.ln_eps = op->f[0];
  a.x = (const bf16_t*)op->p[0];
  a.wp = (const char*)op->p[1];
  a.out = (bf16_t*)op->p[2];
  a.res = (const bf16_t*)op->p[3];
  a.ln_in = (const float2*)op->p[4];
  a.ln_out = (float2*)op->p[5];
  a.vt = (bf16_t*)op->p[6];
  a.gn_ss = (const float*)op->p[7];
  a.dbg = (unsigned long long*)op->p[8];
  a.xwp = epi == RG_GEGLU ? (const char*)op->p[9] : nullptr;
  a.xout = (bf16_t*)op->p[10];
  a.sm_cols = op->i[11];
  a.sm_scale = op->f[1];
  a.inv_k = 1.0 / (double)K;
  a.inv_n = 1.0 / (double)(a.N > 0 ? a.N : 1);
  a.trans_stage = epi == RG_QKV ? trans_from / 64 : (1 << 30);
  a.prio = 1;
  {
    const int nsplit = op->i[12] > 1 ? op->i[12] : 1, nst = a.N >> 6;
    a.spl = (nst + nsplit - 1) / nsplit;
    if (a.spl < 2 && nst >= 2) a.spl = 2;   // (the weight ring is primed two stages deep)
    MG_REQUIRE(nsplit == 1 || !a.ln_out, "rowgemm: row statistics need whole rows in one workgroup (no column split)");
  }
  MG_REQUIRE(epi >= RG_BF16 && epi <= RG_XATTN, "rowgemm: unknown form %d", epi);
  if (a.xwp) {
    MG_REQUIRE(K == 320 && a.xout && a.ln_in && (op->i[12] <= 1) && a.sm_cols > 0 && a.sm_cols % 2 == 0 && a.sm_cols <= 64 &&
               (uintptr_t)a.xwp % 16 == 0 && (uintptr_t)a.xout % 16 == 0,
               "rowgemm: the cross-attention prologue (p[9]) rides on the unsplit K = 320 GEGLU form; p[10] = the updated rows, p[4] = the (mean, rstd) "
               "table of the rows as loaded, i[11] = 2 x heads score columns");
  }
  if (epi == RG_XATTN && (K == 640 || K == 1280)) {
    RgXkArgs x;
    x.x = a.x; x.wp = a.wp; x.out = a.out; x.ln_in = a.ln_in; x.ln_out = a.ln_out;
    x.M = a.M; x.ldx = a.ldx; x.ldo = op->i[4] > 0 ? op->i[4] : K; x.sm_cols = op->i[11];
    x.sm_scale = op->f[1]; x.ln_eps = a.ln_eps; x.inv_n = 1.0 / (double)K;
    MG_REQUIRE(x.x && x.wp && x.out && x.ln_in, "rowgemm: the cross-attention form needs x, packed weights, out and the (mean, rstd) table of x");
    MG_REQUIRE(a.N == 64 && x.M >= 32 && x.M % 32 == 0 && x.ldx >= K && x.ldx % 8 == 0 && x.ldo >= K && x.ldo % 8 == 0, "rowgemm: rows / leading dimensions");
    MG_REQUIRE(x.sm_cols > 0 && x.sm_cols % 2 == 0 && x.sm_cols <= 64, "rowgemm: score columns (2 per head, <= 64)");
    MG_REQUIRE((uintptr_t)x.x % 16 == 0 && (uintptr_t)x.wp % 16 == 0 && (uintptr_t)x.out % 16 == 0 && (uintptr_t)x.ln_in % 8 == 0 &&
               (!x.ln_out || (uintptr_t)x.ln_out % 8 == 0), "rowgemm: alignment");
    return K == 640 ? rg_launch_xattn_ksplit<640>(x, s) : rg_launch_xattn_ksplit<1280>(x, s);
  }
  if (epi == RG_XATTN) {
    RgXArgs x;
    x.x = a.x; x.wp = a.wp; x.out = a.out; x.ln_in = a.ln_in; x.ln_out = a.ln_out;
    x.M = a.M; x.ldx = a.ldx; x.ldo = op->i[4] > 0 ? op->i[4] : K; x.sm_cols = op->i[11];
    x.sm_scale = op->f[1]; x.ln_eps = a.ln_eps; x.inv_n = 1.0 / (double)K;
    MG_REQUIRE(x.x && x.wp && x.out && x.ln_in, "rowgemm: the cross-attention form needs x, packed weights, out and the (mean, rstd) table of x");
    MG_REQUIRE(K == 320 && a.N == 64, "rowgemm: the cross-attention form is instantiated for K = c2 = 320 and 64 score columns (got K %d, N %d)", K, a.N);
    MG_REQUIRE(x.M >= 32 && x.M % 32 == 0 && x.ldx >= K && x.ldx % 8 == 0 && x.ldo >= K && x.ldo % 8 == 0, "rowgemm: rows / leading dimensions");
    MG_REQUIRE(x.sm_cols > 0 && x.sm_cols % 2 == 0 && x.sm_cols <= 64, "rowgemm: score columns (2 per head, <= 64)");
    MG_REQUIRE((uintptr_t)x.x % 16 == 0 && (uintptr_t)x.wp % 16 == 0 && (uintptr_t)x.out % 16 == 0 && (uintptr_t)x.ln_in % 8 == 0 &&
               (!x.ln_out || (uintptr_t)x.ln_out % 8 == 0), "rowgemm: alignment");
    if (nw == 12) return rg_launch_xattn<320, 12>(x, s);
    if (nw == 8) return rg_launch_xattn<320, 8>(x, s);
    MG_REQUIRE(false, "rowgemm: %d waves per workgroup is not instantiated for the cross-attention form (8, 12)", nw);
  }
  MG_REQUIRE(a.x && a.wp && a.out, "rowgemm: null pointer (x, packed weights, out)");
  MG_REQUIRE(K == 320 || K == 640 || (K == 1280 && epi == RG_XATTN), "rowgemm: K = %d is not instantiated (320, 640; 1280 for the cross-attention form)", K);
  MG_REQUIRE(a.M >= 32 && a.M % 32 == 0, "rowgemm: M = %d must be a multiple of 32 (a wave owns 32 whole rows)", a.M);
  MG_REQUIRE(a.N >= 128 && a.N % 64 == 0, "rowgemm: N = %d must be a multiple of 64, >= 128", a.N);
  MG_REQUIRE(a.ldx >= K && a.ldx % 8 == 0 && a.ldo % 8 == 0 && a.ldo >= (epi == RG_GEGLU ? a.N / 2 : (epi == RG_QKV ? trans_from : a.N)),
             "rowgemm: leading dimensions (ldx %d, ldo %d)", a.ldx, a.ldo);
  MG_REQUIRE((uintptr_t)a.x % 16 == 0 && (uintptr_t)a.wp % 16 == 0 && (uintptr_t)a.out % 16 == 0, "rowgemm: 16-byte alignment of x / weights / out");
  if (a.ln_in) MG_REQUIRE((uintptr_t)a.ln_in % 8 == 0, "rowgemm: misaligned (mean, rstd) table of the input rows");
  if (a.res) MG_REQUIRE(epi == RG_BF16 && a.ldr >= a.N && a.ldr % 8 == 0 && (uintptr_t)a.res % 16 == 0, "rowgemm: residual layout");
  if (a.ln_out) MG_REQUIRE((uintptr_t)a.ln_out % 8 == 0, "rowgemm: misaligned (mean, rstd) table");
  if (a.gn_ss) MG_REQUIRE(a.T > 0 && a.T % 32 == 0 && a.M % a.T == 0 && (uintptr_t)a.gn_ss % 16 == 0, "rowgemm: GroupNorm input needs tokens per image %% 32 == 0");
  if (epi == RG_QKV)
    MG_REQUIRE(a.vt && trans_from > 0 && trans_from % 64 == 0 && trans_from < a.N && a.T > 0 && a.T % 32 == 0 && a.M % a.T == 0 &&
               a.ldt >= a.T && a.ldt % 8 == 0 && (uintptr_t)a.vt % 16 == 0,
               "rowgemm: the QKV form needs V^T (p[8]), trans_from %% 64 == 0, tokens per image %% 32 == 0, ldt >= tokens");
  if (K == 640) {   // 160 registers of rows per wave: two waves per SIMD
    const int nw6 = op->i[10] > 0 ? op->i[10] : 8;
    if (nw6 == 8) return rg_dispatch<640, 8>(a, epi, s);
    MG_REQUIRE(false, "rowgemm: K = 640 runs 8 waves per workgroup (got %d)", nw6);
  }
  if (nw == 12) return rg_dispatch<320, 12>(a, epi, s);
  if (nw == 8) return rg_dispatch<320, 8>(a, epi, s);
  if (nw == 4) return rg_dispatch<320, 4>(a, epi, s);
  MG_REQUIRE(false, "rowgemm: %d waves per workgroup is not instantiated (4, 8, 12)", nw);
}
