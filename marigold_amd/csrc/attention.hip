// Attention kernels.
//  * flash_attn64: bf16 MFMA flash attention for head dim 64 (SD-v2 UNet self-attention,
//    seq 9216/2304/576/144 at 768x768).  Replaces diffusers Attention / SDPA / xformers
//    (reference: script/depth/run.py:217-220; reached from marigold_depth_pipeline.py:461-463).
//    gfx950 design: 4 waves x 32 queries per workgroup; K tiles [64 keys][64 d] and V^T tiles
//    [64 d][64 keys] stream global->LDS with global_load_lds (16 B/lane), double buffered,
//    XOR-swizzled (on the source address) so fragment reads are conflict-free.  QK^T is computed
//    "swapped" (S^T = K Q^T) so every lane owns ONE query column: the online-softmax row
//    max/sum are lane-local plus a single lane^32 exchange, and the exponentiated P registers
//    feed the PV MFMA as its B operand with no cross-lane shuffle - V^T is read from LDS in the
//    matching key order (two ds_read_b64 per fragment).  V^T itself is produced by the QKV
//    projection's transposed epilogue (igemm2.hip), never by a transpose pass.
//  * softmax_rows: fp32 -> bf16 row softmax for the VAE's single-head d=512 attention, whose
//    scores are materialised by the GEMM kernel (288 GB HBM: 340 MB/member is cheap).
//  * softmax_pairs: 2-key softmax of the collapsed cross-attention (see marigold_hip.h).
#include <type_traits>

#include "common.h"

namespace {

struct FaArgs {
  const bf16_t* Q;
  const bf16_t* K;
  const bf16_t* Vt;
  bf16_t* O;
  const void* zero;
  int B, heads, Ntok, ldq, ldo, ldvt, nqb;
  long long sQ, sK, sVt, sO;
  float scale_log2;
};

constexpr int FA_QB = 128;   // queries per workgroup (4 waves x 32)
constexpr int FA_KB = 64;    // keys per tile
constexpr int FA_STAGE = 2 * FA_KB * 128;  // K tile + V^T tile, bytes

__global__ __launch_bounds__(256) void flash_attn64_kernel(const FaArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * FA_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int qb = bid % a.nqb;
  const int bh = bid / a.nqb;
  const int h = bh % a.heads, b = bh / a.heads;

  const bf16_t* Qb = a.Q + (long long)b * a.sQ + h * 64;
  const bf16_t* Kb = a.K + (long long)b * a.sK + h * 64;
  const bf16_t* Vb = a.Vt + (long long)b * a.sVt + (long long)h * 64 * a.ldvt;
  const char* zero = (const char*)a.zero;

  // Q fragments (MFMA B operand: column = query, k = d): 4 k-steps x 8 bf16
  const int q_row = qb * FA_QB + wave * 32 + l31;
  const int q_ld = q_row < a.Ntok ? q_row : a.Ntok - 1;
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    qf[ks] = __builtin_bit_cast(bf16x8, *(const uint4*)(Qb + (long long)q_ld * a.ldq + ks * 16 + half * 8));

  // staging: 2 chunks of K and 2 chunks of V^T per thread per tile
  int st_row[2], st_q[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int ci = it * 256 + tid;
    st_row[it] = ci >> 3;
    st_q[it] = (ci & 7) ^ ((st_row[it] >> 1) & 7);
  }
  auto stage = [&](int kt, int buf) {
    const int k0 = kt * FA_KB;
    char* sb = smem + buf * FA_STAGE;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int key = k0 + st_row[it];
      const char* src = key < a.Ntok ? (const char*)(Kb + (long long)key * a.ldq + st_q[it] * 8) : zero;
      glds16(src, sb + (it * 256 + wave * 64) * 16);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const char* src = (const char*)(Vb + (long long)st_row[it] * a.ldvt + k0 + st_q[it] * 8);
      glds16(src, sb + FA_KB * 128 + (it * 256 + wave * 64) * 16);
    }
  };

  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int nkt = (a.Ntok + FA_KB - 1) / FA_KB;
  stage(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nkt) stage(kt + 1, buf ^ 1);
    const char* sK = smem + buf * FA_STAGE;
    const char* sV = sK + FA_KB * 128;

    // ---- S^T = K Q^T : two 32-key sub-tiles ----
    f32x16 s[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t2][r] = 0.f;
      const int row = t2 * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int q = ks * 2 + half;
        const bf16x8 kf = __builtin_bit_cast(
            bf16x8, *(const uint4*)(sK + row * 128 + ((q ^ ((row >> 1) & 7)) << 4)));
        s[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[t2], 0, 0, 0);
      }
    }
    // ---- online softmax (log2 domain); lane owns query l31, keys (r&3)+8(r>>2)+4*half ----
    const int kbase = kt * FA_KB;
    float mx = -1e30f;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = s[t2][r] * a.scale_log2;
        if (kbase + FA_KB > a.Ntok) {
          const int key = kbase + t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (key >= a.Ntok) v = -1e30f;
        }
        s[t2][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f(m_run - m_new);
    m_run = m_new;
    float ps = 0.f;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = exp2f(s[t2][r] - m_new);
        s[t2][r] = p;
        ps += p;
      }
    l_run = l_run * alpha + ps;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;

    // ---- O^T += V^T P^T : k-step (t2, sh) covers keys 32*t2 + 16*sh + [0,16) ----
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
      for (int sh = 0; sh < 2; ++sh) {
        uint4 pw;
        pw.x = pack2bf(s[t2][8 * sh + 0], s[t2][8 * sh + 1]);
        pw.y = pack2bf(s[t2][8 * sh + 2], s[t2][8 * sh + 3]);
        pw.z = pack2bf(s[t2][8 * sh + 4], s[t2][8 * sh + 5]);
        pw.w = pack2bf(s[t2][8 * sh + 6], s[t2][8 * sh + 7]);
        const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
        const int c0 = 4 * t2 + 2 * sh;  // 16-B chunk holding keys 32*t2+16*sh+[0,8)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int row = dt * 32 + l31;
          const int sw = (row >> 1) & 7;
          const uint2 v0 = *(const uint2*)(sV + row * 128 + ((c0 ^ sw) << 4) + 8 * half);
          const uint2 v1 = *(const uint2*)(sV + row * 128 + (((c0 + 1) ^ sw) << 4) + 8 * half);
          uint4 vw;
          vw.x = v0.x; vw.y = v0.y; vw.z = v1.x; vw.w = v1.y;
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vw), pf, o[dt], 0, 0, 0);
        }
      }
    }
  }
  // ---- finalize: O[q][d] = o^T / l ----
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (q_row < a.Ntok) {
    bf16_t* orow = a.O + (long long)b * a.sO + (long long)q_row * a.ldo + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * half;
        uint2 pk;
        pk.x = pack2bf(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv);
        pk.y = pack2bf(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        *(uint2*)(orow + d) = pk;
      }
  }
}

// ---- generation 2 ------------------------------------------------------------------------------
// Same tiling and data flow as above; what changed is the per-tile VALU bill (the first profile had
// the kernel at 14 % of the MFMA roof with ~350 VALU instructions per 16 MFMAs):
//   * softmax in ONE fma + one bare v_exp_f32 per score: p = exp2(s*c - m*c) (c = scale*log2 e),
//     the running max is tracked on the raw scores;
//   * P -> bf16 with v_cvt_pk_bf16_f32 (16 instructions per tile instead of ~100 of integer rounding);
//   * the O / l rescale is skipped (wave-uniform branch) on tiles where no lane's max moved;
//   * 3-deep K / V^T ring with counted vmcnt + raw s_barrier (tile kt+2 in flight while kt computes).
typedef __attribute__((ext_vector_type(2))) __bf16 fa_bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float fa_f32x2_t;
__device__ __forceinline__ uint32_t fa_cvt_pk(float lo, float hi) {
  fa_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, fa_bf16x2_t));
}

constexpr int FA2_NSTAGE = 3;

// SPLIT: the four LDS-DMA pieces of tile kt+2 are issued behind the two QK^T MFMA groups instead
// of in one burst after the barrier (their issue cost then overlaps the wave's own MFMAs).
// NW: waves (x 32 queries) per workgroup.  PV: how P reaches the PV MFMA - 0: straight from the
// QK^T register layout, V^T read as two ds_read_b64 per fragment (2-way LDS bank conflicts);
// 1 / 2: P regrouped across lane^32 (ds_bpermute / v_permlane32_swap) so a lane holds 8 consecutive
// keys and V^T is read with one conflict-free ds_read_b128 per fragment.
template <bool SPLIT, int NW, int PV>
__global__ __launch_bounds__(NW * 64) void flash_attn64_v2_kernel(const FaArgs a) {
  constexpr int NT = NW * 64;
  constexpr int QB = NW * 32;
  constexpr int ITS = 512 / NT;   // 16-byte staging chunks per thread per K (and per V^T) tile
  __shared__ __attribute__((aligned(16))) char smem[FA2_NSTAGE * FA_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int nqb = (a.Ntok + QB - 1) / QB;
  const int qb = bid % nqb;
  const int bh = bid / nqb;
  const int h = bh % a.heads, b = bh / a.heads;

  const bf16_t* Qb = a.Q + (long long)b * a.sQ + h * 64;
  const bf16_t* Kb = a.K + (long long)b * a.sK + h * 64;
  const bf16_t* Vb = a.Vt + (long long)b * a.sVt + (long long)h * 64 * a.ldvt;
  const char* zero = (const char*)a.zero;

  const int q_row = qb * QB + wave * 32 + l31;
  const int q_ld = q_row < a.Ntok ? q_row : a.Ntok - 1;
  bf16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    qf[ks] = __builtin_bit_cast(bf16x8, *(const uint4*)(Qb + (long long)q_ld * a.ldq + ks * 16 + half * 8));

  // per-thread staging sources (2 chunks of K, 2 of V^T per tile), advanced by one tile per issue
  const char* k_src[ITS];
  const char* v_src[ITS];
  int k_row[ITS];
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const int ci = it * NT + tid;
    const int r = ci >> 3;
    const int q = (ci & 7) ^ ((r >> 1) & 7);
    k_row[it] = r;
    k_src[it] = (const char*)(Kb + (long long)r * a.ldq + q * 8);
    v_src[it] = (const char*)(Vb + (long long)r * a.ldvt + q * 8);
  }
  const long long k_step = (long long)FA_KB * a.ldq * 2;  // bytes per key tile
  int i_k0 = 0;                                            // first key of the next tile to issue
  auto issue_k = [&](int stage) {
    char* sb = smem + stage * FA_STAGE;
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      const char* src = (i_k0 + k_row[it] < a.Ntok) ? k_src[it] : zero;
      glds16(src, sb + (it * NT + wave * 64) * 16);
      k_src[it] += k_step;
    }
  };
  auto issue_v = [&](int stage) {
    char* sb = smem + stage * FA_STAGE;
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      glds16(v_src[it], sb + FA_KB * 128 + (it * NT + wave * 64) * 16);
      v_src[it] += FA_KB * 2;
    }
    i_k0 += FA_KB;
  };
  auto issue = [&](int stage) { issue_k(stage); issue_v(stage); };

  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;  // raw-score units
  const float c = a.scale_log2;

  const int nkt = (a.Ntok + FA_KB - 1) / FA_KB;
  issue(0);
  if (nkt > 1) issue(1);
  // One key tile.  st_c / st_i are literal constants in the steady-state loop (unrolled over the three
  // ring stages) so every LDS address is base + immediate; ISSUE / MASK are compile-time tags so the
  // steady state is branch-free (tile kt+2 always exists there, no ragged keys).
  auto tile = [&](int kt, int st_c, int st_i, auto issue_tag, auto mask_tag) {
    constexpr bool do_issue = decltype(issue_tag)::value;
    constexpr bool MASK = decltype(mask_tag)::value;
    if (do_issue || kt + 1 < nkt) {
      if constexpr (ITS == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (!SPLIT && do_issue) issue(st_i);
    const char* sK = smem + st_c * FA_STAGE;
    const char* sV = sK + FA_KB * 128;

    // ---- S^T = K Q^T : two 32-key sub-tiles ----
    f32x16 s[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t2][r] = 0.f;
      const int row = t2 * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int q = ks * 2 + half;
        const bf16x8 kf = __builtin_bit_cast(
            bf16x8, *(const uint4*)(sK + row * 128 + ((q ^ ((row >> 1) & 7)) << 4)));
        s[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[t2], 0, 0, 0);
      }
      if constexpr (SPLIT && do_issue) {
        if (t2 == 0) issue_k(st_i);
        else issue_v(st_i);
      }
    }
    // ---- online softmax; lane owns query l31, keys (r&3)+8(r>>2)+4*half of each sub-tile ----
    const int kbase = kt * FA_KB;
    if (MASK && kbase + FA_KB > a.Ntok) {  // ragged last tile: mask keys beyond Ntok
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (key >= a.Ntok) s[t2][r] = -1e30f;
        }
    }
    float mx = s[0][0];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t2][r]);
    {
      float m0, m1;  // own and partner (lane^32) maxima, in some order
      half_swap(mx, mx, m0, m1);
      mx = fmaxf(m0, m1);
    }
    if (__any(mx > m_run)) {  // some lane's running max moves: rescale O and l (alpha = 1 elsewhere)
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    const float mc = -m_run * c;
    float ps = 0.f;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t2][r], c, mc));
        s[t2][r] = p;
        ps += p;
      }
    l_run += ps;

    // ---- O^T += V^T P^T : k-step (t2, sh) covers keys 32*t2 + 16*sh + [0,16) ----
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
      for (int sh = 0; sh < 2; ++sh) {
        bf16x8 pf;
        if constexpr (PV == 0) {
          uint4 pw;
          pw.x = fa_cvt_pk(s[t2][8 * sh + 0], s[t2][8 * sh + 1]);
          pw.y = fa_cvt_pk(s[t2][8 * sh + 2], s[t2][8 * sh + 3]);
          pw.z = fa_cvt_pk(s[t2][8 * sh + 4], s[t2][8 * sh + 5]);
          pw.w = fa_cvt_pk(s[t2][8 * sh + 6], s[t2][8 * sh + 7]);
          pf = __builtin_bit_cast(bf16x8, pw);
        } else {
          // packed pieces: a = keys 8*(2sh)+4h+{0..3}, b = keys 8*(2sh+1)+4h+{0..3} of this lane
          const uint32_t a0 = fa_cvt_pk(s[t2][8 * sh + 0], s[t2][8 * sh + 1]);
          const uint32_t a1 = fa_cvt_pk(s[t2][8 * sh + 2], s[t2][8 * sh + 3]);
          const uint32_t b0 = fa_cvt_pk(s[t2][8 * sh + 4], s[t2][8 * sh + 5]);
          const uint32_t b1 = fa_cvt_pk(s[t2][8 * sh + 6], s[t2][8 * sh + 7]);
          uint4 pw;  // 8 consecutive keys 32*t2 + 16*sh + 8*half + [0,8)
          if constexpr (PV == 2) {
            const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            pw.x = r0[0]; pw.y = r1[0]; pw.z = r0[1]; pw.w = r1[1];
          } else {
            const uint32_t v0 = __shfl_xor(half ? a0 : b0, 32), v1 = __shfl_xor(half ? a1 : b1, 32);
            pw.x = half ? v0 : a0; pw.y = half ? v1 : a1;
            pw.z = half ? b0 : v0; pw.w = half ? b1 : v1;
          }
          pf = __builtin_bit_cast(bf16x8, pw);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int row = dt * 32 + l31;
          const int sw = (row >> 1) & 7;
          uint4 vw;
          if constexpr (PV == 0) {
            const int c0 = 4 * t2 + 2 * sh;
            const uint2 v0 = *(const uint2*)(sV + row * 128 + ((c0 ^ sw) << 4) + 8 * half);
            const uint2 v1 = *(const uint2*)(sV + row * 128 + (((c0 + 1) ^ sw) << 4) + 8 * half);
            vw.x = v0.x; vw.y = v0.y; vw.z = v1.x; vw.w = v1.y;
          } else {
            const int c0 = 4 * t2 + 2 * sh + half;
            vw = *(const uint4*)(sV + row * 128 + ((c0 ^ sw) << 4));
          }
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vw), pf, o[dt], 0, 0, 0);
        }
      }
    }
  };
  {
    int kt = 0;
    for (; kt + 5 <= nkt; kt += 3) {  // tiles kt..kt+2 all have a tile two ahead and are not the last
      tile(kt, 0, 2, std::true_type{}, std::false_type{});
      tile(kt + 1, 1, 0, std::true_type{}, std::false_type{});
      tile(kt + 2, 2, 1, std::true_type{}, std::false_type{});
    }
    int st_c = 0, st_i = 2;  // kt is a multiple of 3 here
    for (; kt + 2 < nkt; ++kt) {
      tile(kt, st_c, st_i, std::true_type{}, std::false_type{});
      st_c = (st_c + 1 == FA2_NSTAGE) ? 0 : st_c + 1;
      st_i = (st_i + 1 == FA2_NSTAGE) ? 0 : st_i + 1;
    }
    for (; kt < nkt; ++kt) {
      tile(kt, st_c, st_i, std::false_type{}, std::true_type{});
      st_c = (st_c + 1 == FA2_NSTAGE) ? 0 : st_c + 1;
      st_i = (st_i + 1 == FA2_NSTAGE) ? 0 : st_i + 1;
    }
  }
  // ---- finalize: O[q][d] = o^T / l; lane^32 exchange -> 8 consecutive d per lane, 16-byte stores ----
  float l0, l1;
  half_swap(l_run, l_run, l0, l1);
  const float l_tot = l0 + l1;
  const float inv = 1.0f / l_tot;
  bf16_t* orow = a.O + (long long)b * a.sO + (long long)q_row * a.ldo + h * 64;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        half_swap(o[dt][8 * gp + j] * inv, o[dt][8 * gp + 4 + j] * inv, v[j], v[4 + j]);
      }
      if (q_row < a.Ntok) {
        uint4 pk;
        pk.x = fa_cvt_pk(v[0], v[1]); pk.y = fa_cvt_pk(v[2], v[3]);
        pk.z = fa_cvt_pk(v[4], v[5]); pk.w = fa_cvt_pk(v[6], v[7]);
        *(uint4*)(orow + dt * 32 + 16 * gp + 8 * half) = pk;
      }
    }
}

// one workgroup per row; fp32 scores -> bf16 probabilities, pad columns zeroed
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S,
                                                           bf16_t* __restrict__ P, int ncols,
                                                           long long lds_, long long ldp) {
  __shared__ float red[8];
  const long long row = blockIdx.x;
  const float* s = S + row * lds_;
  bf16_t* p = P + row * ldp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float mx = -1e30f;
  for (int c = tid * 4; c < ncols; c += 1024) {
    if (c + 3 < ncols) {
      const float4 v = *(const float4*)(s + c);
      mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    } else {
      for (int j = c; j < ncols; ++j) mx = fmaxf(mx, s[j]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int c = tid * 4; c < ncols; c += 1024) {
    if (c + 3 < ncols) {
      const float4 v = *(const float4*)(s + c);
      sum += __expf(v.x - mx) + __expf(v.y - mx) + __expf(v.z - mx) + __expf(v.w - mx);
    } else {
      for (int j = c; j < ncols; ++j) sum += __expf(s[j] - mx);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int c = tid * 4; c < ldp; c += 1024) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (c + j < ncols) ? __expf(s[c + j] - mx) * inv : 0.f;
    if (c + 3 < ldp) {
      uint2 pk;
      pk.x = pack2bf(v[0], v[1]);
      pk.y = pack2bf(v[2], v[3]);
      *(uint2*)(p + c) = pk;
    } else {
      for (int j = 0; c + j < ldp; ++j) p[c + j] = f2bf(v[j]);
    }
  }
}

__global__ __launch_bounds__(256) void softmax_pairs_kernel(const float* __restrict__ S,
                                                            bf16_t* __restrict__ P, long long M,
                                                            int pairs, int lds_, int ldp,
                                                            float scale) {
  const int ppr = ldp >> 1;  // output pairs per row (incl. zero padding)
  const long long total = M * ppr;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    const long long m = i / ppr;
    const int pr = (int)(i % ppr);
    uint32_t w = 0;
    if (pr < pairs) {
      const float2 s = *(const float2*)(S + m * lds_ + 2 * pr);
      const float a = s.x * scale, b = s.y * scale;
      const float mx = fmaxf(a, b);
      const float ea = __expf(a - mx), eb = __expf(b - mx);
      const float inv = 1.0f / (ea + eb);
      w = pack2bf(ea * inv, eb * inv);
    }
    *(uint32_t*)(P + m * ldp + 2 * pr) = w;
  }
}

}  // namespace

int mg_launch_attention(const mg_op* op, hipStream_t s) {
  switch (op->kind) {
    case MG_OP_FLASH_ATTN64: {
      FaArgs a;
      a.Q = (const bf16_t*)op->p[0];
      a.K = (const bf16_t*)op->p[1];
      a.Vt = (const bf16_t*)op->p[2];
      a.O = (bf16_t*)op->p[3];
      a.zero = g_zero_page;
      a.B = op->i[0]; a.heads = op->i[1]; a.Ntok = op->i[2];
      a.ldq = op->i[3]; a.ldo = op->i[4]; a.ldvt = op->i[5];
      a.sQ = op->l[0]; a.sK = op->l[1]; a.sVt = op->l[2]; a.sO = op->l[3];
      a.scale_log2 = op->f[0] * 1.4426950408889634f;
      a.nqb = (a.Ntok + FA_QB - 1) / FA_QB;
      MG_REQUIRE(g_zero_page || g_dry_run, "flash_attn64: mg_init() not called");
      MG_REQUIRE(a.Q && a.K && a.Vt && a.O, "flash_attn64: null pointer");
      MG_REQUIRE(a.B > 0 && a.heads > 0 && a.Ntok > 0, "flash_attn64: empty problem");
      MG_REQUIRE(a.ldvt % 64 == 0 && a.ldvt >= a.Ntok, "flash_attn64: ldvt must be a multiple of 64 >= Ntok");
      MG_REQUIRE(a.ldq % 8 == 0 && a.ldo % 4 == 0, "flash_attn64: bad leading dims");
      const long long grid = (long long)a.nqb * a.heads * a.B;
      // i[6]: 0 = default; 1 = generation 1; 2.. = generation-2 variants kept for the tuning sweep
      const bool v2ok = (a.ldo % 8 == 0) && ((uintptr_t)a.O % 16 == 0) && (a.sO % 8 == 0);
      const int var = v2ok ? op->i[6] : 1;
      const long long g4 = (long long)((a.Ntok + 127) / 128) * a.heads * a.B;
      const long long g8 = (long long)((a.Ntok + 255) / 256) * a.heads * a.B;
      switch (var) {
        case 1: MG_LAUNCH(flash_attn64_kernel, dim3((unsigned)grid), dim3(256), 0, s, a); break;
        case 2: MG_LAUNCH((flash_attn64_v2_kernel<false, 4, 0>), dim3((unsigned)g4), dim3(256), 0, s, a); break;
        case 3: MG_LAUNCH((flash_attn64_v2_kernel<true, 8, 0>), dim3((unsigned)g8), dim3(512), 0, s, a); break;
        case 4: MG_LAUNCH((flash_attn64_v2_kernel<true, 4, 1>), dim3((unsigned)g4), dim3(256), 0, s, a); break;
        case 5: MG_LAUNCH((flash_attn64_v2_kernel<true, 4, 2>), dim3((unsigned)g4), dim3(256), 0, s, a); break;
        case 6: MG_LAUNCH((flash_attn64_v2_kernel<true, 8, 2>), dim3((unsigned)g8), dim3(512), 0, s, a); break;
        case 7: MG_LAUNCH((flash_attn64_v2_kernel<true, 8, 1>), dim3((unsigned)g8), dim3(512), 0, s, a); break;
        case 8: MG_LAUNCH((flash_attn64_v2_kernel<true, 4, 0>), dim3((unsigned)g4), dim3(256), 0, s, a); break;
        default:  // sweep (profiles/r1_sweep3_flash_variants.log): 8 waves pay off from a few thousand keys on,
                  // provided the grid still covers the 256 CUs twice (small ensembles: 4-wave blocks)
          if (a.Ntok >= 2048 && g8 >= 512) MG_LAUNCH((flash_attn64_v2_kernel<true, 8, 2>), dim3((unsigned)g8), dim3(512), 0, s, a);
          else MG_LAUNCH((flash_attn64_v2_kernel<true, 4, 2>), dim3((unsigned)g4), dim3(256), 0, s, a);
          break;
      }
      break;
    }
    case MG_OP_SOFTMAX_ROWS: {
      const int R = op->i[0], ncols = op->i[1], lds_ = op->i[2], ldp = op->i[3];
      MG_REQUIRE(R > 0 && ncols > 0 && lds_ % 4 == 0 && ldp % 4 == 0 && ldp >= ncols, "softmax_rows: bad dims");
      MG_LAUNCH(softmax_rows_kernel, dim3(R), dim3(256), 0, s, (const float*)op->p[0],
                         (bf16_t*)op->p[1], ncols, (long long)lds_, (long long)ldp);
      break;
    }
    case MG_OP_SOFTMAX_PAIRS: {
      const long long M = op->i[0];
      const int pairs = op->i[1], lds_ = op->i[2], ldp = op->i[3];
      MG_REQUIRE(M > 0 && pairs > 0 && ldp % 2 == 0 && 2 * pairs <= ldp && lds_ % 2 == 0, "softmax_pairs: bad dims");
      const long long total = M * (ldp / 2);
      const int grid = (int)min((total + 255) / 256, (long long)4096);
      MG_LAUNCH(softmax_pairs_kernel, dim3(grid), dim3(256), 0, s, (const float*)op->p[0],
                         (bf16_t*)op->p[1], M, pairs, lds_, ldp, op->f[0]);
      break;
    }
    default: MG_REQUIRE(false, "attention: bad op kind %d", op->kind);
  }
  if (!g_dry_run) MG_CHECK_HIP(hipGetLastError());
  return 0;
}
