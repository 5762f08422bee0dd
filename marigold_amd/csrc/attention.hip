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
#include "flash_args.h"
#include "flash25_body.h"

namespace {


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


// SPLIT: the four LDS-DMA pieces of tile kt+2 are issued behind the two QK^T MFMA groups instead
// of in one burst after the barrier (their issue cost then overlaps the wave's own MFMAs).
// NW: waves (x 32 queries) per workgroup.  PV: how P reaches the PV MFMA - 0: straight from the
// QK^T register layout, V^T read as two ds_read_b64 per fragment (2-way LDS bank conflicts);
// 1 / 2: P regrouped across lane^32 (ds_bpermute / v_permlane32_swap) so a lane holds 8 consecutive
// keys and V^T is read with one conflict-free ds_read_b128 per fragment; 3: as 0, for a V^T whose keys the producer
// stored in the accumulator order (vt_perm) - one ds_read_b128 per fragment and no regroup.
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
        if constexpr (PV == 0 || PV == 3) {
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
          } else {   // PV == 3: V^T's keys are stored in the accumulator order (vt_perm) - the native packing matches
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


// ---- generation 2.5 ----------------------------------------------------------------------------
// The generation-2 loop (one score tile live, 3-4 waves per SIMD: every wait is hidden by another wave) with the VALU
// diet the round-3 counters ask for (profiles/r3_pmc_flash_variants.log: the SIMD issues from ONE wave at a time, the sum
// of the waves' issue cycles IS the kernel time, ~4.5 cycles per VALU instruction, ~9 per v_exp / v_permlane32_swap):
//   * the running max is subtracted by the first QK^T MFMA (C = the lane's -max block, Q pre-scaled by scale * log2 e):
//     p = exp2(s') is one instruction, no fma per score;
//   * the max is raised only past 2^FA3_THR (see generation 3); the first tile takes its own maximum;
//   * V^T in the accumulator key order (PERM): no v_permlane32_swap;
//   * row sums as 16 v_pk_add_f32 instead of 32 v_add_f32.
// 134 registers: three 4-wave workgroups per CU.

// SUMM (round 4): the row sums come off the matrix pipe - a third P V MFMA per 16 keys against a fragment of ones (every
// element of its accumulator is the lane's query's sum over the wave tile's keys) replaces the 17 v_pk_add_f32 per tile:
// the kernel is VALU-issue-bound (~750 issue cycles per tile and wave beside 512 MFMA cycles), the pipe has the room.
template <int NW, bool PERM, int SUMM = 0>   // SUMM: 0 packed adds, 1 matrix pipe, 2 plain v_add_f32 (two chains)
__global__ __launch_bounds__(NW * 64) void flash_attn64_v25_kernel(const FaArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[FA2_NSTAGE * FA_STAGE];
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int nqb = (a.Ntok + NW * 32 - 1) / (NW * 32);
  fa25_body<NW, PERM, SUMM>(a, smem, bid % nqb, bid / nqb);
}

// ---- generation 3 ------------------------------------------------------------------------------
// What the ISA and per-phase cycle stamps of generation 2 showed (round 3, profiles/r3_flash_*.log):
//   * every wave ran its tile as three serial phases - QK^T MFMAs (each behind its own ds_read + lgkmcnt(0)), ~150 VALU
//     instructions of softmax with the matrix pipe idle, PV MFMAs (again one exposed ds_read each);
//   * the kernel is bound by the SIMD's VALU issue, not by the matrix pipe: per wave and 64-key tile 32 v_exp_f32 (8.8
//     cycles each: quarter rate), 32 fma, 32 adds, 16 v_cvt_pk, 8 v_permlane32_swap (9.5 cycles each), 16 v_max3, 16 LDS
//     reads ~ 1000 issue cycles against 512 MFMA cycles, and the two waves of a SIMD do not overlap their VALU streams
//     (tools/ubench/valu_cost.hip for the per-instruction prices).
// Generation 3 therefore removes VALU work and hand-places what is left around the MFMAs:
//   * two-stage software pipeline inside the wave: phase A = QK^T of tile t+1 (8 MFMAs) beside the exponentials of tile t,
//     phase B = PV of tile t beside the row max of tile t+1 and the P -> bf16 conversion;
//   * the running max is SUBTRACTED BY THE MFMA: Q is pre-scaled by scale * log2 e and the first MFMA of a score block
//     takes C = -m (a 16-register block holding the lane's -max, rewritten only when the max moves): a score leaves the
//     matrix pipe as s' = c q.k - c m and p = exp2(s') is ONE instruction (generation 2: fma + exp);
//   * the max moves rarely: it is raised only when a row's new maximum exceeds the running one by more than 2^FA3_THR
//     (then O, l, the -m block and the pending s' are rescaled - a wave-uniform slow path); probabilities are <= 2^FA3_THR
//     instead of <= 1, which changes nothing for bf16 P / fp32 l, O (same relative rounding);
//   * PERM: V^T arrives with its keys permuted inside every group of 16 ([0-3, 8-11, 4-7, 12-15] - the order in which
//     the QK^T accumulators hold them; igemm2's transposed epilogue writes that order for free by NOT regrouping), so the
//     packed probabilities are the PV operand as they are: no v_permlane32_swap;
//   * SUM = 1: the row sums are a fifth MFMA per k-step against an all-ones operand (the matrix pipe has the slack, the
//     VALU does not): l sits in every row of a 32x32 accumulator, already summed over both lane halves;
//   * fragment reads hidden from hipcc's waitcnt pass (which waits lgkmcnt(0) for the first one, i.e. for all sixteen):
//     uncounted asm ds_read_b128 + one counted lgkmcnt per MFMA (LDS returns in order).
// K and V^T tiles have their own 4-deep LDS-DMA rings (tile t+1's K is needed one phase earlier than its V^T): per tile one
// counted vmcnt + one barrier, two DMA pieces issued per wave (dummy pieces from the zero page past the end keep the
// count uniform).  The tile count is rounded up to a multiple of 4 (ring depth x score-buffer parity); surplus tiles are
// fully masked.
constexpr int FA3_NS = 4;
constexpr int FA3_TILE = FA_KB * 128;   // bytes of a K tile [64 keys][64 d] / V^T tile [64 d][64 keys]

// A fragment read hipcc does not count (cdna_hip_programming.md 5.7, form iii)
template <int OFF>
__device__ __forceinline__ bf16x8 fa3_ds_read(uint32_t addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void fa3_lgk_wait() {   // <= N of this wave's LDS reads outstanding; nothing moves across
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int NW, bool PERM, int SUM>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void flash_attn64_v3_kernel(const FaArgs a) {
  static_assert(NW == 8 || NW == 4, "8 or 4 waves");
  const unsigned long long dbg_c0 = a.dbg ? __builtin_amdgcn_s_memtime() : 0ull, dbg_r0 = a.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
  constexpr int NT = NW * 64;
  constexpr int ITS = 512 / NT;   // 16-byte DMA pieces per thread per K (and per V^T) tile
  constexpr int QB = NW * 32;
  __shared__ __attribute__((aligned(16))) char smem[2 * FA3_NS * FA3_TILE];
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

  // Q fragments (MFMA B operand: column = query, k = d), pre-scaled by c = scale * log2 e
  const int q_row = qb * QB + wave * 32 + l31;
  const int q_ld = q_row < a.Ntok ? q_row : a.Ntok - 1;
  bf16x8 qf[4];
  {
    const float c = a.scale_log2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint4 u = *(const uint4*)(Qb + (long long)q_ld * a.ldq + ks * 16 + half * 8);
      uint4 w;
      w.x = fa_cvt_pk(bflo(u.x) * c, bfhi(u.x) * c); w.y = fa_cvt_pk(bflo(u.y) * c, bfhi(u.y) * c);
      w.z = fa_cvt_pk(bflo(u.z) * c, bfhi(u.z) * c); w.w = fa_cvt_pk(bflo(u.w) * c, bfhi(u.w) * c);
      qf[ks] = __builtin_bit_cast(bf16x8, w);
    }
  }

  // staging: ITS 16-byte chunks of the K tile and of the V^T tile per thread and tile
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
  const long long k_step = (long long)FA_KB * a.ldq * 2;
  int i_kk = 0;      // first key of the next K tile to issue
  int i_vk = 0;      // first key column of the next V^T tile to issue
  auto issue_k = [&](int stage) {
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      glds16(i_kk + k_row[it] < a.Ntok ? k_src[it] : zero, smem + stage * FA3_TILE + (it * NT + wave * 64) * 16);
      k_src[it] += k_step;
    }
    i_kk += FA_KB;
  };
  auto issue_v = [&](int stage) {
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      glds16(i_vk < a.ldvt ? v_src[it] : zero, smem + (FA3_NS + stage) * FA3_TILE + (it * NT + wave * 64) * 16);
      v_src[it] += FA_KB * 2;
    }
    i_vk += FA_KB;
  };

  // fragment addresses (LDS bytes): K rows (keys) t2*32 + l31, chunk (2 ks + half) ^ swizzle; V^T rows (d) dt*32 + l31,
  // chunk (4 t2 + 2 sh + half) ^ swizzle - the same four patterns; the swizzle (row >> 1) & 7 only depends on l31
  const int sw = (l31 >> 1) & 7;
  uint32_t fa_[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa_[i] = (uint32_t)(uintptr_t)(LDS_AS char*)smem + (uint32_t)(l31 * 128 + (((i * 2 + half) ^ sw) << 4));

  f32x16 o[2], sA[2], sB[2], negm, lacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; lacc[r] = 0.f; }
  float l_run = 0.f;
  const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));

  const int nkt = (a.Ntok + FA_KB - 1) / FA_KB;
  const int nkt4 = (nkt + 3) & ~3;

  auto mask_scores = [&](int kbase, f32x16 (&s)[2]) {    // keys >= Ntok contribute exp2(-1e30) = 0
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kbase + t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (key >= a.Ntok) s[t2][r] = -1e30f;
      }
  };
  // (chain form on purpose: hipcc folds fmaxf(fmaxf(m, x), y) into one v_max3_f32, a pairwise tree costs three
  // instructions per two values - it canonicalises every MFMA output with v_max x, x first)
  auto row_max = [&](const f32x16 (&s)[2]) {
    float m0 = -1e30f, m1 = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) { m0 = fmaxf(fmaxf(m0, s[0][r]), s[0][r + 1]); m1 = fmaxf(fmaxf(m1, s[1][r]), s[1][r + 1]); }
    float mx = fmaxf(m0, m1), a0, a1;
    half_swap(mx, mx, a0, a1);
    return fmaxf(a0, a1);
  };
  // raise the running max of every lane by d >= 0 (log2 units): O, l, the -m block and the pending scores follow
  auto rescale = [&](float d, f32x16 (&s)[2]) {
    const float alpha = __builtin_amdgcn_exp2f(-d);
    l_run *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o[0][r] *= alpha; o[1][r] *= alpha;
      if constexpr (SUM == 1) lacc[r] *= alpha;
      negm[r] -= d;
      s[0][r] -= d; s[1][r] -= d;
    }
  };

  // ---- prologue: three tiles of K / V^T and the fourth K tile in flight, scores of tile 0 ----
  issue_k(0); issue_v(0); issue_k(1); issue_v(1); issue_k(2); issue_v(2); issue_k(3);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * ITS) : "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 kf = __builtin_bit_cast(bf16x8, *(const uint4*)(smem + t2 * 4096 + l31 * 128 + (((ks * 2 + half) ^ sw) << 4)));
      sA[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? negm : sA[t2], 0, 0, 0);
    }
  if (FA_KB > a.Ntok) mask_scores(0, sA);
  {
    const float mx = row_max(sA);   // first tile: the running max IS the tile's max (whatever its sign)
#pragma unroll
    for (int r = 0; r < 16; ++r) { negm[r] = -mx; sA[0][r] -= mx; sA[1][r] -= mx; }
  }

  unsigned long long dbg_bar = 0, dbg_pa = 0, dbg_pb = 0, dbg_resc = 0;
  // One tile: scores of tile t are in `sc` (relative to the running max), the scores of tile t+1 go to `sn`.
  // J = t % 4 (ring stages are literals), LAST: no tile t+1.
  auto tile = [&](int t, auto j_tag, auto last_tag, f32x16 (&sc)[2], f32x16 (&sn)[2]) {
    constexpr int J = decltype(j_tag)::value;
    constexpr bool LAST = decltype(last_tag)::value;
    const unsigned long long ts0 = a.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * ITS) : "memory");   // K tile t+1 and V^T tile t of this wave have landed
    __builtin_amdgcn_s_barrier();                       // ... of every wave; every wave is done with tile t-1's stages
    const unsigned long long ts1 = a.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long ts2 = ts1;
    issue_k(J);                  // K tile t+4 -> the stage K tile t was read from (phase A of tile t-1)
    issue_v((J + 3) & 3);        // V^T tile t+3 -> the stage of V^T tile t-1
    constexpr int KO = ((J + 1) & 3) * FA3_TILE, VO = (FA3_NS + J) * FA3_TILE;
    bf16x8 kf[8], vf[8];
    if constexpr (!LAST) {
      kf[0] = fa3_ds_read<KO>(fa_[0]); kf[1] = fa3_ds_read<KO>(fa_[1]); kf[2] = fa3_ds_read<KO>(fa_[2]); kf[3] = fa3_ds_read<KO>(fa_[3]);
      kf[4] = fa3_ds_read<KO + 4096>(fa_[0]); kf[5] = fa3_ds_read<KO + 4096>(fa_[1]);
      kf[6] = fa3_ds_read<KO + 4096>(fa_[2]); kf[7] = fa3_ds_read<KO + 4096>(fa_[3]);
    }
    vf[0] = fa3_ds_read<VO>(fa_[0]); vf[1] = fa3_ds_read<VO + 4096>(fa_[0]); vf[2] = fa3_ds_read<VO>(fa_[1]); vf[3] = fa3_ds_read<VO + 4096>(fa_[1]);
    vf[4] = fa3_ds_read<VO>(fa_[2]); vf[5] = fa3_ds_read<VO + 4096>(fa_[2]); vf[6] = fa3_ds_read<VO>(fa_[3]); vf[7] = fa3_ds_read<VO + 4096>(fa_[3]);
    __builtin_amdgcn_sched_barrier(0);
    float ps[4];
    auto exp4 = [&](int e0) {   // scores e0 .. e0 + 3 of the 32 (+ their share of the row sums where the VALU keeps them)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = e0 + j;
        const float p = __builtin_amdgcn_exp2f(sc[idx >> 4][idx & 15]);
        sc[idx >> 4][idx & 15] = p;
        if constexpr (SUM == 0) ps[j] = e0 == 0 ? p : ps[j] + p;
      }
    };
    auto qk = [&](int g) {
      sn[g >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[g], qf[g & 3], (g & 3) == 0 ? negm : sn[g >> 2], 0, 0, 0);
    };
    bf16x8 pf;
    auto make_p = [&](int i) {   // P fragment of k-step i = 2 t2 + sh: keys 32 t2 + 16 sh + [0, 16)
      const int t2 = i >> 1, sh = i & 1;
      const uint32_t a0 = fa_cvt_pk(sc[t2][8 * sh + 0], sc[t2][8 * sh + 1]);   // keys 16 sh + 4 half + {0, 1}
      const uint32_t a1 = fa_cvt_pk(sc[t2][8 * sh + 2], sc[t2][8 * sh + 3]);   //                    + {2, 3}
      const uint32_t b0 = fa_cvt_pk(sc[t2][8 * sh + 4], sc[t2][8 * sh + 5]);   // keys 16 sh + 8 + 4 half + {0, 1}
      const uint32_t b1 = fa_cvt_pk(sc[t2][8 * sh + 6], sc[t2][8 * sh + 7]);
      if constexpr (PERM) {   // V^T holds its keys in exactly this order
        pf = __builtin_bit_cast(bf16x8, make_uint4(a0, a1, b0, b1));
      } else {                // 8 consecutive keys 16 sh + 8 half + [0, 8) per lane
        const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        pf = __builtin_bit_cast(bf16x8, make_uint4(r0[0], r1[0], r0[1], r1[1]));
      }
    };
    float m0 = -1e30f, m1 = -1e30f;
    auto max8 = [&](int i0) {   // scores i0 .. i0 + 7 of tile t+1
      if constexpr (!LAST) {
        m0 = fmaxf(fmaxf(m0, sn[i0 >> 4][i0 & 15]), sn[i0 >> 4][(i0 & 15) + 1]);
        m1 = fmaxf(fmaxf(m1, sn[i0 >> 4][(i0 & 15) + 2]), sn[i0 >> 4][(i0 & 15) + 3]);
        m0 = fmaxf(fmaxf(m0, sn[i0 >> 4][(i0 & 15) + 4]), sn[i0 >> 4][(i0 & 15) + 5]);
        m1 = fmaxf(fmaxf(m1, sn[i0 >> 4][(i0 & 15) + 6]), sn[i0 >> 4][(i0 & 15) + 7]);
      }
    };
    auto pv = [&](int i) {
      o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2 * i], pf, o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2 * i + 1], pf, o[1], 0, 0, 0);
      if constexpr (SUM == 1) lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lacc, 0, 0, 0);
    };
    // ---- phase A: S(t+1) = K Q^T beside p = exp2(s') of tile t ----
    if constexpr (!LAST) {
      exp4(0); exp4(4);
      fa3_lgk_wait<15>(); qk(0); exp4(8);
      fa3_lgk_wait<14>(); qk(1); exp4(12);
      fa3_lgk_wait<13>(); qk(2); exp4(16);
      fa3_lgk_wait<12>(); qk(3); exp4(20);
      fa3_lgk_wait<11>(); qk(4); exp4(24);
      fa3_lgk_wait<10>(); qk(5); exp4(28);
      fa3_lgk_wait<9>(); qk(6); make_p(0);
      fa3_lgk_wait<8>(); qk(7);
    } else {
#pragma unroll
      for (int e0 = 0; e0 < 32; e0 += 4) exp4(e0);
      make_p(0);
    }
    if constexpr (SUM == 0) l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
    if (a.dbg) ts2 = __builtin_amdgcn_s_memtime();
    // ---- phase B: O^T += V^T P^T of tile t beside the row max of tile t+1 ----
    fa3_lgk_wait<6>(); pv(0); make_p(1); max8(0);
    fa3_lgk_wait<4>(); pv(1); make_p(2); max8(8);
    fa3_lgk_wait<2>(); pv(2); make_p(3); max8(16);
    fa3_lgk_wait<0>(); pv(3); max8(24);
    if constexpr (!LAST) {
      float mx = fmaxf(m0, m1), x0, x1;
      half_swap(mx, mx, x0, x1);
      mx = fmaxf(x0, x1);
      // ragged / surplus tile t+1 (wave-uniform, only near the end; behind phase B so that the branch does not cut the
      // MFMA / VALU interleave): keys >= Ntok leave the row max and get p = 0
      if ((t + 2) * FA_KB > a.Ntok) {
        mask_scores((t + 1) * FA_KB, sn);
        mx = row_max(sn);
      }
      if (__any(mx > FA3_THR)) { rescale(fmaxf(mx, 0.f), sn); ++dbg_resc; }
    }
    if (a.dbg) {
      const unsigned long long ts3 = __builtin_amdgcn_s_memtime();
      dbg_bar += ts1 - ts0; dbg_pa += ts2 - ts1; dbg_pb += ts3 - ts2;
    }
  };
  {
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    int t = 0;
    for (; t + 4 < nkt4; t += 4) {
      tile(t, I0{}, std::false_type{}, sA, sB);
      tile(t + 1, I1{}, std::false_type{}, sB, sA);
      tile(t + 2, I2{}, std::false_type{}, sA, sB);
      tile(t + 3, I3{}, std::false_type{}, sB, sA);
    }
    tile(t, I0{}, std::false_type{}, sA, sB);
    tile(t + 1, I1{}, std::false_type{}, sB, sA);
    tile(t + 2, I2{}, std::false_type{}, sA, sB);
    tile(t + 3, I3{}, std::true_type{}, sB, sA);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy pieces of the last tiles (nothing reads them)
  if (a.dbg && lane == 0) {   // [workgroup][wave][8]: total cycles, 100 MHz ticks, barrier wait, phase A, phase B, rescales
    unsigned long long* d = a.dbg + ((long long)blockIdx.x * NW + wave) * 8;
    d[0] = __builtin_amdgcn_s_memtime() - dbg_c0;
    d[1] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
    d[2] = dbg_bar; d[3] = dbg_pa; d[4] = dbg_pb; d[5] = dbg_resc;
  }
  // ---- finalize: O[q][d] = o^T / l; lane^32 exchange -> 8 consecutive d per lane, 16-byte stores ----
  float l_tot;
  if constexpr (SUM == 1) {
    l_tot = lacc[0];   // every row of the all-ones product holds the sum over all keys of the lane's query
  } else {
    float l0, l1;
    half_swap(l_run, l_run, l0, l1);
    l_tot = l0 + l1;
  }
  const float inv = 1.0f / l_tot;
  bf16_t* orow = a.O + (long long)b * a.sO + (long long)q_row * a.ldo + h * 64;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) half_swap(o[dt][8 * gp + j] * inv, o[dt][8 * gp + 4 + j] * inv, v[j], v[4 + j]);
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
      a.dbg = (unsigned long long*)op->p[4];
      a.redo_thr = op->f[1] > 0.f ? op->f[1] : 1.2676506e30f;   // 2^100 (tests force the fallback with a tiny value)
      a.ws = op->p[5];                                            // variant 26: workspace of the key-split blocks (i[8] KB)
      a.ws_bytes = (long long)op->i[8] * 1024;
      a.split = op->i[9];
      a.n_full = a.n_rem = a.n_rem_wg = 0;
      MG_REQUIRE(!a.ws || ((uintptr_t)a.ws % 16 == 0 && a.ws_bytes > 0), "flash_attn64: workspace must be 16-byte aligned, its size (KB) in i[8]");
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
      const bool vt_perm = op->i[7] != 0;   // V^T keys permuted inside every group of 16: [0-3, 8-11, 4-7, 12-15]
      MG_REQUIRE(vt_perm == ((var >= 13 && var <= 20) || var == 22 || var == 23 || var == 25 || var == 26) || var == 0, "flash_attn64: variant %d and the V^T key order (i[7] = %d) do not match", var, op->i[7]);
      MG_REQUIRE(!vt_perm || a.Ntok % 16 == 0, "flash_attn64: the permuted V^T layout needs Ntok %% 16 == 0");
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
        case 19: MG_LAUNCH((flash_attn64_v25_kernel<4, true>), dim3((unsigned)g4), dim3(256), 0, s, a); break;   // generation 2.5, vt_perm
        case 20: MG_LAUNCH((flash_attn64_v25_kernel<8, true>), dim3((unsigned)g8), dim3(512), 0, s, a); break;
        case 21: MG_LAUNCH((flash_attn64_v25_kernel<4, false>), dim3((unsigned)g4), dim3(256), 0, s, a); break;  // natural V^T
        case 22: MG_LAUNCH((flash_attn64_v25_kernel<4, true, 1>), dim3((unsigned)g4), dim3(256), 0, s, a); break;   // + row sums on the matrix pipe
        case 23: MG_LAUNCH((flash_attn64_v25_kernel<8, true, 1>), dim3((unsigned)g8), dim3(512), 0, s, a); break;
        case 24: MG_LAUNCH((flash_attn64_v25_kernel<4, false, 1>), dim3((unsigned)g4), dim3(256), 0, s, a); break;
        case 25: MG_LAUNCH((flash_attn64_v25_kernel<4, true, 2>), dim3((unsigned)g4), dim3(256), 0, s, a); break;   // plain v_add_f32 row sums
        case 26:   // the hand-placed one-wave-per-SIMD stream (flash4w.hip)
          MG_REQUIRE(mg_flash4w_ok(a, vt_perm), "flash_attn64 variant 26: Ntok %d must be a multiple of 256 (even number of key tiles), V^T permuted", a.Ntok);
          return mg_launch_flash4w(a, s);
        case 17: MG_LAUNCH((flash_attn64_v2_kernel<true, 4, 3>), dim3((unsigned)g4), dim3(256), 0, s, a); break;   // vt_perm
        case 18: MG_LAUNCH((flash_attn64_v2_kernel<true, 8, 3>), dim3((unsigned)g8), dim3(512), 0, s, a); break;
        // generation 3: 9 / 10 = 8 / 4 waves, natural V^T; 11 / 12 = + row sums on the matrix pipe; 13-16 = the same four with
        // V^T's keys permuted inside groups of 16 (i[7] = 1 required: the producer wrote that order)
        case 9: MG_LAUNCH((flash_attn64_v3_kernel<8, false, 0>), dim3((unsigned)g8), dim3(512), 0, s, a); break;
        case 10: MG_LAUNCH((flash_attn64_v3_kernel<4, false, 0>), dim3((unsigned)g4), dim3(256), 0, s, a); break;
        case 11: MG_LAUNCH((flash_attn64_v3_kernel<8, false, 1>), dim3((unsigned)g8), dim3(512), 0, s, a); break;
        case 12: MG_LAUNCH((flash_attn64_v3_kernel<4, false, 1>), dim3((unsigned)g4), dim3(256), 0, s, a); break;
        case 13: MG_LAUNCH((flash_attn64_v3_kernel<8, true, 0>), dim3((unsigned)g8), dim3(512), 0, s, a); break;
        case 14: MG_LAUNCH((flash_attn64_v3_kernel<4, true, 0>), dim3((unsigned)g4), dim3(256), 0, s, a); break;
        case 15: MG_LAUNCH((flash_attn64_v3_kernel<8, true, 1>), dim3((unsigned)g8), dim3(512), 0, s, a); break;
        case 16: MG_LAUNCH((flash_attn64_v3_kernel<4, true, 1>), dim3((unsigned)g4), dim3(256), 0, s, a); break;
        default:
          // round 3 (profiles/r3_flash_variants*.log, TFLOP/s at E = 10): generation 2.5 on 4-wave workgroups (three per CU)
          // is ahead at every sequence length - 9216 tokens 903 (generation 3 860-890, generation 2 827-853), 2304: 798
          // (782 / 748), 576: 562 (538 / 403), 144: 119 (122 / 109) - with either V^T order
          // round 4: the row sums as plain v_add_f32 on two chains instead of v_pk_add_f32 (a packed fp32 add beside MFMAs costs
          // more than the two adds it replaces, MI355X_MICROARCH.md): 9216 tokens 975 vs 947 TFLOP/s, 2304: 795 vs 743
          // (the sums on the matrix pipe, SUMM = 1: 945 / 828) - profiles/r4_flash_rowsum_mfma.log
          // round 4: the hand-placed one-wave-per-SIMD stream (variant 26, flash4w.hip) where its shape constraints hold
          // (the 96 x 96 and 48 x 48 levels: 9 216 / 2 304 tokens): 1 022-1 067 vs 907-928 TFLOP/s at 9 216 tokens, 893 vs 782
          // at 2 304 (profiles/r4_flash4w.log).  MARIGOLD_FLASH4W=0 switches it off (A/B).
          static const int f4w = [] { const char* e = getenv("MARIGOLD_FLASH4W"); return e ? atoi(e) : 1; }();
          if (f4w && vt_perm && mg_flash4w_ok(a, true)) return mg_launch_flash4w(a, s);
          if (vt_perm) MG_LAUNCH((flash_attn64_v25_kernel<4, true, 2>), dim3((unsigned)g4), dim3(256), 0, s, a);
          else MG_LAUNCH((flash_attn64_v25_kernel<4, false>), dim3((unsigned)g4), dim3(256), 0, s, a);
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
