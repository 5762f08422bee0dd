// The generation-2.5 flash attention loop at head width 64 (online softmax with a running maximum, hipcc-scheduled) as a device
// function over one block of NW x 32 queries: the body of flash_attn64_v25_kernel (attention.hip, where its design is
// described) and the exact fallback of flash_attn64_4w_kernel (flash4w.hip) for a workgroup whose fixed reference was too low.
// `smem`: FA2_NSTAGE * FA_STAGE bytes of LDS.
#pragma once
#include <type_traits>

#include "flash_args.h"

template <int NW, bool PERM, int SUMM>   // SUMM: 0 packed adds, 1 matrix pipe, 2 plain v_add_f32 (two chains)
__device__ __forceinline__ void fa25_body(const FaArgs& a, char* smem, const int qb, const int bh) {
  constexpr int NT = NW * 64;
  constexpr int QB = NW * 32;
  constexpr int ITS = 512 / NT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  const int h = bh % a.heads, b = bh / a.heads;

  const bf16_t* Qb = a.Q + (long long)b * a.sQ + h * 64;
  const bf16_t* Kb = a.K + (long long)b * a.sK + h * 64;
  const bf16_t* Vb = a.Vt + (long long)b * a.sVt + (long long)h * 64 * a.ldvt;
  const char* zero = (const char*)a.zero;

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
  int i_k0 = 0;
  // only the last key tile can be ragged: every other tile takes its source as it is (no per-lane select)
  auto issue = [&](int stage, bool ragged) {
    char* sb = smem + stage * FA_STAGE;
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      const char* src = (!ragged || i_k0 + k_row[it] < a.Ntok) ? k_src[it] : zero;
      glds16(src, sb + (it * NT + wave * 64) * 16);
      k_src[it] += k_step;
    }
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      glds16(v_src[it], sb + FA_KB * 128 + (it * NT + wave * 64) * 16);
      v_src[it] += FA_KB * 2;
    }
    i_k0 += FA_KB;
  };

  f32x16 o[2], negm, osum;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; osum[r] = 0.f; }
  fa_f32x2 l2 = {0.f, 0.f};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(MG_OP16_ONE_X2, MG_OP16_ONE_X2, MG_OP16_ONE_X2, MG_OP16_ONE_X2));

  const int nkt = (a.Ntok + FA_KB - 1) / FA_KB;
  const bool ragged_end = (a.Ntok & (FA_KB - 1)) != 0;
  issue(0, nkt == 1 && ragged_end);
  if (nkt > 1) issue(1, nkt == 2 && ragged_end);
  auto tile = [&](int kt, int st_c, int st_i, auto issue_tag, auto mask_tag, auto first_tag) {
    constexpr bool do_issue = decltype(issue_tag)::value;
    constexpr bool MASK = decltype(mask_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    if (do_issue || kt + 1 < nkt) {
      if constexpr (ITS == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (do_issue) issue(st_i, kt + 3 == nkt && ragged_end);
    const char* sK = smem + st_c * FA_STAGE;
    const char* sV = sK + FA_KB * 128;
    f32x16 s[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const int row = t2 * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int q = ks * 2 + half;
        const bf16x8 kf = __builtin_bit_cast(bf16x8, *(const uint4*)(sK + row * 128 + ((q ^ ((row >> 1) & 7)) << 4)));
        s[t2] = mg_mfma32(kf, qf[ks], ks == 0 ? negm : s[t2]);
      }
    }
    const int kbase = kt * FA_KB;
    if (MASK && kbase + FA_KB > a.Ntok) {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (key >= a.Ntok) s[t2][r] = -1e30f;
        }
    }
    float m0 = -1e30f, m1 = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) { m0 = fmaxf(fmaxf(m0, s[0][r]), s[0][r + 1]); m1 = fmaxf(fmaxf(m1, s[1][r]), s[1][r + 1]); }
    float mx = fmaxf(m0, m1);
    {
      float x0, x1;
      half_swap(mx, mx, x0, x1);
      mx = fmaxf(x0, x1);
    }
    if (FIRST || __any(mx > FA3_THR)) {   // raise the running max (first tile: set it, whatever its sign)
      const float d = FIRST ? mx : fmaxf(mx, 0.f);
      const float alpha = __builtin_amdgcn_exp2f(-d);
      if constexpr (!FIRST) {
        if constexpr (SUMM == 1) osum[0] *= alpha;   // (only element 0 is read at the end; the others run on unscaled, unused)
        else l2 *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) { negm[r] -= d; s[0][r] -= d; s[1][r] -= d; }
    }
    fa_f32x2 ps = {0.f, 0.f};
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        fa_f32x2 p2;
        p2.x = __builtin_amdgcn_exp2f(s[t2][r]);
        p2.y = __builtin_amdgcn_exp2f(s[t2][r + 1]);
        s[t2][r] = p2.x;
        s[t2][r + 1] = p2.y;
        if constexpr (SUMM == 0) ps += p2;   // v_pk_add_f32
        if constexpr (SUMM == 2) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(ps.x) : "v"(p2.x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(ps.y) : "v"(p2.y)); }
      }
    if constexpr (SUMM != 1) l2 += ps;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
      for (int sh = 0; sh < 2; ++sh) {
        const uint32_t a0 = fa_cvt_pk(s[t2][8 * sh + 0], s[t2][8 * sh + 1]);
        const uint32_t a1 = fa_cvt_pk(s[t2][8 * sh + 2], s[t2][8 * sh + 3]);
        const uint32_t b0 = fa_cvt_pk(s[t2][8 * sh + 4], s[t2][8 * sh + 5]);
        const uint32_t b1 = fa_cvt_pk(s[t2][8 * sh + 6], s[t2][8 * sh + 7]);
        uint4 pw;
        if constexpr (PERM) {
          pw = make_uint4(a0, a1, b0, b1);
        } else {
          const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
          pw = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        }
        const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int row = dt * 32 + l31;
          const int c0 = 4 * t2 + 2 * sh + half;
          const uint4 vw = *(const uint4*)(sV + row * 128 + ((c0 ^ ((row >> 1) & 7)) << 4));
          o[dt] = mg_mfma32(__builtin_bit_cast(bf16x8, vw), pf, o[dt]);
        }
        if constexpr (SUMM == 1) osum = mg_mfma32(ones, pf, osum);
      }
    }
  };
  {
    using T = std::true_type;
    using F = std::false_type;
    int kt = 0, st_c = 0, st_i = 2;
    auto rot = [&]() {
      st_c = (st_c + 1 == FA2_NSTAGE) ? 0 : st_c + 1;
      st_i = (st_i + 1 == FA2_NSTAGE) ? 0 : st_i + 1;
    };
    if (nkt > 2) tile(0, 0, 2, T{}, F{}, T{});
    else tile(0, 0, 2, F{}, T{}, T{});
    kt = 1; rot();
    for (; kt + 5 <= nkt; kt += 3) {   // literal ring stages in the steady state: kt = 1 (mod 3) here
      tile(kt, 1, 0, T{}, F{}, F{});
      tile(kt + 1, 2, 1, T{}, F{}, F{});
      tile(kt + 2, 0, 2, T{}, F{}, F{});
    }
    for (; kt + 2 < nkt; ++kt) { tile(kt, st_c, st_i, T{}, F{}, F{}); rot(); }
    for (; kt < nkt; ++kt) { tile(kt, st_c, st_i, F{}, T{}, F{}); rot(); }
  }
  float inv;
  if constexpr (SUMM == 1) {
    inv = 1.0f / osum[0];   // the MFMA summed over both key halves already
  } else {
    float l0, l1;
    const float l_lane = l2.x + l2.y;
    half_swap(l_lane, l_lane, l0, l1);
    inv = 1.0f / (l0 + l1);
  }
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
