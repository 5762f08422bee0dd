// Flash attention at head width 64, hand-placed (MG_OP_FLASH_ATTN64 variant 26): the UNet's self-attention at 9 216 and
// 2 304 tokens (marigold/marigold_depth_pipeline.py:461-463 -> diffusers Attention).
//
// flash_attn64_v25 (attention.hip) leaves hipcc to order the tile's work and relies on three waves per SIMD to overlap the
// softmax's VALU stream with the MFMAs.  On gfx950 that overlap is only partial (tools/ubench/coissue2.hip: three waves that
// alternate whole phases reach 1.44 x one wave, an instruction-by-instruction interleave in ONE wave 1.57 x), and the compiled
// loop waits lgkmcnt(0) in front of every MFMA.  Here:
//   * 64 queries per wave = two 32-query blocks that share every K / V^T fragment read from LDS (half the LDS traffic per MFMA:
//     at 32 queries per wave the fragment reads alone need the CU's whole 128 B / cycle), four waves per workgroup, TWO
//     workgroups per CU (256 registers per lane, all of them VGPRs: hipcc splits the file 128 / 128 as soon as a function
//     touches AGPRs, so the output accumulators stay in VGPRs and the file is built in the VGPR MFMA form) - one wave alone is bound
//     by its own instruction issue (~1 600 cycles per tile against 1 024 of MFMA time), a second one fills its gaps;
//   * the key loop is ONE asm statement written by gen_fa4w.py (flash4w.inc, where the schedule is described): per 64-key tile
//     32 MFMAs with five VALU instructions of the softmax behind each, the QK^T MFMAs of the next 32 keys and the P V MFMAs of
//     the current ones in the same stream, fragments held in registers for both query blocks and reloaded with counted waits,
//     the LDS-DMA pieces of ring slot t + 3 among them, one barrier per tile;
//   * softmax against a FIXED per-query reference (the row maximum over the first 32 keys, subtracted by the first QK^T MFMA's
//     C operand): no running maximum, no rescaling - 16 v_max3 and the branch per tile are gone and the output accumulators
//     are only ever touched by MFMAs.  Exact as long as no later score tops its reference by more than ~2^100; the row sums
//     tell (>= 2^100 or not finite) and such a workgroup redoes its queries with the running-maximum loop of flash_attn64_v25
//     (flash25_body.h; FaArgs::redo_thr lets the tests force that path).
// Round 6 - the same loop on v_mfma_f32_16x16x32 (template M16, variant 27; stream FA4W16_ASM, layout notes in gen_fa4w.py): the chip
// is power-limited under this kernel (1.6 GHz), and a 16x16x32 MFMA draws less than a 32x32x16 one for the same product
// (profiles/r6_mfma_shape.log).  Four 16-query blocks per wave, the K fragment rows read in the order that makes a lane's eight
// packed probabilities the B operand of its P V MFMA against the permuted V^T, the V^T rows in the order that lets eight
// v_permlane16_swap per 32 x 32 block rebuild the 32x32x16 accumulator layout for the unchanged finalize, and the ROW SUMS ON THE
// MATRIX PIPE (P against ones: the stream is issue-bound, and an MFMA costs the wave ~8 issue cycles whatever its size).  Chosen
// by mg_launch_flash4w where two workgroups per CU run for several rounds at 9 216 tokens: 851 vs 911 us at E = 10.
// Ring slot t of the four 16 KB slots holds what iteration t reads: K rows 64 t + 32 ... 64 t + 95 and V^T tile t; the K rows
// of the prologue (the first 32 keys of the sequence: the reference; the first 32 keys of the segment: the stream's first
// scores) go to two 4 KB regions of their own.  Needs Ntok % 256 == 0, at least four key tiles and V^T in the accumulator's
// key order (vt_perm).
//
// Workgroups: a block of 256 queries keeps a workgroup busy for ~200 us, so a launch whose number of blocks is not a multiple of
// the chip's slots ends in a round that a few CUs run alone (E = 10 at 9 216 tokens: 1 800 blocks on 512 slots = 3.52 rounds,
// paid as 4).  With a workspace from the caller (FaArgs::ws) the blocks beyond the last multiple of the CU count are therefore
// split along the KEYS over the slots: the fixed reference depends on the sequence's first 32 keys only - every piece of a block
// computes the same one - so the pieces' unnormalised outputs and row sums simply ADD.  Each piece leaves its partial result
// in the workspace and draws a ticket; the block's last piece adds the others to its registers, normalises and stores (no
// spinning: nothing waits for a workgroup that may not be resident).
#include "flash_args.h"
#include "flash25_body.h"
#include "flash4w.inc"

// fp16 build: probabilities are fp16 (max 65504) where bf16 reaches 2^127 - the reference sits 2^6 above the first keys' maximum
// (p = 2^(s - max32 - 6): scores may top the first 32 keys' maximum by 2^21 before the row sums reach the 2^15 that sends the
// workgroup to the running-maximum loop; fp16 subnormals keep 2^-24), every piece of a split block still computes the same reference
constexpr float F4_REF_BIAS = MG_F16 ? 6.0f : 0.0f;

// the score set Y of the stream (v[160:191]): physical registers, scratch to the compiler
#define F4_CLOBBER_Y "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", \
                     "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191"

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;
constexpr int F4_SLOT = 16384, F4_PRE = 4 * F4_SLOT, F4_LDS = F4_PRE + 2 * 4096;
constexpr int F4_PIECES = 4;                                   // at most four key pieces per block of queries
constexpr int F4_PART_BYTES = 4 * (16384 + 1024);               // one piece's partial result: per wave 64 x 64 fp32 + 16 bytes of row sums per lane
constexpr int F4_CTR_BYTES = 4096;                             // tickets (zero between launches) in front of the partial results

__device__ __forceinline__ void f4_mfma(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile(MG_MFMA32_ASM " %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
typedef __attribute__((ext_vector_type(8))) float f32x8;
__device__ __forceinline__ void f4_mfma16(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile(MG_MFMA16_ASM " %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void f4_dma(unsigned voff, i32x4 srd, unsigned soff, unsigned m0v) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(srd), "s"(soff), "s"(m0v) : "memory");
}
__device__ __forceinline__ unsigned f4_sgpr(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

// Piece boundaries of the split blocks, in key tiles over the concatenated blocks: workgroup r of `nwg` takes [f4_bound(r),
// f4_bound(r + 1)).  A boundary closer than four tiles to a block's edge moves onto it (the stream needs four tiles).
// (32-bit: the launcher keeps r total below 2^31)
__device__ __host__ __forceinline__ unsigned f4_bound(unsigned r, unsigned nwg, unsigned total, unsigned nkt) {
  unsigned s = (r * total + nwg / 2) / nwg;
  const unsigned m = s % nkt;
  if (m > 0 && m < 4) s -= m;
  else if (m + 4 > nkt) s += nkt - m;
  return s;
}

// One segment: key tiles [t0, t1) of the 256 queries of block `blk`.  `part` < 0: the whole block (normalise and store);
// else: piece `part` of `npieces` of split block `rem` (partial result + ticket, the last piece finishes).
template <bool M16>
__device__ __forceinline__ void f4_segment(const FaArgs& a, char* smem, const int blk, const int t0, const int t1, const int part,
                                           const int npieces, const int rem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int nqb = a.Ntok >> 8;
  const int qb = blk % nqb, bh = blk / nqb;
  const int h = bh % a.heads, b = bh / a.heads;
  const bf16_t* Qb = a.Q + (long long)b * a.sQ + h * 64;
  const bf16_t* Kb = a.K + (long long)b * a.sK + h * 64;
  const bf16_t* Vb = a.Vt + (long long)b * a.sVt + (long long)h * 64 * a.ldvt;

  // LDS-DMA: 16-byte element ci = it * 256 + tid of a tile = row ci >> 3, chunk position ci & 7 (source chunk XOR-swizzled)
  unsigned vk[2], vv[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int ci = it * 256 + tid, r = ci >> 3, q = (ci & 7) ^ ((r >> 1) & 7);
    vk[it] = (unsigned)(r * a.ldq * 2 + q * 16);
    vv[it] = (unsigned)(r * a.ldvt * 2 + q * 16);
  }
  auto srd_of = [](const void* p, unsigned bytes) {   // raw buffer: offsets >= bytes read as zero
    const unsigned long long u = (unsigned long long)(uintptr_t)p;
    const i32x4 r = {(int)f4_sgpr((unsigned)u), (int)(f4_sgpr((unsigned)(u >> 32)) & 0xffffu), (int)f4_sgpr(bytes), 0x00020000};
    return r;
  };
  // (the last slot's K piece reaches 32 rows past the sequence - the next tile's half 0, never used: zeros, not a stray read)
  const i32x4 srk = srd_of(Kb, (unsigned)((a.Ntok - 1) * a.ldq * 2 + 128)), srv = srd_of(Vb, 0x80000000u);
  const unsigned kst = f4_sgpr((unsigned)(64 * a.ldq * 2));   // one key tile of K rows, bytes
  const unsigned mw = f4_sgpr((unsigned)(wave * 1024));
  const unsigned k0 = f4_sgpr((unsigned)t0 * kst), v0 = f4_sgpr((unsigned)t0 * 128u);
  unsigned long long* dbgw = a.dbg ? a.dbg + ((long long)blockIdx.x * 4 + wave) * 8 : nullptr;   // tuning only: phase time stamps
  if (dbgw && lane == 0) dbgw[2] = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_barrier();   // (a previous segment's / the fallback's readers are done with the ring)
  // K rows 0 ... 31 (the reference) and 64 t0 ... + 31 (the stream's first scores), then slots 0, 1, 2
  f4_dma(vk[0], srk, 0u, f4_sgpr(mw + F4_PRE));
  f4_dma(vk[0], srk, k0, f4_sgpr(mw + F4_PRE + 4096));
#pragma unroll
  for (int sl = 0; sl < 3; ++sl) {
#pragma unroll
    for (int it = 0; it < 2; ++it) f4_dma(vk[it], srk, f4_sgpr(k0 + (unsigned)sl * kst + (kst >> 1)), f4_sgpr(mw + sl * F4_SLOT + it * 4096));
#pragma unroll
    for (int it = 0; it < 2; ++it) f4_dma(vv[it], srv, f4_sgpr(v0 + (unsigned)(sl * 128)), f4_sgpr(mw + sl * F4_SLOT + 8192 + it * 4096));
  }
  // (the queries after the LDS-DMA pieces: their latency runs beside the DMA's; the counted waits below and in the stream only
  // ever see fewer loads in flight than they allow)
  const float qscale = a.scale_log2;
  auto load_q = [&](int q_row, int d0) {   // eight pre-scaled d values of a query row
    const uint4 u = *(const uint4*)(Qb + (long long)q_row * a.ldq + d0);
    uint4 w;
    w.x = cvt_pk_bf16_f32(bflo(u.x) * qscale, bfhi(u.x) * qscale); w.y = cvt_pk_bf16_f32(bflo(u.y) * qscale, bfhi(u.y) * qscale);
    w.z = cvt_pk_bf16_f32(bflo(u.z) * qscale, bfhi(u.z) * qscale); w.w = cvt_pk_bf16_f32(bflo(u.w) * qscale, bfhi(u.w) * qscale);
    return __builtin_bit_cast(bf16x8, w);
  };
  // 32x32x16: lane = (query l31 of a 32-query block, d group half: 8 of a k-step's 16); 16x16x32: lane = (query l15 of a 16-query
  // block, d group kq = lane >> 4: 8 of a k-step's 32)
  const int l15 = lane & 15, kq = lane >> 4;
  bf16x8 qf[2][4], qg[4][2];
  if constexpr (!M16) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) qf[q][ks] = load_q(qb * 256 + wave * 64 + q * 32 + l31, ks * 16 + half * 8);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) qg[q][ds] = load_q(qb * 256 + wave * 64 + q * 16 + l15, ds * 32 + kq * 8);
  }
  if (dbgw && lane == 0) dbgw[3] = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (dbgw && lane == 0) dbgw[4] = __builtin_amdgcn_s_memrealtime();
  f32x16 o[2][2];
  float ll[2];
  unsigned long long tm0 = 0, rm0 = 0;
  if constexpr (!M16) {
    // S^T = K Q^T (lane = query, registers = keys): the reference = the row maximum over the sequence's first 32 keys
    const int sw = (l31 >> 1) & 7;
    unsigned ad[4];
  #pragma unroll
    for (int ks = 0; ks < 4; ++ks) ad[ks] = (unsigned)(l31 * 128 + (((2 * ks + half) ^ sw) << 4));
    f32x16 s[2], negm[2];
    auto first_scores = [&](int region) {
  #pragma unroll
      for (int q = 0; q < 2; ++q)
  #pragma unroll
        for (int r = 0; r < 16; ++r) s[q][r] = 0.f;
  #pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf = __builtin_bit_cast(bf16x8, *(const uint4*)(smem + region + ad[ks]));
        f4_mfma(s[0], kf, qf[0][ks]);
        f4_mfma(s[1], kf, qf[1][ks]);
      }
      asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    };
    first_scores(F4_PRE);
    float ref[2];
  #pragma unroll
    for (int q = 0; q < 2; ++q) {
      float mx = -1e30f;
  #pragma unroll
      for (int r = 0; r < 16; r += 2) mx = fmaxf(mx, fmaxf(s[q][r], s[q][r + 1]));
      float x0, x1;
      half_swap(mx, mx, x0, x1);
      ref[q] = fmaxf(x0, x1) + F4_REF_BIAS;
    }
    if (t0 != 0) first_scores(F4_PRE + 4096);   // (wave-uniform)
  #pragma unroll
    for (int q = 0; q < 2; ++q)
  #pragma unroll
      for (int r = 0; r < 16; ++r) { negm[q][r] = -ref[q]; s[q][r] -= ref[q]; }
  #pragma unroll
    for (int q = 0; q < 2; ++q)
  #pragma unroll
      for (int d = 0; d < 2; ++d)
  #pragma unroll
        for (int r = 0; r < 16; ++r) o[q][d][r] = 0.f;
    float l00 = 0.f, l01 = 0.f, l10 = 0.f, l11 = 0.f;
    unsigned sok = f4_sgpr(k0 + 3u * kst + (kst >> 1)), sov = f4_sgpr(v0 + 3u * 128u), mb = f4_sgpr(mw + 3 * F4_SLOT),
             cnt = f4_sgpr((unsigned)(t1 - t0 - 3));
    uint4 kf0, kf1, kf2, kf3, vf0, vf1, vf2, vf3;
    if (a.dbg) { tm0 = __builtin_amdgcn_s_memtime(); rm0 = __builtin_amdgcn_s_memrealtime(); }
    asm volatile(FA4W_ASM
                 : "+{v[128:143]}"(s[0]), "+{v[144:159]}"(s[1]),
                   [o00] "+v"(o[0][0]), [o01] "+v"(o[0][1]), [o10] "+v"(o[1][0]), [o11] "+v"(o[1][1]),
                   [l00] "+v"(l00), [l01] "+v"(l01), [l10] "+v"(l10), [l11] "+v"(l11),
                   [ad0] "+v"(ad[0]), [ad1] "+v"(ad[1]), [ad2] "+v"(ad[2]), [ad3] "+v"(ad[3]),
                   [kf0] "=&v"(kf0), [kf1] "=&v"(kf1), [kf2] "=&v"(kf2), [kf3] "=&v"(kf3),
                   [vf0] "=&v"(vf0), [vf1] "=&v"(vf1), [vf2] "=&v"(vf2), [vf3] "=&v"(vf3),
                   [sok] "+s"(sok), [sov] "+s"(sov), [mb] "+s"(mb), [cnt] "+s"(cnt)
                 : [q00] "v"(qf[0][0]), [q01] "v"(qf[0][1]), [q02] "v"(qf[0][2]), [q03] "v"(qf[0][3]),
                   [q10] "v"(qf[1][0]), [q11] "v"(qf[1][1]), [q12] "v"(qf[1][2]), [q13] "v"(qf[1][3]),
                   [ng0] "v"(negm[0]), [ng1] "v"(negm[1]),
                   [vk0] "v"(vk[0]), [vk1] "v"(vk[1]), [vv0] "v"(vv[0]), [vv1] "v"(vv[1]),
                   [srk] "s"(srk), [srv] "s"(srv), [kst] "s"(kst)
                 : "memory", "scc", F4_CLOBBER_Y);
    ll[0] = l00 + l01;   // the lane's share of its queries' row sums (the other half-wave holds the rest)
    ll[1] = l10 + l11;
  } else {
    // ---- the 16x16x32 form (gen_fa4w.py, FA4W16_ASM) ----
    // K fragment (key block kb, d-step ds): lane (l15, kq) reads d chunk 4 ds + kq of key ROW krow(kb) of the 32-key half - the
    // key whose probability the lane then holds as element 4 kb + r (r = MFMA row & 3) of its eight: position 8 kq + 4 kb + r of
    // the half in the V^T order ([0-3, 8-11, 4-7, 12-15] inside every 16): rows 0-3, 4-7, 16-19, 20-23 (kb 0), + 8 (kb 1).
    // V^T fragment (d block, half hh): lane reads key chunk 4 hh + kq of d ROW pw (bits 2 <-> 3 of l15 swapped), so that eight
    // v_permlane16_swap per 32 x 32 block rebuild the 32x32x16 accumulator layout for the unchanged finalize (igemm2_body.h).
    unsigned adk[4], adv[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        const int kr = (l15 & 7) + 2 * (l15 & 8) + 8 * kb;
        adk[2 * kb + ds] = (unsigned)(kr * 128 + (((4 * ds + kq) ^ ((kr >> 1) & 7)) << 4));
      }
    const int pw = (l15 & 3) | ((l15 & 4) << 1) | ((l15 & 8) >> 1);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) adv[hh] = (unsigned)(pw * 128 + (((4 * hh + kq) ^ ((pw >> 1) & 7)) << 4));
    f32x4 sc[4][2], ng[4];
    auto first_scores = [&](int region) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) sc[q][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) {
          const bf16x8 kf = __builtin_bit_cast(bf16x8, *(const uint4*)(smem + region + adk[2 * kb + ds]));
#pragma unroll
          for (int q = 0; q < 4; ++q) f4_mfma16(sc[q][kb], kf, qg[q][ds]);
        }
      asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    };
    first_scores(F4_PRE);
    float ref[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float mx = -1e30f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; r += 2) mx = fmaxf(mx, fmaxf(sc[q][kb][r], sc[q][kb][r + 1]));
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      ref[q] = mx + F4_REF_BIAS;
    }
    if (t0 != 0) first_scores(F4_PRE + 4096);   // (wave-uniform)
    f32x8 sx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ng[q] = f32x4{-ref[q], -ref[q], -ref[q], -ref[q]};
#pragma unroll
      for (int r = 0; r < 4; ++r) { sx[q][r] = sc[q][0][r] - ref[q]; sx[q][4 + r] = sc[q][1][r] - ref[q]; }
    }
    f32x4 og[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int d = 0; d < 4; ++d) og[q][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ls[4];   // row sums on the matrix pipe: P against ones - every register of a lane = the sum over the keys so far of its query
#pragma unroll
    for (int q = 0; q < 4; ++q) ls[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, MG_F16 ? make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u)
                                                          : make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
    unsigned sok = f4_sgpr(k0 + 3u * kst + (kst >> 1)), sov = f4_sgpr(v0 + 3u * 128u), mb = f4_sgpr(mw + 3 * F4_SLOT),
             cnt = f4_sgpr((unsigned)(t1 - t0 - 3));
    uint4 kf0, kf1, kf2, kf3, vf0, vf1, vf2, vf3;
    if (a.dbg) { tm0 = __builtin_amdgcn_s_memtime(); rm0 = __builtin_amdgcn_s_memrealtime(); }
    asm volatile(FA4W16_ASM
                 : "+{v[128:135]}"(sx[0]), "+{v[136:143]}"(sx[1]), "+{v[144:151]}"(sx[2]), "+{v[152:159]}"(sx[3]),
                   [o00] "+v"(og[0][0]), [o01] "+v"(og[0][1]), [o02] "+v"(og[0][2]), [o03] "+v"(og[0][3]),
                   [o10] "+v"(og[1][0]), [o11] "+v"(og[1][1]), [o12] "+v"(og[1][2]), [o13] "+v"(og[1][3]),
                   [o20] "+v"(og[2][0]), [o21] "+v"(og[2][1]), [o22] "+v"(og[2][2]), [o23] "+v"(og[2][3]),
                   [o30] "+v"(og[3][0]), [o31] "+v"(og[3][1]), [o32] "+v"(og[3][2]), [o33] "+v"(og[3][3]),
                   [ls0] "+v"(ls[0]), [ls1] "+v"(ls[1]), [ls2] "+v"(ls[2]), [ls3] "+v"(ls[3]),
                   [adk0] "+v"(adk[0]), [adk1] "+v"(adk[1]), [adk2] "+v"(adk[2]), [adk3] "+v"(adk[3]),
                   [adv0] "+v"(adv[0]), [adv1] "+v"(adv[1]),
                   [kf0] "=&v"(kf0), [kf1] "=&v"(kf1), [kf2] "=&v"(kf2), [kf3] "=&v"(kf3),
                   [vf0] "=&v"(vf0), [vf1] "=&v"(vf1), [vf2] "=&v"(vf2), [vf3] "=&v"(vf3),
                   [sok] "+s"(sok), [sov] "+s"(sov), [mb] "+s"(mb), [cnt] "+s"(cnt)
                 : [q00] "v"(qg[0][0]), [q01] "v"(qg[0][1]), [q10] "v"(qg[1][0]), [q11] "v"(qg[1][1]),
                   [q20] "v"(qg[2][0]), [q21] "v"(qg[2][1]), [q30] "v"(qg[3][0]), [q31] "v"(qg[3][1]),
                   [ng0] "v"(ng[0]), [ng1] "v"(ng[1]), [ng2] "v"(ng[2]), [ng3] "v"(ng[3]), [ones] "v"(ones),
                   [vk0] "v"(vk[0]), [vk1] "v"(vk[1]), [vv0] "v"(vv[0]), [vv1] "v"(vv[1]),
                   [srk] "s"(srk), [srv] "s"(srv), [kst] "s"(kst)
                 : "memory", "scc", F4_CLOBBER_Y);
    // -> the 32x32x16 layout of the finalize: o[Q][dt][4 g + j] = O^T[d = 32 dt + 8 g + 4 half + j][query 32 Q + l31]
#pragma unroll
    for (int Q = 0; Q < 2; ++Q)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const auto sw2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(og[2 * Q][2 * dt + cb][r]),
                                                              __float_as_uint(og[2 * Q + 1][2 * dt + cb][r]), false, false);
            o[Q][dt][8 * cb + r] = __uint_as_float(sw2[0]);
            o[Q][dt][8 * cb + 4 + r] = __uint_as_float(sw2[1]);
          }
    // row sums: every lane of column l15 holds the whole sum of its four queries; the finalize wants, per lane (query l31, half), a
    // share whose two halves add up to the query's sum
    const bool up = (lane >> 4) & 1;
    ll[0] = half == 0 ? (up ? ls[1][0] : ls[0][0]) : 0.f;
    ll[1] = half == 0 ? (up ? ls[3][0] : ls[2][0]) : 0.f;
  }
  if (a.dbg) {   // tuning only: shader cycles / 100 MHz ticks (+ tiles << 40) of the key loop, per wave of the workgroup's last segment
    const unsigned long long tm1 = __builtin_amdgcn_s_memtime(), rm1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      dbgw[0] = tm1 - tm0;
      dbgw[1] = (rm1 - rm0) | ((unsigned long long)(t1 - t0) << 40);
      dbgw[5] = rm0;
      dbgw[6] = rm1;
    }
  }
  bool finish = true;
  if (part >= 0) {
    // ---- a piece: leave the partial result, draw a ticket; the block's last piece adds the others to its registers ----
    // Hand-off (cdna_hip_programming.md, the split-K slab recipe in its write-through form): 16-byte sc1 stores -> every wave
    // drains vmcnt -> __syncthreads -> ONE relaxed agent-scope ticket; the last piece reads the slabs with sc1 loads (they
    // bypass its L1; the stores went through L2) - a valid form per MI355X_MICROARCH.md; mg_handoff_release / _acquire add the
    // fences in the A/B build (common.h).  With __threadfence() in EVERY wave here the split launch ran 1.7 x SLOWER than the
    // unsplit one (round 4).
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const __amdgpu_buffer_rsrc_t wsr = __builtin_amdgcn_make_buffer_rsrc((char*)a.ws + F4_CTR_BYTES, 0, 0x7fffffff, 0x00020000);
    const unsigned mine = (unsigned)(((long long)rem * F4_PIECES + part) * F4_PART_BYTES + wave * (16384 + 1024));
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32x4 v = {__float_as_uint(o[q][d][4 * j]), __float_as_uint(o[q][d][4 * j + 1]), __float_as_uint(o[q][d][4 * j + 2]),
                           __float_as_uint(o[q][d][4 * j + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, wsr, mine + ((q * 2 + d) * 4 + j) * 1024 + lane * 16, 0, 16);
        }
    {
      const u32x4 v = {__float_as_uint(ll[0]), __float_as_uint(ll[1]), 0u, 0u};
      __builtin_amdgcn_raw_buffer_store_b128(v, wsr, mine + 16384 + lane * 16, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = (int*)(smem + F4_PRE);
    if (tid == 0) {
      unsigned* ctr = (unsigned*)a.ws + rem;
      mg_handoff_release();
      const unsigned ticket = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool lastp = ticket == (unsigned)(npieces - 1);
      if (lastp) {
        mg_handoff_acquire();
        __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch (stream-ordered)
      }
      *flag = lastp ? 1 : 0;
    }
    __syncthreads();
    finish = *flag != 0;
    __syncthreads();
    if (finish) {
      // the sum in the order of the pieces whichever arrived last - its own slab is read back like the others - so that the
      // launch is bit-reproducible (fp32 addition is not associative)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        ll[q] = 0.f;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[q][d][r] = 0.f;
      }
      for (int pp = 0; pp < npieces; ++pp) {
        const unsigned other = (unsigned)(((long long)rem * F4_PIECES + pp) * F4_PART_BYTES + wave * (16384 + 1024));
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wsr, other + ((q * 2 + d) * 4 + j) * 1024 + lane * 16, 0, 16);
              o[q][d][4 * j] += __uint_as_float(v.x); o[q][d][4 * j + 1] += __uint_as_float(v.y);
              o[q][d][4 * j + 2] += __uint_as_float(v.z); o[q][d][4 * j + 3] += __uint_as_float(v.w);
            }
        const u32x4 lv = __builtin_amdgcn_raw_buffer_load_b128(wsr, other + 16384 + lane * 16, 0, 16);
        ll[0] += __uint_as_float(lv.x); ll[1] += __uint_as_float(lv.y);
      }
    }
  }
  if (!finish) return;   // (workgroup-uniform)
  // ---- finalize: O[q][d] = o^T / l; lane^32 exchange -> 8 consecutive d per lane, 16-byte stores ----
  // A block whose reference was too low (row sums >= redo_thr - 2^100, 2^15 in the fp16 build - or not finite) redoes its 256 queries
  // with the running-maximum form (two blocks of 128) INSTEAD of storing: decided for the whole workgroup before anything is stored
  // (round 6: the fixed-reference result used to be stored first and overwritten - two stores of one address from different waves
  // with only a barrier between them; in the fp16 build, where an overflowed probability is inf, the stale store showed).
  bool bad = false;
  float inv[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    float la, lb;
    half_swap(ll[q], ll[q], la, lb);
    const float l_tot = la + lb;
    bad = bad || !(l_tot < a.redo_thr);
    inv[q] = 1.0f / l_tot;
  }
  __syncthreads();   // (no __syncthreads_or: its static LDS word would move the ring off LDS address 0)
  const int wave_bad = __any(bad) ? 1 : 0;   // over ALL lanes (inside `if (lane == 0)` the vote would see lane 0 alone - the form of rounds 4-5)
  if (lane == 0) ((int*)smem)[wave] = wave_bad;
  __syncthreads();
  const int4 flags = *(const int4*)smem;
  __syncthreads();
  if (flags.x | flags.y | flags.z | flags.w) {
    fa25_body<4, true, 2>(a, smem, 2 * qb, bh);
    __syncthreads();
    fa25_body<4, true, 2>(a, smem, 2 * qb + 1, bh);
    __syncthreads();
    return;
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int q_row = qb * 256 + wave * 64 + q * 32 + l31;
    bf16_t* orow = a.O + (long long)b * a.sO + (long long)q_row * a.ldo + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) half_swap(o[q][dt][8 * gp + j] * inv[q], o[q][dt][8 * gp + 4 + j] * inv[q], v[j], v[4 + j]);
        uint4 pk;
        pk.x = cvt_pk_bf16_f32(v[0], v[1]); pk.y = cvt_pk_bf16_f32(v[2], v[3]);
        pk.z = cvt_pk_bf16_f32(v[4], v[5]); pk.w = cvt_pk_bf16_f32(v[6], v[7]);
        *(uint4*)(orow + dt * 32 + 16 * gp + 8 * half) = pk;
      }
  }
  if (dbgw && lane == 0) dbgw[7] = __builtin_amdgcn_s_memrealtime();
}

template <bool M16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void flash_attn64_4w_kernel(const FaArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // the ring: LDS address 0 (the stream wraps addresses at 64 KB)
  const int nkt = a.Ntok >> 6;
  const int bid = (int)blockIdx.x;
  // a whole block of 256 queries (bid < n_full), or a range of key tiles over the concatenated split blocks beyond them
  const bool whole = bid < a.n_full;
  const unsigned r = (unsigned)(bid - a.n_full), nwg = (unsigned)a.n_rem_wg, unkt = (unsigned)nkt;
  const unsigned total = (unsigned)a.n_rem * unkt;
  unsigned s0 = whole ? 0u : f4_bound(r, nwg, total, unkt);
  const unsigned s1 = whole ? unkt : f4_bound(r + 1, nwg, total, unkt);
  while (s0 < s1) {   // (one call site: the stream is ~1 300 instructions)
    const unsigned rem = s0 / unkt, t0 = s0 - rem * unkt;
    const unsigned t1 = (s1 - s0) < (unkt - t0) ? t0 + (s1 - s0) : unkt;
    int part = -1, npieces = 1;
    if (!(t0 == 0 && t1 == unkt)) {   // this block's pieces: the workgroups whose ranges meet [rem nkt, (rem + 1) nkt)
      const unsigned b0 = rem * unkt, b1 = b0 + unkt;
      unsigned rf = r, rl = r;
      while (rf > 0 && f4_bound(rf, nwg, total, unkt) > b0) --rf;
      while (rl + 1 < nwg && f4_bound(rl + 1, nwg, total, unkt) < b1) ++rl;
      part = (int)(r - rf);
      npieces = (int)(rl - rf + 1);
    }
    f4_segment<M16>(a, smem, whole ? xcd_remap(bid, a.n_full) : a.n_full + (int)rem, (int)t0, (int)t1, part, npieces, (int)rem);
    s0 += t1 - t0;
  }
}

}  // namespace

bool mg_flash4w_ok(const FaArgs& a, bool vt_perm) {
  const int nkt = a.Ntok / 64;
  return vt_perm && a.Ntok % 256 == 0 && nkt >= 4 && a.ldq % 8 == 0 && a.ldo % 8 == 0 && a.ldvt % 8 == 0 &&
         (uintptr_t)a.Q % 16 == 0 && (uintptr_t)a.K % 16 == 0 && (uintptr_t)a.Vt % 16 == 0 && (uintptr_t)a.O % 16 == 0 &&
         a.sQ % 8 == 0 && a.sK % 8 == 0 && a.sVt % 8 == 0 && a.sO % 8 == 0 &&
         (long long)a.Ntok * a.ldq * 2 < (1ll << 31) && 64ll * a.ldvt * 2 < (1ll << 31);
}

// Whole blocks in multiples of the CU count (two workgroups share a CU: 3 1/2 blocks per slot end in a round that a few CUs run
// alone); the rest in key pieces of at least a third of a block (at most four pieces per block) over up to two workgroups per CU.
// (measured, profiles/r4_flash4w.log: a piece costs ~15-20 us of prologue / partial result / combine on top of its key loop -
// splitting pays for a FEW left-over blocks behind many whole ones (1 800 = 7 x 256 + 8: 919 -> 863 us), not for a left-over of
// half a round (900 blocks at 2 304 tokens: 142 -> 154 us) or for a launch that does not fill the chip anyway)
// (FaArgs::split: 0 = that rule, 1 = split whatever is left over - the tests, 2 = never)
void mg_flash4w_plan(FaArgs* a, int n_cu) {
  const int nkt = a->Ntok / 64;
  const long long nb = (long long)(a->Ntok / 256) * a->heads * a->B;
  a->n_full = (int)nb;
  a->n_rem = 0;
  a->n_rem_wg = 0;
  static const int split = mg_tuning_int("MARIGOLD_FLASH4W_SPLIT", 1);
  if (split && a->split != 2 && a->ws && nb % n_cu != 0 && (a->split == 1 || (nb > n_cu && (nb % n_cu) * 8 <= n_cu))) {
    const long long rem = nb % n_cu;
    const long long cap = (a->ws_bytes - F4_CTR_BYTES) / ((long long)F4_PIECES * F4_PART_BYTES);
    long long wg = 2ll * n_cu;
    if (wg > 3 * rem) wg = 3 * rem;                               // pieces of >= a third of a block: at most four per block
    if (wg > (long long)rem * nkt / 4) wg = (long long)rem * nkt / 4;   // ... and of four tiles at least
    if (rem <= cap && rem * 4 <= F4_CTR_BYTES && wg > rem && (wg + 1) * rem * nkt < (1ll << 31)) {
      a->n_full = (int)(nb - rem);
      a->n_rem = (int)rem;
      a->n_rem_wg = (int)wg;
    }
  }
}

// (host tests) the plan for B x heads sequences of Ntok tokens and the piece boundaries of its split blocks:
// out[0..2] = whole blocks, split blocks, workgroups over them; bounds[0 .. out[2]] (if not null) in key tiles
extern "C" int mg_flash4w_plan_test(int B, int heads, int Ntok, int n_cu, long long ws_bytes, int split, int* out, unsigned* bounds) {
  FaArgs a = {};
  a.B = B; a.heads = heads; a.Ntok = Ntok;
  a.ws = ws_bytes > 0 ? (void*)(uintptr_t)16 : nullptr;
  a.ws_bytes = ws_bytes;
  a.split = split;
  mg_flash4w_plan(&a, n_cu);
  out[0] = a.n_full; out[1] = a.n_rem; out[2] = a.n_rem_wg;
  if (bounds && a.n_rem_wg > 0)
    for (int r = 0; r <= a.n_rem_wg; ++r) bounds[r] = f4_bound((unsigned)r, (unsigned)a.n_rem_wg, (unsigned)a.n_rem * (unsigned)(Ntok / 64), (unsigned)(Ntok / 64));
  return 0;
}

int mg_launch_flash4w(const FaArgs& a_in, hipStream_t s) {
  FaArgs a = a_in;
  MG_REQUIRE(mg_flash4w_ok(a, true), "flash_attn64 (hand-placed form): Ntok %d must be a multiple of 256, "
             "16-byte aligned operands", a.Ntok);
  const int LDS = F4_LDS;
  static bool attr_set = false;
  static int n_cu = 256;
  if (!attr_set && !g_dry_run) {
    for (const void* fn : {(const void*)flash_attn64_4w_kernel<false>, (const void*)flash_attn64_4w_kernel<true>}) {
      hipFuncAttributes fa;
      MG_CHECK_HIP(hipFuncGetAttributes(&fa, fn));
      MG_REQUIRE(fa.sharedSizeBytes == 0, "flash_attn64 (hand-placed form): the ring must start at LDS address 0 (static LDS %d bytes)", (int)fa.sharedSizeBytes);
      MG_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
    attr_set = true;
  }
  MG_REQUIRE((long long)(a.Ntok / 256) * a.heads * a.B < (1ll << 30), "flash_attn64: too many query blocks");
  mg_flash4w_plan(&a, n_cu);
  const long long grid = (long long)a.n_full + a.n_rem_wg;
  // The 16x16x32 stream (variant 27) where the launch keeps two workgroups on every CU for several rounds at 9 216 tokens - the
  // power-limited regime it was built for: E = 10: 851 vs 911 us; a tie at 2 304 tokens and for two members, 2 % slower for one
  // (profiles/r6_flash_mfma16.log).  m16 < 0: this rule; MARIGOLD_FLASH4W_M16 = 0 / 1 (tuning gate) forces a form.
  if (a.m16 < 0) a.m16 = (a.Ntok >= 4096 && (long long)(a.Ntok / 256) * a.heads * a.B >= 2ll * n_cu) ? 1 : 0;
  if (a.m16) MG_LAUNCH(flash_attn64_4w_kernel<true>, dim3((unsigned)grid), dim3(256), LDS, s, a);   // the stream on 16x16x32 MFMAs (variant 27)
  else MG_LAUNCH(flash_attn64_4w_kernel<false>, dim3((unsigned)grid), dim3(256), LDS, s, a);
  return 0;
}
