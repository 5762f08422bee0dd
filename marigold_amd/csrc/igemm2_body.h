// Body of the implicit-GEMM kernels (shared by igemm2.hip and igemm2_big.hip, which is built with the MFMA accumulators in
// the AGPR half of the register file - see the Makefile).  Design notes: igemm2.hip.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "igemm2_k4w.inc"

namespace {

struct Igemm2Args {
  const bf16_t* A;
  const bf16_t* A1;   // second channel source (channels [C0, Cin) of every tap; the UNet's skip concat) or nullptr
  int C0, lda1, c0t;  // channels / row stride of the sources; c0t = K tiles per tap that come from A (= cpt without A1)
  const bf16_t* X0;   // a 1x1 convolution folded in as extra K (the ResNet block's conv_shortcut): K tiles [taps * cpt, + xcpt) read pixel
  const bf16_t* X1;   // (y, x) of X0 (channels [0, xc0)) and X1 ([xc0, xcin)); nullptr = none
  int xcin, xc0, ldx0, ldx1, xcpt, xc0t;
  const bf16_t* Wt;
  void* out;
  const float* bias;
  const float* rowvec;
  const bf16_t* res;
  const void* zero;
  int H, W, Cin, Ho, Wo, N, taps, stride, pad, Hu, Wu, epi, ldo, ldr, lda, ldt, ldw;
  int tw;      // tap window width: 3 (taps = 9), 2 (taps = 4, the sub-pixel form of nearest-2x + conv3x3), 1
  int subpix;  // 1: batch entry z = 2a+b is output parity (a, b): window rows y-1+a.., cols x-1+b.., output pixel (2y+a, 2x+b)
  int M, rows_per_img, tiles_m, tiles_n, cpt, KT, rv_stride, up2, ctr;
  int n_begin, n_end;  // output-column range of this launch (tiles start at n_begin, bound n_end <= N)
  int splits, kps;     // split-K: `splits` workgroups per tile, each `kps` K steps; partials go to `ws`
  float* ws;           // fp32 [splits][M][N] (scale / bias / residual are applied by the reduce kernel)
  long long sA, sW, sO, sR;
  float scale;
  // LayerNorm folded into this GEMM (Linear layers that consume LN(x)): A = the RAW rows x, weights = W * gamma, and
  //   out = rstd[m] * (acc - mean[m] * ln_g[n]) + ln_c[n],  ln_g[n] = sum_k (W gamma)[n][k],  ln_c[n] = sum_k beta[k] W[n][k] + bias[n]
  // with (mean, rstd) of row m = ln_in[m], written by the GEMM launch that produced x: its epilogue stores (sum, sum of
  // squares) of every output row over each 32-column slot into ln_out [M][N/32], and the LAST column tile of a row block
  // to finish (ticket in ln_ctr[tile_m]) reduces the block's slots to (mean, rstd) at ln_out + M * (N/32) - the ln_in of
  // the consumers.  (Round 2, first form: every consumer tile reduced the slots itself - 10-40 strided 8-byte loads per
  // row and fp64 arithmetic in front of each of its N / BN tiles: GEGLU 320->2560 +30 us, 640->5120 +100 us.)
  const float2* ln_in;
  const float* ln_g;
  const float* ln_c;
  float2* ln_out;
  unsigned* ln_ctr;
  float ln_eps;
  float sm_scale;  // MG_EPI_SOFTMAX2: softmax scale and the number of real score columns (2 x heads)
  int sm_cols;
  // Launch-time constants of the index arithmetic (common.h: fdiv) and the straight-row switch: `lin` = Linear layer /
  // conv1x1 (taps 1, stride 1, no padding / up-sampling), whose output row m reads input row m - no (image, y, x) split.
  mg_fastdiv fd_per_z, fd_tiles, fd_tiles_n, fd_rpi, fd_wo, fd_cpt;
  int lin;
  double inv_n;     // 1 / N (row statistics of the output -> mean, rstd)
  int tperm;         // transposed section: tokens stored in ACCUMULATOR order inside every group of 16 ([0-3, 8-11, 4-7, 12-15] -
                     // what flash_attn64's generation 3 consumes without a lane exchange; no regroup here either)
  const bf16_t* w2;  // MG_EPI_XATTN2: second-stage weights [c2][64]
  unsigned long long* stamps;   // tuning only (MARIGOLD_IGEMM_STAMPS=1, tools/igemm_phases.py; tile variants 72 / 73): per workgroup 8 x s_memrealtime (100 MHz)
                                // at kernel entry / operands addressed / first tile landed / K loop done / outputs stored / exit
  int c2;
  double inv_c2;
};

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) { return cvt_pk_bf16_f32(lo, hi); }   // (common.h: RNE; the fp16 build saturates)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// TRANS = false: weights are the MFMA "A" operand, pixels "B": acc[ni][mi][4g+j] = C[m = mb+l31][n = nb+8g+4h+j]
// TRANS = true : pixels "A", weights "B":                      acc[ni][mi][4g+j] = C[m = mb+8g+4h+j][n = nb+l31]
// SPLIT: the LDS-DMA pieces of the next tile are issued in four portions BETWEEN the k-substeps'
// fragment reads and their MFMAs instead of in one burst after the barrier, so the ~60-100 issue
// cycles each piece costs overlap the wave's own MFMAs (all waves of a workgroup leave the barrier
// together - a burst leaves every SIMD's matrix pipe idle at the same time).
// ABL (tuning sweeps only; results are WRONG for ABL != 0): 1 = no LDS-DMA in the steady state,
// 2 = fragment reads but no MFMAs, 3 = MFMAs on fixed registers (no fragment reads).
// BK: K-tile depth (64: 128-byte LDS rows; 32: 64-byte rows, half the LDS per stage - lets two 4-wave
// workgroups with 128x64 wave tiles share a CU).
// LOOP: K-loop schedule - 0 = one barrier per K tile (the default family), 1 = half-K-step software pipeline,
// 2 = "ping-pong": 256x256 tile, 8 waves in two groups of four (rows 0-127 / 128-255) that run ONE BARRIER
// APART, four phases per K tile, each phase {fragment reads + LDS-DMA issue | barrier | 8 MFMAs | barrier} -
// while one group's waves are in the MFMA segment, their SIMD partners (the other group) are in the
// load segment, so the matrix pipe always has a wave to run (cdna_hip_programming.md "256^2 8-phase").
// PPOPT (ping-pong only): bit 0 = no s_setprio around the MFMA segments; bit 2 = in phases 1-3 the second
// LDS-DMA piece of the half tile is issued in the middle of the MFMA segment instead of the load segment
// (a piece costs its wave 60-185 issue cycles - two of them make the load segment longer than the partner's
// eight MFMAs).
// STAMP (tuning only): the instrumented instantiations of the hand-placed tiles (igemm2_big.hip) write their phase stamps.
template <int BM, int BN, int WGM, int WGN, int NSTAGE, bool TRANS, bool SPLIT, int LOOP, int ABL, int BK, int PPOPT = 0, bool STAMP = false>
__device__ __forceinline__ void igemm2_body(const Igemm2Args& a) {
  constexpr bool PF = LOOP == 1, PP = LOOP == 2, K4 = LOOP == 3;
  static_assert(!K4 || (((BM == 256 && BN == 256) || (BM == 192 && BN == 320)) && WGM == 2 && WGN == 2 && NSTAGE == 2 && BK == 64 &&
                        ABL == 0 && !TRANS),
                "the hand-placed one-wave-per-SIMD schedule is written for the 256x256 and 192x320 / 4-wave / 2-buffer tiles");
  static_assert(!PP || (BM == 256 && BN == 256 && WGM == 2 && WGN == 4 && NSTAGE == 2 && BK == 64 && ABL == 0),
                "the ping-pong schedule is written for the 256x256 / 8-wave / 2-stage tile");
  constexpr int NT = WGM * WGN * 64;
  constexpr int TM = BM / WGM, TN = BN / WGN, MI = TM / 32, NI = TN / 32;
  constexpr int ROWB = BK * 2;        // bytes per LDS row (one pixel's / one weight row's K tile)
  constexpr int CPR = BK / 8;         // 16-byte chunks per row
  constexpr int KS = BK / 16;         // MFMA k-substeps per K tile
  constexpr int A_IT = BM * CPR / NT, B_IT = BN * CPR / NT;
  constexpr int LOADS = A_IT + B_IT;  // LDS-DMA instructions per wave per K tile
  constexpr int STAGE = (BM + BN) * ROWB;
  static_assert(BK == 64 || BK == 32, "BK is 64 or 32");
  static_assert((BM * CPR) % NT == 0 && (BN * CPR) % NT == 0, "every thread stages a whole number of 16-byte chunks");
  // XOR swizzle of the 16-byte chunk index (applied on the DMA source address and on the fragment
  // read): conflict-free ds_read_b128 for 128-byte rows (r>>1)&7 and for 64-byte rows (r>>2)&3
  auto swz = [](int chunk, int r) { return BK == 64 ? (chunk ^ ((r >> 1) & 7)) : (chunk ^ ((r >> 2) & 3)); };
  constexpr int D = NSTAGE - 1;       // prefetch distance (tiles in flight)
  static_assert(A_IT >= 1 && B_IT >= 1 && MI >= 1 && NI >= 1, "tile too small for the block");
  static_assert(NSTAGE >= 2 && NSTAGE <= 4, "2..4 LDS stages");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, half = lane >> 5;

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  auto stamp = [&](int k) {
    if constexpr (STAMP) {
      if (a.stamps && tid == 0) a.stamps[(long long)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memrealtime();
    }
    (void)k;
  };
  stamp(0);
  const int tiles = a.tiles_m * a.tiles_n;
  const int per_z = tiles * a.splits;
  const int z = fdiv(bid, a.fd_per_z);
  const int rz = bid - z * per_z;
  const int split = fdiv(rz, a.fd_tiles);
  const int t = rz - split * tiles;
  const int tile_m = fdiv(t, a.fd_tiles_n), tile_n = t - tile_m * a.tiles_n;
  const int m0 = tile_m * BM, n0 = a.n_begin + tile_n * BN;

  const bf16_t* Ab = a.A + (long long)z * a.sA;   // (no __restrict__: MG_EPI_XATTN2 runs in place, out == A)
  const bf16_t* A1b = a.A1;
  const bf16_t* __restrict__ Wb = a.Wt + (long long)z * a.sW;
  const char* zero = (const char*)a.zero;

  // LDS row slot -> tile row.  Identity except for the ping-pong schedule, whose K tile is staged as four
  // half tiles in the order the phases consume them: A slots [0,128) = the first 64 rows of both wave groups,
  // [128,256) their second 64 rows; B slots [0,128) = the first 32 columns of the four wave columns, ...
  auto a_row_of = [](int slot) { return PP ? ((slot >> 6) & 1) * 128 + (slot >> 7) * 64 + (slot & 63) : slot; };
  auto b_row_of = [](int slot) { return PP ? ((slot >> 5) & 3) * 64 + (slot >> 7) * 32 + (slot & 31) : slot; };
  // sub-pixel form (nearest-2x up-sampling folded into the weights): parity (a, b) = (z >> 1, z & 1) reads source rows
  // y - 1 + a + ty, ty in {0, 1}, i.e. pads 1 - a on top / 1 - b on the left
  const int pad_y = a.subpix ? 1 - (z >> 1) : a.pad, pad_x = a.subpix ? 1 - (z & 1) : a.pad;
  // ---- staging rows owned by this thread (fixed over the K loop) ----
  int a_by[A_IT], a_bx[A_IT], a_qoff[A_IT];
  int a_img[A_IT];   // first pixel of the row's image (B * H * W < 2^31: M is an int)
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int ci = it * NT + tid;
    const int r = ci / CPR, p = ci % CPR;
    a_qoff[it] = swz(p, r) * 8;
    const int m = m0 + a_row_of(r);
    const bool ok = m < a.M;
    const int mm = ok ? m : 0;
    if (a.lin) {   // input row = output row: a_by carries it (negative beyond M), the (image, y, x) split is not needed
      a_by[it] = ok ? m : -1;
      a_bx[it] = 0;
      a_img[it] = 0;
    } else {
      const int img = fdiv(mm, a.fd_rpi);
      const int rem = mm - img * a.rows_per_img;
      const int oy = fdiv(rem, a.fd_wo), ox = rem - oy * a.Wo;
      a_by[it] = ok ? oy * a.stride - pad_y : -(1 << 28);  // rows beyond M never pass the bounds test
      a_bx[it] = ox * a.stride - pad_x;
      a_img[it] = img * a.H * a.W;
    }
  }
  const char* b_ptr[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int ci = it * NT + tid;
    const int r = ci / CPR, p = ci % CPR;
    const int n = n0 + b_row_of(r);
    b_ptr[it] = (n < a.n_end) ? (const char*)(Wb + (long long)n * a.ldw + swz(p, r) * 8) : zero;
  }
  const int kt0 = split * a.kps;                                   // this workgroup's K-step range
  const int KT = min(a.KT, kt0 + a.kps) - kt0;
  if (kt0) {
#pragma unroll
    for (int it = 0; it < B_IT; ++it)
      if (b_ptr[it] != zero) b_ptr[it] += (long long)kt0 * ROWB;
  }
  const char* a_ptr[A_IT];
  const int hb = a.Hu ? a.Hu : a.H, wb = a.Hu ? a.Wu : a.W;  // bounds in (virtual) input space
  auto tap_setup = [&](int tap, bool second = false) {
    int dy = 0, dx = 0;
    const bf16_t* Sb = second ? A1b : Ab;
    const int ld = second ? a.lda1 : a.lda;
    if (a.lin) {
#pragma unroll
      for (int it = 0; it < A_IT; ++it)
        a_ptr[it] = a_by[it] >= 0 ? (const char*)(Sb + (long long)a_by[it] * ld + a_qoff[it]) : zero;
      return;
    }
    if (a.tw > 1) { dy = a.tw == 3 ? (tap * 11) >> 5 : tap >> 1; dx = tap - dy * a.tw; }   // tap / tw for tap < 9
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      int iy = a_by[it] + dy, ix = a_bx[it] + dx;
      const bool ok = (unsigned)iy < (unsigned)hb && (unsigned)ix < (unsigned)wb;
      if (a.Hu) {
        if (a.up2) { iy >>= 1; ix >>= 1; }
        else { iy = ok ? (iy * a.H) / a.Hu : 0; ix = ok ? (ix * a.W) / a.Wu : 0; }
      }
      const char* pv = (const char*)(Sb + (long long)(a_img[it] + iy * a.W + ix) * ld + a_qoff[it]);
      a_ptr[it] = ok ? pv : zero;
    }
  };
  // the folded 1x1 convolution's K tiles (behind the taps): the window's centre pixel of X0 / X1 - i_tap == taps marks them
  auto fold_setup = [&](bool second) {
    const bf16_t* Sb = second ? a.X1 : a.X0;
    const int ld = second ? a.ldx1 : a.ldx0;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int iy = a_by[it] + a.pad, ix = a_bx[it] + a.pad;
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      a_ptr[it] = ok ? (const char*)(Sb + (long long)(a_img[it] + iy * a.W + ix) * ld + a_qoff[it]) : zero;
    }
  };
  const int KTm = a.taps * a.cpt;   // K tiles of the taps
  int i_tap = kt0 ? fdiv(kt0, a.fd_cpt) : 0, i_c = kt0 - i_tap * a.cpt;  // (tap, channel tile) of the NEXT tile to issue
  if (kt0 >= KTm) { i_tap = a.taps; i_c = kt0 - KTm; }
  {
    const bool fold = i_tap >= a.taps;
    const bool second = fold ? i_c >= a.xc0t : i_c >= a.c0t;
    if (fold) fold_setup(second);
    else tap_setup(i_tap, second);
    if (const int skip = second ? i_c - (fold ? a.xc0t : a.c0t) : i_c) {
#pragma unroll
      for (int it = 0; it < A_IT; ++it)
        if (a_ptr[it] != zero) a_ptr[it] += skip * ROWB;
    }
  }
  // LDS map: [stage][A rows | B rows]; the ping-pong schedule keeps [A stage 0 | A stage 1 | B stage 0 | B stage 1]
  // so that both stages of an operand are within the 16-bit immediate offset of one base address.
  auto a_stage = [&](int stage) { return smem + stage * (PP ? BM * ROWB : STAGE); };
  auto b_stage = [&](int stage) { return smem + (PP ? NSTAGE * BM * ROWB + stage * (BN * ROWB) : stage * STAGE + BM * ROWB); };
  auto issue_piece = [&](int stage, int idx) {  // idx in [0, LOADS): A pieces first, then B pieces
    if constexpr (ABL == 1) return;
    if (idx < A_IT) {
      glds16(a_ptr[idx], a_stage(stage) + (idx * NT + wave * 64) * 16);
      a_ptr[idx] += ROWB;
    } else {
      const int it = idx - A_IT;
      glds16(b_ptr[it], b_stage(stage) + (it * NT + wave * 64) * 16);
      b_ptr[it] += ROWB;
    }
  };
  auto advance = [&]() {
    if (i_tap >= a.taps) {   // inside the folded 1x1 convolution: X0, then X1
      if (++i_c == a.xc0t && a.xc0t < a.xcpt) fold_setup(true);
      return;
    }
    if (++i_c == a.cpt) {
      i_c = 0;
      ++i_tap;
      if (i_tap < a.taps) tap_setup(i_tap);
      else if (a.xcpt) fold_setup(false);
    } else if (i_c == a.c0t) {
      tap_setup(i_tap, true);   // the remaining channel tiles of this tap come from the second source
    }
  };
  auto issue = [&](int stage) {
#pragma unroll
    for (int idx = 0; idx < LOADS; ++idx) issue_piece(stage, idx);
    advance();
  };

  f32x16 acc[NI][MI];
  auto zero_acc = [&]() {   // called after ln_publish: the statistics slots' registers are dead by then
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
  };

  // folded LayerNorm: (mean, rstd) of the tile's BM rows (ln_in, finalized by the producer's last column tile), kept in
  // LDS behind the ring for the epilogue.  One 8-byte load per row, issued ahead of the first LDS-DMA stage and published
  // right after it (ln_publish); the K loop's barriers order the LDS write before the epilogue's reads.
  float2* const lnst = (float2*)(smem + NSTAGE * STAGE);
  static_assert(NT >= BM, "one statistics row per thread (or less)");
  float2 ln_mr = make_float2(0.f, 1.f);
  if (a.ln_in && tid < BM && m0 + tid < a.M) ln_mr = a.ln_in[m0 + tid];
  auto ln_publish = [&]() {
    if (!a.ln_in) return;
    if (tid < BM) lnst[tid] = ln_mr;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS write has left before this wave's next barrier
  };

  int rowA[MI], rowB[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
    rowA[mi] = PP ? (mi >> 1) * 128 + wm * 64 + (mi & 1) * 32 + l31 : wm * TM + mi * 32 + l31;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) rowB[ni] = PP ? ni * 128 + wn * 32 + l31 : wn * TN + ni * 32 + l31;

  if constexpr (!PP && !K4) {
#pragma unroll
    for (int d = 0; d < D; ++d)
      if (d < KT) issue(d);
  }
  if constexpr (!PP && !K4) {
    ln_publish();
    zero_acc();
    stamp(1);
    stamp(2);
  }
  int st_c = 0;                 // stage holding tile kt
  int st_i = D % NSTAGE;        // stage receiving tile kt + D
  // One K step.  ISSUE is a compile-time tag: the steady-state loop (a tile to prefetch every step) is
  // straight-line code - no branch between the fragment reads and the MFMAs, so hipcc's waitcnt
  // pass keeps the partial lgkmcnt waits - and the last D steps run the same body without the DMA.
  auto load_frags = [&](const char* sA, int ks, bf16x8(&fa)[MI], bf16x8(&fb)[NI]) {
    if constexpr (ABL == 3) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) fa[mi] = __builtin_bit_cast(bf16x8, make_uint4(ks, lane, mi, 1));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) fb[ni] = __builtin_bit_cast(bf16x8, make_uint4(ks, lane, ni, 2));
      return;
    }
    const char* sB = sA + BM * ROWB;
    const int q = ks * 2 + half;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int r = rowA[mi];
      fa[mi] = __builtin_bit_cast(bf16x8, *(const uint4*)(sA + r * ROWB + (swz(q, r) << 4)));
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int r = rowB[ni];
      fb[ni] = __builtin_bit_cast(bf16x8, *(const uint4*)(sB + r * ROWB + (swz(q, r) << 4)));
    }
  };
  auto mfmas = [&](const bf16x8(&fa)[MI], const bf16x8(&fb)[NI]) {
    if constexpr (ABL == 2) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(fa[mi]));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(fb[ni]));
      return;
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        if constexpr (!TRANS)
          acc[ni][mi] = mg_mfma32(fb[ni], fa[mi], acc[ni][mi]);
        else
          acc[ni][mi] = mg_mfma32(fa[mi], fb[ni], acc[ni][mi]);
      }
  };

  if constexpr (K4) {
    // ---- LOOP == 3: 256 x 256 tile on FOUR waves (one per SIMD, 128 x 128 wave tile, accumulators in AGPRs), the K tile's
    // instruction stream placed by hand (gen_k4w.py -> igemm2_k4w.inc; schedule and its reasons in the generator's header).
    // Staging differs from the other loops in two ways that take every VALU instruction out of the steady state:
    //   * LDS-DMA through BUFFER loads: a wave-uniform resource (base in SGPRs, advanced by SALU adds - 128 bytes per K
    //     tile) + a per-lane 32-bit byte offset that only changes with the tap / source; padding rows carry an offset beyond
    //     num_records (2 GiB), for which the buffer load writes zeros - no zero page, no per-lane 64-bit pointer arithmetic;
    //   * fragment reads at immediate offsets from eight address registers (one per k-step and operand), the buffer
    //     selected by XOR 0x8000.
    // LDS: pixel rows of buffer 0 / 1 at 0 / 32768, weight rows at 65536 / 98304, 128-byte rows, 16-byte chunks XOR-swizzled
    // as in the other loops (the DMA source address carries the swizzle).
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    constexpr unsigned OOB = 0x80000000u;
    static_assert(((A_IT == 8 && B_IT == 8 && MI == 4 && NI == 4) || (A_IT == 6 && B_IT == 10 && MI == 3 && NI == 5)) && KS == 4, "k4w geometry");
    constexpr int ABUF = BM * ROWB, BBUF = BN * ROWB;   // LDS: [pixel rows 0 | pixel rows 1 | weight rows 0 | weight rows 1]
    unsigned va[A_IT], vb[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int ci = it * NT + tid;
      const int r = ci / CPR, p = ci % CPR;
      const int n = n0 + r;
      vb[it] = n < a.n_end ? (unsigned)(n * a.ldw + swz(p, r) * 8) * 2u : OOB;
    }
    // Operand bases live in SGPRs as (lo, hi) halves (readfirstlane'd: hipcc otherwise keeps a 64-bit induction variable in
    // VGPRs and hands the asm a VGPR tuple for its "s" operand).
    auto sgpr = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    unsigned loA = 0, hiA = 0, loB, hiB;
    {
      const unsigned long long b = (unsigned long long)(uintptr_t)Wb + (unsigned long long)kt0 * ROWB;
      loB = sgpr((unsigned)b);
      hiB = sgpr((unsigned)(b >> 32));
    }
    auto add_base = [](unsigned& lo, unsigned& hi, unsigned bytes) {
      const unsigned n = lo + bytes;
      hi += n < lo ? 1u : 0u;
      lo = n;
    };
    auto tap_setup4 = [&](int tap, bool second) {
      int dy = 0, dx = 0;
      const int ld = second ? a.lda1 : a.lda;
      const unsigned long long b = (unsigned long long)(uintptr_t)(second ? A1b : Ab);
      loA = sgpr((unsigned)b);
      hiA = sgpr((unsigned)(b >> 32));
      if (a.lin) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) va[it] = a_by[it] >= 0 ? (unsigned)(a_by[it] * ld + a_qoff[it]) * 2u : OOB;
        return;
      }
      if (a.tw > 1) { dy = a.tw == 3 ? (tap * 11) >> 5 : tap >> 1; dx = tap - dy * a.tw; }
#pragma unroll
      for (int it = 0; it < A_IT; ++it) {
        int iy = a_by[it] + dy, ix = a_bx[it] + dx;
        const bool ok = (unsigned)iy < (unsigned)hb && (unsigned)ix < (unsigned)wb;
        if (a.Hu) {
          if (a.up2) { iy >>= 1; ix >>= 1; }
          else { iy = ok ? (iy * a.H) / a.Hu : 0; ix = ok ? (ix * a.W) / a.Wu : 0; }
        }
        va[it] = ok ? (unsigned)((a_img[it] + iy * a.W + ix) * ld + a_qoff[it]) * 2u : OOB;
      }
    };
    // The K tiles are staged in SEGMENTS (one tap of one source): inside a segment the per-lane offsets are fixed and a tile
    // costs two SALU adds; (j_tap, j_c) = the next tile to stage, seg_left = tiles left in its segment.
    auto fold_setup4 = [&](bool second) {   // the folded 1x1 convolution's segments: the window's centre pixel of X0 / X1
      const int ld = second ? a.ldx1 : a.ldx0;
      const unsigned long long b = (unsigned long long)(uintptr_t)(second ? a.X1 : a.X0);
      loA = sgpr((unsigned)b);
      hiA = sgpr((unsigned)(b >> 32));
#pragma unroll
      for (int it = 0; it < A_IT; ++it) {
        const int iy = a_by[it] + a.pad, ix = a_bx[it] + a.pad;
        const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        va[it] = ok ? (unsigned)((a_img[it] + iy * a.W + ix) * ld + a_qoff[it]) * 2u : OOB;
      }
    };
    int j_tap = kt0 ? fdiv(kt0, a.fd_cpt) : 0, j_c = kt0 - j_tap * a.cpt;
    if (kt0 >= KTm) { j_tap = a.taps; j_c = kt0 - KTm; }
    int seg_left;
    if (j_tap >= a.taps) {
      const bool second = j_c >= a.xc0t;
      fold_setup4(second);
      add_base(loA, hiA, (unsigned)(second ? j_c - a.xc0t : j_c) * ROWB);
      seg_left = second ? a.xcpt - j_c : a.xc0t - j_c;
    } else {
      const bool second = j_c >= a.c0t;
      tap_setup4(j_tap, second);
      add_base(loA, hiA, (unsigned)(second ? j_c - a.c0t : j_c) * ROWB);
      seg_left = second ? a.cpt - j_c : a.c0t - j_c;
    }
    auto next_segment = [&]() {   // called with seg_left == 0 and at least one more tile to stage
      if (j_tap >= a.taps) {        // inside the folded 1x1 convolution: X0 is done, X1 follows
        fold_setup4(true);
        seg_left = a.xcpt - a.xc0t;
      } else if (j_c == a.cpt) {
        j_c = 0;
        ++j_tap;
        if (j_tap < a.taps) {
          tap_setup4(j_tap, false);
          seg_left = a.c0t;
        } else {
          fold_setup4(false);
          seg_left = a.xc0t;
        }
      } else {
        tap_setup4(j_tap, true);
        seg_left = a.cpt - a.c0t;
      }
    };
    auto srd_of = [](unsigned lo, unsigned hi) {
      const i32x4 r = {(int)lo, (int)(hi & 0xffffu), (int)OOB, 0x00020000};
      return r;
    };
    const int sw = (l31 >> 1) & 7;
    int la[KS], lb[KS], xa[KS], xb[KS];   // fragment read addresses and their buffer toggles (XOR)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int q = (((2 * ks + half) ^ sw) << 4) + l31 * ROWB;
      la[ks] = wm * (TM * ROWB) + q;
      lb[ks] = 2 * ABUF + wn * (TN * ROWB) + q;
      xa[ks] = la[ks] ^ (la[ks] + ABUF);
      xb[ks] = lb[ks] ^ (lb[ks] + BBUF);
    }
    int fill = 0;           // buffer being filled
    int ma = wave * 1024, mb = 2 * ABUF + wave * 1024;   // LDS addresses of this wave's first pixel / weight piece in it
    auto flip = [&]() {
      fill ^= 1;
      ma = __builtin_amdgcn_readfirstlane(wave * 1024 + fill * ABUF);
      mb = __builtin_amdgcn_readfirstlane(2 * ABUF + wave * 1024 + fill * BBUF);
    };
    int staged = 0;         // tiles staged so far
    auto stage_tile = [&]() {   // prologue only: the steady state stages from inside the hand-placed stream
      const i32x4 sa = srd_of(loA, hiA), sb = srd_of(loB, hiB);
#pragma unroll
      for (int it = 0; it < A_IT; ++it)
        asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                     :: "v"(va[it]), "s"(sa), "s"(ma), "n"(it * 4096) : "memory", "scc");
#pragma unroll
      for (int it = 0; it < B_IT; ++it)
        asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                     :: "v"(vb[it]), "s"(sb), "s"(mb), "n"(it * 4096) : "memory", "scc");
      add_base(loA, hiA, ROWB);
      add_base(loB, hiB, ROWB);
      flip();
      ++staged;
      ++j_c;
      if (--seg_left == 0 && staged < KT) next_segment();
    };
    bf16x8 fa4[KS][MI], fb4[KS][NI];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int i = 0; i < MI; ++i) fa4[ks][i] = __builtin_bit_cast(bf16x8, make_uint4(0, 0, 0, 0));
#pragma unroll
      for (int i = 0; i < NI; ++i) fb4[ks][i] = __builtin_bit_cast(bf16x8, make_uint4(0, 0, 0, 0));
    }
    stamp(1);
    stage_tile();                 // tile 0 -> buffer 0
    if (KT > 1) stage_tile();     // tile 1 -> buffer 1
    ln_publish();
    zero_acc();
    if (KT > 1) wait_vmcnt<A_IT + B_IT>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    stamp(2);
#define K4W_C(ni, mi) [c##ni##mi] "+a"(acc[ni][mi])
#define K4W_A(ks, i) [a##ks##i] "+v"(fa4[ks][i])
#define K4W_B(ks, i) [b##ks##i] "+v"(fb4[ks][i])
#define K4W_ADDR                                                                                                            \
  [la0] "+v"(la[0]), [la1] "+v"(la[1]), [la2] "+v"(la[2]), [la3] "+v"(la[3]),                                              \
  [lb0] "+v"(lb[0]), [lb1] "+v"(lb[1]), [lb2] "+v"(lb[2]), [lb3] "+v"(lb[3])
#define K4W_TOGGLES                                                                                                         \
  [xa0] "v"(xa[0]), [xa1] "v"(xa[1]), [xa2] "v"(xa[2]), [xa3] "v"(xa[3]), [xb0] "v"(xb[0]), [xb1] "v"(xb[1]),              \
  [xb2] "v"(xb[2]), [xb3] "v"(xb[3]), [sa] "s"(sa), [sb] "s"(sb), [ma] "s"(ma), [mb] "s"(mb)
    // 256 x 256: 4 x 4 fragments per wave, 8 + 8 DMA pieces
#define K4W_OUT_A                                                                                                           \
  K4W_C(0, 0), K4W_C(0, 1), K4W_C(0, 2), K4W_C(0, 3), K4W_C(1, 0), K4W_C(1, 1), K4W_C(1, 2), K4W_C(1, 3), K4W_C(2, 0),      \
  K4W_C(2, 1), K4W_C(2, 2), K4W_C(2, 3), K4W_C(3, 0), K4W_C(3, 1), K4W_C(3, 2), K4W_C(3, 3),                                \
  K4W_A(0, 0), K4W_A(0, 1), K4W_A(0, 2), K4W_A(0, 3), K4W_A(1, 0), K4W_A(1, 1), K4W_A(1, 2), K4W_A(1, 3), K4W_A(2, 0),      \
  K4W_A(2, 1), K4W_A(2, 2), K4W_A(2, 3), K4W_A(3, 0), K4W_A(3, 1), K4W_A(3, 2), K4W_A(3, 3),                                \
  K4W_B(0, 0), K4W_B(0, 1), K4W_B(0, 2), K4W_B(0, 3), K4W_B(1, 0), K4W_B(1, 1), K4W_B(1, 2), K4W_B(1, 3), K4W_B(2, 0),      \
  K4W_B(2, 1), K4W_B(2, 2), K4W_B(2, 3), K4W_B(3, 0), K4W_B(3, 1), K4W_B(3, 2), K4W_B(3, 3), K4W_ADDR
#define K4W_IN_A                                                                                                            \
  [va0] "v"(va[0]), [va1] "v"(va[1]), [va2] "v"(va[2]), [va3] "v"(va[3]), [va4] "v"(va[4]), [va5] "v"(va[5]),              \
  [va6] "v"(va[6]), [va7] "v"(va[7]), [vb0] "v"(vb[0]), [vb1] "v"(vb[1]), [vb2] "v"(vb[2]), [vb3] "v"(vb[3]),  \
  [vb4] "v"(vb[4]), [vb5] "v"(vb[5]), [vb6] "v"(vb[6]), [vb7] "v"(vb[7]), K4W_TOGGLES
    // 192 x 320: 3 x 5 fragments per wave, 6 + 10 DMA pieces
#define K4W_OUT_B                                                                                                           \
  K4W_C(0, 0), K4W_C(0, 1), K4W_C(0, 2), K4W_C(1, 0), K4W_C(1, 1), K4W_C(1, 2), K4W_C(2, 0), K4W_C(2, 1), K4W_C(2, 2),      \
  K4W_C(3, 0), K4W_C(3, 1), K4W_C(3, 2), K4W_C(4, 0), K4W_C(4, 1), K4W_C(4, 2),                              \
  K4W_A(0, 0), K4W_A(0, 1), K4W_A(0, 2), K4W_A(1, 0), K4W_A(1, 1), K4W_A(1, 2), K4W_A(2, 0), K4W_A(2, 1), K4W_A(2, 2),      \
  K4W_A(3, 0), K4W_A(3, 1), K4W_A(3, 2),                                                                                    \
  K4W_B(0, 0), K4W_B(0, 1), K4W_B(0, 2), K4W_B(0, 3), K4W_B(0, 4), K4W_B(1, 0), K4W_B(1, 1), K4W_B(1, 2), K4W_B(1, 3), \
  K4W_B(1, 4), K4W_B(2, 0), K4W_B(2, 1), K4W_B(2, 2), K4W_B(2, 3), K4W_B(2, 4), K4W_B(3, 0), K4W_B(3, 1),         \
  K4W_B(3, 2), K4W_B(3, 3), K4W_B(3, 4), K4W_ADDR
#define K4W_IN_B                                                                                                            \
  [va0] "v"(va[0]), [va1] "v"(va[1]), [va2] "v"(va[2]), [va3] "v"(va[3]), [va4] "v"(va[4]), [va5] "v"(va[5]),              \
  [vb0] "v"(vb[0]), [vb1] "v"(vb[1]), [vb2] "v"(vb[2]), [vb3] "v"(vb[3]), [vb4] "v"(vb[4]), [vb5] "v"(vb[5]),              \
  [vb6] "v"(vb[6]), [vb7] "v"(vb[7]), [vb8] "v"(vb[8]), [vb9] "v"(vb[9]), K4W_TOGGLES
#define K4W_RUN(text_a, text_b)                                                                  \
  do {                                                                                           \
    if constexpr (MI == 4) asm volatile(text_a : K4W_OUT_A : K4W_IN_A : "memory", "scc");        \
    else asm volatile(text_b : K4W_OUT_B : K4W_IN_B : "memory", "scc");                          \
  } while (0)
    {
      const i32x4 sa = srd_of(loA, hiA), sb = srd_of(loB, hiB);
      K4W_RUN(K4W_ASM_PROLOGUE, K4WB_ASM_PROLOGUE);
    }
    // steady state: while tile t is computed, tile t + 2 is staged into the buffer it is read from - one stream per tile,
    // grouped by the segment of the STAGED tile (the computed tile needs no addressing state at all)
    while (staged < KT) {
      const int n = min(seg_left, KT - staged);
      for (int i = 0; i < n; ++i) {
        const i32x4 sa = srd_of(loA, hiA), sb = srd_of(loB, hiB);
        K4W_RUN(K4W_ASM_FULL, K4WB_ASM_FULL);
        add_base(loA, hiA, ROWB);
        add_base(loB, hiB, ROWB);
        flip();
      }
      staged += n;
      j_c += n;
      seg_left -= n;
      if (seg_left == 0 && staged < KT) next_segment();
    }
    {
      const i32x4 sa = srd_of(loA, hiA), sb = srd_of(loB, hiB);
      if (KT > 1) K4W_RUN(K4W_ASM_NODMA, K4WB_ASM_NODMA);
      K4W_RUN(K4W_ASM_LAST, K4WB_ASM_LAST);
    }
#undef K4W_ADDR
#undef K4W_TOGGLES
#undef K4W_OUT_A
#undef K4W_IN_A
#undef K4W_OUT_B
#undef K4W_IN_B
#undef K4W_RUN
#undef K4W_C
#undef K4W_A
#undef K4W_B
  } else if constexpr (PP) {
    // Half tiles of K tile t (each 128 LDS rows = two DMA pieces per wave), in consumption order:
    //   H0 = A rows {mi 0,1}  H1 = B rows {ni 0}  H2 = B rows {ni 1}  H3 = A rows {mi 2,3}
    // Phase p of tile t:  P1 reads H0+H1 -> acc(mi 0,1 x ni 0)   P2 reads H2 -> acc(mi 0,1 x ni 1)
    //                     P3 reads H3 -> acc(mi 2,3 x ni 1)      P4 reads -  -> acc(mi 2,3 x ni 0)
    // and issues half tile H(p-1) of tile t+1 into the other stage.  A half tile is read one phase AFTER the
    // counted vmcnt that retires it (wait in the load segment of phase p-1, barrier, read in phase p): with the
    // two groups one barrier apart that is the earliest point at which every wave's part has landed.  Stage
    // reuse is 4-5 phases behind the last read of the slot.  Waits in a tile that still issues: P1/P2/P4
    // leave the two youngest half tiles (4 loads) in flight; the last tile drains 2 -> 0.
    static_assert(A_IT == 4 && B_IT == 4 && MI == 4 && NI == 2 && KS == 4, "ping-pong geometry");
    auto issue_half = [&](int stage, int h) {
      if (h == 0) { issue_piece(stage, 0); issue_piece(stage, 1); }
      else if (h == 1) { issue_piece(stage, A_IT + 0); issue_piece(stage, A_IT + 1); }
      else if (h == 2) { issue_piece(stage, A_IT + 2); issue_piece(stage, A_IT + 3); }
      else { issue_piece(stage, 2); issue_piece(stage, 3); }
    };
    bf16x8 fa[2][KS], fb[NI][KS];
    auto read_a = [&](int st, int pair) {
      const char* sA = a_stage(st);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int r = rowA[pair * 2 + m];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          fa[m][ks] = __builtin_bit_cast(bf16x8, *(const uint4*)(sA + r * ROWB + (swz(ks * 2 + half, r) << 4)));
      }
    };
    auto read_b = [&](int st, int ni) {
      const char* sB = b_stage(st);
      const int r = rowB[ni];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        fb[ni][ks] = __builtin_bit_cast(bf16x8, *(const uint4*)(sB + r * ROWB + (swz(ks * 2 + half, r) << 4)));
    };
    auto quadrant = [&](int pair, int ni, int mid_stage, int mid_piece) {   // 8 MFMAs, two accumulators alternating
      if constexpr (!(PPOPT & 1)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks == 2 && mid_piece >= 0) issue_piece(mid_stage, mid_piece);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          if constexpr (!TRANS)
            acc[ni][pair * 2 + m] = mg_mfma32(fb[ni][ks], fa[m][ks], acc[ni][pair * 2 + m]);
          else
            acc[ni][pair * 2 + m] = mg_mfma32(fa[m][ks], fb[ni][ks], acc[ni][pair * 2 + m]);
        }
      }
      if constexpr (!(PPOPT & 1)) __builtin_amdgcn_s_setprio(0);
    };
    auto fence = [&]() {   // nothing moves across: neither the compiler's memory ops nor the machine scheduler
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    auto rendezvous = [&]() {
      fence();
      __builtin_amdgcn_s_barrier();
      fence();
    };
    auto compute = [&](int pair, int ni, int mid_stage = 0, int mid_piece = -1) {   // barrier | MFMA segment | barrier
      rendezvous();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      quadrant(pair, ni, mid_stage, mid_piece);
      rendezvous();
    };
    // prologue: the whole first tile, H0 and H1 landed for everyone before the first phase
    issue_half(0, 0); issue_half(0, 1); issue_half(0, 2); issue_half(0, 3);
    advance();
    ln_publish();
    zero_acc();
    wait_vmcnt<4>();
    rendezvous();
    if (wm == 1) rendezvous();   // the second wave group runs one barrier behind the first
    // DMA pieces of the half tiles: H0 = A pieces 0,1; H1 = B pieces 0,1; H2 = B pieces 2,3; H3 = A pieces 2,3
    constexpr int HP[4][2] = {{0, 1}, {A_IT + 0, A_IT + 1}, {A_IT + 2, A_IT + 3}, {2, 3}};
    constexpr bool MID = (PPOPT & 4) != 0;
    // counted waits in a tile that stages its successor: loads that may stay in flight after the wait of P1 / P2
    // (the half tile read next phase is older than these); P4 always leaves H2', H3' = 4 pieces
    constexpr int W12 = MID ? 3 : 4;
    auto tile = [&](int st, auto issue_tag) {
      constexpr bool ISSUE = decltype(issue_tag)::value;   // a next tile exists: stage it into the other stage
      const int sn = st ^ 1;
      // P1
      read_a(st, 0); read_b(st, 0);
      if constexpr (ISSUE) {
        issue_piece(sn, HP[0][0]);
        if constexpr (!MID) issue_piece(sn, HP[0][1]);
        wait_vmcnt<W12>();
      } else {
        wait_vmcnt<2>();
      }
      compute(0, 0, sn, (ISSUE && MID) ? HP[0][1] : -1);
      // P2
      read_b(st, 1);
      if constexpr (ISSUE) {
        issue_piece(sn, HP[1][0]);
        if constexpr (!MID) issue_piece(sn, HP[1][1]);
        wait_vmcnt<W12>();
      } else {
        wait_vmcnt<0>();
      }
      compute(0, 1, sn, (ISSUE && MID) ? HP[1][1] : -1);
      // P3
      read_a(st, 1);
      if constexpr (ISSUE) {
        issue_piece(sn, HP[2][0]);
        if constexpr (!MID) issue_piece(sn, HP[2][1]);
      }
      compute(1, 1, sn, (ISSUE && MID) ? HP[2][1] : -1);
      // P4 (no fragment reads: both pieces of H3' and the tap bookkeeping go here)
      if constexpr (ISSUE) {
        issue_piece(sn, HP[3][0]);
        issue_piece(sn, HP[3][1]);
        advance();
        wait_vmcnt<4>();
      }
      compute(1, 0);
    };
    {
      int kt = 0, st = 0;
      for (; kt + 1 < KT; ++kt, st ^= 1) tile(st, std::true_type{});
      tile(st, std::false_type{});
    }
    if (wm == 0) rendezvous();   // pair the second group's last barrier
  } else if constexpr (!PF) {
    // One K step.  ISSUE is a compile-time tag: the steady-state loop (a tile to prefetch every step)
    // is straight-line code and the last D steps run the same body without the DMA.
    auto k_step = [&](int kt, auto issue_tag) {
      constexpr bool ISSUE = decltype(issue_tag)::value;
      if constexpr (ISSUE) {
        wait_vmcnt<(D - 1) * LOADS>();  // tiles kt+1 .. kt+D-1 may stay in flight
      } else {
        const int younger = KT - 1 - kt;
        if (D >= 3 && younger >= 2) wait_vmcnt<2 * LOADS>();
        else if (D >= 2 && younger >= 1) wait_vmcnt<LOADS>();
        else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();  // everyone's part of tile kt landed; everyone left stage st_i
      if constexpr (!SPLIT && ISSUE) issue(st_i);
      const char* sA = smem + st_c * STAGE;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 fa[MI], fb[NI];
        load_frags(sA, ks, fa, fb);
        if constexpr (SPLIT && ISSUE) {
#pragma unroll
          for (int idx = ks; idx < LOADS; idx += KS) issue_piece(st_i, idx);
        }
        mfmas(fa, fb);
      }
      if constexpr (SPLIT && ISSUE) advance();
      st_c = (st_c + 1 == NSTAGE) ? 0 : st_c + 1;
      st_i = (st_i + 1 == NSTAGE) ? 0 : st_i + 1;
    };
    int kt = 0;
    for (; kt + D < KT; ++kt) k_step(kt, std::true_type{});
    for (; kt < KT; ++kt) k_step(kt, std::false_type{});
  } else {
    // Software pipeline at half-K-step granularity (needs the 3-stage ring): the fragments of the
    // NEXT eight MFMAs are always in flight while eight MFMAs run -
    //   read F1 = (tile kt, k-substeps 2,3) | MFMA F0 + DMA pieces of tile kt+2 | wait + barrier (tile
    //   kt+1 landed, every wave retired its reads of tile kt-1's stage) | read F0 = (tile kt+1, k-substeps
    //   0,1) | MFMA F1.
    // The profile that motivated it: MFMA, fragment reads and DMA issue of the plain loop add up
    // (0.44 + 0.35 + 0.22 of the step time) instead of overlapping.
    static_assert(NSTAGE == 3 && BK == 64, "the pipelined loop is written for the 3-stage ring, BK = 64");
    bf16x8 fa0[2][MI], fb0[2][NI], fa1[2][MI], fb1[2][NI];
    if (KT > 1) wait_vmcnt<LOADS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    load_frags(smem, 0, fa0[0], fb0[0]);
    load_frags(smem, 1, fa0[1], fb0[1]);
    int st_n = 1;  // stage of tile kt + 1
    auto k_step = [&](auto issue_tag, auto next_tag) {
      constexpr bool ISSUE = decltype(issue_tag)::value;  // tile kt+2 exists
      constexpr bool NEXT = decltype(next_tag)::value;    // tile kt+1 exists
      const char* sA = smem + st_c * STAGE;
      load_frags(sA, 2, fa1[0], fb1[0]);
      load_frags(sA, 3, fa1[1], fb1[1]);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(fa0[0], fb0[0]);
      if constexpr (ISSUE) {
#pragma unroll
        for (int idx = 0; idx < LOADS; idx += 2) issue_piece(st_i, idx);
      }
      mfmas(fa0[1], fb0[1]);
      if constexpr (ISSUE) {
#pragma unroll
        for (int idx = 1; idx < LOADS; idx += 2) issue_piece(st_i, idx);
        advance();
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NEXT) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of stage st_c's first half and of
                                                             // every older stage are retired before the barrier
        if constexpr (ISSUE) wait_vmcnt<LOADS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        const char* sN = smem + st_n * STAGE;
        load_frags(sN, 0, fa0[0], fb0[0]);
        load_frags(sN, 1, fa0[1], fb0[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfmas(fa1[0], fb1[0]);
      mfmas(fa1[1], fb1[1]);
      __builtin_amdgcn_sched_barrier(0);
      st_c = (st_c + 1 == NSTAGE) ? 0 : st_c + 1;
      st_n = (st_n + 1 == NSTAGE) ? 0 : st_n + 1;
      st_i = (st_i + 1 == NSTAGE) ? 0 : st_i + 1;
    };
    int kt = 0;
    for (; kt + 2 < KT; ++kt) k_step(std::true_type{}, std::true_type{});
    for (; kt + 1 < KT; ++kt) k_step(std::false_type{}, std::true_type{});
    for (; kt < KT; ++kt) k_step(std::false_type{}, std::false_type{});
  }

  stamp(3);
  // ---------------- epilogue ----------------
  // Latency structure (round 2: the first version loaded bias / LayerNorm vectors / residual inside the innermost
  // (mi, ni, gp) iteration, behind branches and behind the previous iteration's stores - ~50 serialized L2 round trips
  // per tile, 10 us that no K loop of 5-20 steps could hide).  Now every optional per-column vector is loaded
  // UNCONDITIONALLY (absent ones point at the zero page; clamped column index), once per 16-column block before its row
  // loop, together with the block's residual rows: one batch of loads per (ni, gp), i.e. 2 NI round trips per tile.
  const float scale = a.scale;
  const float* const zf = (const float*)a.zero;
  // ---- interior tiles (every row < M, column blocks of 16 wholly inside or outside n_end), bf16 / GEGLU output ----
  // What a short-K tile costs is the instruction count of its prologue + epilogue (PMC, profiles/r2_shortk_pmc_*.log:
  // ~1065 VALU + ~570 SALU per wave of a 64x64 wave tile at K = 64, against 16 MFMAs), so this path is written for
  // count: per-row pointers built once (lane column included; the (ni, gp) block offset is an instruction immediate),
  // no bounds masks, and the optional terms (folded LayerNorm, residual, time-embedding row, row statistics) are
  // compile-time flags of a generic lambda, selected by a wave-uniform branch.  Edge tiles, fp32 / pair-softmax /
  // split-K / sub-pixel outputs take the general epilogue below.
  // Hand-off form (cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "valid forms": {sc1 stores AND sc1 loads on both
  // sides} + every storing wave drains vmcnt + __syncthreads + ONE relaxed agent-scope ticket): no release / acquire fence
  // is needed because neither side ever holds the payload in a non-coherent cache line - the stores write through, the
  // reducer's loads bypass its L1.  tests/test_gpu_kernels.py::test_igemm_row_statistics_bit_stable_at_scale gates it.
  // Row statistics of the output (ln_out): every tile has stored its slots with write-through (sc1) stores; the row
  // block's last column tile to get here (one relaxed agent-scope ticket per workgroup - the hand-off of
  // norm.hip::gn_stats_kernel) reduces the block's N / 32 slots per row to (mean, rstd) in a FIXED order (8 threads per
  // row, fp64 for the cancellation): bit-reproducible whichever tile is last, no extra launch, and the consumers' tiles
  // read 8 bytes per row.
  auto store_slot = [&](float2* p, float s2, float q2) {
    const unsigned long long bits = ((unsigned long long)__float_as_uint(q2) << 32) | __float_as_uint(s2);
    __hip_atomic_store((unsigned long long*)p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto ln_out_finish = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's slot stores have been written through
    __syncthreads();                                     // (also: every wave is past the K loop - the ring is free)
    int* const flag = (int*)smem;
    if (tid == 0) {
      mg_handoff_release();
      const unsigned ticket = __hip_atomic_fetch_add(&a.ln_ctr[tile_m], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool last = ticket == (unsigned)(a.tiles_n - 1);
      if (last) {
        mg_handoff_acquire();
        __hip_atomic_store(&a.ln_ctr[tile_m], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      *flag = last ? 1 : 0;
    }
    __syncthreads();
    if (*flag == 0) return;
    const int slots = a.N >> 5;
    float2* const mr = a.ln_out + (long long)a.M * slots;
    static_assert(BM % (NT / 8) == 0, "whole passes of NT / 8 rows");
    for (int r = tid >> 3; r < BM; r += NT >> 3) {
      const int m = m0 + r, sub = tid & 7;
      double sd = 0.0, qd = 0.0;
      if (m < a.M) {   // up to 5 slots per thread (1280 channels): all in flight before any is consumed
        const unsigned long long* p = (const unsigned long long*)(a.ln_out + (long long)m * slots);
        for (int sl0 = sub; sl0 < slots; sl0 += 40) {
          unsigned long long v[5];
#pragma unroll
          for (int i = 0; i < 5; ++i)
            v[i] = sl0 + 8 * i < slots ? __hip_atomic_load(p + sl0 + 8 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            sd += (double)__uint_as_float((unsigned)v[i]);
            qd += (double)__uint_as_float((unsigned)(v[i] >> 32));
          }
        }
      }
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) { sd += __shfl_xor(sd, o); qd += __shfl_xor(qd, o); }
      if (sub == 0 && m < a.M) {   // fp64 for the cancellation E[x^2] - mean^2 only; 1/sqrt in fp32 (v_rsq_f32, ~1 ulp)
        const double mean = sd * a.inv_n;
        const float var = fmaxf((float)__builtin_fma(qd, a.inv_n, -mean * mean), 0.f);
        mr[m] = make_float2((float)mean, __builtin_amdgcn_rsqf(var + a.ln_eps));
      }
    }
  };
  // ---- MG_EPI_XATTN2: the collapsed 2-token cross-attention in one launch (tile 128 x 64, one wave = 32 whole rows) ----
  // Stage 1 is the pair-softmax epilogue; its packed probabilities of the (ni, gp) block ARE the pixel-side MFMA fragment
  // of k-block 2 ni + gp (lane = row, 8 consecutive k per half), so stage 2 (P x W2^T, K = 64) runs straight from
  // registers: per 32 output channels four weight fragments from global (L2-resident, next block prefetched), four MFMAs,
  // then the usual regroup + bias + residual + store.  The wave holds whole rows: (mean, rstd) of the new residual stream
  // come out directly.  In-place on the residual stream is safe - this tile's rows were staged before its last barrier and
  // no other workgroup touches them.
  if constexpr (!TRANS && WGN == 1 && TN == 64 && MI == 1) {
    if (a.epi == MG_EPI_XATTN2) {
      const int lrow = wm * TM + l31, m = m0 + lrow;
      const bool row_ok = m < a.M;
      const int mc = row_ok ? m : a.M - 1;
      float l_sc = scale, l_mr = 0.f;
      if (a.ln_in) {
        const float2 lst = lnst[lrow];
        l_sc = lst.y * scale;
        l_mr = -lst.y * lst.x;
      }
      const float* const pg = a.ln_in ? a.ln_g + 8 * half : zf;
      const float* const pc = a.ln_in ? a.ln_c + 8 * half : zf;
      bf16x8 pf[4];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int c = ni * 32 + gp * 16;
          const float4 g0 = *(const float4*)(pg + c), g1 = *(const float4*)(pg + c + 4);
          const float4 c0 = *(const float4*)(pc + c), c1 = *(const float4*)(pc + c + 4);
          const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
          const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
          float v[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) half_swap(acc[ni][0][8 * gp + j], acc[ni][0][8 * gp + 4 + j], v[j], v[4 + j]);
          uint32_t w4[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float s0 = __builtin_fmaf(v[2 * k], l_sc, __builtin_fmaf(l_mr, gg[2 * k], cc[2 * k])) * a.sm_scale;
            const float s1 = __builtin_fmaf(v[2 * k + 1], l_sc, __builtin_fmaf(l_mr, gg[2 * k + 1], cc[2 * k + 1])) * a.sm_scale;
            const float mx = fmaxf(s0, s1);
            const float e0 = __expf(s0 - mx), e1 = __expf(s1 - mx);
            const float inv = 1.0f / (e0 + e1);
            w4[k] = (c + 8 * half + 2 * k < a.sm_cols) ? pack2bf(e0 * inv, e1 * inv) : 0u;
          }
          pf[ni * 2 + gp] = __builtin_bit_cast(bf16x8, make_uint4(w4[0], w4[1], w4[2], w4[3]));
        }
      const bf16_t* const wrow = a.w2 + (long long)l31 * 64 + 8 * half;           // + cb * 2048 + kb * 16
      bf16_t* const po = (bf16_t*)a.out + (long long)mc * a.ldo + 8 * half;      // + cb * 32 + gp * 16
      const bf16_t* const pr = a.res ? a.res + (long long)mc * a.ldr + 8 * half : (const bf16_t*)zf;
      const int rstep = a.res ? 32 : 0;
      const float* const pb = (a.bias ? a.bias : zf) + 8 * half;
      const int bstep = a.bias ? 32 : 0;
      const int ncb = a.c2 >> 5;
      double sd = 0.0, qd = 0.0;
      // XG column blocks per pass, every load of the pass in flight before the first MFMA: one block's work is ~0.3 us
      // against ~2 us for its residual rows to arrive from HBM (first form, one block ahead: 2 us per block, 85 us for the
      // 40 blocks of a 1280-channel level on 45 workgroups)
      constexpr int XG = 4;
      for (int cb0 = 0; cb0 < ncb; cb0 += XG) {
        uint4 wf[XG][4], rr[XG][2];
        float4 bb[XG][4];
#pragma unroll
        for (int u = 0; u < XG; ++u) {
          const int cb = cb0 + u < ncb ? cb0 + u : ncb - 1;
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) wf[u][kb] = *(const uint4*)(wrow + (long long)cb * 2048 + kb * 16);
          rr[u][0] = *(const uint4*)(pr + cb * rstep);
          rr[u][1] = *(const uint4*)(pr + cb * rstep + (a.res ? 16 : 0));
          bb[u][0] = *(const float4*)(pb + cb * bstep);
          bb[u][1] = *(const float4*)(pb + cb * bstep + 4);
          bb[u][2] = *(const float4*)(pb + cb * bstep + (a.bias ? 16 : 0));
          bb[u][3] = *(const float4*)(pb + cb * bstep + (a.bias ? 20 : 4));
        }
#pragma unroll
        for (int u = 0; u < XG; ++u) {
          const int cb = cb0 + u;
          if (cb >= ncb) break;   // wave-uniform
          f32x16 acc2;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
            acc2 = mg_mfma32(__builtin_bit_cast(bf16x8, wf[u][kb]), pf[kb], acc2);
          float ps = 0.f, pq = 0.f;
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) half_swap(acc2[8 * gp + j], acc2[8 * gp + 4 + j], v[j], v[4 + j]);
            const float4 ba = bb[u][2 * gp], bc = bb[u][2 * gp + 1];
            const uint4 r4 = a.res ? rr[u][gp] : make_uint4(0, 0, 0, 0);
            v[0] += ba.x + bflo(r4.x); v[1] += ba.y + bfhi(r4.x); v[2] += ba.z + bflo(r4.y); v[3] += ba.w + bfhi(r4.y);
            v[4] += bc.x + bflo(r4.z); v[5] += bc.y + bfhi(r4.z); v[6] += bc.z + bflo(r4.w); v[7] += bc.w + bfhi(r4.w);
#pragma unroll
            for (int j = 0; j < 8; ++j) { ps += v[j]; pq = __builtin_fmaf(v[j], v[j], pq); }
            uint4 pk;
            pk.x = cvt_pk_bf16(v[0], v[1]); pk.y = cvt_pk_bf16(v[2], v[3]);
            pk.z = cvt_pk_bf16(v[4], v[5]); pk.w = cvt_pk_bf16(v[6], v[7]);
            if (row_ok) *(uint4*)(po + cb * 32 + gp * 16) = pk;
          }
          sd += (double)ps;
          qd += (double)pq;
        }
      }
      if (a.ln_out) {   // here: [M] (mean, rstd) rows of the output, written directly
        sd += __shfl_xor(sd, 32);
        qd += __shfl_xor(qd, 32);
        if (half == 0 && row_ok) {
          const double mean = sd * a.inv_c2;
          const float var = fmaxf((float)__builtin_fma(qd, a.inv_c2, -mean * mean), 0.f);
          a.ln_out[m] = make_float2((float)mean, __builtin_amdgcn_rsqf(var + a.ln_eps));
        }
      }
      return;
    }
  }
  bool single = a.splits <= 1;   // this workgroup holds the tile's complete sums
  if constexpr (!TRANS) {
    const bool interior = m0 + BM <= a.M && (a.n_end & 15) == 0 && single && !a.subpix;
    const int lrow = wm * TM + l31;          // the lane's first row inside the tile
    const int colw = n0 + wn * TN;           // the wave's first column
    if (interior && a.epi == MG_EPI_GEGLU) {
      auto geglu_fast = [&](auto LN) {
        constexpr bool kLN = decltype(LN)::value;
        const float* const pb = (a.bias ? a.bias + colw : zf) + 4 * half;
        const float* const pg = kLN ? a.ln_g + colw + 4 * half : zf;
        const float* const pc = kLN ? a.ln_c + colw + 4 * half : zf;
        bf16_t* po[MI];
        float l_sc[MI], l_mr[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          po[mi] = (bf16_t*)a.out + (long long)z * a.sO + (long long)(m0 + lrow + mi * 32) * a.ldo + (colw >> 1) + 8 * half;
          l_sc[mi] = scale;
          l_mr[mi] = 0.f;
          if constexpr (kLN) {
            const float2 lst = lnst[lrow + mi * 32];
            l_sc[mi] = lst.y * scale;
            l_mr[mi] = -lst.y * lst.x;
          }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          if (colw + ni * 32 >= a.n_end) break;   // wave-uniform
          float bq[4][4], gq[4][4];
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const float4 bv = *(const float4*)(pb + ni * 32 + 8 * q4);
            bq[q4][0] = bv.x; bq[q4][1] = bv.y; bq[q4][2] = bv.z; bq[q4][3] = bv.w;
            if constexpr (kLN) {
              const float4 gv = *(const float4*)(pg + ni * 32 + 8 * q4);
              const float4 cv = *(const float4*)(pc + ni * 32 + 8 * q4);
              bq[q4][0] += cv.x; bq[q4][1] += cv.y; bq[q4][2] += cv.z; bq[q4][3] += cv.w;
              gq[q4][0] = gv.x; gq[q4][1] = gv.y; gq[q4][2] = gv.z; gq[q4][3] = gv.w;
            }
          }
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            float r[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float o4[4];
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) {
                const float off = kLN ? __builtin_fmaf(l_mr[mi], gq[q4][j], bq[q4][j]) : bq[q4][j];
                o4[q4] = __builtin_fmaf(acc[ni][mi][4 * q4 + j], l_sc[mi], off);
              }
              const float o0 = o4[0] * gelu_poly_f(o4[2]);  // channel 16i + 4h + j
              const float o1 = o4[1] * gelu_poly_f(o4[3]);  // channel 16i + 8 + 4h + j
              half_swap(o0, o1, r[j], r[4 + j]);
            }
            uint4 pk;
            pk.x = cvt_pk_bf16(r[0], r[1]); pk.y = cvt_pk_bf16(r[2], r[3]);
            pk.z = cvt_pk_bf16(r[4], r[5]); pk.w = cvt_pk_bf16(r[6], r[7]);
            *(uint4*)(po[mi] + ni * 16) = pk;
          }
        }
      };
      if (a.ln_in) geglu_fast(std::true_type{});
      else geglu_fast(std::false_type{});
      return;
    }
    if (interior && a.epi == MG_EPI_BF16) {
      auto bf16_fast = [&](auto LN, auto RES, auto RV, auto LNO) {
        constexpr bool kLN = decltype(LN)::value, kRES = decltype(RES)::value, kRV = decltype(RV)::value, kLNO = decltype(LNO)::value;
        const float* const pb = (a.bias ? a.bias + colw : zf) + 8 * half;
        const float* const pg = kLN ? a.ln_g + colw + 8 * half : zf;
        const float* const pc = kLN ? a.ln_c + colw + 8 * half : zf;
        bf16_t* po[MI];
        const bf16_t* pr[MI];
        const float* pv[MI];
        float2* pl[MI];
        float l_sc[MI], l_mr[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int m = m0 + lrow + mi * 32;
          po[mi] = (bf16_t*)a.out + (long long)z * a.sO + (long long)m * a.ldo + colw + 8 * half;
          if constexpr (kRES) pr[mi] = a.res + (long long)z * a.sR + (long long)m * a.ldr + colw + 8 * half;
          if constexpr (kRV) pv[mi] = a.rowvec + (long long)fdiv(m, a.fd_rpi) * a.rv_stride + colw + 8 * half;
          if constexpr (kLNO) pl[mi] = a.ln_out + (long long)m * (a.N >> 5) + (colw >> 5);
          l_sc[mi] = scale;
          l_mr[mi] = 0.f;
          if constexpr (kLN) {
            const float2 lst = lnst[lrow + mi * 32];
            l_sc[mi] = lst.y * scale;
            l_mr[mi] = -lst.y * lst.x;
          }
        }
        // residual rows of EVERY block up front where the registers allow (<= 12 x 16 bytes per lane): one HBM round trip
        // per tile instead of one per 16-column block (the loads of a block cannot pass the previous block's stores - the
        // residual may alias the output - so they were 2 NI dependent round trips of ~1.5 us each)
        constexpr bool PRE = kRES && NI * 2 * MI <= 12;
        uint4 rpre[PRE ? NI * 2 : 1][MI];
        if constexpr (PRE) {
#pragma unroll
          for (int b = 0; b < NI * 2; ++b) {
            const int c = (b >> 1) * 32 + (b & 1) * 16;
            const bool in = colw + c < a.n_end;   // wave-uniform
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) rpre[b][mi] = in ? *(const uint4*)(pr[mi] + c) : make_uint4(0, 0, 0, 0);
          }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          float ps[MI], pq[MI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) ps[mi] = pq[mi] = 0.f;
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            constexpr int dummy = 0; (void)dummy;
            const int c = ni * 32 + gp * 16;              // compile-time after unrolling: an immediate offset
            if (colw + c >= a.n_end) break;               // wave-uniform
            const float4 b0 = *(const float4*)(pb + c), b1 = *(const float4*)(pb + c + 4);
            float cb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            float gg[8];
            if constexpr (kLN) {
              const float4 g0 = *(const float4*)(pg + c), g1 = *(const float4*)(pg + c + 4);
              const float4 c0 = *(const float4*)(pc + c), c1 = *(const float4*)(pc + c + 4);
              gg[0] = g0.x; gg[1] = g0.y; gg[2] = g0.z; gg[3] = g0.w; gg[4] = g1.x; gg[5] = g1.y; gg[6] = g1.z; gg[7] = g1.w;
              cb[0] += c0.x; cb[1] += c0.y; cb[2] += c0.z; cb[3] += c0.w; cb[4] += c1.x; cb[5] += c1.y; cb[6] += c1.z; cb[7] += c1.w;
            }
            uint4 rr[MI];
            float4 rv0[MI], rv1[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
              if constexpr (kRES) {
                if constexpr (PRE) rr[mi] = rpre[ni * 2 + gp][mi];
                else rr[mi] = *(const uint4*)(pr[mi] + c);
              }
              if constexpr (kRV) { rv0[mi] = *(const float4*)(pv[mi] + c); rv1[mi] = *(const float4*)(pv[mi] + c + 4); }
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
              float v[8];
#pragma unroll
              for (int j = 0; j < 4; ++j) half_swap(acc[ni][mi][8 * gp + j], acc[ni][mi][8 * gp + 4 + j], v[j], v[4 + j]);
#pragma unroll
              for (int j = 0; j < 8; ++j)
                v[j] = __builtin_fmaf(v[j], l_sc[mi], kLN ? __builtin_fmaf(l_mr[mi], gg[j], cb[j]) : cb[j]);
              if constexpr (kRV) {
                v[0] += rv0[mi].x; v[1] += rv0[mi].y; v[2] += rv0[mi].z; v[3] += rv0[mi].w;
                v[4] += rv1[mi].x; v[5] += rv1[mi].y; v[6] += rv1[mi].z; v[7] += rv1[mi].w;
              }
              if constexpr (kRES) {
                const uint4 r4 = rr[mi];
                v[0] += bflo(r4.x); v[1] += bfhi(r4.x); v[2] += bflo(r4.y); v[3] += bfhi(r4.y);
                v[4] += bflo(r4.z); v[5] += bfhi(r4.z); v[6] += bflo(r4.w); v[7] += bfhi(r4.w);
              }
              if constexpr (kLNO) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { ps[mi] += v[j]; pq[mi] = __builtin_fmaf(v[j], v[j], pq[mi]); }
              }
              uint4 pk;
              pk.x = cvt_pk_bf16(v[0], v[1]); pk.y = cvt_pk_bf16(v[2], v[3]);
              pk.z = cvt_pk_bf16(v[4], v[5]); pk.w = cvt_pk_bf16(v[6], v[7]);
              *(uint4*)(po[mi] + c) = pk;
            }
          }
          if constexpr (kLNO) {   // both halves of the lane pair hold 16 of the row's 32 columns (N % 32 == 0: whole blocks)
            if (colw + ni * 32 < a.n_end) {
#pragma unroll
              for (int mi = 0; mi < MI; ++mi) {
                const float s2 = ps[mi] + __shfl_xor(ps[mi], 32), q2 = pq[mi] + __shfl_xor(pq[mi], 32);
                if (half == 0) store_slot(pl[mi] + ni, s2, q2);
              }
            }
          }
        }
      };
      using T = std::true_type;
      using F = std::false_type;
      const bool fLN = a.ln_in != nullptr, fRES = a.res != nullptr, fRV = a.rowvec != nullptr, fLNO = a.ln_out != nullptr;
      const int key = (fLN ? 1 : 0) | (fRES ? 2 : 0) | (fRV ? 4 : 0) | (fLNO ? 8 : 0);
      bool done = true;
      switch (key) {
        case 0: bf16_fast(F{}, F{}, F{}, F{}); break;    // projections, shortcuts
        case 1: bf16_fast(T{}, F{}, F{}, F{}); break;    // Linear on LayerNorm(x), folded
        case 2: bf16_fast(F{}, T{}, F{}, F{}); break;    // + residual
        case 10: bf16_fast(F{}, T{}, F{}, T{}); break;   // + residual, row statistics for the next folded LayerNorm
        case 8: bf16_fast(F{}, F{}, F{}, T{}); break;    // proj_in: row statistics
        case 4: bf16_fast(F{}, F{}, T{}, F{}); break;    // resnet conv1: + time-embedding row
        default: done = false;
      }
      if (done) {
        stamp(4);
        if (fLNO) ln_out_finish();
        stamp(5);
        return;
      }
    }
  }
  if constexpr (!TRANS) {
    if (a.epi == MG_EPI_GEGLU) {
      // Weight rows are interleaved in 32-row groups (weights.py::pack_geglu): rows [32i,32i+16) =
      // u(16i..16i+15), rows [32i+16,32i+32) = their gates.  acc groups g = 0,1 are u(16i+8g+4h+j),
      // g = 2,3 the gates of the same channels; a lane^32 exchange then leaves 8 consecutive output
      // channels per lane -> one 16-byte store per (mi, ni).
      const float* const pb = a.bias ? a.bias : zf;
      const float* const pg = a.ln_in ? a.ln_g : zf;
      const float* const pc = a.ln_in ? a.ln_c : zf;
      float l_sc[MI], l_mr[MI];   // rstd * scale, -mean * rstd of the lane's rows
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const float2 lst = a.ln_in ? lnst[wm * TM + mi * 32 + l31] : make_float2(0.f, 1.f);
        l_sc[mi] = lst.y * scale;
        l_mr[mi] = -lst.y * lst.x;
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int nb = n0 + wn * TN + ni * 32;
        const bool nok = nb < a.n_end;
        const int nu0 = (nok ? nb : n0) + 4 * half;   // clamped: the loads below are unconditional
        float bq[4][4], gq[4][4], cq[4][4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 bv = *(const float4*)(pb + nu0 + 8 * q4);
          const float4 gv = *(const float4*)(pg + nu0 + 8 * q4);
          const float4 cv = *(const float4*)(pc + nu0 + 8 * q4);
          bq[q4][0] = bv.x + cv.x; bq[q4][1] = bv.y + cv.y; bq[q4][2] = bv.z + cv.z; bq[q4][3] = bv.w + cv.w;
          gq[q4][0] = gv.x; gq[q4][1] = gv.y; gq[q4][2] = gv.z; gq[q4][3] = gv.w;
          (void)cq;
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int m = m0 + wm * TM + mi * 32 + l31;
          float r[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // folded LayerNorm: rstd * (acc - mean g) + c = (rstd scale) acc + (c - mean rstd g); the Linear's own bias
            // is part of c there, and the only offset otherwise
            const float u0 = __builtin_fmaf(acc[ni][mi][j], l_sc[mi], __builtin_fmaf(l_mr[mi], gq[0][j], bq[0][j]));
            const float u1 = __builtin_fmaf(acc[ni][mi][4 + j], l_sc[mi], __builtin_fmaf(l_mr[mi], gq[1][j], bq[1][j]));
            const float t0 = __builtin_fmaf(acc[ni][mi][8 + j], l_sc[mi], __builtin_fmaf(l_mr[mi], gq[2][j], bq[2][j]));
            const float t1 = __builtin_fmaf(acc[ni][mi][12 + j], l_sc[mi], __builtin_fmaf(l_mr[mi], gq[3][j], bq[3][j]));
            const float o0 = u0 * gelu_poly_f(t0);  // channel 16i + 4h + j
            const float o1 = u1 * gelu_poly_f(t1);  // channel 16i + 8 + 4h + j
            half_swap(o0, o1, r[j], r[4 + j]);
          }
          if (m < a.M && nok) {
            const int oc = (nb >> 1) + 8 * half;  // 16 output channels per 32 weight rows
            uint4 pk;
            pk.x = cvt_pk_bf16(r[0], r[1]); pk.y = cvt_pk_bf16(r[2], r[3]);
            pk.z = cvt_pk_bf16(r[4], r[5]); pk.w = cvt_pk_bf16(r[6], r[7]);
            *(uint4*)((bf16_t*)a.out + (long long)z * a.sO + (long long)m * a.ldo + oc) = pk;
          }
        }
      }
      return;
    }
  }
  if constexpr (!TRANS) {
    // lane^32 exchange: afterwards v[0..7] are 8 consecutive output channels starting at 16*gp + 8*half of pixel l31
    const float* const pb = a.bias ? a.bias : zf;
    const float* const pg = a.ln_in ? a.ln_g : zf;
    const float* const pc = a.ln_in ? a.ln_c : zf;
    const float* const prv = a.rowvec ? a.rowvec : zf;
    const int rvs = a.rowvec ? a.rv_stride : 0;
    float l_sc[MI], l_mr[MI];
    long long e_row[MI];   // output row of the lane's pixel (identity unless sub-pixel)
    int e_img[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const float2 lst = a.ln_in ? lnst[wm * TM + mi * 32 + l31] : make_float2(0.f, 1.f);
      l_sc[mi] = lst.y * (single ? scale : 1.f);
      l_mr[mi] = -lst.y * lst.x;
      const int m = m0 + wm * TM + mi * 32 + l31;
      const int mc = m < a.M ? m : a.M - 1;          // clamped: loads stay in range, stores are masked
      const int img = fdiv(mc, a.fd_rpi);
      e_img[mi] = img;
      e_row[mi] = mc;
      if (a.subpix) {   // low-resolution pixel (img, y, x) of parity (a, b) -> pixel (2y + a, 2x + b) of the 2H x 2W map
        const int rem = mc - img * a.rows_per_img;
        const int y = fdiv(rem, a.fd_wo), x = rem - y * a.Wo;
        e_row[mi] = (long long)img * 4 * a.rows_per_img + (long long)(2 * y + (z >> 1)) * (2 * a.Wo) + 2 * x + (z & 1);
      }
    }
    float ps[MI], pq[MI];   // ln_out: the lane's share of (sum, sum of squares) of its rows over the current 32 columns
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int nb = n0 + wn * TN + ni * 32;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) ps[mi] = pq[mi] = 0.f;
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const int n = nb + 16 * gp + 8 * half;
        const bool nok = n < a.n_end;
        const int nc = nok ? n : n0;   // clamped column for the unconditional loads
        if (!single) {  // split-K: raw partial sums, everything else happens in splitk_reduce_kernel
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) half_swap(acc[ni][mi][8 * gp + j], acc[ni][mi][8 * gp + 4 + j], v[j], v[4 + j]);
            const int m = m0 + wm * TM + mi * 32 + l31;
            if (m < a.M && nok) {
              float* o = a.ws + ((long long)split * a.M + m) * a.N + n;
              *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
              *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
          }
          continue;
        }
        // ---- the block's loads, all issued before anything is consumed ----
        const float4 b0 = *(const float4*)(pb + nc), b1 = *(const float4*)(pb + nc + 4);
        const float4 g0 = *(const float4*)(pg + nc), g1 = *(const float4*)(pg + nc + 4);
        const float4 c0 = *(const float4*)(pc + nc), c1 = *(const float4*)(pc + nc + 4);
        uint4 rr[MI];
        float4 rv0[MI], rv1[MI];
        const bool has_res = a.res != nullptr && a.epi == MG_EPI_BF16;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          rr[mi] = has_res ? *(const uint4*)(a.res + (long long)z * a.sR + e_row[mi] * a.ldr + nc) : make_uint4(0, 0, 0, 0);
          const float* rv = prv + (long long)e_img[mi] * rvs + (a.rowvec ? nc : 0);
          rv0[mi] = *(const float4*)rv;
          rv1[mi] = *(const float4*)(rv + 4);
        }
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float cb[8] = {c0.x + b0.x, c0.y + b0.y, c0.z + b0.z, c0.w + b0.w, c1.x + b1.x, c1.y + b1.y, c1.z + b1.z, c1.w + b1.w};
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) half_swap(acc[ni][mi][8 * gp + j], acc[ni][mi][8 * gp + 4 + j], v[j], v[4 + j]);
          const float rvv[8] = {rv0[mi].x, rv0[mi].y, rv0[mi].z, rv0[mi].w, rv1[mi].x, rv1[mi].y, rv1[mi].z, rv1[mi].w};
          // scale, folded LayerNorm (rstd acc - mean rstd g + c), bias, time-embedding row: two fused multiply-adds + an add
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf(v[j], l_sc[mi], __builtin_fmaf(l_mr[mi], gg[j], cb[j])) + rvv[j];
          const int m = m0 + wm * TM + mi * 32 + l31;
          const bool ok = m < a.M && nok;
          if (a.epi == MG_EPI_F32) {
            if (ok) {
              float* o = (float*)a.out + (long long)z * a.sO + e_row[mi] * a.ldo + n;
              *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
              *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
          } else if (a.epi == MG_EPI_SOFTMAX2) {
            // 2-key softmax of the collapsed cross-attention (columns 2h, 2h+1 = the two context tokens of head h) taken
            // on the accumulators: the fp32 scores never reach HBM and no softmax launch follows.  Pad columns -> 0.
            uint32_t w4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float s0 = v[2 * k] * a.sm_scale, s1 = v[2 * k + 1] * a.sm_scale;
              const float mx = fmaxf(s0, s1);
              const float e0 = __expf(s0 - mx), e1 = __expf(s1 - mx);
              const float inv = 1.0f / (e0 + e1);
              w4[k] = (n + 2 * k < a.sm_cols) ? pack2bf(e0 * inv, e1 * inv) : 0u;
            }
            if (ok) *(uint4*)((bf16_t*)a.out + (long long)z * a.sO + e_row[mi] * a.ldo + n) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
          } else {
            const uint4 r4 = rr[mi];
            v[0] += bflo(r4.x); v[1] += bfhi(r4.x); v[2] += bflo(r4.y); v[3] += bfhi(r4.y);
            v[4] += bflo(r4.z); v[5] += bfhi(r4.z); v[6] += bflo(r4.w); v[7] += bfhi(r4.w);
            if (a.ln_out && ok) {
#pragma unroll
              for (int j = 0; j < 8; ++j) { ps[mi] += v[j]; pq[mi] = __builtin_fmaf(v[j], v[j], pq[mi]); }
            }
            uint4 pk;
            pk.x = cvt_pk_bf16(v[0], v[1]); pk.y = cvt_pk_bf16(v[2], v[3]);
            pk.z = cvt_pk_bf16(v[4], v[5]); pk.w = cvt_pk_bf16(v[6], v[7]);
            if (ok) *(uint4*)((bf16_t*)a.out + (long long)z * a.sO + e_row[mi] * a.ldo + n) = pk;
          }
        }
      }
      if (a.ln_out && single) {   // both halves of the lane pair hold 16 of the row's 32 columns
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const float s2 = ps[mi] + __shfl_xor(ps[mi], 32), q2 = pq[mi] + __shfl_xor(pq[mi], 32);
          const int m = m0 + wm * TM + mi * 32 + l31;
          if (half == 0 && m < a.M && nb < a.n_end) store_slot(a.ln_out + (long long)m * (a.N >> 5) + (nb >> 5), s2, q2);
        }
      }
    }
    if (a.ln_out && single) ln_out_finish();
  } else {
    // transposed section (V^T of the fused QKV projection): out[z][img][n][tok], 8 consecutive tokens per lane; the
    // lane's column n = nb + l31 is fixed per ni: its bias / LayerNorm constants are loaded once per ni
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int nb = n0 + wn * TN + ni * 32;
      const int n = nb + l31;
      const bool nok = n < a.n_end;
      const int nc = nok ? n : n0;
      const float bv = a.bias ? a.bias[nc] : 0.f;
      const float gn = a.ln_in ? a.ln_g[nc] : 0.f, cn = (a.ln_in ? a.ln_c[nc] : 0.f) + bv;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int mb = m0 + wm * TM + mi * 32;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          float v[8];
          if (a.tperm) {   // accumulator order: v[j] = token 16 gp + 4 half + j (j < 4), 16 gp + 8 + 4 half + j - 4 (j >= 4)
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = acc[ni][mi][8 * gp + j]; v[4 + j] = acc[ni][mi][8 * gp + 4 + j]; }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) half_swap(acc[ni][mi][8 * gp + j], acc[ni][mi][8 * gp + 4 + j], v[j], v[4 + j]);
          }
          const int m = mb + 16 * gp + 8 * half;   // position of v[0] in the row of tokens (and its token, natural order)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int tj = a.tperm ? mb + 16 * gp + 4 * half + j + (j >= 4 ? 4 : 0) : m + j;   // the token v[j] belongs to
            const float2 lst = a.ln_in ? lnst[tj - m0] : make_float2(0.f, 1.f);
            v[j] = __builtin_fmaf(v[j] * scale, lst.y, __builtin_fmaf(-lst.y * lst.x, gn, cn));
          }
          if (nok && m < a.M) {
            const int img = fdiv(m, a.fd_rpi);
            const int tok = m - img * a.rows_per_img;
            if (tok + 8 <= a.rows_per_img && (a.rows_per_img & 7) == 0) {
              bf16_t* o = (bf16_t*)a.out + (long long)z * a.sO + ((long long)img * a.ctr + n) * a.ldt + tok;
              uint4 pk;
              pk.x = cvt_pk_bf16(v[0], v[1]); pk.y = cvt_pk_bf16(v[2], v[3]);
              pk.z = cvt_pk_bf16(v[4], v[5]); pk.w = cvt_pk_bf16(v[6], v[7]);
              *(uint4*)o = pk;
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const int mj = m + j;
                if (mj < a.M) {
                  const int im = fdiv(mj, a.fd_rpi);
                  const int tk = mj - im * a.rows_per_img;
                  ((bf16_t*)a.out)[(long long)z * a.sO + ((long long)im * a.ctr + n) * a.ldt + tk] = f2bf(v[j]);
                }
              }
            }
          }
        }
      }
    }
  }
}

}  // namespace
