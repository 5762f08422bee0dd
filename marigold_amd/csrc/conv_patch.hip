// Patch-resident 3x3 convolution (MG_OP_CONV3X3): the stride-1 / pad-1 convolutions of the ResNet blocks
// (diffusers ResnetBlock2D conv1 / conv2, Upsample2D conv - reached from marigold/marigold_depth_pipeline.py:461-463,
// 491-492, 512-513) with the GroupNorm scale/shift + SiLU of the block fused into the operand staging, the skip
// concat of the UNet's up blocks folded into the channel loop, and nearest-2x up-sampling in sub-pixel form.
//
// Why a second convolution kernel beside the implicit GEMM of igemm2.hip.  The implicit GEMM streams one 64-channel
// K tile per (tap, channel tile): every input pixel travels global -> LDS nine times (once per tap), which is what
// (a) made the taps re-fetch from HBM once the activation outgrew L2/MALL (round 1: 2.1x the algorithmic traffic),
// (b) costs nine LDS-DMA pieces per pixel row and (c) rules out touching the operand on its way in (LDS-DMA bypasses
// the registers, and a fix-up per staged tile would repeat the GroupNorm + SiLU arithmetic nine times).  Here a
// workgroup owns a TH x TW tile of output pixels and keeps the (TH+2) x (TW+2) input patch of one 64-channel tile
// RESIDENT in LDS while all nine taps run over it: each input pixel is staged once per channel tile (1.27x instead
// of 9x for 16 x 16), the nine taps are nine shifted fragment addresses into the same patch ("im2col" happens in the
// ds_read addresses of the wavefront), and the GroupNorm affine + SiLU is applied ONCE per staged element, in place,
// by the waves between the MFMA groups of the previous channel tile.
//
//   LDS   [patch buffer 0 | patch buffer 1 | weight ring: NSTB stages of BN rows x 128 B]
//   K loop  for channel tile c:  for tap t:  { wait + barrier | patch DMA of tile c+1 (taps 0-2) | fix-up slices of
//           patch c+1 (taps 3-5) | 4 k-substeps: fragment reads (patch rows shifted by the tap, weight rows), the next
//           weight tile's LDS-DMA pieces, MFMAs }
//   weights k = tap * Cin + c (the igemm layout); sub-pixel mode: four 2x2 windows, [z][N][4*Cin] (weights.py).
// Numerics are those of the unfused chain: the fix-up computes silu(x * scale + shift) in fp32 and rounds it to bf16
// exactly as gn_apply_kernel does, the MFMAs accumulate the same products in fp32 (channel tile outermost instead of
// tap outermost, so sums differ in the last fp32 bits only).
#include <type_traits>

#include <stdlib.h>

#include "common.h"
#include "conv_patch_args.h"

namespace {


template <int N>
__device__ __forceinline__ void cp_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// TH x TW output pixels x BN output channels per workgroup; WGM x WGN waves (wave tile TM x TN); NSTB weight stages.
// GNS: the epilogue also leaves the tile's GroupNorm partial sums of the output (ConvPArgs::gn_part) - what MG_OP_GN_STATS would
// compute by re-reading the tensor (4.6 ms per VAE decode at E = 10, HBM-bound); on the 12-wave tiles the extra VALU hides
// under the other waves' MFMAs.  Per wave: sums over its pixels in registers, a 32-lane butterfly, one LDS row per wave; the
// waves' rows are added in wave order (bit-reproducible) and the tile's groups written as one table slot.
template <int TH, int TW, int BN, int WGM, int WGN, int NSTB, bool FIX, bool GNS = false>
__global__ __launch_bounds__(WGM* WGN * 64) void conv_patch_kernel(const ConvPArgs a) {
  constexpr int NW = WGM * WGN, NT = NW * 64;
  constexpr int BM = TH * TW, TM = BM / WGM, TN = BN / WGN, MI = TM / 32, NI = TN / 32;
  constexpr int ROWB = 128;                                  // bytes per LDS row: 64 bf16 channels
  constexpr int PRMAX = ((TH + 2) * (TW + 2) + 7) / 8 * 8;   // patch rows, padded to whole 8-row DMA pieces
  constexpr int NPW = PRMAX / 8;                             // wave-pieces (64 lanes x 16 B = 8 rows) per patch
  constexpr int NSLOT = (NPW + NW - 1) / NW;                 // patch pieces per wave
  constexpr int PATCH = PRMAX * ROWB;
  constexpr int B_IT = (BN * 8 + NT - 1) / NT;               // weight pieces per wave and K step (the last one only for
  constexpr bool B_EVEN = (BN * 8) % NT == 0;                // the first waves when the stage does not divide evenly)
  constexpr int BSTAGE = BN * ROWB;
  constexpr int D = NSTB - 1;
  constexpr int NFIX = (PRMAX * 8 + NT - 1) / NT;            // fix-up vectors per thread
  static_assert(TW == 16 && BM % (32 * WGM) == 0 && BN % (32 * WGN) == 0 && (BN * 8) % 64 == 0, "tile geometry");
  static_assert(B_EVEN || NSTB == 2, "an uneven weight stage needs the complete (vmcnt(0)) waits of the two-stage ring");
  static_assert(NSTB >= 2 && NSTB <= 3, "2 or 3 weight stages");
  static_assert((NT / 8) % 16 == 0, "the fix-up's channel group must not depend on the iteration");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const patch0 = smem;
  char* const ring = smem + 2 * PATCH;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, half = lane >> 5;

  // ---- which tile: output channels fastest (the blocks that share an input patch are neighbours on one XCD) ----
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = bid % a.tiles_n; bid /= a.tiles_n;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; bid /= a.tiles_y;
  const int img = bid % a.B;
  const int z = bid / a.B;                                   // output parity 2a+b in sub-pixel mode, else 0
  const int n0 = tile_n * BN, y0 = ty * TH, x0 = tx * TW;
  const int pad_y = a.subpix ? 1 - (z >> 1) : 1, pad_x = a.subpix ? 1 - (z & 1) : 1;
  const int PW = TW + a.tw - 1, PR = (TH + a.tw - 1) * PW;
  const bf16_t* __restrict__ Wb = a.Wt + (long long)z * a.sW;
  const char* zero = (const char*)a.zero;

  // ---- patch staging slots of this thread: LDS row r, 16-byte slot ps <- channel group j of input pixel (iy, ix) ----
  int p_off[NSLOT];     // pixel index (img * H + iy) * W + ix, or -1 (outside the image / beyond the patch)
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) {
    const int i = k * NW + wave;
    const int r = i * 8 + (lane >> 3);
    const int pr = r / PW, pc = r - pr * PW;
    const int iy = y0 - pad_y + pr, ix = x0 - pad_x + pc;
    const bool ok = i < NPW && r < PR && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    p_off[k] = ok ? (img * a.H + iy) * a.W + ix : -1;
  }
  // XOR swizzle of the 16-byte slot (applied on the DMA source, undone by the fragment read): slot ps of row r holds
  // channel group ps ^ ((r >> 1) & 7)
  auto slot_j8 = [&](int k) {
    const int i = k * NW + wave;
    const int r = i * 8 + (lane >> 3);
    return ((lane & 7) ^ ((r >> 1) & 7)) * 8;
  };
  auto issue_patch_slot = [&](int k, int c, int pb) {
    const int i = k * NW + wave;
    if (i >= NPW) return;                                   // wave-uniform
    const char* src = zero;
    if (p_off[k] >= 0) {
      const int j8 = slot_j8(k);
      src = c < a.c0t ? (const char*)(a.A0 + (long long)p_off[k] * a.lda0 + c * 64 + j8)
                      : (const char*)(a.A1 + (long long)p_off[k] * a.lda1 + (c - a.c0t) * 64 + j8);
    }
    glds16(src, patch0 + pb * PATCH + i * 1024);
  };

  // ---- weight staging rows of this thread ----
  // (uneven stages - the 12 / 6-wave 320-channel tiles - keep ONE pointer: piece `it` is NT / 8 weight rows further on, the
  // swizzle term (r >> 1) & 7 is the same for it * NT / 8 rows later (NT / 8 % 16 == 0), and N % BN == 0 is required)
  const char* b_base[B_EVEN ? B_IT : 1];
#pragma unroll
  for (int it = 0; it < (B_EVEN ? B_IT : 1); ++it) {
    const int ci = it * NT + tid;
    const int r = ci >> 3, p = ci & 7;
    const int n = n0 + r;
    b_base[it] = n < a.N ? (const char*)(Wb + (long long)n * a.ldw + (p ^ ((r >> 1) & 7)) * 8) : zero;
  }
  const long long b_stride = (long long)(NT / 8) * a.ldw * 2;
  auto issue_b_piece = [&](int it, long long koff_bytes, int stage) {
    if (!B_EVEN && it * NT + wave * 64 >= BN * 8) return;   // wave-uniform
    long long bs = b_stride;
    if constexpr (!B_EVEN) asm volatile("" : "+s"(bs));   // recomputed at the use: hoisted, the B_IT addresses are spilled (168-register budget)
    const char* src = B_EVEN ? b_base[it] : b_base[0] + it * bs;
    glds16(src + koff_bytes, ring + stage * BSTAGE + (it * NT + wave * 64) * 16);
  };
  auto koff_of = [&](int c, int t) { return ((long long)t * a.Cin + c * 64) * 2; };

  // ---- fragment rows ----
  int prow0[MI], rowB[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int p = wm * TM + mi * 32 + l31;
    prow0[mi] = (p >> 4) * PW + (p & 15);
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) rowB[ni] = wn * TN + ni * 32 + l31;

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  // ---- fix-up (fused GroupNorm affine + SiLU on the staged patch, in place) ----
  // vector e = it NT + tid: row e >> 3, slot e & 7 = tid & 7, channel group (tid & 7) ^ ((row >> 1) & 7) where
  // (row >> 1) & 7 = (tid >> 4) & 7 for every it (NT / 8 is a multiple of 16): one set of 8 scales / shifts per thread
  unsigned fix_mask = 0;     // bit it: the row is an in-image pixel (padding rows must stay zero)
  const int fix_j8 = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;
  if constexpr (FIX) {
#pragma unroll
    for (int it = 0; it < NFIX; ++it) {
      const int r = (it * NT + tid) >> 3;
      const int pr = r / PW, pc = r - pr * PW;
      const int iy = y0 - pad_y + pr, ix = x0 - pad_x + pc;
      if (r < PR && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) fix_mask |= 1u << it;
    }
  }
  // SSL (the 12-wave 320-channel tile, 168 registers per lane): the image's scale / shift vectors live in LDS behind the
  // weight ring and are read at the use - held in registers across the K steps they cost 16 of the budget (18 spilled
  // registers, 7 scratch reloads per K step before)
  constexpr bool SSL = FIX && TH == 12 && BN == 320;
  float* const ssl = (float*)(smem + 2 * PATCH + NSTB * BSTAGE);   // [2][Cin]
  if constexpr (SSL) {
    const float* src = a.ss + (long long)img * 2 * a.Cin;
    for (int i = tid; i < 2 * a.Cin; i += NT) ssl[i] = src[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // published by the barrier ahead of the first fix-up
  }
  float fsc[8], fsh[8];
  int fix_c = 0;   // SSL: channel tile the next fix_vec calls belong to
  auto fix_load = [&](int c) {
    if constexpr (SSL) { fix_c = c; return; }
    const float* scp = a.ss + (long long)img * 2 * a.Cin + c * 64 + fix_j8;
    const float4 s0 = *(const float4*)scp, s1 = *(const float4*)(scp + 4);
    const float4 h0 = *(const float4*)(scp + a.Cin), h1 = *(const float4*)(scp + a.Cin + 4);
    fsc[0] = s0.x; fsc[1] = s0.y; fsc[2] = s0.z; fsc[3] = s0.w; fsc[4] = s1.x; fsc[5] = s1.y; fsc[6] = s1.z; fsc[7] = s1.w;
    fsh[0] = h0.x; fsh[1] = h0.y; fsh[2] = h0.z; fsh[3] = h0.w; fsh[4] = h1.x; fsh[5] = h1.y; fsh[6] = h1.z; fsh[7] = h1.w;
  };
  auto fix_vec = [&](int it, int pb) {
    if (!((fix_mask >> it) & 1u)) return;
    uint4* p = (uint4*)(patch0 + pb * PATCH + (it * NT + tid) * 16);
    const uint4 u = *p;
    float v[8] = {bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y), bflo(u.z), bfhi(u.z), bflo(u.w), bfhi(u.w)};
    if constexpr (SSL) {
      const float* scp = ssl + fix_c * 64 + fix_j8;
      const float4 s0 = *(const float4*)scp, s1 = *(const float4*)(scp + 4);
      const float4 h0 = *(const float4*)(scp + a.Cin), h1 = *(const float4*)(scp + a.Cin + 4);
      fsc[0] = s0.x; fsc[1] = s0.y; fsc[2] = s0.z; fsc[3] = s0.w; fsc[4] = s1.x; fsc[5] = s1.y; fsc[6] = s1.z; fsc[7] = s1.w;
      fsh[0] = h0.x; fsh[1] = h0.y; fsh[2] = h0.z; fsh[3] = h0.w; fsh[4] = h1.x; fsh[5] = h1.y; fsh[6] = h1.z; fsh[7] = h1.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = __builtin_fmaf(v[j], fsc[j], fsh[j]);
      if (a.silu) v[j] = silu_fast_f(v[j]);
    }
    uint4 o;
    o.x = cvt_pk_bf16_f32(v[0], v[1]); o.y = cvt_pk_bf16_f32(v[2], v[3]);
    o.z = cvt_pk_bf16_f32(v[4], v[5]); o.w = cvt_pk_bf16_f32(v[6], v[7]);
    *p = o;
  };

  const int T = a.T, chunks = a.chunks;
  const int S = chunks * T;

  // ---- prologue: patch of channel tile 0, the first D weight tiles ----
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) issue_patch_slot(k, 0, 0);
  {
    int c = 0, t = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (d < S) {
#pragma unroll
        for (int it = 0; it < B_IT; ++it) issue_b_piece(it, koff_of(c, t), d);
      }
      if (++t == T) { t = 0; ++c; }
    }
  }
  if constexpr (FIX) {
    fix_load(0);
    cp_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int it = 0; it < NFIX; ++it) fix_vec(it, 0);
    // the K loop's first barrier (after its lgkmcnt-complete wait) publishes the fixed patch
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  // K-step state: (c, t) of the step being computed, (ci, ti) of the weight tile to issue (D steps ahead)
  int c = 0, t = 0, ci = 0, ti = 0;
  for (int d = 0; d < D; ++d) { if (++ti == T) { ti = 0; ++ci; } }
  int st_c = 0, st_i = D % NSTB;

  auto k_step = [&](int s, auto issue_tag) {
    constexpr bool ISSUE = decltype(issue_tag)::value;      // a weight tile s + D exists
    if constexpr (ISSUE) {
      cp_wait_vmcnt<(D - 1) * B_IT>();
    } else {
      const int younger = S - 1 - s;
      if (D >= 2 && younger >= 1) cp_wait_vmcnt<B_IT>();
      else cp_wait_vmcnt<0>();
    }
    if constexpr (FIX) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fix-up stores have left
    __builtin_amdgcn_s_barrier();
    const int pb = c & 1;
    const bool next_chunk = c + 1 < chunks;
    // stage the next channel tile's patch (its buffer was last read in the previous channel tile)
    if (next_chunk && t < 3) {
#pragma unroll
      for (int k = 0; k < NSLOT; ++k)
        if (k % 3 == t) issue_patch_slot(k, c + 1, pb ^ 1);
    }
    if constexpr (FIX) {
      // slices of the fix-up of patch c+1: it landed for every wave at the barrier of tap 3 (issued in taps 0-2, waited
      // with the weight tiles); taps 3, 4, 5 each fix a third
      if (next_chunk && t >= 3 && t < 6) {
        if (t == 3) fix_load(c + 1);
#pragma unroll
        for (int it = 0; it < NFIX; ++it)
          if (it % 3 == t - 3) fix_vec(it, pb ^ 1);
      }
    }
    const char* sP = patch0 + pb * PATCH;
    const char* sB = ring + st_c * BSTAGE;
    const int dy = t / a.tw, dx = t - dy * a.tw;
    const int toff = dy * PW + dx;
    int abase[MI], af[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int r = prow0[mi] + toff;
      abase[mi] = r * ROWB;
      af[mi] = ((r >> 1) & 7) << 4;
    }
    const long long koff_i = ISSUE ? koff_of(ci, ti) : 0;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int q16 = (ks * 2 + half) << 4;
      bf16x8 fa[MI], fb[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) fa[mi] = __builtin_bit_cast(bf16x8, *(const uint4*)(sP + abase[mi] + (af[mi] ^ q16)));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int r = rowB[ni];
        fb[ni] = __builtin_bit_cast(bf16x8, *(const uint4*)(sB + r * ROWB + ((((r >> 1) & 7) << 4) ^ q16)));
      }
      if constexpr (ISSUE) {
#pragma unroll
        for (int it = ks; it < B_IT; it += 4) issue_b_piece(it, koff_i, st_i);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = mg_mfma32(fb[ni], fa[mi], acc[ni][mi]);
    }
    if (++t == T) { t = 0; ++c; }
    if constexpr (ISSUE) { if (++ti == T) { ti = 0; ++ci; } }
    st_c = (st_c + 1 == NSTB) ? 0 : st_c + 1;
    st_i = (st_i + 1 == NSTB) ? 0 : st_i + 1;
  };
  {
    int s = 0;
    for (; s + D < S; ++s) k_step(s, std::true_type{});
    for (; s < S; ++s) k_step(s, std::false_type{});
  }

  // ---------------- epilogue (the non-transposed epilogue of igemm2.hip with the tile's pixel map) ----------------
  float gsum[GNS ? NI : 1][2][2], gsq[GNS ? NI : 1][2][2];   // [ni][gp][k]: this lane's pixels, channels 4 k ... of its 8-channel vector
  if constexpr (GNS) {                                        // (k = 1 only with 4-channel groups)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) gsum[ni][gp][0] = gsum[ni][gp][1] = gsq[ni][gp][0] = gsq[ni][gp][1] = 0.f;
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int p = wm * TM + mi * 32 + l31;
    const int oy = y0 + (p >> 4), ox = x0 + (p & 15);
    const bool pix_ok = oy < a.H && ox < a.W;
    long long orow = ((long long)img * a.H + oy) * a.W + ox;
    if (a.subpix) orow = ((long long)img * 2 * a.H + 2 * oy + (z >> 1)) * (2 * a.W) + 2 * ox + (z & 1);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int nb = n0 + wn * TN + ni * 32;
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) half_swap(acc[ni][mi][8 * gp + j], acc[ni][mi][8 * gp + 4 + j], v[j], v[4 + j]);
        const int n = nb + 16 * gp + 8 * half;
        if (pix_ok && n < a.N) {
          if (a.bias) {
            const float4 b0 = *(const float4*)(a.bias + n), b1 = *(const float4*)(a.bias + n + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
          }
          if (a.rowvec) {
            const float* rv = a.rowvec + (long long)img * a.rv_stride + n;
            const float4 r0 = *(const float4*)rv, r1 = *(const float4*)(rv + 4);
            v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
            v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
          }
          if (a.res) {
            const uint4 r4 = *(const uint4*)(a.res + orow * a.ldr + n);
            v[0] += bflo(r4.x); v[1] += bfhi(r4.x); v[2] += bflo(r4.y); v[3] += bfhi(r4.y);
            v[4] += bflo(r4.z); v[5] += bfhi(r4.z); v[6] += bflo(r4.w); v[7] += bfhi(r4.w);
          }
          uint4 pk;
          pk.x = cvt_pk_bf16_f32(v[0], v[1]); pk.y = cvt_pk_bf16_f32(v[2], v[3]);
          pk.z = cvt_pk_bf16_f32(v[4], v[5]); pk.w = cvt_pk_bf16_f32(v[6], v[7]);
          *(uint4*)(a.out + orow * a.ldo + n) = pk;
          if constexpr (GNS) {   // of the values as stored (bf16): what a statistics pass over the tensor would see
            const float w[8] = {bflo(pk.x), bfhi(pk.x), bflo(pk.y), bfhi(pk.y), bflo(pk.z), bfhi(pk.z), bflo(pk.w), bfhi(pk.w)};
            gsum[ni][gp][0] += (w[0] + w[1]) + (w[2] + w[3]);
            gsq[ni][gp][0] += (w[0] * w[0] + w[1] * w[1]) + (w[2] * w[2] + w[3] * w[3]);
            gsum[ni][gp][1] += (w[4] + w[5]) + (w[6] + w[7]);
            gsq[ni][gp][1] += (w[4] * w[4] + w[5] * w[5]) + (w[6] * w[6] + w[7] * w[7]);
          }
        }
      }
    }
  }
  if constexpr (GNS) {
    // the wave's rows: [ni][gp][half][k] (sum, sum of squares) - 8-channel vector c8 = wn TN + 32 ni + 16 gp + 8 half of the tile
    float* const gl = (float*)smem;                       // [NW][TN / 4][2]; the ring is free behind this barrier
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          // over the 32 pixels of a half-wave: an inclusive scan inside each row of 16 lanes (DPP row_shr 1, 2, 4, 8, zeros
          // shifted in), then lane 15's total into the odd rows (row_bcast:15) - lanes 31 and 63 hold the half-waves' sums; five
          // VALU adds per value where the ds_bpermute butterfly took fifteen instructions
          float s_ = gsum[ni][gp][k], q_ = gsq[ni][gp][k];
          auto dpp_sum = [](float v) {
#define CP_DPP(ctrl, rmask) __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), ctrl, rmask, 0xf, true))
            v += CP_DPP(0x111, 0xf);
            v += CP_DPP(0x112, 0xf);
            v += CP_DPP(0x114, 0xf);
            v += CP_DPP(0x118, 0xf);
            v += CP_DPP(0x142, 0xa);
#undef CP_DPP
            return v;
          };
          s_ = dpp_sum(s_);
          q_ = dpp_sum(q_);
          if (l31 == 31) {
            float* d = gl + ((wave * (TN / 4) + (ni * 8 + gp * 4 + half * 2 + k)) << 1);
            d[0] = s_;
            d[1] = q_;
          }
        }
    __syncthreads();
    // group g of the tile's BN / cpg: its 4-channel quarters q4 = g cpg / 4 ... of column wave wn, over the WGM pixel waves
    const int cpg = a.gn_cpg, ngt = BN / cpg, gtot = a.N / cpg;
    const int slot = (z * a.tiles_y + ty) * a.tiles_x + tx;
    for (int g = tid; g < ngt; g += NT) {
      float s_ = 0.f, q_ = 0.f;
      for (int q4 = g * cpg / 4; q4 < (g + 1) * cpg / 4; ++q4) {
        const int wn_ = q4 / (TN / 4), qq = q4 % (TN / 4);
        for (int wm_ = 0; wm_ < WGM; ++wm_) {
          const float* d = gl + (((wm_ * WGN + wn_) * (TN / 4) + qq) << 1);
          s_ += d[0];
          q_ += d[1];
        }
      }
      float* out = a.gn_part + ((((long long)img * a.gn_slots + slot) * gtot + tile_n * ngt + g) << 1);
      out[0] = s_;
      out[1] = q_;
    }
  }
}

template <int TH, int TW, int BN, int WGM, int WGN, int NSTB>
int launch_patch(const ConvPArgs& a0, hipStream_t s) {
  constexpr int NT = WGM * WGN * 64;
  constexpr int PRMAX = ((TH + 2) * (TW + 2) + 7) / 8 * 8;
  constexpr int LDS = 2 * PRMAX * 128 + NSTB * BN * 128 + ((TH == 12 && BN == 320) ? 8192 : 0);   // + scale / shift [2][Cin <= 1024] (SSL)
  static_assert(LDS <= 160 * 1024, "LDS budget exceeds 160 KiB");
  ConvPArgs a = a0;
  a.tiles_x = (a.W + TW - 1) / TW;
  a.tiles_y = (a.H + TH - 1) / TH;
  a.tiles_n = (a.N + BN - 1) / BN;
  const long long grid = (long long)a.tiles_x * a.tiles_y * a.tiles_n * a.B * (a.subpix ? 4 : 1);
  MG_REQUIRE(grid > 0 && grid < (1ll << 31), "conv3x3: bad grid %lld", grid);
  void (*kern)(const ConvPArgs) = a.ss ? conv_patch_kernel<TH, TW, BN, WGM, WGN, NSTB, true>
                                       : conv_patch_kernel<TH, TW, BN, WGM, WGN, NSTB, false>;
  int ki = a.ss ? 1 : 0;
  if constexpr ((WGM * WGN == 12 && BN != 320) || (TH == 16 && BN == 256 && NSTB == 2)) {   // the output's GroupNorm statistics as a
    // by-product: the VAE's 12-wave tiles and the 16 x 16 x 256 tile its 256-channel sub-pixel up-sampling runs on
    if (a.gn_part) {
      kern = a.ss ? conv_patch_kernel<TH, TW, BN, WGM, WGN, NSTB, true, true> : conv_patch_kernel<TH, TW, BN, WGM, WGN, NSTB, false, true>;
      ki += 2;
      MG_REQUIRE((a.gn_cpg == 4 || a.gn_cpg == 8 || a.gn_cpg == 16 || a.gn_cpg == 32) && a.N % a.gn_cpg == 0 && a.N % BN == 0 &&
                 a.gn_slots == a.tiles_x * a.tiles_y * (a.subpix ? 4 : 1),
                 "conv3x3: output statistics need 4 / 8 / 16 / 32 channels per group, N %% %d == 0 and %d table slots (got %d)", BN,
                 a.tiles_x * a.tiles_y * (a.subpix ? 4 : 1), a.gn_slots);
    }
  } else {
    MG_REQUIRE(!a.gn_part, "conv3x3: this tile variant does not produce output statistics (mg_conv3x3_gn_slots() tells)");
  }
  static bool attr_set[4] = {false, false, false, false};
  if (!attr_set[ki] && !g_dry_run) {
    MG_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set[ki] = true;
  }
  MG_LAUNCH(kern, dim3((unsigned)grid), dim3(NT), LDS, s, a);
  if (!g_dry_run) MG_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// Tile choice (profiles/r2_sweep3_patch_conv.log, TFLOP/s at the benchmark batch): 16 x 16 pixels x 256 channels /
// 8 waves (wave tile 128 x 64) where N is a multiple of 256 (VAE 512 / 256 channels: 1100-1130); 8 x 16 x 320 / 8 waves
// (wave tile 32 x 160) for the 320-channel level (930-1050) and the 640-channel sub-pixel convolution; 16 x 16 x 128 /
// 8 waves / 2 weight stages (wave tile 64 x 64) otherwise (N = 640: 990-1060, N = 128: 830-1020).
int mg_conv3x3_auto_variant(int N, int subpix, int B, int H, int W, int has_ss, int Cin, int allow4w) {
  {
    // Round 4: the four-wave hand-placed kernels (conv_patch4w.hip, one workgroup per CU) where their tile count fills the chip
    // (profiles/r4_conv_patch4w.log, TFLOP/s at E = 10): 12 x 16 x 320 for the plain N = 320 / 640 convolutions (640 -> 640
    // @48: 1 147 vs 1 016, 1280 -> 640: 1 267 vs 1 102) and for the fused-norm ones from 640 input channels (960 -> 320 @96:
    // 1 141 vs 1 078; at 320 -> 320 the in-stream fix-up costs what the schedule gains); 16 x 16 x 256 for the plain
    // N = 256 k convolutions (512 -> 512 @96: 1 202 vs 1 050) and the 512-channel sub-pixel up-sampling (1 104 vs 1 031).
    // MARIGOLD_CP4W=0 switches them off (A/B).
    static const int cp4 = mg_tuning_int("MARIGOLD_CP4W", 1);
    const long long par = subpix ? 4 : 1;
    if (cp4 && allow4w && N % 320 == 0 && !subpix && (long long)B * ((H + 11) / 12) * ((W + 15) / 16) * (N / 320) >= 200 &&
        (!has_ss || (!MG_F16 && Cin >= 640 && Cin <= 1024)))   // (the in-stream fix-up unpacks bf16: plain convolutions only in the fp16 build)
      return 11;
    if (cp4 && allow4w && N % 256 == 0 && !has_ss && (long long)B * ((H + 15) / 16) * ((W + 15) / 16) * (N / 256) * par >= 512 && (!subpix || N >= 512))
      return 10;
  }
  static const int n320 = mg_tuning_int("MARIGOLD_PATCH_N320", 0);   // A/B: 3 = the round-2 choice
  const bool old = n320 == 3;
  // N = 256 / 512 (VAE): 12 x 16 x 256 on 12 waves (wave tile 64 x 64) where the GroupNorm is fused / the map is large
  // (512 -> 512 @192 with the fused norm: 668 vs 758 us; 256 -> 256 @384: 712 vs 741; plain @96: equal); the sub-pixel forms stay
  if (N % 256 == 0) return (!subpix && !old && (long long)B * ((H + 11) / 12) * ((W + 15) / 16) * (N / 256) >= 256) ? 9 : 1;
  const long long par = subpix ? 4 : 1;
  // (round 3) 12-wave tiles - three waves per SIMD, and a K step's weights serve 1.5x the pixels - where they still fill the
  // chip (profiles/r3_conv_patch_12row_tiles.log):
  //   12 x 16 pixels x 320 channels for N = 320 / 640 (320 -> 320 @96: 168 vs 198 us; 640 -> 640 @48: 160 vs 193; the
  //   sub-pixel 640 -> 640 up-sampling: 296 vs 355), 24 x 16 x 128 for N = 128 (128 -> 128 @768: 378 vs 438)
  if (N % 320 == 0 && !old && (long long)B * ((H + 11) / 12) * ((W + 15) / 16) * (N / 320) * par >= 200) return 6;
  if (N % 128 == 0 && !old && (long long)B * ((H + 23) / 24) * ((W + 15) / 16) * (N / 128) * par >= 256) return 8;
  if (N == 320 || (subpix && N % 320 == 0)) return 3;
  return 4;
}

// the tile variant MG_OP_CONV3X3 `op` runs on (i[14], or the automatic choice)
static int mg_conv3x3_variant_of(const mg_op* op) {
  const int C0 = op->i[3], C1 = op->i[4], N = op->i[5], subpix = op->i[6], silu = op->i[7];
  const int lda0 = op->i[8] > 0 ? op->i[8] : C0, lda1 = op->i[9] > 0 ? op->i[9] : C1;
  const bool ss = op->p[7] != nullptr;
  // (the four-wave kernels: SiLU with the fused norm, 31-bit byte offsets into the operands)
  const int allow4w = (!ss || silu) && (long long)op->i[0] * op->i[1] * op->i[2] * (lda0 > lda1 ? lda0 : lda1) < (1ll << 30) &&
                      (long long)N * (op->i[12] > 0 ? op->i[12] : (subpix ? 4 : 9) * (C0 + C1)) < (1ll << 30);
  int variant = op->i[14] ? op->i[14] : mg_conv3x3_auto_variant(N, subpix, op->i[0], op->i[1], op->i[2], ss, C0 + C1, allow4w);
  if (!op->i[14] && variant == 6 && ss && C0 + C1 > 1024) variant = 3;   // (variant 6 keeps the fused norm's [2][Cin] vectors in 8 KB of LDS)
  return variant;
}

// Table slots per image of the output-statistics by-product (p[8]) for this op - 0: its tile variant does not produce them.
int mg_conv3x3_gn_slots_of(const mg_op* op) {
  const int variant = mg_conv3x3_variant_of(op);
  const int H = op->i[1], W = op->i[2], N = op->i[5], par = op->i[6] ? 4 : 1;
  if (variant == 1 && N % 256 == 0) return ((H + 15) / 16) * ((W + 15) / 16) * par;
  if (variant == 8 && N % 128 == 0) return ((H + 23) / 24) * ((W + 15) / 16) * par;
  if (variant == 9 && N % 256 == 0) return ((H + 11) / 12) * ((W + 15) / 16) * par;
  return 0;
}

int mg_launch_conv_patch(const mg_op* op, hipStream_t s) {
  ConvPArgs a;
  a.A0 = (const bf16_t*)op->p[0];
  a.Wt = (const bf16_t*)op->p[1];
  a.out = (bf16_t*)op->p[2];
  a.bias = (const float*)op->p[3];
  a.rowvec = (const float*)op->p[4];
  a.res = (const bf16_t*)op->p[5];
  a.A1 = (const bf16_t*)op->p[6];
  a.ss = (const float*)op->p[7];
  a.gn_part = (float*)op->p[8];
  a.gn_cpg = op->i[15];
  a.gn_slots = op->i[16];
  a.zero = g_zero_page;
  a.B = op->i[0]; a.H = op->i[1]; a.W = op->i[2]; a.C0 = op->i[3];
  const int C1 = op->i[4];
  a.Cin = a.C0 + C1;
  a.N = op->i[5];
  a.subpix = op->i[6];
  a.silu = op->i[7];
  a.lda0 = op->i[8] > 0 ? op->i[8] : a.C0;
  a.lda1 = op->i[9] > 0 ? op->i[9] : C1;
  a.ldo = op->i[10] > 0 ? op->i[10] : a.N;
  a.ldr = op->i[11] > 0 ? op->i[11] : a.N;
  a.T = a.subpix ? 4 : 9;
  a.tw = a.subpix ? 2 : 3;
  a.ldw = op->i[12] > 0 ? op->i[12] : a.T * a.Cin;
  a.rv_stride = op->i[13] ? 0 : a.N;
  const int variant = mg_conv3x3_variant_of(op);
  a.sW = op->l[0];
  a.chunks = a.Cin / 64;
  a.c0t = a.C0 / 64;
  a.tiles_x = a.tiles_y = a.tiles_n = 0;
  MG_REQUIRE(g_zero_page || g_dry_run, "conv3x3: mg_init() not called");
  MG_REQUIRE(a.A0 && a.Wt && a.out, "conv3x3: null pointer");
  MG_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0, "conv3x3: empty problem");
  MG_REQUIRE(a.C0 > 0 && a.C0 % 64 == 0 && C1 >= 0 && C1 % 64 == 0, "conv3x3: C0 %d / C1 %d must be multiples of 64", a.C0, C1);
  MG_REQUIRE((C1 == 0) == (a.A1 == nullptr), "conv3x3: second source given without channels (or the reverse)");
  MG_REQUIRE(a.N > 0 && a.N % 8 == 0 && a.ldo % 8 == 0 && a.ldr % 8 == 0, "conv3x3: N %d, ldo, ldr must be multiples of 8", a.N);
  MG_REQUIRE(a.lda0 % 8 == 0 && a.lda1 % 8 == 0 && a.ldw % 8 == 0, "conv3x3: lda/ldw must be multiples of 8");
  MG_REQUIRE((uintptr_t)a.A0 % 16 == 0 && (uintptr_t)a.A1 % 16 == 0 && (uintptr_t)a.Wt % 16 == 0 && (uintptr_t)a.out % 16 == 0 &&
             (uintptr_t)a.res % 16 == 0 && (uintptr_t)a.bias % 16 == 0 && (uintptr_t)a.rowvec % 16 == 0 && (uintptr_t)a.ss % 16 == 0,
             "conv3x3: pointers need 16-byte alignment");
  MG_REQUIRE((long long)a.T * a.Cin * 2 + 256 <= MG_ZERO_BYTES, "conv3x3: K %d exceeds the zero region", a.T * a.Cin);
  MG_REQUIRE((long long)a.B * a.H * a.W * 4 < (1ll << 31), "conv3x3: too many pixels");
  if (a.subpix) MG_REQUIRE(!a.ss && !a.res && !a.rowvec && a.sW > 0, "conv3x3: the sub-pixel mode takes bias only and a parity stride");
  switch (variant) {
    case 1: return launch_patch<16, 16, 256, 2, 4, 2>(a, s);
    case 2: return launch_patch<16, 16, 128, 4, 2, 3>(a, s);
    case 3: return launch_patch<8, 16, 320, 4, 2, 2>(a, s);
    case 4: return launch_patch<16, 16, 128, 4, 2, 2>(a, s);
    case 5: return launch_patch<8, 16, 128, 2, 2, 3>(a, s);
    case 6:   // 192 pixels x 320 channels / 12 waves (wave tile 32 x 160)
      MG_REQUIRE(a.N % 320 == 0 && (!a.ss || a.Cin <= 1024), "conv3x3: tile variant 6 needs N %% 320 == 0 (and Cin <= 1024 with the fused norm)");
      return launch_patch<12, 16, 320, 6, 2, 2>(a, s);
    case 7:   // 192 pixels x 320 channels / 6 waves (wave tile 64 x 160)
      MG_REQUIRE(a.N % 320 == 0 && (!a.ss || a.Cin <= 1024), "conv3x3: tile variant 7 needs N %% 320 == 0 (and Cin <= 1024 with the fused norm)");
      return launch_patch<12, 16, 320, 3, 2, 2>(a, s);
    case 8:   // 24 x 16 pixels x 128 channels / 12 waves (wave tile 64 x 64): the VAE's 128-channel 768^2 level
      MG_REQUIRE(a.N % 128 == 0, "conv3x3: tile variant 8 needs N %% 128 == 0");
      return launch_patch<24, 16, 128, 6, 2, 2>(a, s);
    case 9:   // 12 x 16 pixels x 256 channels / 12 waves (wave tile 64 x 64)
      MG_REQUIRE(a.N % 256 == 0, "conv3x3: tile variant 9 needs N %% 256 == 0");
      return launch_patch<12, 16, 256, 3, 4, 2>(a, s);
    case 10:   // four waves, one per SIMD, hand-placed streams (conv_patch4w.hip): 16 x 16 pixels x 256 channels
      return mg_launch_conv_patch4w(a, 0, s);
    case 11:   //   12 x 16 pixels x 320 channels
      return mg_launch_conv_patch4w(a, 1, s);
    default: MG_REQUIRE(false, "conv3x3: unknown tile variant %d", variant);
  }
  return 0;
}
