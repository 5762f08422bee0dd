// Patch-resident 3x3 convolution on FOUR waves (one per SIMD, 128 x 128 / 96 x 160 wave tiles, accumulators in AGPRs) with the
// K-tile instruction streams placed by hand - round 4's answer to "every MFMA kernel plateaus at 1.0 PFLOP/s": the schedule
// that took the implicit-GEMM tile from 1.06 to 1.29 PFLOP/s (igemm2 variant 72, gen_k4w.py) on the kernel that carries the
// ResNet convolutions (diffusers ResnetBlock2D conv1 / conv2, Upsample2D conv - marigold/marigold_depth_pipeline.py:461-463,
// 491-492, 512-513).  Same operator contract as conv_patch.hip (MG_OP_CONV3X3: fused GroupNorm scale / shift + SiLU on the
// staged patch, second channel source, sub-pixel 2x up-sampling, bias / time-embedding row / residual epilogue), same
// reduction order (channel tile outermost, taps inside) - outputs match the 8- / 12-wave kernels to the last fp32 bits of the
// accumulation.
//
//   LDS   [weight stage 0 | weight stage 1 | patch buffer 0 | patch buffer 1 | scale / shift of the image]
//   staging  buffer-load LDS-DMA on SGPR bases (padding pixels / channels = out-of-range offsets -> zeros); the weights of
//            tile k + 2 while tile k computes, the whole patch of channel tile c + 1 during tap 0 of channel tile c
//   reads    whole-tile fragment sets (F[2], F[3] of the tile at its start, F[0], F[1] of the next one at its end); the
//            pixel-side addresses (tap shift + XOR swizzle) of the next tile computed by VALU fillers
//   fix-up   GroupNorm affine + SiLU on the staged patch of channel tile c + 1, in place, as VALU fillers of taps 2-7
// Built without -amdgpu-mfma-vgpr-form (Makefile).  Streams: gen_cp4w.py -> conv_patch4w.inc.
#include <stdlib.h>

#include "common.h"
#include "conv_patch_args.h"
#include "conv_patch4w.inc"

namespace {

#define CP4_SCRATCH "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"

// ---- geometry A: 16 x 16 pixels x 256 channels, wave tile 128 x 128 ----
#define CP4_OUT \
  [c00] "+a"(acc[0][0]), [c01] "+a"(acc[0][1]), [c02] "+a"(acc[0][2]), [c03] "+a"(acc[0][3]), [c10] "+a"(acc[1][0]), \
  [c11] "+a"(acc[1][1]), [c12] "+a"(acc[1][2]), [c13] "+a"(acc[1][3]), [c20] "+a"(acc[2][0]), [c21] "+a"(acc[2][1]), \
  [c22] "+a"(acc[2][2]), [c23] "+a"(acc[2][3]), [c30] "+a"(acc[3][0]), [c31] "+a"(acc[3][1]), [c32] "+a"(acc[3][2]), \
  [c33] "+a"(acc[3][3]), [a00] "+v"(fa4[0][0]), [a01] "+v"(fa4[0][1]), [a02] "+v"(fa4[0][2]), [a03] "+v"(fa4[0][3]), \
  [a10] "+v"(fa4[1][0]), [a11] "+v"(fa4[1][1]), [a12] "+v"(fa4[1][2]), [a13] "+v"(fa4[1][3]), [a20] "+v"(fa4[2][0]), \
  [a21] "+v"(fa4[2][1]), [a22] "+v"(fa4[2][2]), [a23] "+v"(fa4[2][3]), [a30] "+v"(fa4[3][0]), [a31] "+v"(fa4[3][1]), \
  [a32] "+v"(fa4[3][2]), [a33] "+v"(fa4[3][3]), [b00] "+v"(fb4[0][0]), [b01] "+v"(fb4[0][1]), [b02] "+v"(fb4[0][2]), \
  [b03] "+v"(fb4[0][3]), [b10] "+v"(fb4[1][0]), [b11] "+v"(fb4[1][1]), [b12] "+v"(fb4[1][2]), [b13] "+v"(fb4[1][3]), \
  [b20] "+v"(fb4[2][0]), [b21] "+v"(fb4[2][1]), [b22] "+v"(fb4[2][2]), [b23] "+v"(fb4[2][3]), [b30] "+v"(fb4[3][0]), \
  [b31] "+v"(fb4[3][1]), [b32] "+v"(fb4[3][2]), [b33] "+v"(fb4[3][3]), [pa00] "+v"(pa[0][0]), [pa01] "+v"(pa[0][1]), \
  [pa02] "+v"(pa[0][2]), [pa03] "+v"(pa[0][3]), [pa10] "+v"(pa[1][0]), [pa11] "+v"(pa[1][1]), [pa12] "+v"(pa[1][2]), \
  [pa13] "+v"(pa[1][3]), [pa20] "+v"(pa[2][0]), [pa21] "+v"(pa[2][1]), [pa22] "+v"(pa[2][2]), [pa23] "+v"(pa[2][3]), \
  [pa30] "+v"(pa[3][0]), [pa31] "+v"(pa[3][1]), [pa32] "+v"(pa[3][2]), [pa33] "+v"(pa[3][3]), [lb0] "+v"(lb[0]), \
  [lb1] "+v"(lb[1]), [lb2] "+v"(lb[2]), [lb3] "+v"(lb[3])
#define CP4_IN \
  [xb0] "v"(xb[0]), [xb1] "v"(xb[1]), [xb2] "v"(xb[2]), [xb3] "v"(xb[3]), [r00] "v"(r0[0]), [r01] "v"(r0[1]), \
  [r02] "v"(r0[2]), [r03] "v"(r0[3]), [hx] "v"(hx), [vb0] "v"(vb[0]), [vb1] "v"(vb[1]), [vb2] "v"(vb[2]), \
  [vb3] "v"(vb[3]), [vb4] "v"(vb[4]), [vb5] "v"(vb[5]), [vb6] "v"(vb[6]), [vb7] "v"(vb[7]), [vp0] "v"(vp[0]), \
  [vp1] "v"(vp[1]), [vp2] "v"(vp[2]), [vp3] "v"(vp[3]), [vp4] "v"(vp[4]), [vp5] "v"(vp[5]), [vp6] "v"(vp[6]), \
  [vp7] "v"(vp[7]), [vp8] "v"(vp[8]), [vp9] "v"(vp[9]), [vp10] "v"(vp[10]), [sw] "s"(swt), [sp] "s"(spt), \
  [mw] "s"(mw), [mp] "s"(mp), [stoff] "s"(stoff), [spb] "s"(spb), [s96] "s"(s96), [fxa] "v"(fxa), [fmask] "v"(fmask), \
  [fs0] "v"(fs[0]), [fs1] "v"(fs[1]), [fs2] "v"(fs[2]), [fs3] "v"(fs[3]), [fs4] "v"(fs[4]), [fs5] "v"(fs[5]), \
  [fs6] "v"(fs[6]), [fs7] "v"(fs[7]), [fh0] "v"(fh[0]), [fh1] "v"(fh[1]), [fh2] "v"(fh[2]), [fh3] "v"(fh[3]), \
  [fh4] "v"(fh[4]), [fh5] "v"(fh[5]), [fh6] "v"(fh[6]), [fh7] "v"(fh[7])
#define CP4_NAME conv_patch4w_a_kernel
#define CP4_TH 16
#define CP4_BN 256
#define CP4_MI 4
#define CP4_NI 4
#define CP4_NPW 44
#define CP4_T(x) CP4A_##x
#include "conv_patch4w_body.h"
#undef CP4_OUT
#undef CP4_IN
#undef CP4_NAME
#undef CP4_TH
#undef CP4_BN
#undef CP4_MI
#undef CP4_NI
#undef CP4_NPW
#undef CP4_T

// ---- geometry B: 12 x 16 pixels x 320 channels, wave tile 96 x 160 ----
#define CP4_OUT \
  [c00] "+a"(acc[0][0]), [c01] "+a"(acc[0][1]), [c02] "+a"(acc[0][2]), [c10] "+a"(acc[1][0]), [c11] "+a"(acc[1][1]), \
  [c12] "+a"(acc[1][2]), [c20] "+a"(acc[2][0]), [c21] "+a"(acc[2][1]), [c22] "+a"(acc[2][2]), [c30] "+a"(acc[3][0]), \
  [c31] "+a"(acc[3][1]), [c32] "+a"(acc[3][2]), [c40] "+a"(acc[4][0]), [c41] "+a"(acc[4][1]), [c42] "+a"(acc[4][2]), \
  [a00] "+v"(fa4[0][0]), [a01] "+v"(fa4[0][1]), [a02] "+v"(fa4[0][2]), [a10] "+v"(fa4[1][0]), [a11] "+v"(fa4[1][1]), \
  [a12] "+v"(fa4[1][2]), [a20] "+v"(fa4[2][0]), [a21] "+v"(fa4[2][1]), [a22] "+v"(fa4[2][2]), [a30] "+v"(fa4[3][0]), \
  [a31] "+v"(fa4[3][1]), [a32] "+v"(fa4[3][2]), [b00] "+v"(fb4[0][0]), [b01] "+v"(fb4[0][1]), [b02] "+v"(fb4[0][2]), \
  [b03] "+v"(fb4[0][3]), [b04] "+v"(fb4[0][4]), [b10] "+v"(fb4[1][0]), [b11] "+v"(fb4[1][1]), [b12] "+v"(fb4[1][2]), \
  [b13] "+v"(fb4[1][3]), [b14] "+v"(fb4[1][4]), [b20] "+v"(fb4[2][0]), [b21] "+v"(fb4[2][1]), [b22] "+v"(fb4[2][2]), \
  [b23] "+v"(fb4[2][3]), [b24] "+v"(fb4[2][4]), [b30] "+v"(fb4[3][0]), [b31] "+v"(fb4[3][1]), [b32] "+v"(fb4[3][2]), \
  [b33] "+v"(fb4[3][3]), [b34] "+v"(fb4[3][4]), [pa00] "+v"(pa[0][0]), [pa01] "+v"(pa[0][1]), [pa02] "+v"(pa[0][2]), \
  [pa03] "+v"(pa[0][3]), [pa10] "+v"(pa[1][0]), [pa11] "+v"(pa[1][1]), [pa12] "+v"(pa[1][2]), [pa13] "+v"(pa[1][3]), \
  [pa20] "+v"(pa[2][0]), [pa21] "+v"(pa[2][1]), [pa22] "+v"(pa[2][2]), [pa23] "+v"(pa[2][3]), [lb0] "+v"(lb[0]), \
  [lb1] "+v"(lb[1]), [lb2] "+v"(lb[2]), [lb3] "+v"(lb[3])
#define CP4_IN \
  [xb0] "v"(xb[0]), [xb1] "v"(xb[1]), [xb2] "v"(xb[2]), [xb3] "v"(xb[3]), [r00] "v"(r0[0]), [r01] "v"(r0[1]), \
  [r02] "v"(r0[2]), [hx] "v"(hx), [vb0] "v"(vb[0]), [vb1] "v"(vb[1]), [vb2] "v"(vb[2]), [vb3] "v"(vb[3]), \
  [vb4] "v"(vb[4]), [vb5] "v"(vb[5]), [vb6] "v"(vb[6]), [vb7] "v"(vb[7]), [vb8] "v"(vb[8]), [vb9] "v"(vb[9]), \
  [vp0] "v"(vp[0]), [vp1] "v"(vp[1]), [vp2] "v"(vp[2]), [vp3] "v"(vp[3]), [vp4] "v"(vp[4]), [vp5] "v"(vp[5]), \
  [vp6] "v"(vp[6]), [vp7] "v"(vp[7]), [sw] "s"(swt), [sp] "s"(spt), [mw] "s"(mw), [mp] "s"(mp), [stoff] "s"(stoff), \
  [spb] "s"(spb), [s96] "s"(s96), [fxa] "v"(fxa), [fmask] "v"(fmask), [fs0] "v"(fs[0]), [fs1] "v"(fs[1]), \
  [fs2] "v"(fs[2]), [fs3] "v"(fs[3]), [fs4] "v"(fs[4]), [fs5] "v"(fs[5]), [fs6] "v"(fs[6]), [fs7] "v"(fs[7]), \
  [fh0] "v"(fh[0]), [fh1] "v"(fh[1]), [fh2] "v"(fh[2]), [fh3] "v"(fh[3]), [fh4] "v"(fh[4]), [fh5] "v"(fh[5]), \
  [fh6] "v"(fh[6]), [fh7] "v"(fh[7])
#define CP4_NAME conv_patch4w_b_kernel
#define CP4_TH 12
#define CP4_BN 320
#define CP4_MI 3
#define CP4_NI 5
#define CP4_NPW 32
#define CP4_T(x) CP4B_##x
#include "conv_patch4w_body.h"

template <int TH, int BN, int NPW>
int launch4w(const ConvPArgs& a0, void (*plain)(const ConvPArgs), void (*fixed)(const ConvPArgs), hipStream_t s) {
  ConvPArgs a = a0;
  const int LDS = 2 * BN * 128 + 2 * NPW * 1024 + (a.ss ? 8 * a.Cin : 0);
  MG_REQUIRE(LDS <= 160 * 1024, "conv3x3: %d input channels exceed the four-wave tile's LDS budget with the fused norm", a.Cin);
  MG_REQUIRE(a.N % BN == 0, "conv3x3: the four-wave tile needs N %% %d == 0", BN);
  MG_REQUIRE(!a.ss || a.silu, "conv3x3: the four-wave tile's fused norm includes the SiLU");
  MG_REQUIRE(!a.ss || !MG_F16, "conv3x3: the four-wave tile's in-stream fix-up unpacks bf16 (the fp16 build runs the fused norm on the 12-wave tiles)");
  MG_REQUIRE(!a.gn_part, "conv3x3: the four-wave tiles do not produce output statistics (p[8]: mg_conv3x3_gn_slots() returns 0 for them)");
  MG_REQUIRE((long long)a.B * a.H * a.W * (a.lda0 > a.lda1 ? a.lda0 : a.lda1) < (1ll << 30) && (long long)a.N * a.ldw < (1ll << 30),
             "conv3x3: the four-wave tile addresses its operands with 31-bit byte offsets");
  a.tiles_x = (a.W + 15) / 16;
  a.tiles_y = (a.H + TH - 1) / TH;
  a.tiles_n = a.N / BN;
  const long long grid = (long long)a.tiles_x * a.tiles_y * a.tiles_n * a.B * (a.subpix ? 4 : 1);
  MG_REQUIRE(grid > 0 && grid < (1ll << 31), "conv3x3: bad grid %lld", grid);
  void (*kern)(const ConvPArgs) = a.ss ? fixed : plain;
  static bool attr_set[2] = {false, false};
  if (!attr_set[a.ss ? 1 : 0] && !g_dry_run) {
    MG_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set[a.ss ? 1 : 0] = true;
  }
  MG_LAUNCH(kern, dim3((unsigned)grid), dim3(256), LDS, s, a);
  if (!g_dry_run) MG_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int mg_launch_conv_patch4w(const ConvPArgs& a, int geo, hipStream_t s) {
  if (geo == 0) return launch4w<16, 256, 44>(a, conv_patch4w_a_kernel<false>, conv_patch4w_a_kernel<true>, s);
  return launch4w<12, 320, 32>(a, conv_patch4w_b_kernel<false>, conv_patch4w_b_kernel<true>, s);
}
