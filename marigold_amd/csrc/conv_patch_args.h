// Argument block of the patch-resident 3x3 convolution kernels (conv_patch.hip, conv_patch4w.hip).
#pragma once
#include "common.h"

struct ConvPArgs {
  const bf16_t* A0;
  const bf16_t* A1;
  const bf16_t* Wt;
  bf16_t* out;
  const float* bias;
  const float* rowvec;
  const bf16_t* res;
  const float* ss;       // fused GroupNorm: [B][2][Cin] fp32 (scale, shift) or nullptr
  float* gn_part;        // or nullptr: GroupNorm statistics of the OUTPUT as a by-product - (sum, sum of squares) of every group of
  int gn_cpg, gn_slots;  // gn_cpg (4 | 8 | 16 | 32) channels over the tile's pixels into [B][gn_slots][N / gn_cpg][2], slot = the tile
  const void* zero;
  int B, H, W, C0, Cin, N, lda0, lda1, ldo, ldr, ldw, rv_stride, silu, subpix;
  int tiles_x, tiles_y, tiles_n, chunks, c0t, T, tw;
  long long sW;
};

// the hand-placed four-wave kernels (conv_patch4w.hip); geo 0 = 16 x 16 pixels x 256 channels, 1 = 12 x 16 x 320
int mg_launch_conv_patch4w(const ConvPArgs& a, int geo, hipStream_t s);
