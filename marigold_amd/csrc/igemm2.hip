// Implicit-GEMM bf16 MFMA kernel
// (MG_OP_IGEMM in include/marigold_hip.h: conv3x3 / conv1x1 / Linear / batched GEMM replacing the
// torch conv2d / linear / matmul calls inside diffusers' UNet2DConditionModel / AutoencoderKL,
// reference call sites marigold/marigold_depth_pipeline.py:461-463, 491-492, 512-513).
//
// Design points, all of them aimed at the two things the first profile of a plain tile loop showed
// (UNet GEMMs at 14 % of the MFMA roof, the loop draining its DMA queue at every barrier):
//   * NSTAGE-deep LDS ring filled by global_load_lds_dwordx4 with COUNTED `s_waitcnt vmcnt(N)` and a
//     raw `s_barrier`: up to NSTAGE-1 K tiles stay in flight across the barrier (one barrier per
//     K step).  All LDS lives in one dynamic array so hipcc adds no vmcnt(0) of its own.
//   * im2col addressing is hoisted: per-row source pointers are rebuilt once per TAP (branch-free,
//     padding rows point at a zero region), the per-K-step work is one 64-bit add per row.
//   * the transposed (V^T) section is a template parameter (own launch) - no dual code path, the
//     256x128 tile needs ~120 VGPRs.
//   * epilogue: a lane^32 exchange regroups the MFMA accumulators so every lane owns 8 consecutive
//     output channels of one pixel -> 16-byte NHWC stores / residual loads (32-byte for fp32),
//     v_cvt_pk_bf16_f32 instead of integer rounding.  GEGLU pairs u/gate rows 16 apart (32-row
//     interleave, weights.py::pack_geglu) so it keeps the 16-byte stores too.
//   * prologue / epilogue are written for INSTRUCTION COUNT (round 2): a short-K tile executed ~1600 instructions around
//     its 16 MFMAs.  Launch-time constant division (common.h: fdiv), a straight-row path for Linear layers, an interior
//     epilogue with compile-time optional terms and per-row pointers; LayerNorm statistics are finalized by the producer's
//     last column tile (ticket hand-off), consumers read 8 bytes per row.
//   * MG_EPI_XATTN2: the pair-softmax epilogue's packed probabilities are the register operand of a second MFMA stage
//     (the collapsed cross-attention in one launch).
#include "igemm2_body.h"

void* mg_igemm2_big_kernel(int which);   // igemm2_big.hip

namespace {

template <int BM, int BN, int WGM, int WGN, int NSTAGE, bool TRANS, bool SPLIT, bool PF, int ABL = 0, int BK = 64>
__global__ __launch_bounds__(WGM* WGN * 64) void igemm2_kernel(const Igemm2Args a) {
  igemm2_body<BM, BN, WGM, WGN, NSTAGE, TRANS, SPLIT, PF ? 1 : 0, ABL, BK>(a);
}

// The ping-pong schedule on the 256x256 / 8-wave / 2-stage tile (LOOP = 2 above).
template <bool TRANS, int PPOPT>
__global__ __launch_bounds__(512) void igemm2_pingpong_kernel(const Igemm2Args a) {
  igemm2_body<256, 256, 2, 4, 2, TRANS, false, 2, 0, 64, PPOPT>(a);
}

// out[m][n] = bf16( scale * sum_s ws[s][m][n] + bias[n] + rowvec[img(m)][n] + residual[m][n] ): the
// epilogue of a split-K launch (fixed summation order -> bit-reproducible), 8 channels per thread.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const Igemm2Args a) {
  const int nv = a.N >> 3;
  const long long total = (long long)a.M * nv;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / nv);
    const int n = (int)(i - (long long)m * nv) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    for (int sidx = 0; sidx < a.splits; ++sidx) {
      const float* p = a.ws + ((long long)sidx * a.M + m) * a.N + n;
      const float4 x0 = *(const float4*)p, x1 = *(const float4*)(p + 4);
      v[0] += x0.x; v[1] += x0.y; v[2] += x0.z; v[3] += x0.w;
      v[4] += x1.x; v[5] += x1.y; v[6] += x1.z; v[7] += x1.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= a.scale;
    if (a.bias) {
      const float4 b0 = *(const float4*)(a.bias + n), b1 = *(const float4*)(a.bias + n + 4);
      v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
      v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    if (a.rowvec) {
      const float* rv = a.rowvec + (long long)(m / a.rows_per_img) * a.rv_stride + n;
      const float4 r0 = *(const float4*)rv, r1 = *(const float4*)(rv + 4);
      v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
      v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
    }
    if (a.res) {
      const uint4 r4 = *(const uint4*)(a.res + (long long)m * a.ldr + n);
      v[0] += bflo(r4.x); v[1] += bfhi(r4.x); v[2] += bflo(r4.y); v[3] += bfhi(r4.y);
      v[4] += bflo(r4.z); v[5] += bfhi(r4.z); v[6] += bflo(r4.w); v[7] += bfhi(r4.w);
    }
    uint4 pk;
    pk.x = cvt_pk_bf16(v[0], v[1]); pk.y = cvt_pk_bf16(v[2], v[3]);
    pk.z = cvt_pk_bf16(v[4], v[5]); pk.w = cvt_pk_bf16(v[6], v[7]);
    *(uint4*)((bf16_t*)a.out + (long long)m * a.ldo + n) = pk;
  }
}

template <int BM, int BN, int WGM, int WGN, int NSTAGE, bool TRANS, bool SPLIT = false, bool PF = false, int ABL = 0,
          int BK = 64, int PPOPT = -1, int BIG = -1>
int launch2(const Igemm2Args& a, int batch_z, hipStream_t s) {
  constexpr int NT = WGM * WGN * 64;
  constexpr int LDS = NSTAGE * (BM + BN) * BK * 2 + BM * 8;   // ring + (mean, rstd) of the tile's rows (folded LayerNorm)
  static_assert(LDS <= 160 * 1024, "LDS ring exceeds 160 KiB");
  static bool attr_set[2] = {false, false};   // [1]: the instrumented instantiation (tuning only)
  const int ai = a.stamps ? 1 : 0;
  void (*kern)(const Igemm2Args);
  if constexpr (BIG >= 0) kern = (void (*)(const Igemm2Args))mg_igemm2_big_kernel(a.stamps ? BIG + 2 : BIG);
  else if constexpr (PPOPT >= 0) kern = igemm2_pingpong_kernel<TRANS, PPOPT>;
  else kern = igemm2_kernel<BM, BN, WGM, WGN, NSTAGE, TRANS, SPLIT, PF, ABL, BK>;
  if (!attr_set[ai] && !g_dry_run) {
    MG_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set[ai] = true;
  }
  Igemm2Args b = a;
  b.cpt = a.Cin / BK;
  b.c0t = a.A1 ? a.C0 / BK : b.cpt;
  b.xcpt = a.xcin / BK;                       // the folded 1x1 convolution's K tiles, behind the taps
  b.xc0t = a.X1 ? a.xc0 / BK : b.xcpt;
  b.KT = a.taps * b.cpt + b.xcpt;
  b.tiles_m = (a.M + BM - 1) / BM;
  b.tiles_n = (a.n_end - a.n_begin + BN - 1) / BN;
  // split-K for the deep UNet levels (a few hundred pixels x thousands of input channels): too few
  // output tiles to fill 256 CUs, so the K loop is cut into `splits` workgroups per tile.
  b.splits = 1;
  b.kps = b.KT;
  b.ws = nullptr;
  const long long tiles = (long long)b.tiles_m * b.tiles_n;
  // a.splits: -1 = never (forced tile / epilogues the reduce launch cannot finish), 0 = the automatic rule, n >= 1 = exactly n
  // (op i[31]: the tuning sweeps and the per-batch rules of mg_igemm_auto_split)
  const bool sk_ok = !TRANS && a.splits >= 0 && batch_z == 1 && a.epi == MG_EPI_BF16 && a.n_begin == 0 && a.n_end == a.N && (a.ws || g_splitk_ws);
  if (sk_ok && (a.splits > 1 || (a.splits == 0 && tiles < 160 && b.KT >= 32))) {   // (a wider window, < 256 tiles, measured no gain at E = 10)
    int sp;
    if (a.splits > 1) {
      sp = min(a.splits, max(1, b.KT / 2));
    } else {
      sp = (int)min((long long)8, (320 + tiles - 1) / tiles);
      if constexpr (BIG == 2 || BIG == 3) sp = (int)min((long long)8, max(1ll, 256 / tiles));   // one workgroup per CU: a single round of <= 256
      sp = min(sp, b.KT / 12);
    }
    // (round 4, tried and NOT kept) combining the splits inside the launch by the tile's last split instead of the splitk_reduce
    // launch: correct, but 25 ms per map SLOWER (igemm 110.5 -> 135.5 ms, profiles/r4_ab_splitk_fused.log) - ONE workgroup then
    // reads splits x 256 KB of slabs per tile at the 60-100 GB/s a single workgroup gets, on the launch's critical path, where
    // the reduce launch spreads the same bytes over the whole chip in 13 us.
    {   // tuning only (A/B under maps in flight): MARIGOLD_SPLITK_MAX = 1: no split-K, n > 1: at most n K ranges per tile
      static const int skmax = mg_tuning_int("MARIGOLD_SPLITK_MAX", 0);
      if (skmax > 0) sp = min(sp, skmax);
    }
    const long long per_split = (long long)a.M * a.N * 4;
    while (sp > 1 && (long long)sp * per_split > MG_SPLITK_WS_BYTES) --sp;
    if (sp > 1) {
      b.kps = (b.KT + sp - 1) / sp;
      b.splits = (b.KT + b.kps - 1) / b.kps;
      b.ws = a.ws ? a.ws : (float*)g_splitk_ws;   // op p[14]: the program's own workspace (programs on concurrent streams)
    }
  }
  const long long grid = tiles * b.splits * batch_z;
  MG_REQUIRE(grid > 0 && grid < (1ll << 31), "igemm: bad grid %lld", grid);
  b.fd_per_z = mg_make_fastdiv(tiles * b.splits);
  b.fd_tiles = mg_make_fastdiv(tiles);
  b.fd_tiles_n = mg_make_fastdiv(b.tiles_n);
  b.fd_cpt = mg_make_fastdiv(b.cpt);
  MG_LAUNCH(kern, dim3((unsigned)grid), dim3(NT), LDS, s, b);
  if (b.splits > 1) {
    const long long nvec = (long long)a.M * (a.N / 8);
    MG_LAUNCH(splitk_reduce_kernel, dim3((unsigned)min((nvec + 255) / 256, (long long)4096)), dim3(256), 0, s, b);
  }
  if (!g_dry_run) MG_CHECK_HIP(hipGetLastError());
  return 0;
}

// The tiles mg_igemm_auto_variant and the tuning table (marigold_amd/tuning) can name.  (Rounds 1-4 carried ~35 variants for
// the sweeps - 20-28, 30-34, 37-39, 47-53, 60-63, 70-71, the ablation forms 40-45; their numbers are in profiles/r1_sweep* ...
// r4_k4w_sweep.log.  Round 5 pruned the list to what a launch can actually run on.)
template <bool TRANS>
int dispatch_tile(const Igemm2Args& a, int batch_z, int variant, hipStream_t s) {
  switch (variant) {
    case 22: return launch2<128, 128, 4, 2, 3, TRANS>(a, batch_z, s);
    case 23: return launch2<64, 64, 2, 2, 2, TRANS>(a, batch_z, s);
    case 24: return launch2<64, 64, 2, 2, 4, TRANS>(a, batch_z, s);       // 23 / 35 / 32 with a deeper ring: the split-K launches of a
    case 25: return launch2<128, 64, 2, 2, 4, TRANS, true>(a, batch_z, s);   // single member's 12^2 / 24^2 levels are bound by the
    case 26: return launch2<128, 128, 2, 2, 3, TRANS, true>(a, batch_z, s);  // latency of a K step, not by bandwidth
    case 29: return launch2<128, 32, 4, 1, 3, TRANS>(a, batch_z, s);  // N <= 32 (4 <-> C boundary convs)
    case 54: return launch2<128, 64, 4, 1, 3, TRANS, true>(a, batch_z, s);   // one wave = 32 rows x all 64 columns (MG_EPI_XATTN2)
    case 32: return launch2<128, 128, 2, 2, 2, TRANS, true>(a, batch_z, s);
    case 35: return launch2<128, 64, 2, 2, 3, TRANS, true>(a, batch_z, s);
    case 36: return launch2<256, 128, 4, 2, 3, TRANS, true, true>(a, batch_z, s);   // half-K-step pipelined loop
    case 51: return launch2<256, 128, 2, 2, 3, TRANS, true, false, 0, 32>(a, batch_z, s);  // BK = 32: 4 waves, wave tile 128x64
    case 62: return launch2<256, 256, 2, 4, 2, TRANS, false, false, 0, 64, 4>(a, batch_z, s);   // ping-pong schedule, 2nd DMA piece among the MFMAs
    case 72:   // 256 x 256 on four waves (one per SIMD, 128 x 128 wave tile), K loop placed by hand (LOOP == 3): offsets are 32-bit, relative to the operand bases
      if constexpr (!TRANS) {
        MG_REQUIRE((long long)(a.M / a.rows_per_img) * a.H * a.W * max(max(a.lda, a.lda1), max(a.ldx0, a.ldx1)) < (1ll << 30) && (long long)a.N * a.ldw < (1ll << 30),
                   "igemm: tile variant 72 addresses its operands with 31-bit byte offsets");
        return launch2<256, 256, 2, 2, 2, false, false, false, 0, 64, -1, 2>(a, batch_z, s);
      } else MG_REQUIRE(false, "igemm: tile variant 72 has no transposed section");
      return 0;
    case 73:   // the hand-placed K loop on a 192 x 320 tile (wave tile 96 x 160): full width for the N = 320 k layers
      if constexpr (!TRANS) {
        MG_REQUIRE((long long)(a.M / a.rows_per_img) * a.H * a.W * max(max(a.lda, a.lda1), max(a.ldx0, a.ldx1)) < (1ll << 30) && (long long)a.N * a.ldw < (1ll << 30),
                   "igemm: tile variant 73 addresses its operands with 31-bit byte offsets");
        return launch2<192, 320, 2, 2, 2, false, false, false, 0, 64, -1, 3>(a, batch_z, s);
      } else MG_REQUIRE(false, "igemm: tile variant 73 has no transposed section");
      return 0;
    case 46: return launch2<128, 320, 4, 2, 2, TRANS, true>(a, batch_z, s);   // full-width tiles for N = 320
    default: MG_REQUIRE(false, "igemm: unknown tile variant %d (22 - 26, 29, 32, 35, 36, 46, 51, 54, 62, 72, 73)", variant);
  }
  return 0;
}

}  // namespace

int mg_igemm_auto_variant(long long M, int N, int K, int batch_z, int geglu);

// returns -1 when the shape is outside the kernel's contract (mg_launch_igemm turns that into an error)
int mg_launch_igemm2(const mg_op* op, hipStream_t s, int variant) {
  Igemm2Args a;
  a.A = (const bf16_t*)op->p[0];
  a.Wt = (const bf16_t*)op->p[1];
  a.out = op->p[2];
  a.bias = (const float*)op->p[3];
  a.rowvec = (const float*)op->p[4];
  a.res = (const bf16_t*)op->p[5];
  void* out2 = op->p[6];
  a.zero = g_zero_page;
  const int B = op->i[0];
  a.H = op->i[1]; a.W = op->i[2]; a.Cin = op->i[3]; a.Ho = op->i[4]; a.Wo = op->i[5];
  a.N = op->i[6]; a.taps = op->i[7]; a.stride = op->i[8]; a.pad = op->i[9];
  a.Hu = op->i[10]; a.Wu = op->i[11]; a.epi = op->i[12]; a.ldo = op->i[13];
  const int trans_from = op->i[14];
  const int batch_z = op->i[15] > 0 ? op->i[15] : 1;
  a.ldr = op->i[16] > 0 ? op->i[16] : a.N;
  a.A1 = (const bf16_t*)op->p[7];
  a.C0 = a.A1 ? op->i[24] : a.Cin;
  a.lda1 = a.A1 ? (op->i[25] > 0 ? op->i[25] : a.Cin - a.C0) : 0;
  a.c0t = 0;
  a.lda = op->i[17] > 0 ? op->i[17] : a.C0;
  a.ldt = op->i[18];
  a.ldw = op->i[20] > 0 ? op->i[20] : a.taps * a.Cin + (op->p[12] ? op->i[32] : 0);
  a.rv_stride = op->i[21] ? 0 : a.N;
  a.sA = op->l[0]; a.sW = op->l[1]; a.sO = op->l[2]; a.sR = op->l[3];
  a.scale = op->f[0] == 0.f ? 1.f : op->f[0];
  a.rows_per_img = a.Ho * a.Wo;
  a.M = B * a.rows_per_img;
  a.cpt = a.Cin / 64;
  a.KT = a.taps * a.cpt;
  a.up2 = (a.Hu == 2 * a.H) && (a.Wu == 2 * a.W);
  a.tw = a.taps == 9 ? 3 : (a.taps == 4 ? 2 : 1);
  a.subpix = a.taps == 4;
  // split-K: i[31] = n >= 1 asks for exactly n K ranges per tile (1 = none); 0 = the automatic rule, which only applies under
  // the automatic tile choice (a forced tile runs unsplit unless i[31] says otherwise)
  a.splits = op->i[31] > 0 ? op->i[31] : (variant ? -1 : 0);
  a.ln_out = (float2*)op->p[8];
  a.ln_in = (const float2*)op->p[9];
  a.ln_g = (const float*)op->p[10];
  a.ln_c = (const float*)op->p[11];
  a.ln_eps = op->f[1];
  a.sm_scale = op->f[2];
  a.sm_cols = op->i[27];
  a.tperm = op->i[26];
  a.w2 = nullptr;
  a.X0 = (const bf16_t*)op->p[12];
  a.X1 = (const bf16_t*)op->p[13];
  a.xcin = a.X0 ? op->i[32] : 0;
  a.xc0 = a.X1 ? op->i[33] : a.xcin;
  a.ldx0 = op->i[34] > 0 ? op->i[34] : a.xc0;
  a.ldx1 = a.X1 ? (op->i[35] > 0 ? op->i[35] : a.xcin - a.xc0) : 0;
  a.xcpt = a.xc0t = 0;
  if (a.X0) {
    MG_REQUIRE(a.taps == 9 && a.stride == 1 && a.pad == 1 && a.Hu == 0 && batch_z == 1 && trans_from < 0 && a.epi == MG_EPI_BF16,
               "igemm: a folded 1x1 convolution (p[12]) rides on a plain 3x3 / stride 1 / pad 1 convolution");
    MG_REQUIRE(a.xcin > 0 && a.xcin % 64 == 0 && a.xc0 > 0 && a.xc0 % 64 == 0 && a.xc0 <= a.xcin && (a.X1 != nullptr) == (a.xc0 < a.xcin) &&
               a.ldx0 % 8 == 0 && a.ldx1 % 8 == 0 && (uintptr_t)a.X0 % 16 == 0 && (uintptr_t)a.X1 % 16 == 0 && a.ldw >= a.taps * a.Cin + a.xcin,
               "igemm: bad folded source (Cx %d, Cx0 %d: multiples of 64; weight rows hold taps * Cin + Cx columns)", a.xcin, a.xc0);
  }
  {   // tuning only: phase stamps of every workgroup into the (otherwise idle) split-K workspace - tools/igemm_phases.py
    static const int st = mg_tuning_int("MARIGOLD_IGEMM_STAMPS", 0);
    a.stamps = (st && op->i[31] <= 1 && (variant == 72 || variant == 73)) ? (unsigned long long*)g_splitk_ws : nullptr;
  }
  a.c2 = 0;
  a.inv_c2 = 0.0;
  if (a.epi == MG_EPI_XATTN2) {
    a.w2 = (const bf16_t*)op->p[6];
    a.c2 = op->i[28];
    out2 = nullptr;
    MG_REQUIRE(trans_from < 0 && batch_z == 1 && a.taps == 1 && a.N == 64 && a.sm_cols > 0 && a.sm_cols % 2 == 0 && a.sm_cols <= 64 &&
               a.w2 && (uintptr_t)a.w2 % 16 == 0 && a.c2 > 0 && a.c2 % 32 == 0 && a.ldo >= a.c2 && (long long)a.c2 * 4 + 64 <= MG_ZERO_BYTES &&
               (variant == 0 || variant == 54) && (!a.bias || (uintptr_t)a.bias % 16 == 0) && (!a.res || a.ldr >= a.c2),
               "igemm: the fused cross-attention epilogue takes N = 64 score columns, second-stage weights [c2][64] (c2 %% 32 == 0) in p[6]");
    variant = 54;
    a.inv_c2 = 1.0 / (double)a.c2;
  }
  if (a.epi == MG_EPI_SOFTMAX2)
    MG_REQUIRE(trans_from < 0 && batch_z == 1 && a.sm_cols > 0 && a.sm_cols % 2 == 0 && a.sm_cols <= a.N && !a.res,
               "igemm: the pair-softmax epilogue takes an even number of score columns <= N, no residual / transposed section");
  if (a.ln_out && a.epi == MG_EPI_XATTN2) {
    MG_REQUIRE((uintptr_t)a.ln_out % 8 == 0, "igemm: misaligned (mean, rstd) table");
    a.splits = -1;
  } else if (a.ln_out) {
    MG_REQUIRE((a.M + 63) / 64 <= MG_LN_COUNTERS, "igemm: too many row blocks for the row-statistics tickets (M %d)", a.M);
    MG_REQUIRE(a.epi == MG_EPI_BF16 && trans_from < 0 && a.N % 32 == 0 && batch_z == 1 && (uintptr_t)a.ln_out % 8 == 0,
               "igemm: row statistics (ln_out) need the bf16 epilogue, N %% 32 == 0, no transposed section / batching");
    a.splits = -1;   // the statistics are taken in the tile epilogue, not in the split-K reduction
  }
  if (a.ln_in) {
    MG_REQUIRE(a.taps == 1 && batch_z == 1 && !a.A1 && a.ln_g && a.ln_c && (uintptr_t)a.ln_in % 8 == 0 &&
               (uintptr_t)a.ln_g % 16 == 0 && (uintptr_t)a.ln_c % 16 == 0,
               "igemm: folded LayerNorm needs a Linear layer (taps 1), ln_g / ln_c and the (mean, rstd) table of its input rows");
    a.splits = -1;
  }
  a.kps = 0;
  a.ws = (float*)op->p[14];   // the caller's split-K workspace (MG_SPLITK_WS_BYTES) | NULL = the library's (one stream only)
  a.ctr = 0;
  a.fd_rpi = mg_make_fastdiv(a.rows_per_img > 0 ? a.rows_per_img : 1);
  a.fd_wo = mg_make_fastdiv(a.Wo > 0 ? a.Wo : 1);
  a.fd_per_z = a.fd_tiles = a.fd_tiles_n = a.fd_cpt = mg_make_fastdiv(1);
  a.lin = a.taps == 1 && a.stride == 1 && a.pad == 0 && a.Hu == 0 && a.Ho == a.H && a.Wo == a.W;
  a.inv_n = 1.0 / (double)(a.N > 0 ? a.N : 1);
  // row-block tickets of the ln_out hand-off: the caller's own buffer (i[29] / i[30] = low / high half of its device address;
  // MG_LN_COUNTERS zeroed uint32, one per program / stream - engine.py::Builder) or, absent, the library's global one, which
  // is only safe while every program that writes row statistics runs on ONE stream (the tickets are self-resetting and
  // stream-ordered, but two streams would draw from the same slots)
  a.ln_ctr = (op->i[29] | op->i[30]) ? (unsigned*)(uintptr_t)((uint64_t)(uint32_t)op->i[29] | ((uint64_t)(uint32_t)op->i[30] << 32))
                                     : g_ln_counters;
  a.tiles_m = a.tiles_n = 0;
  MG_REQUIRE(g_zero_page || g_dry_run, "igemm: mg_init() not called");
  MG_REQUIRE(a.A && a.Wt && (a.out || out2), "igemm: null pointer");
  MG_REQUIRE(a.taps == 1 || a.taps == 9 || a.taps == 4, "igemm: taps must be 1, 9 or 4 (got %d)", a.taps);
  if (a.subpix) {
    MG_REQUIRE(batch_z == 4 && a.stride == 1 && a.Hu == 0 && a.Ho == a.H && a.Wo == a.W && trans_from < 0 && !a.res && !a.rowvec &&
               a.epi != MG_EPI_GEGLU && a.sA == 0 && a.sO == 0,
               "igemm: the sub-pixel form (taps = 4) takes batch_z = 4 parities of one stride-1 input, bias only");
  }
  MG_REQUIRE(a.Cin > 0 && a.Cin % 64 == 0, "igemm: Cin %d must be a multiple of 64", a.Cin);
  MG_REQUIRE(a.N > 0 && a.N % 4 == 0, "igemm: N %d must be a multiple of 4", a.N);
  MG_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0, "igemm: lda/ldw must be multiples of 8");
  if (a.A1) {
    MG_REQUIRE(a.C0 > 0 && a.C0 < a.Cin && a.C0 % 64 == 0 && a.lda1 % 8 == 0 && (uintptr_t)a.A1 % 16 == 0 && batch_z == 1 && a.Hu == 0,
               "igemm: bad second source (C0 %d of Cin %d must be a multiple of 64, no up-sampling / batching)", a.C0, a.Cin);
  }
  MG_REQUIRE(a.M > 0 && a.stride >= 1, "igemm: empty problem");
  MG_REQUIRE(((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.Wt % 16 == 0), "igemm: A/Wt need 16-B alignment");
  // generation-2 preconditions (16-byte epilogue accesses, zero-region reach)
  const bool geglu = a.epi == MG_EPI_GEGLU;
  if (a.N % 8 != 0 || a.ldo % 8 != 0 || (a.res && a.ldr % 8 != 0)) return -1;
  if ((long long)a.Cin * 2 + 256 > MG_ZERO_BYTES || (long long)a.taps * a.Cin * 2 + 256 > MG_ZERO_BYTES) return -1;
  if (geglu && a.N % 32 != 0) return -1;
  if (a.out && ((uintptr_t)a.out % 16 != 0)) return -1;
  if (a.res && ((uintptr_t)a.res % 16 != 0)) return -1;
  int rc = 0;
  const int nmain = trans_from >= 0 ? trans_from : a.N;
  const int K = a.taps * a.Cin + a.xcin;
  if (nmain > 0) {
    Igemm2Args m = a;
    m.N = nmain;
    m.n_begin = 0;
    m.n_end = nmain;
    int v = variant ? variant : mg_igemm_auto_variant(m.M, m.n_end, K, batch_z, geglu);
    if (!variant && (v == 72 || v == 73) && !((long long)(a.M / a.rows_per_img) * a.H * a.W * max(max(a.lda, a.lda1), max(a.ldx0, a.ldx1)) < (1ll << 30) && (long long)a.N * a.ldw < (1ll << 30)))
      v = v == 72 ? 62 : 46;   // operands beyond the hand-placed loops' 31-bit byte offsets
    rc = dispatch_tile<false>(m, batch_z, v, s);
    if (rc) return rc;
  }
  if (trans_from >= 0) {
    MG_REQUIRE(out2 && a.ldt > 0 && a.ldt % 8 == 0, "igemm: bad transposed section");
    MG_REQUIRE(!a.tperm || a.rows_per_img % 16 == 0, "igemm: the permuted transposed section needs a multiple of 16 tokens per image (%d)", a.rows_per_img);
    MG_REQUIRE(trans_from % 8 == 0 && (uintptr_t)out2 % 16 == 0, "igemm: transposed section misaligned");
    Igemm2Args tns = a;
    tns.out = out2;
    tns.Wt = a.Wt + (long long)trans_from * a.ldw;
    tns.bias = a.bias ? a.bias + trans_from : nullptr;
    tns.ln_g = a.ln_g ? a.ln_g + trans_from : nullptr;
    tns.ln_c = a.ln_c ? a.ln_c + trans_from : nullptr;
    tns.ln_out = nullptr;
    tns.N = a.N - trans_from;
    tns.n_begin = 0;
    tns.n_end = tns.N;
    tns.ctr = a.N - trans_from;
    tns.rowvec = nullptr;
    tns.res = nullptr;
    tns.epi = MG_EPI_BF16;
    int v = variant ? variant : mg_igemm_auto_variant(tns.M, tns.N, K, batch_z, 0);
    if (!variant && (v == 72 || v == 73)) v = 62;   // (the hand-placed loops have no transposed instantiation)
    rc = dispatch_tile<true>(tns, batch_z, v, s);
  }
  return rc;
}

// Tile choice, from the round-1 sweeps on MI355X (profiles/r1_sweep*_*.log; TFLOP/s at E = 10):
//   * 256x256 / 8 waves (wave tile 128x64) when N is a multiple of 256 and there are >= 512 tiles:
//     VAE 512/256-channel layers 890-1080, GEGLU projections 440-750 - since sweep 10 with the ping-pong
//     K loop (variant 62; variant 34 is the same tile with one barrier per K tile);
//   * 128x320 / 8 waves (wave tile 32x160) for the 320-channel UNet level at large M (730-920);
//   * 128x64 / 4 waves / 3 stages for other N = 128k+64 and for the deep levels where M is a few
//     thousand pixels;
//   * 256x128 / 8 waves / 3 stages for the long-K layers (870-1000), 128x128 / 4 waves / 2 stages
//     (two workgroups per CU) when K <= 1536;
//   all with the next tile's LDS-DMA pieces issued between the k-substeps' MFMA groups.
int mg_igemm_auto_variant(long long M, int N, int K, int batch_z, int geglu) {
  if (N <= 32) return 29;
  {
    // The 12 x 12 level's 3x3 convolutions (M = 1 440 pixels at E = 10, K = 11 520 / 23 040): with 128 x 64 tiles every one
    // of the 12 row tiles re-reads the 30-59 MB of weights; 256 x 256 tiles read them 6 times and split-K (30 tiles x 8
    // splits) fills the chip: 77 -> 68 us (1280 -> 1280), 144 -> 105 us (2560 -> 1280), profiles/r3_deep_conv_tiles.log.
    // MARIGOLD_DEEP_TILE=<variant> | 0 (off) for A/B runs.
    static const int deep = mg_tuning_int("MARIGOLD_DEEP_TILE", 72);
    if (deep && !geglu && batch_z == 1 && M >= 1152 && M <= 2048 && K >= 5760 && N % 256 == 0) return deep;   // (E >= 8 at 12 x 12)
  }
  {
    // Round 4: the hand-placed four-wave K loop (variant 72: one wave per SIMD, 128 x 128 wave tile) for the long-K convolutions
    // whose tile count fits its ONE workgroup per CU - the 24 x 24 level (M = 5 760 at E = 10: 23 x 5 tiles x 2 K splits) and
    // the sub-pixel up-sampling convolutions (4 parities): 2560 -> 1280: 1 237 vs 981 TFLOP/s, 1280 -> 1280: 976 vs 957 with
    // three splits (profiles/r4_k4w_sweep.log).  MARIGOLD_K4W=0 switches it off (A/B).
    static const int k4w = mg_tuning_int("MARIGOLD_K4W", 1);
    const long long t = ((M + 255) / 256) * (N / 256) * batch_z;
    if (k4w && !geglu && N % 256 == 0 && K >= 4096 && ((batch_z == 1 && t >= 64 && t <= 128) || (t >= 200 && t <= 256) || (t >= 400 && t <= 512) || t >= 720)) return 72;
    // ... and its 192 x 320 sibling (variant 73) for the N = 320 k Linear layers and 1x1 convolutions with at least ten K tiles
    // and a chip's worth of tiles: ff.out of the 96 x 96 level 1280 -> 320: 661 vs 584 TFLOP/s, of the 48 x 48 level
    // 2560 -> 640: 997 vs 841, conv_shortcut 960 -> 320: 608 vs 550, 640 -> 640: 537 vs 506.  MARIGOLD_K4WB=0: off.
    static const int k4wb = mg_tuning_int("MARIGOLD_K4WB", 1);
    const long long tb = ((M + 191) / 192) * (N / 320) * batch_z;
    if (k4w && k4wb && !geglu && N % 320 == 0 && N % 256 != 0 && K >= 640 && tb >= 200) return 73;
    // (round 5) the long-K convolutions of the N = 320 k levels keep the tile below a chip's worth of workgroups too (split-K fills
    // the rest: launch2): 72 tiles x 3 splits 64.8 vs 88.5 us for the 256 x 128 tile (640 -> 640 @ 48 x 48, three members),
    // 120 tiles 152 vs 171 us (1280 -> 640, five members) - profiles/r5_sweep_program_E3.tsv, _E5.tsv
    if (k4w && k4wb && !geglu && batch_z == 1 && N % 320 == 0 && N % 256 != 0 && K >= 2560 && tb >= 64) return 73;
  }
  const long long tm256 = (M + 255) / 256;
  // short K (GEGLU projections, K = C linears): 256x128 with 32-deep K tiles - half the LDS per stage,
  // two 8-wave workgroups per CU (GEGLU 525-780 vs 485-770 for 256x256 and 400-700 for the 64-deep tile)
  const bool many256 = N % 256 == 0 && tm256 * (N / 256) * batch_z >= 512;
  if (geglu && K >= 1280 && many256) return 62;   // 1280 -> 10240: 831 vs 760-790
  // (round 2, profiles/r2_sweep5_short_k_tiles.log: the 4-wave form of the same tile - wave tile 128x64, half the
  // fragment reads per MFMA - is 10-13 % ahead of the 8-wave one on the GEGLU projections and the 640-channel linears)
  // (K = 640 linears on 128x128 / 2 stages instead: ahead in the isolated sweep 8, no change in the pipeline - not kept)
  if (N % 128 == 0 && tm256 * (N / 128) * batch_z >= 400 && (geglu || K <= 768)) return 51;
  // 256x256 with the ping-pong schedule (two wave groups one barrier apart, 2nd DMA piece among the MFMAs):
  // +3...10 % over the one-barrier 256x256 tile in interleaved rounds (VAE 512-channel convs 1068 vs 996)
  if (many256) return 62;
  // N = 320 (UNet level 0): a full-width 128x320 tile reads the activation tile once for all output
  // channels (2.1x fewer LDS-DMA bytes per MFMA than 128x64): 900 vs 740 TFLOP/s on the 640->320 convs
  if (N == 320 && ((M + 127) / 128) * batch_z >= 400) return 46;
  const long long t256 = tm256 * ((N + 127) / 128) * batch_z;
  const long long t128x64 = ((M + 127) / 128) * ((N + 63) / 64) * batch_z;
  if (t256 >= 200) {
    if (N % 128 == 64) return 35;
    // long K: the 256x128 tile with the half-K-step software pipeline (round 2, profiles/r2_sweep10_conv_igemm_lean.log:
    // 956-1021 vs 911-976 TFLOP/s for the one-barrier-per-tile loop on the 24x24 / 48x48 convolutions)
    return K <= 1536 ? 32 : 36;
  }
  if (t128x64 >= 96) return 35;
  return 23;
}

int mg_launch_igemm(const mg_op* op, hipStream_t s) {
  const int rc = mg_launch_igemm2(op, s, op->i[19]);
  MG_REQUIRE(rc >= 0, "igemm: unsupported shape (N %d, ldo %d, residual stride %d must be multiples of 8; GEGLU needs N %% 32 == 0; "
             "out / residual 16-byte aligned; K within the zero region)", op->i[6], op->i[13], op->i[16]);
  return rc;
}
