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
//    (The compiled kernel here is generation 2.5, flash25_body.h; the 9 216 / 2 304-token launches run the hand-placed
//    stream of flash4w.hip.  Generations 1-3 were retired in round 5: profiles/r3_flash_variants.log holds their numbers.)
//  * softmax_rows: fp32 -> bf16 row softmax for single-head attention of a width other than 512 (MG_OP_FLASH_ATTN512
//    covers the published VAE), whose scores are materialised by the GEMM kernel.
#include <type_traits>

#include "common.h"
#include "flash_args.h"
#include "flash25_body.h"

namespace {

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
      a.redo_thr = op->f[1] > 0.f ? op->f[1] : (MG_F16 ? 32768.0f : 1.2676506e30f);   // 2^100 (tests force the fallback with a tiny value)
      a.ws = op->p[5];                                            // variant 26: workspace of the key-split blocks (i[8] KB)
      a.ws_bytes = (long long)op->i[8] * 1024;
      a.split = op->i[9];
      a.n_full = a.n_rem = a.n_rem_wg = 0;
      a.m16 = 0;
      MG_REQUIRE(!a.ws || ((uintptr_t)a.ws % 16 == 0 && a.ws_bytes > 0), "flash_attn64: workspace must be 16-byte aligned, its size (KB) in i[8]");
      a.nqb = (a.Ntok + FA_QB - 1) / FA_QB;
      MG_REQUIRE(g_zero_page || g_dry_run, "flash_attn64: mg_init() not called");
      MG_REQUIRE(a.Q && a.K && a.Vt && a.O, "flash_attn64: null pointer");
      MG_REQUIRE(a.B > 0 && a.heads > 0 && a.Ntok > 0, "flash_attn64: empty problem");
      MG_REQUIRE(a.ldvt % 64 == 0 && a.ldvt >= a.Ntok, "flash_attn64: ldvt must be a multiple of 64 >= Ntok");
      MG_REQUIRE(a.ldq % 8 == 0 && a.ldo % 4 == 0, "flash_attn64: bad leading dims");
      // i[6]: 0 = automatic; 19 / 20 / 21 / 25 force a form of the compiled kernel (generation 2.5: 4 / 8 waves with the permuted
      // V^T, 4 waves with the natural one, 4 waves + plain v_add_f32 row sums), 26 the hand-placed stream - tests and sweeps
      MG_REQUIRE((a.ldo % 8 == 0) && ((uintptr_t)a.O % 16 == 0) && (a.sO % 8 == 0), "flash_attn64: O needs 16-byte rows (ldo %% 8 == 0, aligned base / batch stride)");
      const int var = op->i[6];
      const bool vt_perm = op->i[7] != 0;   // V^T keys permuted inside every group of 16: [0-3, 8-11, 4-7, 12-15]
      MG_REQUIRE(var == 0 || vt_perm == (var == 19 || var == 20 || var == 25 || var == 26 || var == 27), "flash_attn64: variant %d and the V^T key order (i[7] = %d) do not match", var, op->i[7]);
      MG_REQUIRE(!vt_perm || a.Ntok % 16 == 0, "flash_attn64: the permuted V^T layout needs Ntok %% 16 == 0");
      const long long g4 = (long long)((a.Ntok + 127) / 128) * a.heads * a.B;
      const long long g8 = (long long)((a.Ntok + 255) / 256) * a.heads * a.B;
      switch (var) {
        case 19: MG_LAUNCH((flash_attn64_v25_kernel<4, true>), dim3((unsigned)g4), dim3(256), 0, s, a); break;   // generation 2.5, vt_perm
        case 20: MG_LAUNCH((flash_attn64_v25_kernel<8, true>), dim3((unsigned)g8), dim3(512), 0, s, a); break;
        case 21: MG_LAUNCH((flash_attn64_v25_kernel<4, false>), dim3((unsigned)g4), dim3(256), 0, s, a); break;  // natural V^T
        case 25: MG_LAUNCH((flash_attn64_v25_kernel<4, true, 2>), dim3((unsigned)g4), dim3(256), 0, s, a); break;   // plain v_add_f32 row sums
        case 26:   // the hand-placed stream (flash4w.hip) on 32x32x16 MFMAs
        case 27:   // ... on 16x16x32 MFMAs
          MG_REQUIRE(mg_flash4w_ok(a, vt_perm), "flash_attn64 variant %d: Ntok %d must be a multiple of 256 (even number of key tiles), V^T permuted", var, a.Ntok);
          a.m16 = var == 27;
          return mg_launch_flash4w(a, s);
        case 0: {
          // round 3 (profiles/r3_flash_variants*.log, TFLOP/s at E = 10): generation 2.5 on 4-wave workgroups (three per CU)
          // beat the earlier generations at every sequence length (9216 tokens: 903 vs 827-890) with either V^T order
          // round 4: the row sums as plain v_add_f32 on two chains instead of v_pk_add_f32 (a packed fp32 add beside MFMAs costs
          // more than the two adds it replaces, MI355X_MICROARCH.md): 9216 tokens 975 vs 947 TFLOP/s, 2304: 795 vs 743
          // round 4: the hand-placed one-wave-per-SIMD stream (flash4w.hip) where its shape constraints hold (the 96 x 96 and
          // 48 x 48 levels: 9 216 / 2 304 tokens): 1 160-1 260 vs 907-928 TFLOP/s at 9 216 tokens (profiles/r4_flash4w.log).
          static const int f4w = mg_tuning_int("MARIGOLD_FLASH4W", 1);
          static const int f4w_m16 = mg_tuning_int("MARIGOLD_FLASH4W_M16", -1);   // -1: by shape (mg_launch_flash4w)
          a.m16 = f4w_m16;
          if (f4w && vt_perm && mg_flash4w_ok(a, true)) return mg_launch_flash4w(a, s);
          if (vt_perm) MG_LAUNCH((flash_attn64_v25_kernel<4, true, 2>), dim3((unsigned)g4), dim3(256), 0, s, a);
          else MG_LAUNCH((flash_attn64_v25_kernel<4, false>), dim3((unsigned)g4), dim3(256), 0, s, a);
          break;
        }
        default: MG_REQUIRE(false, "flash_attn64: unknown variant %d", var);
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
    default: MG_REQUIRE(false, "attention: bad op kind %d", op->kind);
  }
  if (!g_dry_run) MG_CHECK_HIP(hipGetLastError());
  return 0;
}
