// Implicit-GEMM bf16 MFMA kernel: conv3x3 / conv1x1 / Linear / batched GEMM.
//
//   out[m][n] = epilogue( sum_k A_im2col[m][k] * Wt[n][k] )      m = (img, oy, ox), k = (tap, c)
//
// Replaces torch conv2d / linear / matmul inside diffusers' UNet2DConditionModel and
// AutoencoderKL (reference call sites marigold/marigold_depth_pipeline.py:461-463, 491-492,
// 512-513).  gfx950 design:
//   * NHWC activations: one output pixel row of the im2col matrix is Cin contiguous bf16 per
//     tap, so a 64-wide K tile is ONE 128-byte line per pixel -> fully coalesced 16 B/lane loads.
//   * K tiles stream global -> LDS with `global_load_lds_dwordx4` (no VGPR round trip), double
//     buffered; zero padding / edge rows are sourced from a zero page so the copy stays uniform.
//   * LDS image is [rows][128 B] with the 16-B chunk index XOR-swizzled by (row>>1)&7 (applied on
//     the per-lane SOURCE address, LDS destination stays lane-linear) -> conflict-free
//     ds_read_b128 fragment reads for v_mfma_f32_32x32x16_bf16.
//   * Operands are swapped (weights = MFMA "A", pixels = MFMA "B") so each lane ends up with 4
//     consecutive output channels of one pixel: 8-byte NHWC stores, bias/temb/residual/GEGLU
//     fused in the epilogue.  Column blocks >= trans_from use the un-swapped order and store
//     transposed ([channel][token]) - that is how V^T for the attention kernels is produced.
//   * nearest-2x up-sampling (or up-sampling to an explicit size), stride 2 and the VAE's
//     asymmetric padding are folded into the im2col addressing.
//   * workgroup ids are remapped XCD-aware so tiles sharing activation rows share an L2.
#include "common.h"
extern "C" int mg_igemm_generation(void);

namespace {

struct IgemmArgs {
  const bf16_t* A;
  const bf16_t* Wt;
  void* out;
  const float* bias;
  const float* rowvec;
  const bf16_t* res;
  void* out2;
  const void* zero;
  int H, W, Cin, Ho, Wo, N, taps, stride, pad, Hu, Wu, epi, ldo, trans_from, ldr, lda, ldt, ldw;
  int M, rows_per_img, tiles_m, tiles_n, cpt, KT, rv_stride;
  long long sA, sW, sO, sR;
  float scale;
};

template <int BM, int BN, int WGM, int WGN, bool GLDS>
__global__ __launch_bounds__(WGM* WGN * 64) void igemm_kernel(const IgemmArgs a) {
  constexpr int NT = WGM * WGN * 64;
  constexpr int TM = BM / WGM, TN = BN / WGN, MI = TM / 32, NI = TN / 32;
  constexpr int A_IT = BM * 8 / NT, B_IT = BN * 8 / NT;
  constexpr int STAGE = (BM + BN) * 128;
  static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for the block");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, half = lane >> 5;

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int per_z = a.tiles_m * a.tiles_n;
  const int z = bid / per_z;
  const int t = bid - z * per_z;
  const int tile_m = t / a.tiles_n, tile_n = t - tile_m * a.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const bf16_t* __restrict__ Ab = a.A + (long long)z * a.sA;
  const bf16_t* __restrict__ Wb = a.Wt + (long long)z * a.sW;
  const char* zero = (const char*)a.zero;

  // ---- per-thread staging rows (fixed over the K loop) ----
  int a_by[A_IT], a_bx[A_IT], a_q[A_IT];
  long long a_img[A_IT];
  bool a_ok[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int ci = it * NT + tid;
    const int r = ci >> 3, p = ci & 7;
    a_q[it] = p ^ ((r >> 1) & 7);
    const int m = m0 + r;
    a_ok[it] = m < a.M;
    const int mm = a_ok[it] ? m : 0;
    const int img = mm / a.rows_per_img;
    const int rem = mm - img * a.rows_per_img;
    const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
    a_by[it] = oy * a.stride - a.pad;
    a_bx[it] = ox * a.stride - a.pad;
    a_img[it] = (long long)img * a.H * a.W;
  }
  const bf16_t* b_src[B_IT];
  bool b_ok[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int ci = it * NT + tid;
    const int r = ci >> 3, p = ci & 7;
    const int n = n0 + r;
    b_ok[it] = n < a.N;
    b_src[it] = Wb + (long long)(b_ok[it] ? n : 0) * a.ldw + (p ^ ((r >> 1) & 7)) * 8;
  }
  const bool up2 = (a.Hu == 2 * a.H) && (a.Wu == 2 * a.W);

  uint4 regA[GLDS ? 1 : A_IT], regB[GLDS ? 1 : B_IT];

  auto stage_issue = [&](int kt, int buf) {
    const int tap = kt / a.cpt;
    const int c0 = (kt - tap * a.cpt) * 64;
    int dy = 0, dx = 0;
    if (a.taps == 9) { dy = tap / 3; dx = tap - dy * 3; }
    char* sbase = smem + buf * STAGE;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      int iy = a_by[it] + dy, ix = a_bx[it] + dx;
      bool ok = a_ok[it];
      if (a.Hu) {
        ok = ok && (unsigned)iy < (unsigned)a.Hu && (unsigned)ix < (unsigned)a.Wu;
        if (up2) { iy >>= 1; ix >>= 1; }
        else { iy = ok ? (iy * a.H) / a.Hu : 0; ix = ok ? (ix * a.W) / a.Wu : 0; }
      } else {
        ok = ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      }
      const char* src = ok ? (const char*)(Ab + (a_img[it] + (long long)iy * a.W + ix) * a.lda + c0 +
                                           a_q[it] * 8)
                           : zero;
      if constexpr (GLDS) glds16(src, sbase + (it * NT + wave * 64) * 16);
      else regA[it] = *(const uint4*)src;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const char* src = b_ok[it] ? (const char*)(b_src[it] + (long long)kt * 64) : zero;
      if constexpr (GLDS) glds16(src, sbase + BM * 128 + (it * NT + wave * 64) * 16);
      else regB[it] = *(const uint4*)src;
    }
  };
  auto stage_commit = [&](int buf) {  // register-staged variant only
    if constexpr (!GLDS) {
      char* sbase = smem + buf * STAGE;
#pragma unroll
      for (int it = 0; it < A_IT; ++it) *(uint4*)(sbase + (it * NT + tid) * 16) = regA[it];
#pragma unroll
      for (int it = 0; it < B_IT; ++it)
        *(uint4*)(sbase + BM * 128 + (it * NT + tid) * 16) = regB[it];
    }
  };

  const bool trans_blk = (a.trans_from >= 0) && (n0 >= a.trans_from);

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  int rowA[MI], rowB[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) rowA[mi] = wm * TM + mi * 32 + l31;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) rowB[ni] = wn * TN + ni * 32 + l31;

  stage_issue(0, 0);
  stage_commit(0);
  for (int kt = 0; kt < a.KT; ++kt) {
    const int buf = kt & 1;
    if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < a.KT) stage_issue(kt + 1, buf ^ 1);
    const char* sA = smem + buf * STAGE;
    const char* sB = sA + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int q = ks * 2 + half;
      bf16x8 fa[MI], fb[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int r = rowA[mi];
        fa[mi] = __builtin_bit_cast(bf16x8, *(const uint4*)(sA + r * 128 + ((q ^ ((r >> 1) & 7)) << 4)));
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int r = rowB[ni];
        fb[ni] = __builtin_bit_cast(bf16x8, *(const uint4*)(sB + r * 128 + ((q ^ ((r >> 1) & 7)) << 4)));
      }
      if (!trans_blk) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ni], fa[mi], acc[ni][mi], 0, 0, 0);
      } else {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[ni], acc[ni][mi], 0, 0, 0);
      }
    }
    if (kt + 1 < a.KT) stage_commit(buf ^ 1);
  }

  // ---------------- epilogue ----------------
  const float scale = a.scale;
  if (!trans_blk) {
    // acc[ni][mi][4g+j]: n = nb + 8g + 4*half + j (4 consecutive channels), m = mb + l31
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + wm * TM + mi * 32 + l31;
      if (m >= a.M) continue;
      const int img = m / a.rows_per_img;
      const float* rv = a.rowvec ? a.rowvec + (long long)img * a.rv_stride : nullptr;
      if (a.epi == MG_EPI_GEGLU) {
        // Weight rows are pre-interleaved in 16-row groups: rows [16i,16i+8) = u(8i..8i+7),
        // rows [16i+8,16i+16) = gate(8i..8i+7); a lane's register groups g and g+1 pair up.
        bf16_t* o = (bf16_t*)a.out + (long long)z * a.sO + (long long)m * a.ldo;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int nb = n0 + wn * TN + ni * 32;
#pragma unroll
          for (int g = 0; g < 4; g += 2) {
            const int n = nb + 8 * g + 4 * half;  // u rows n..n+3, gate rows n+8..n+11
            if (n >= a.N) continue;
            float r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float u = acc[ni][mi][4 * g + j] * scale + (a.bias ? a.bias[n + j] : 0.f);
              const float gt = acc[ni][mi][4 * g + 4 + j] * scale + (a.bias ? a.bias[n + 8 + j] : 0.f);
              r[j] = u * gelu_erf_f(gt);
            }
            const int oc = (nb >> 1) + 4 * g + 4 * half;
            uint2 v;
            v.x = pack2bf(r[0], r[1]);
            v.y = pack2bf(r[2], r[3]);
            *(uint2*)(o + oc) = v;
          }
        }
      } else {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int nb = n0 + wn * TN + ni * 32;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = nb + 8 * g + 4 * half;
            if (n >= a.N) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[ni][mi][4 * g + j] * scale;
            if (a.bias) {
              const float4 b4 = *(const float4*)(a.bias + n);
              v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
            }
            if (rv) {
              const float4 r4 = *(const float4*)(rv + n);
              v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            }
            if (a.epi == MG_EPI_F32) {
              float* o = (float*)a.out + (long long)z * a.sO + (long long)m * a.ldo + n;
              *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
            } else {
              if (a.res) {
                const uint2 r2 = *(const uint2*)(a.res + (long long)z * a.sR + (long long)m * a.ldr + n);
                v[0] += bflo(r2.x); v[1] += bfhi(r2.x); v[2] += bflo(r2.y); v[3] += bfhi(r2.y);
              }
              uint2 pk;
              pk.x = pack2bf(v[0], v[1]);
              pk.y = pack2bf(v[2], v[3]);
              *(uint2*)((bf16_t*)a.out + (long long)z * a.sO + (long long)m * a.ldo + n) = pk;
            }
          }
        }
      }
    }
  } else {
    // transposed section: acc[ni][mi][4g+j]: n = nb + l31, m = mb + 8g + 4*half + j
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + wn * TN + ni * 32 + l31;
      if (n >= a.N) continue;
      const float bv = a.bias ? a.bias[n] : 0.f;
      const int nn = n - a.trans_from;
      const int ctr = a.N - a.trans_from;  // channels in the transposed section
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int m = m0 + wm * TM + mi * 32 + 8 * g + 4 * half;
          if (m >= a.M) continue;
          if ((a.rows_per_img & 3) == 0) {  // 4 consecutive tokens of one image: one 8-byte store
            const int img = m / a.rows_per_img;
            const int tok = m - img * a.rows_per_img;
            uint2 pk;
            pk.x = pack2bf(acc[ni][mi][4 * g + 0] * scale + bv, acc[ni][mi][4 * g + 1] * scale + bv);
            pk.y = pack2bf(acc[ni][mi][4 * g + 2] * scale + bv, acc[ni][mi][4 * g + 3] * scale + bv);
            bf16_t* o = (bf16_t*)a.out2 + (long long)z * a.sO + ((long long)img * ctr + nn) * a.ldt + tok;
            *(uint2*)o = pk;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int mj = m + j;
              if (mj >= a.M) break;
              const int img = mj / a.rows_per_img;
              const int tok = mj - img * a.rows_per_img;
              ((bf16_t*)a.out2)[(long long)z * a.sO + ((long long)img * ctr + nn) * a.ldt + tok] =
                  f2bf(acc[ni][mi][4 * g + j] * scale + bv);
            }
          }
        }
      }
    }
  }
}

template <int BM, int BN, int WGM, int WGN, bool GLDS>
int launch_variant(const IgemmArgs& a, int batch_z, hipStream_t s) {
  constexpr int NT = WGM * WGN * 64;
  constexpr int LDS = 2 * (BM + BN) * 128;
  static bool attr_set = false;
  auto kern = igemm_kernel<BM, BN, WGM, WGN, GLDS>;
  if (!attr_set && !g_dry_run) {
    MG_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  IgemmArgs b = a;
  b.tiles_m = (a.M + BM - 1) / BM;
  b.tiles_n = (a.N + BN - 1) / BN;
  const long long grid = (long long)b.tiles_m * b.tiles_n * batch_z;
  MG_REQUIRE(grid > 0 && grid < (1ll << 31), "igemm: bad grid %lld", grid);
  if (a.trans_from >= 0)
    MG_REQUIRE(a.trans_from % BN == 0, "igemm: trans_from %d not a multiple of BN %d", a.trans_from, BN);
  MG_LAUNCH(kern, dim3((unsigned)grid), dim3(NT), LDS, s, b);
  if (!g_dry_run) MG_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int mg_launch_igemm(const mg_op* op, hipStream_t s) {
  {
    const int v = op->i[19];
    if ((v == 0 && mg_igemm_generation() == 2) || v >= 20) {
      const int rc = mg_launch_igemm2(op, s, v);
      if (rc >= 0) return rc;
      MG_REQUIRE(op->i[12] != MG_EPI_GEGLU, "igemm: GEGLU shape unsupported by the generation-2 kernel");
    }
  }
  MG_REQUIRE(!op->p[7], "igemm: the generation-1 kernel has no second channel source");
  IgemmArgs a;
  a.A = (const bf16_t*)op->p[0];
  a.Wt = (const bf16_t*)op->p[1];
  a.out = op->p[2];
  a.bias = (const float*)op->p[3];
  a.rowvec = (const float*)op->p[4];
  a.res = (const bf16_t*)op->p[5];
  a.out2 = op->p[6];
  a.zero = g_zero_page;
  const int B = op->i[0];
  a.H = op->i[1]; a.W = op->i[2]; a.Cin = op->i[3]; a.Ho = op->i[4]; a.Wo = op->i[5];
  a.N = op->i[6]; a.taps = op->i[7]; a.stride = op->i[8]; a.pad = op->i[9];
  a.Hu = op->i[10]; a.Wu = op->i[11]; a.epi = op->i[12]; a.ldo = op->i[13];
  a.trans_from = op->i[14];
  const int batch_z = op->i[15] > 0 ? op->i[15] : 1;
  a.ldr = op->i[16] > 0 ? op->i[16] : a.N;
  a.lda = op->i[17] > 0 ? op->i[17] : a.Cin;
  a.ldt = op->i[18];
  int variant = op->i[19];
  a.ldw = op->i[20] > 0 ? op->i[20] : a.taps * a.Cin;
  a.rv_stride = op->i[21] ? 0 : a.N;  // i[21] != 0: one rowvec row shared by all images
  a.sA = op->l[0]; a.sW = op->l[1]; a.sO = op->l[2]; a.sR = op->l[3];
  a.scale = op->f[0] == 0.f ? 1.f : op->f[0];
  a.rows_per_img = a.Ho * a.Wo;
  a.M = B * a.rows_per_img;
  a.cpt = a.Cin / 64;
  a.KT = a.taps * a.cpt;
  a.tiles_m = a.tiles_n = 0;
  MG_REQUIRE(g_zero_page || g_dry_run, "igemm: mg_init() not called");
  MG_REQUIRE(a.A && a.Wt && (a.out || a.out2), "igemm: null pointer");
  MG_REQUIRE(a.taps == 1 || a.taps == 9, "igemm: taps must be 1 or 9 (got %d)", a.taps);
  MG_REQUIRE(a.Cin > 0 && a.Cin % 64 == 0, "igemm: Cin %d must be a multiple of 64", a.Cin);
  MG_REQUIRE(a.N > 0 && a.N % 4 == 0, "igemm: N %d must be a multiple of 4", a.N);
  MG_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0, "igemm: lda/ldw must be multiples of 8");
  MG_REQUIRE(a.M > 0 && a.stride >= 1, "igemm: empty problem");
  MG_REQUIRE(a.epi != MG_EPI_GEGLU || (a.N % 32 == 0), "igemm: GEGLU needs N %% 32 == 0");
  if (a.trans_from >= 0)
    MG_REQUIRE(a.out2 && a.ldt > 0 && a.ldt % 4 == 0, "igemm: bad transposed section");
  MG_REQUIRE(((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.Wt % 16 == 0), "igemm: A/Wt need 16-B alignment");
  if (variant == 0) {
    const long long t128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128) * batch_z;
    const long long t256 = (long long)((a.M + 255) / 256) * ((a.N + 127) / 128) * batch_z;
    if (t256 >= 512) variant = 2;
    else if (t128 >= 200) variant = 1;
    else variant = 3;
  }
  switch (variant) {
    case 1: return launch_variant<128, 128, 2, 2, true>(a, batch_z, s);
    case 2: return launch_variant<256, 128, 4, 2, true>(a, batch_z, s);
    case 3: return launch_variant<64, 64, 2, 2, true>(a, batch_z, s);
    case 4: return launch_variant<128, 64, 2, 2, true>(a, batch_z, s);
    case 11: return launch_variant<128, 128, 2, 2, false>(a, batch_z, s);  // register-staged A/B
    case 13: return launch_variant<64, 64, 2, 2, false>(a, batch_z, s);
    default: MG_REQUIRE(false, "igemm: unknown tile variant %d", variant);
  }
  return 0;
}
