// EXPERIMENTAL (tile variants 70..; never chosen automatically).
//
// 3x3 / stride 1 / pad 1 convolution as implicit GEMM with a HALO-SHARED activation tile: same contract
// and epilogue as MG_OP_IGEMM (include/marigold_hip.h; reference call sites: the conv1 / conv2 of every
// diffusers ResnetBlock2D, marigold_depth_pipeline.py:461-463, 512-513) for the subset
// {taps = 9, stride 1, pad 1, no virtual up-sampling, bf16 / fp32 epilogue}.
//
// Why: the generation-2 K loop stages a fresh [256 pixels x 64 channels] activation tile for every one of
// the nine taps although the three taps of one kernel row read the SAME pixels shifted by -1 / 0 / +1.  The
// LDS-DMA pieces are what bounds that loop, so here the K order is (ky, channel tile, kx)
// and one activation tile of 256 + 2 halo pixels serves the three kx steps: the MFMA fragment of output
// pixel r for tap kx is LDS row r + kx.  Per three K steps a wave issues 5 + 3 x 2 = 11 pieces instead of
// 3 x (4 + 2) = 18, and the activation is read from L2 / HBM three times instead of nine.
//   * a pixel whose x + kx - 1 falls outside the image row must contribute zero although its neighbour row
//     in LDS holds a real pixel (the end of the previous image row): its fragment address is redirected to a
//     zero row kept at the end of each activation stage (decided once per tile - x does not change);
//   * rows outside the image in y (and beyond the tensor) are staged from the global zero page, decided per
//     staged LDS row once per ky;
//   * every wave issues the same number of pieces every step (dummy pieces from the zero page past the end
//     of the K loop), so the counted `s_waitcnt vmcnt(2 / 5 / 7)` of the three kx steps are constants.
#include <type_traits>

#include "common.h"

namespace {

struct Igemm3Args {
  const bf16_t* A;
  const bf16_t* Wt;
  void* out;
  const float* bias;
  const float* rowvec;
  const bf16_t* res;
  const void* zero;
  int H, W, Cin, N, epi, ldo, ldr, lda, ldw;
  int M, HW, cpt, rv_stride, tiles_m, tiles_n;
  float scale;
};

typedef __attribute__((ext_vector_type(2))) __bf16 h_bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float h_f32x2_t;
__device__ __forceinline__ uint32_t h_cvt_pk_bf16(float lo, float hi) {
  h_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h_bf16x2_t));
}
template <int N>
__device__ __forceinline__ void h_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int H_BM = 256, H_NT = 512, H_ROWB = 128;
constexpr int H_A_BLOCKS = 33;                       // 8-row DMA blocks of the activation tile: 264 rows >= 256 + 2
constexpr int H_ZERO_ROW = H_A_BLOCKS * 8;           // LDS row 264 of every activation stage holds zeros
constexpr int H_A_STAGE = (H_ZERO_ROW + 1) * H_ROWB;  // 33 920 B
constexpr int h_lds_bytes(int bn, int nstb) { return 2 * H_A_STAGE + nstb * bn * H_ROWB; }   // 256x128x3: 116 992 B
// Counted wait at the top of a kx step: the loads that may still be in flight once the step's weight tile (issued
// D = NSTB - 1 steps ago as the FIRST pieces of its step) and, for kx = 0, the whole activation tile (last pieces
// issued in the kx = 1 step of the previous triple) have landed.  PB = weight pieces per wave per step; the next
// activation tile goes out as 3 pieces in the kx = 0 step and 2 in the kx = 1 step, after the weights.
constexpr int h_wait_count(int kx, int d, int pb) {
  const int a[3] = {3, 2, 0};
  int younger = a[(kx + 3 - d % 3) % 3];                 // activation pieces that followed the tile in its own step
  for (int j = 1; j < d; ++j) younger += pb + a[(kx + 3 - j % 3) % 3];
  if (kx == 0) {                                          // activation tile: everything up to step k-2 must be back
    const int act = d >= 2 ? pb : 0;                      // (with D = 1 nothing younger than it has been issued)
    younger = younger < act ? younger : act;
  }
  return younger;
}
static_assert(h_wait_count(0, 2, 2) == 2 && h_wait_count(1, 2, 2) == 5 && h_wait_count(2, 2, 2) == 7, "256x128x3 waits");
static_assert(h_wait_count(0, 1, 4) == 0 && h_wait_count(1, 1, 4) == 3 && h_wait_count(2, 1, 4) == 2, "256x256x2 waits");

// XOR swizzle of the 16-byte chunk index by the LDS row (both on the DMA source address and on the fragment
// read), as in igemm2.hip: conflict-free ds_read_b128 on 128-byte rows.
__device__ __forceinline__ int h_swz(int chunk, int row) { return chunk ^ ((row >> 1) & 7); }

// SPLIT: the step's LDS-DMA pieces are issued between the k-substeps' fragment reads and MFMAs (same issue ORDER as
// the burst form, so the counted waits are unchanged) instead of in one burst after the barrier.
// Geometry: 256 output pixels x BN channels per workgroup, 8 waves as WGM x WGN; NSTB weight stages.
//   <128, 4, 2, 3>: wave tile 64 x 64, 117 KB LDS  (variants 70 / 71 - the form whose first GPU run was correct)
//   <256, 2, 4, 2>: wave tile 128 x 64, 132 KB LDS (variants 72 / 73 - not yet run)
template <int BN, int WGM, int WGN, int NSTB, bool SPLIT>
__global__ __launch_bounds__(H_NT) void igemm3_halo_kernel(const Igemm3Args a) {
  static_assert(WGM * WGN * 64 == H_NT && (NSTB == 2 || NSTB == 3), "8 waves, 2 or 3 weight stages");
  constexpr int TM = H_BM / WGM, TN = BN / WGN, MI = TM / 32, NI = TN / 32;
  constexpr int PB = BN * 8 / H_NT;            // weight pieces per wave per K step
  constexpr int D = NSTB - 1;                  // weight prefetch distance in K steps
  constexpr int H_B_STAGE = BN * H_ROWB;
  static_assert(MI >= 1 && NI >= 1 && PB >= 1 && h_lds_bytes(BN, NSTB) <= 160 * 1024, "tile geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sA = smem;                       // [2][265 rows][128 B]
  char* const sB = smem + 2 * H_A_STAGE;       // [NSTB][BN rows][128 B]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int l31 = lane & 31, half = lane >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = bid / a.tiles_n, tile_n = bid - tile_m * a.tiles_n;
  const int m0 = tile_m * H_BM, n0 = tile_n * BN;
  const char* zero = (const char*)a.zero;

  // zero rows of both activation stages, visible before any DMA is in flight
  if (tid < 16) *(uint4*)(sA + (tid >> 3) * H_A_STAGE + H_ZERO_ROW * H_ROWB + (tid & 7) * 16) = make_uint4(0, 0, 0, 0);
  __syncthreads();

  // ---- activation staging: 5 pieces per wave per (ky, channel tile); piece i covers LDS rows blk*8 .. +7,
  //      blk = min(i*8 + wave, 32) (the pieces past block 32 repeat it: same bytes to the same place) ----
  uint32_t a_ptr[5];   // byte offsets from a.A (the tensor is < 4 GiB, checked by the launcher)
  unsigned a_valid = 0;
  auto a_blk = [&](int i) { return min(i * 8 + wave, H_A_BLOCKS - 1); };   // wave-uniform: the LDS-DMA base
  auto a_slot = [&](int i) { return a_blk(i) * 8 + (lane >> 3); };          // this lane's LDS row of piece i
  auto a_setup = [&](int ky) {   // LDS row `slot` holds input pixel m0 - 1 + slot + (ky - 1) * W
    a_valid = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const long long q = (long long)m0 - 1 + a_slot(i) + (long long)(ky - 1) * a.W;
      bool ok = q >= 0 && q < a.M;
      const int qq = ok ? (int)q : 0;
      const int ys = (qq % a.HW) / a.W;        // the output row this input row serves is ys - (ky - 1)
      ok = ok && (unsigned)(ys - ky + 1) < (unsigned)a.H;
      a_ptr[i] = (uint32_t)(((long long)qq * a.lda + h_swz(lane & 7, a_slot(i)) * 8) * 2);
      a_valid |= ok ? (1u << i) : 0u;
    }
  };
  auto a_issue = [&](int stage, int i) {
    glds16((a_valid >> i) & 1 ? (const char*)a.A + a_ptr[i] : zero, sA + stage * H_A_STAGE + a_blk(i) * 8 * H_ROWB);
  };
  int i_ky = 0, i_c = 0;   // the (ky, channel tile) whose activation tile is issued next
  auto a_advance = [&]() {
    if (++i_c == a.cpt) {
      i_c = 0;
      ++i_ky;
      if (i_ky < 3) a_setup(i_ky);
      else a_valid = 0;   // past the end: dummy pieces from the zero page keep the vmcnt arithmetic uniform
    } else {
#pragma unroll
      for (int i = 0; i < 5; ++i) a_ptr[i] += H_ROWB;
    }
  };

  // ---- weight staging: PB pieces per wave per K step; K step (ky, c, kx) reads k = (ky*3 + kx)*Cin + c*64 ----
  uint32_t b_base[PB];   // byte offsets from a.Wt (< 4 GiB, checked by the launcher); ~0 = row beyond N
#pragma unroll
  for (int it = 0; it < PB; ++it) {
    const int ci = it * H_NT + tid;
    const int row = ci >> 3, p = ci & 7;
    const int n = n0 + row;
    b_base[it] = n < a.N ? (uint32_t)(((long long)n * a.ldw + h_swz(p, row) * 8) * 2) : 0xffffffffu;
  }
  int b_ky = 0, b_c = 0, b_kx = 0;   // the K step whose weight tile is issued next
  auto b_piece = [&](int stage, int it) {
    const uint32_t kofs = (uint32_t)(((b_ky * 3 + b_kx) * a.Cin + b_c * 64) * 2);
    glds16((b_ky < 3 && b_base[it] != 0xffffffffu) ? (const char*)a.Wt + (b_base[it] + kofs) : zero,
           sB + stage * H_B_STAGE + (it * H_NT + wave * 64) * 16);
  };
  auto b_advance = [&]() {
    if (++b_kx == 3) {
      b_kx = 0;
      if (++b_c == a.cpt) { b_c = 0; ++b_ky; }
    }
  };
  auto b_issue = [&](int stage) {
#pragma unroll
    for (int it = 0; it < PB; ++it) b_piece(stage, it);
    b_advance();
  };

  // ---- fragment addresses (fixed for the tile): output row r, tap kx -> LDS row r + kx, or the zero row ----
  int a_off[MI][3], b_off[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int r = wm * TM + mi * 32 + l31;
    const int x = ((m0 + r) % a.HW) % a.W;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const bool ok = (unsigned)(x + kx - 1) < (unsigned)a.W;
      const int j = ok ? r + kx : H_ZERO_ROW;
      a_off[mi][kx] = j * H_ROWB + (h_swz(half, j) << 4);   // k-substep ks adds (ks << 5) by XOR
    }
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int r = wn * TN + ni * 32 + l31;
    b_off[ni] = r * H_ROWB + (h_swz(half, r) << 4);
  }

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  // ---- prologue: activation tile of (ky 0, c 0), weight tiles of the first D K steps ----
  a_setup(0);
#pragma unroll
  for (int i = 0; i < 5; ++i) a_issue(0, i);
  a_advance();
#pragma unroll
  for (int d = 0; d < D; ++d) b_issue(d);

  int st_b = 0;   // weight stage of the current K step; the step issues into (st_b + D) % NSTB
  int st_a = 0;   // activation stage of the current (ky, c)
  auto step = [&](auto kx_tag) {
    constexpr int KX = decltype(kx_tag)::value;
    // Nothing of the previous step may drift below this point: its fragment reads must have RETURNED before the
    // barrier, because right after it other waves issue DMA into the stage those reads come from.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // younger loads that may stay in flight: see the header (issue order inside a step: weights, then activation)
    h_wait_vmcnt<h_wait_count(KX, D, PB)>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int st_i = st_b >= 1 ? st_b - 1 : NSTB - 1;   // (st_b + D) % NSTB
    // the step's pieces, in issue order: the weights, then the next activation tile's 0..2 (kx 0) / 3, 4 (kx 1)
    constexpr int NP = PB + (KX == 0 ? 3 : KX == 1 ? 2 : 0);
    auto piece = [&](int i) {
      if (i < PB) b_piece(st_i, i);
      else a_issue(st_a ^ 1, (KX == 0 ? 0 : 3) + i - PB);
      if (i == PB - 1) b_advance();
      if (KX == 1 && i == NP - 1) a_advance();
    };
    if constexpr (!SPLIT) {
#pragma unroll
      for (int i = 0; i < NP; ++i) piece(i);
    }
    const char* pa = sA + st_a * H_A_STAGE;
    const char* pb = sB + st_b * H_B_STAGE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 fa[MI], fb[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) fa[mi] = __builtin_bit_cast(bf16x8, *(const uint4*)(pa + (a_off[mi][KX] ^ (ks << 5))));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) fb[ni] = __builtin_bit_cast(bf16x8, *(const uint4*)(pb + (b_off[ni] ^ (ks << 5))));
      if constexpr (SPLIT) {   // pieces i with i * 4 / NP == ks, in order
#pragma unroll
        for (int i = 0; i < NP; ++i)
          if (i * 4 / NP == ks) piece(i);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ni], fa[mi], acc[ni][mi], 0, 0, 0);
    }
    st_b = st_b == NSTB - 1 ? 0 : st_b + 1;
  };
  const int triples = 3 * a.cpt;
  for (int t = 0; t < triples; ++t) {
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
    st_a ^= 1;
  }
  h_wait_vmcnt<0>();   // the dummy pieces of the last steps must not outlive the workgroup's LDS

  // ---- epilogue (igemm2.hip's bf16 / fp32 path): acc[ni][mi][4g+j] = C[m = mb + l31][n = nb + 8g + 4*half + j];
  //      a lane^32 exchange leaves 8 consecutive output channels per lane -> 16-byte stores ----
  const float scale = a.scale;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int mb = m0 + wm * TM + mi * 32, nb = n0 + wn * TN + ni * 32;
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) half_swap(acc[ni][mi][8 * gp + j], acc[ni][mi][8 * gp + 4 + j], v[j], v[4 + j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= scale;
        const int m = mb + l31, n = nb + 16 * gp + 8 * half;
        if (m < a.M && n < a.N) {
          if (a.bias) {
            const float4 b0 = *(const float4*)(a.bias + n), b1 = *(const float4*)(a.bias + n + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
          }
          if (a.rowvec) {
            const float* rv = a.rowvec + (long long)(m / a.HW) * a.rv_stride + n;
            const float4 r0 = *(const float4*)rv, r1 = *(const float4*)(rv + 4);
            v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
            v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
          }
          if (a.epi == MG_EPI_F32) {
            float* o = (float*)a.out + (long long)m * a.ldo + n;
            *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
          } else {
            if (a.res) {
              const uint4 r4 = *(const uint4*)(a.res + (long long)m * a.ldr + n);
              v[0] += bflo(r4.x); v[1] += bfhi(r4.x); v[2] += bflo(r4.y); v[3] += bfhi(r4.y);
              v[4] += bflo(r4.z); v[5] += bfhi(r4.z); v[6] += bflo(r4.w); v[7] += bfhi(r4.w);
            }
            uint4 pk;
            pk.x = h_cvt_pk_bf16(v[0], v[1]); pk.y = h_cvt_pk_bf16(v[2], v[3]);
            pk.z = h_cvt_pk_bf16(v[4], v[5]); pk.w = h_cvt_pk_bf16(v[6], v[7]);
            *(uint4*)((bf16_t*)a.out + (long long)m * a.ldo + n) = pk;
          }
        }
      }
    }
  }
}

}  // namespace

// Entry for tile variants 70..79 of MG_OP_IGEMM (explicit only).  Returns an error for shapes outside the subset.
int mg_launch_igemm3(const mg_op* op, hipStream_t s, int variant) {
  Igemm3Args a;
  a.A = (const bf16_t*)op->p[0];
  a.Wt = (const bf16_t*)op->p[1];
  a.out = op->p[2];
  a.bias = (const float*)op->p[3];
  a.rowvec = (const float*)op->p[4];
  a.res = (const bf16_t*)op->p[5];
  a.zero = g_zero_page;
  const int B = op->i[0];
  a.H = op->i[1]; a.W = op->i[2]; a.Cin = op->i[3];
  a.N = op->i[6]; a.epi = op->i[12]; a.ldo = op->i[13];
  a.ldr = op->i[16] > 0 ? op->i[16] : a.N;
  a.lda = op->i[17] > 0 ? op->i[17] : a.Cin;
  a.ldw = op->i[20] > 0 ? op->i[20] : 9 * a.Cin;
  a.rv_stride = op->i[21] ? 0 : a.N;
  a.scale = op->f[0] == 0.f ? 1.f : op->f[0];
  a.HW = a.H * a.W;
  a.M = B * a.HW;
  a.cpt = a.Cin / 64;
  MG_REQUIRE(variant >= 70 && variant <= 73, "igemm: unknown halo tile variant %d", variant);
  MG_REQUIRE(g_zero_page || g_dry_run, "igemm: mg_init() not called");
  MG_REQUIRE(a.A && a.Wt && a.out, "igemm(halo): null pointer");
  MG_REQUIRE(op->i[7] == 9 && op->i[8] == 1 && op->i[9] == 1 && op->i[10] == 0 && op->i[11] == 0 &&
                 op->i[4] == a.H && op->i[5] == a.W,
             "igemm(halo): only 3x3 / stride 1 / pad 1 convolutions without up-sampling");
  MG_REQUIRE(op->i[14] < 0 && op->i[15] <= 1, "igemm(halo): no transposed section, no batch");
  MG_REQUIRE(a.epi == MG_EPI_BF16 || a.epi == MG_EPI_F32, "igemm(halo): bf16 / fp32 epilogue only");
  MG_REQUIRE(a.Cin > 0 && a.Cin % 64 == 0 && a.N > 0 && a.N % 8 == 0, "igemm(halo): Cin %% 64, N %% 8");
  MG_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldo % 8 == 0 && a.ldr % 8 == 0, "igemm(halo): leading dims %% 8");
  MG_REQUIRE((long long)B * a.HW < (1ll << 31) - 1024, "igemm(halo): too many pixels");
  MG_REQUIRE((long long)a.M * a.lda * 2 < (1ll << 32) - 65536 && (long long)a.N * a.ldw * 2 < (1ll << 32) - 65536,
             "igemm(halo): operands must be smaller than 4 GiB (32-bit staging offsets)");
  MG_REQUIRE(((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.Wt % 16 == 0) && ((uintptr_t)a.out % 16 == 0) &&
                 (!a.res || (uintptr_t)a.res % 16 == 0),
             "igemm(halo): 16-byte alignment");
  const int bn = variant >= 72 ? 256 : 128;
  const int lds = variant >= 72 ? h_lds_bytes(256, 2) : h_lds_bytes(128, 3);
  a.tiles_m = (a.M + H_BM - 1) / H_BM;
  a.tiles_n = (a.N + bn - 1) / bn;
  void (*kern)(const Igemm3Args);
  switch (variant) {
    case 70: kern = igemm3_halo_kernel<128, 4, 2, 3, false>; break;   // burst DMA issue after the barrier
    case 71: kern = igemm3_halo_kernel<128, 4, 2, 3, true>; break;    // pieces issued between the k-substeps
    case 72: kern = igemm3_halo_kernel<256, 2, 4, 2, false>; break;   // 256 x 256, two weight stages
    default: kern = igemm3_halo_kernel<256, 2, 4, 2, true>; break;
  }
  static bool attr_set[4] = {false, false, false, false};
  if (!attr_set[variant - 70] && !g_dry_run) {
    MG_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set[variant - 70] = true;
  }
  const long long grid = (long long)a.tiles_m * a.tiles_n;
  MG_REQUIRE(grid > 0 && grid < (1ll << 31), "igemm(halo): bad grid %lld", grid);
  MG_LAUNCH(kern, dim3((unsigned)grid), dim3(H_NT), lds, s, a);
  if (!g_dry_run) MG_CHECK_HIP(hipGetLastError());
  return 0;
}
