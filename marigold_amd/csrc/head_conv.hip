// MG_OP_CONV3X3_HEAD: the output heads - GroupNorm apply + SiLU + conv3x3 (pad 1) to <= 4 channels in ONE launch
// (diffusers: conv_norm_out -> conv_act -> conv_out of UNet2DConditionModel and of the VAE decoder; reference call sites
// marigold_depth_pipeline.py:461-463 and :498-516 via the modules' forward).
//
// Why its own kernel: with <= 4 output channels the convolution is not MFMA work (the implicit GEMM padded them to 8 and
// fetched every input pixel nine times - once per tap - from L2: 13.6 GB for the decoder's ten 768^2 x 128 maps, 1.2 ms, after a
// 0.6 ms pass that had materialised the normalised tensor).  Here a workgroup stages the RAW input patch of a 16 x (16 | 32)
// pixel tile once, 32 channels at a time, normalising on the way into LDS (same arithmetic and bf16 rounding as gn_apply_kernel),
// and every thread walks the nine taps of its one or two output pixels out of LDS with packed bf16 dot products
// (v_dot2c_f32_bf16), fp32 accumulation.  HBM: one read of the input; bound by it.
//
//   x bf16 [B][H][W][C]   ss f32 [B][2][C] (scale, shift; NULL = no normalisation)   w bf16 [>= Cout][9 C], k = tap * C + c
//   bias f32   out f32 [B H W][ldo] (columns [0, Cout))      C % 32 == 0, 1 <= Cout <= 4
#include "common.h"

namespace {

typedef __bf16 hc_v2bf16 __attribute__((ext_vector_type(2)));

struct HeadArgs {
  const bf16_t* x;
  const float* ss;
  const bf16_t* w;
  const float* bias;
  float* out;
  int B, H, W, C, ldo, silu, tiles_x, tiles_y;
};

constexpr int HC_CK = 32;          // channels per LDS pass
constexpr int HC_PS = 80;          // bytes per patch pixel in LDS: 64 of data + 16 (a 16-lane read group then hits every bank once)
constexpr int HC_TW = 16;

typedef _Float16 hc_v2f16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float hc_dot2(uint32_t a, uint32_t b, float c) {   // v_dot2c_f32_bf16 / v_dot2c_f32_f16
  if constexpr (MG_F16) return __builtin_amdgcn_fdot2(__builtin_bit_cast(hc_v2f16, a), __builtin_bit_cast(hc_v2f16, b), c, false);
  else return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hc_v2bf16, a), __builtin_bit_cast(hc_v2bf16, b), c, false);
}

template <int COUT, int PPT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PPT == 2 ? 2 : 4))) void head_conv_kernel(const HeadArgs a) {
  constexpr int TH = 16 * PPT, PH = TH + 2, PW = HC_TW + 2;
  constexpr int PATCH = PH * PW * HC_PS, WCH = COUT * 9 * HC_CK * 2;
  __shared__ __attribute__((aligned(16))) char smem[PATCH + WCH + 2 * HC_CK * 4];
  char* const patch = smem;
  char* const wl = smem + PATCH;
  float* const ssl = (float*)(smem + PATCH + WCH);   // [2][32] scale, shift of the pass
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int txi = t % a.tiles_x; t /= a.tiles_x;
  const int tyi = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int x0 = txi * HC_TW, y0 = tyi * TH;
  const int px = tid & 15, py = (tid >> 4) * PPT;
  const bf16_t* const xb = a.x + (long long)b * a.H * a.W * a.C;
  float acc[PPT][COUT];
#pragma unroll
  for (int p = 0; p < PPT; ++p)
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[p][co] = 0.f;

  // The raw operands of a pass travel global -> registers -> (normalise) -> LDS, and the registers of pass c + 1 are requested
  // BEFORE pass c is computed: as a loop of dependent loads (first version) a pass cost ten HBM round trips per thread, 60 us per
  // tile against 6 us of dot products.  The weights stay LDS broadcast reads (144 of a thread's 180 LDS reads per pass at four
  // output channels: the kernel is LDS-bound, 0.80 ms for the decoder's ten maps); as scalar loads with SGPR operands - tried - the
  // dot products wait on lgkmcnt, which scalar memory returns out of order and shares with LDS: 1.04 ms.
  constexpr int NIT = (PH * PW * 4 + 255) / 256;   // 16-byte patch items per thread and pass
  constexpr int NWI = COUT * 9 * 4;                // 16-byte weight items per pass (<= 144: one per thread)
  uint4 u[NIT], wreg = make_uint4(0, 0, 0, 0);
  float ssreg = 0.f;
  auto request = [&](int c0) {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = tid + k * 256;
      const int part = i & 3, pix = i >> 2;
      const int pr = pix / PW, pc = pix - pr * PW;
      const int gy = y0 + pr - 1, gx = x0 + pc - 1;
      const bool in = i < PH * PW * 4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      u[k] = in ? *(const uint4*)(xb + ((long long)gy * a.W + gx) * a.C + c0 + part * 8) : make_uint4(0, 0, 0, 0);
    }
    if (tid < NWI) {   // [co][tap][32 channels]
      const int part = tid & 3, ct = tid >> 2, co = ct / 9, tap = ct - co * 9;
      wreg = *(const uint4*)(a.w + (long long)co * 9 * a.C + (long long)tap * a.C + c0 + part * 8);
    }
    if (a.ss && tid < 2 * HC_CK) ssreg = a.ss[((long long)b * 2 + (tid >> 5)) * a.C + c0 + (tid & 31)];
  };
  request(0);
  for (int c0 = 0; c0 < a.C; c0 += HC_CK) {
    __syncthreads();   // the previous pass' reads of LDS are done
    if (tid < 2 * HC_CK) ssl[tid] = ssreg;
    if (tid < NWI) *(uint4*)(wl + tid * 16) = wreg;
    __syncthreads();   // (scale / shift visible)
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = tid + k * 256;
      if (i >= PH * PW * 4) break;
      const int part = i & 3, pix = i >> 2;
      const int pr = pix / PW, pc = pix - pr * PW;
      const int gy = y0 + pr - 1, gx = x0 + pc - 1;
      uint4 o = u[k];   // out-of-image pixels were requested as zeros: the convolution's zero padding sits BEHIND norm + SiLU
      if (a.ss && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
        float v[8] = {bflo(o.x), bfhi(o.x), bflo(o.y), bfhi(o.y), bflo(o.z), bfhi(o.z), bflo(o.w), bfhi(o.w)};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[j] = __builtin_fmaf(v[j], ssl[part * 8 + j], ssl[HC_CK + part * 8 + j]);
          if (a.silu) v[j] = silu_fast_f(v[j]);
        }
        o.x = cvt_pk_bf16_f32(v[0], v[1]); o.y = cvt_pk_bf16_f32(v[2], v[3]);
        o.z = cvt_pk_bf16_f32(v[4], v[5]); o.w = cvt_pk_bf16_f32(v[6], v[7]);
      }
      *(uint4*)(patch + pix * HC_PS + part * 16) = o;
    }
    __syncthreads();
    if (c0 + HC_CK < a.C) request(c0 + HC_CK);   // in flight under this pass' dot products
    // rows py .. py + PPT + 1 of the patch feed this thread's PPT pixels: a row is read once and used by both
#pragma unroll
    for (int r = 0; r < PPT + 2; ++r) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const char* pp = patch + ((py + r) * PW + px + dx) * HC_PS;
#pragma unroll
        for (int part = 0; part < 4; ++part) {
          const uint4 xv = *(const uint4*)(pp + part * 16);
#pragma unroll
          for (int p = 0; p < PPT; ++p) {
            const int dy = r - p;   // compile-time after unrolling
            if (dy < 0 || dy > 2) continue;
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
              const uint4 wv = *(const uint4*)(wl + ((co * 9 + dy * 3 + dx) * 4 + part) * 16);   // wave-uniform: a broadcast read
              float s = acc[p][co];
              s = hc_dot2(xv.x, wv.x, s); s = hc_dot2(xv.y, wv.y, s);
              s = hc_dot2(xv.z, wv.z, s); s = hc_dot2(xv.w, wv.w, s);
              acc[p][co] = s;
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);   // a tap column at a time: hipcc otherwise hoists every weight read of the pass (250+ registers)
      }
    }
  }
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int gy = y0 + py + p, gx = x0 + px;
    if (gy < a.H && gx < a.W) {
      float* po = a.out + (((long long)b * a.H + gy) * a.W + gx) * a.ldo;
#pragma unroll
      for (int co = 0; co < COUT; ++co) po[co] = acc[p][co] + (a.bias ? a.bias[co] : 0.f);
    }
  }
}

template <int COUT>
int launch_head(const HeadArgs& a0, hipStream_t s) {
  HeadArgs a = a0;
  a.tiles_x = (a.W + HC_TW - 1) / HC_TW;
  // two pixels per thread (16 x 32 tiles: a patch row feeds both, the weight reads are shared) while that still leaves the
  // chip >= 2 workgroups per CU; else 16 x 16 tiles
  const long long t32 = (long long)a.B * a.tiles_x * ((a.H + 31) / 32);
  if (t32 >= 512) {
    a.tiles_y = (a.H + 31) / 32;
    MG_LAUNCH((head_conv_kernel<COUT, 2>), dim3((unsigned)t32), dim3(256), 0, s, a);
  } else {
    a.tiles_y = (a.H + 15) / 16;
    MG_LAUNCH((head_conv_kernel<COUT, 1>), dim3((unsigned)((long long)a.B * a.tiles_x * a.tiles_y)), dim3(256), 0, s, a);
  }
  return 0;
}

}  // namespace

int mg_launch_head_conv(const mg_op* op, hipStream_t s) {
  HeadArgs a;
  a.x = (const bf16_t*)op->p[0];
  a.ss = (const float*)op->p[1];
  a.w = (const bf16_t*)op->p[2];
  a.bias = (const float*)op->p[3];
  a.out = (float*)op->p[4];
  a.B = op->i[0]; a.H = op->i[1]; a.W = op->i[2]; a.C = op->i[3];
  const int cout = op->i[4];
  a.ldo = op->i[5] > 0 ? op->i[5] : cout;
  a.silu = op->i[6];
  a.tiles_x = a.tiles_y = 0;
  MG_REQUIRE(a.x && a.w && a.out && a.B > 0 && a.H > 0 && a.W > 0, "conv3x3_head: bad arguments");
  // the activation is applied where the norm is (the staging pass): a SiLU without scale / shift is not a form this kernel has
  MG_REQUIRE(a.ss || !a.silu, "conv3x3_head: i[6] (SiLU) needs the GroupNorm scale / shift in p[1] (pass scale 1, shift 0 for a bare activation)");
  MG_REQUIRE(a.C > 0 && a.C % HC_CK == 0 && cout >= 1 && cout <= 4 && a.ldo >= cout,
             "conv3x3_head: C %d must be a multiple of %d, 1 <= Cout %d <= 4 <= ldo %d", a.C, HC_CK, cout, a.ldo);
  MG_REQUIRE((uintptr_t)a.x % 16 == 0 && (uintptr_t)a.w % 16 == 0 && (uintptr_t)a.out % 4 == 0, "conv3x3_head: misaligned operands");
  MG_REQUIRE((long long)a.B * ((a.W + 15) / 16) * ((a.H + 15) / 16) < (1ll << 31), "conv3x3_head: grid too large");
  int rc;
  switch (cout) {
    case 1: rc = launch_head<1>(a, s); break;
    case 2: rc = launch_head<2>(a, s); break;
    case 3: rc = launch_head<3>(a, s); break;
    default: rc = launch_head<4>(a, s); break;
  }
  if (rc) return rc;
  if (!g_dry_run) MG_CHECK_HIP(hipGetLastError());
  return 0;
}
