// Small / HBM-bound kernels around the MFMA path: im2col / pointwise tails of the 4<->C channel convolutions at the
// latent and image boundaries (fp32 NCHW <-> bf16 NHWC), the scheduler update, the time-embedding dense layers and
// the post_quant 1x1 conv.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sched_step_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ mo,
                                                         const float* __restrict__ nz,
                                                         float* __restrict__ out, long long n, float cx,
                                                         float cm, float cn) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (long long)gridDim.x * 256) {
    float v = cx * x[i] + cm * mo[i];
    if (nz) v += cn * nz[i];
    out[i] = v;
  }
}

// one wave per output column n and block of 16 rows: the weight row is read once for all of them (the UNet's time-embedding
// projections are M = T rows x 20 k columns of fp32 weights: ~100 MB that a wave per (m, n) fetched M times).  A lane sums
// k = lane, lane + 64, ... in order, then the wave's butterfly - the order does not depend on the row blocking.
#define LSM_ROWS 16
__global__ __launch_bounds__(256) void linear_small_m_kernel(const float* __restrict__ in,
                                                             const float* __restrict__ Wt,
                                                             const float* __restrict__ bias,
                                                             float* __restrict__ out, int M, int N,
                                                             int K, int act_in, int act_out, int ldo) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int m0 = blockIdx.y * LSM_ROWS;
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  const int rows = min(LSM_ROWS, M - m0);
  const float* wr = Wt + (long long)n * K;
  float s[LSM_ROWS];
#pragma unroll
  for (int r = 0; r < LSM_ROWS; ++r) s[r] = 0.f;
  // four k steps per trip: the weight loads of a trip, then each row's four inputs, are independent loads in flight together
  // (one k step per trip was one ~2 us round trip per step: 260 us for the 103 MB of the stacked projections)
  int k = lane;
  for (; k + 192 < K; k += 256) {
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) w[u] = wr[k + 64 * u];
#pragma unroll
    for (int r = 0; r < LSM_ROWS; ++r) {
      if (r < rows) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = in[(long long)(m0 + r) * K + k + 64 * u];
#pragma unroll
        for (int u = 0; u < 4; ++u) s[r] += (act_in ? silu_f(v[u]) : v[u]) * w[u];
      }
    }
  }
  for (; k < K; k += 64) {
    const float w = wr[k];
#pragma unroll
    for (int r = 0; r < LSM_ROWS; ++r) {
      if (r < rows) {
        float v = in[(long long)(m0 + r) * K + k];
        if (act_in) v = silu_f(v);
        s[r] += v * w;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < LSM_ROWS; ++r) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s[r] += __shfl_xor(s[r], o);
  }
  if (lane == 0) {
    const float bn = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < LSM_ROWS; ++r) {
      if (r < rows) {
        float v = s[r] + bn;
        if (act_out) v = silu_f(v);
        out[(long long)(m0 + r) * ldo + n] = v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void latent_1x1_kernel(const float* __restrict__ in,
                                                         const float* __restrict__ Wt,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ out, int B, int Ci, int Co,
                                                         long long HW, float scale) {
  const long long total = (long long)B * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    const long long b = i / HW, p = i % HW;
    for (int co = 0; co < Co; ++co) {
      float s = bias[co];
      for (int ci = 0; ci < Ci; ++ci) s += Wt[co * Ci + ci] * (in[(b * Ci + ci) * HW + p] * scale);
      out[(b * Co + co) * HW + p] = s;
    }
  }
}

// one thread per output pixel: gather the 3x3 x CIN fp32 neighbourhood, round to bf16, write the
// Kp-wide row with 16-byte stores (reads are coalesced along W, writes are whole rows)
template <int CIN, int KP>
__global__ __launch_bounds__(256) void im2col_small_kernel(const float* __restrict__ src0,
                                                           const float* __restrict__ src1,
                                                           bf16_t* __restrict__ out, int B, int H, int W,
                                                           int C0, int bcast0) {
  const long long hw = (long long)H * W;
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= (long long)B * hw) return;
  const int b = (int)(pix / hw);
  const int rem = (int)(pix % hw);
  const int y = rem / W, x = rem % W;
  float v[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) v[k] = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
    const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      if (ok) {
        v[t * CIN + c] = c < C0 ? src0[((long long)(bcast0 ? 0 : b) * C0 + c) * hw + (long long)iy * W + ix]
                                : src1[((long long)b * (CIN - C0) + (c - C0)) * hw + (long long)iy * W + ix];
      }
    }
  }
  uint4* o = (uint4*)(out + pix * KP);
#pragma unroll
  for (int k = 0; k < KP; k += 8) {
    uint4 pk;
    pk.x = pack2bf(v[k], v[k + 1]); pk.y = pack2bf(v[k + 2], v[k + 3]);
    pk.z = pack2bf(v[k + 4], v[k + 5]); pk.w = pack2bf(v[k + 6], v[k + 7]);
    o[k >> 3] = pk;
  }
}

template <int COUT>
__global__ __launch_bounds__(256) void post_nchw_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        long long npix, long long HW, int ldi, int post,
                                                        float scale, const float* __restrict__ noise, float cx, float cm,
                                                        float cn) {
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < npix;
       pix += (long long)gridDim.x * 256) {
    const long long b = pix / HW, p = pix - b * HW;
    float acc[COUT];
#pragma unroll
    for (int c4 = 0; c4 < COUT; c4 += 4) {
      const float4 v = *(const float4*)(in + pix * ldi + c4);
      acc[c4] = v.x * scale;
      if (c4 + 1 < COUT) acc[c4 + 1] = v.y * scale;
      if (c4 + 2 < COUT) acc[c4 + 2] = v.z * scale;
      if (c4 + 3 < COUT) acc[c4 + 3] = v.w * scale;
    }
    if (post == MG_POST_DEPTH) {
      float m = 0.f;
#pragma unroll
      for (int co = 0; co < COUT; ++co) m += acc[co];
      m = m / (float)COUT;
      m = fminf(fmaxf(m, -1.f), 1.f);
      out[b * HW + p] = (m + 1.0f) * 0.5f;
    } else if (post == MG_POST_NORMALS) {
      float n2 = 0.f;
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        acc[co] = fminf(fmaxf(acc[co], -1.f), 1.f);
        n2 += acc[co] * acc[co];
      }
      const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-6f);
#pragma unroll
      for (int co = 0; co < COUT; ++co) out[(b * COUT + co) * HW + p] = acc[co] * inv;
    } else if (post == MG_POST_SCHED) {
      // the DDIM / LCM update applied to the UNet's output in place of storing it: x <- cx x + cm model_out + cn noise
      // (same operation order as sched_step_kernel: bit-identical to the two-launch form)
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        const long long i = (b * COUT + co) * HW + p;
        float v = cx * out[i] + cm * acc[co];
        if (noise) v += cn * noise[i];
        out[i] = v;
      }
    } else if (post == MG_POST_UNIT) {   // IID: clip to [-1,1], shift to [0,1] (marigold_iid_pipeline.py:523-526)
#pragma unroll
      for (int co = 0; co < COUT; ++co)
        out[(b * COUT + co) * HW + p] = (fminf(fmaxf(acc[co], -1.f), 1.f) + 1.0f) * 0.5f;
    } else {
#pragma unroll
      for (int co = 0; co < COUT; ++co) out[(b * COUT + co) * HW + p] = acc[co];
    }
  }
}

}  // namespace

int mg_launch_misc(const mg_op* op, hipStream_t s) {
  switch (op->kind) {
    case MG_OP_SCHED_STEP: {
      const long long n = op->l[0];
      const int grid = (int)min((n + 255) / 256, (long long)4096);
      MG_LAUNCH(sched_step_kernel, dim3(grid), dim3(256), 0, s, (const float*)op->p[0],
                         (const float*)op->p[1], (const float*)op->p[2], (float*)op->p[3], n,
                         op->f[0], op->f[1], op->f[2]);
      break;
    }
    case MG_OP_LINEAR_SMALL_M: {
      const int M = op->i[0], N = op->i[1], K = op->i[2];
      MG_REQUIRE(M > 0 && M < 65536 && N > 0 && K > 0, "linear_small_m: bad dims");
      MG_LAUNCH(linear_small_m_kernel, dim3((N + 3) / 4, (M + LSM_ROWS - 1) / LSM_ROWS), dim3(256), 0, s,
                         (const float*)op->p[0], (const float*)op->p[1], (const float*)op->p[2],
                         (float*)op->p[3], M, N, K, op->i[3], op->i[4], op->i[5] > 0 ? op->i[5] : N);
      break;
    }
    case MG_OP_LATENT_1X1: {
      const int B = op->i[0], Ci = op->i[1], Co = op->i[2];
      const long long HW = op->i[3];
      const long long total = (long long)B * HW;
      const int grid = (int)min((total + 255) / 256, (long long)4096);
      MG_LAUNCH(latent_1x1_kernel, dim3(grid), dim3(256), 0, s, (const float*)op->p[0],
                         (const float*)op->p[1], (const float*)op->p[2], (float*)op->p[3], B, Ci, Co,
                         HW, op->f[0] == 0.f ? 1.f : op->f[0]);
      break;
    }
    case MG_OP_IM2COL_SMALL: {
      const int B = op->i[0], H = op->i[1], W = op->i[2], C0 = op->i[3], C1 = op->i[4], Kp = op->i[5];
      const int cin = C0 + C1;
      MG_REQUIRE(C1 == 0 || op->p[1], "im2col_small: src1 missing");
      const long long npix = (long long)B * H * W;
      const dim3 grid((unsigned)((npix + 255) / 256));
#define I2C_CASE(N, K)                                                                             \
  if (cin == N && Kp == K) {                                                                       \
    MG_LAUNCH((im2col_small_kernel<N, K>), grid, dim3(256), 0, s, (const float*)op->p[0],         \
              (const float*)op->p[1], (bf16_t*)op->p[2], B, H, W, C0, op->i[6]);                   \
    break;                                                                                         \
  }
      I2C_CASE(3, 64) I2C_CASE(4, 64) I2C_CASE(8, 128) I2C_CASE(12, 128) I2C_CASE(16, 192)
#undef I2C_CASE
      MG_REQUIRE(false, "im2col_small: unsupported (Cin %d, Kp %d): (3,64), (4,64), (8,128), (12,128), (16,192)", cin, Kp);
      break;
    }
    case MG_OP_POST_NCHW: {
      const long long B = op->i[0], HW = op->i[1];
      const int Cout = op->i[2], ldi = op->i[3], post = op->i[4];
      MG_REQUIRE(ldi % 4 == 0 && ldi >= Cout, "post_nchw: ldi %d must be a multiple of 4 >= Cout", ldi);
      const long long npix = B * HW;
      const int grid = (int)min((npix + 255) / 256, (long long)8192);
      const float sc = op->f[0] == 0.f ? 1.f : op->f[0];
      switch (Cout) {
        case 1: MG_LAUNCH(post_nchw_kernel<1>, dim3(grid), dim3(256), 0, s, (const float*)op->p[0], (float*)op->p[1], npix, HW, ldi, post, sc, (const float*)op->p[2], op->f[1], op->f[2], op->f[3]); break;
        case 3: MG_LAUNCH(post_nchw_kernel<3>, dim3(grid), dim3(256), 0, s, (const float*)op->p[0], (float*)op->p[1], npix, HW, ldi, post, sc, (const float*)op->p[2], op->f[1], op->f[2], op->f[3]); break;
        case 4: MG_LAUNCH(post_nchw_kernel<4>, dim3(grid), dim3(256), 0, s, (const float*)op->p[0], (float*)op->p[1], npix, HW, ldi, post, sc, (const float*)op->p[2], op->f[1], op->f[2], op->f[3]); break;
        case 8: MG_LAUNCH(post_nchw_kernel<8>, dim3(grid), dim3(256), 0, s, (const float*)op->p[0], (float*)op->p[1], npix, HW, ldi, post, sc, (const float*)op->p[2], op->f[1], op->f[2], op->f[3]); break;
        case 12: MG_LAUNCH(post_nchw_kernel<12>, dim3(grid), dim3(256), 0, s, (const float*)op->p[0], (float*)op->p[1], npix, HW, ldi, post, sc, (const float*)op->p[2], op->f[1], op->f[2], op->f[3]); break;
        default: MG_REQUIRE(false, "post_nchw: unsupported Cout %d (1, 3, 4, 8 or 12)", Cout);
      }
      break;
    }
    case MG_OP_MEMSET:
      if (!g_dry_run) MG_CHECK_HIP(hipMemsetAsync(op->p[0], op->i[0], (size_t)op->l[0], s));
      break;
    case MG_OP_COPY:
      if (!g_dry_run) MG_CHECK_HIP(hipMemcpyAsync(op->p[1], op->p[0], (size_t)op->l[0], hipMemcpyDeviceToDevice, s));
      break;
    default: MG_REQUIRE(false, "misc: bad op kind %d", op->kind);
  }
  if (!g_dry_run) MG_CHECK_HIP(hipGetLastError());
  return 0;
}
