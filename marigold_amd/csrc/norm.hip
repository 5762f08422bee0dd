// GroupNorm (stats partials -> per-(image,channel) scale/shift -> apply [+SiLU]) and LayerNorm
// on bf16 NHWC activations.  All HBM-bound streaming kernels: 16-byte (8 x bf16) accesses,
// fp32 statistics, bit-reproducible (no atomics anywhere).
// Replaces torch group_norm / silu / layer_norm inside diffusers ResnetBlock2D,
// Transformer2DModel and BasicTransformerBlock (reached from
// marigold/marigold_depth_pipeline.py:461-463, 491-492, 512-513).
#include "common.h"

namespace {

constexpr int GN_NV = 4;  // channel vectors per thread: supports C <= 256*8*GN_NV

// grid (chunks, B); block 256.  partials[b][chunk][c][2] = (sum, sumsq) over the chunk's rows.
// Threads are laid out txn (channel vectors) x tyn (row lanes); every thread keeps fp32 partials
// of its rows in registers (4 rows in flight), then the row lanes are combined through LDS in a
// FIXED order - bit-reproducible, no atomics.
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x,
                                                       float* __restrict__ partials, int HW, int C,
                                                       int chunks) {
  extern __shared__ float lds[];  // [tyn][C][2]
  const int cv = C >> 3;
  const int txn = cv < 256 ? cv : 256;
  const int tyn = 256 / txn;
  const int tx = threadIdx.x % txn, ty = threadIdx.x / txn;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int rpc = (HW + chunks - 1) / chunks;
  const int r0 = chunk * rpc;
  const int r1 = min(HW, r0 + rpc);
  float s[GN_NV][8], q[GN_NV][8];
#pragma unroll
  for (int v = 0; v < GN_NV; ++v)
#pragma unroll
    for (int j = 0; j < 8; ++j) s[v][j] = q[v][j] = 0.f;
  const bool active = ty < tyn;
  if (active) {
    const bf16_t* xb = x + (long long)b * HW * C;
    auto acc8 = [&](int v, const uint4& u) {
      const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = bflo(w[j]), c = bfhi(w[j]);
        s[v][2 * j] += a; q[v][2 * j] += a * a;
        s[v][2 * j + 1] += c; q[v][2 * j + 1] += c * c;
      }
    };
#pragma unroll
    for (int v = 0; v < GN_NV; ++v) {
      const int vc = tx + v * txn;
      if (vc >= cv) break;
      const bf16_t* col = xb + vc * 8;
      int r = r0 + ty;
      for (; r + 3 * tyn < r1; r += 4 * tyn) {  // 4 independent 16-byte loads in flight
        const uint4 u0 = *(const uint4*)(col + (long long)r * C);
        const uint4 u1 = *(const uint4*)(col + (long long)(r + tyn) * C);
        const uint4 u2 = *(const uint4*)(col + (long long)(r + 2 * tyn) * C);
        const uint4 u3 = *(const uint4*)(col + (long long)(r + 3 * tyn) * C);
        acc8(v, u0); acc8(v, u1); acc8(v, u2); acc8(v, u3);
      }
      for (; r < r1; r += tyn) acc8(v, *(const uint4*)(col + (long long)r * C));
    }
#pragma unroll
    for (int v = 0; v < GN_NV; ++v) {
      const int vc = tx + v * txn;
      if (vc < cv) {
        float* d = lds + ((long long)ty * C + vc * 8) * 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) { d[2 * j] = s[v][j]; d[2 * j + 1] = q[v][j]; }
      }
    }
  }
  __syncthreads();
  float* out = partials + ((long long)b * chunks + chunk) * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float t = 0.f;
    for (int y = 0; y < tyn; ++y) t += lds[(long long)y * 2 * C + i];
    out[i] = t;
  }
}

// one workgroup per (b, group): fixed-order tree over the chunk partials (fp64), then
// ss[b][0][c] = scale = rstd * gamma, ss[b][1][c] = shift = beta - mean * scale
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partials,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          float* __restrict__ ss, int B, int C,
                                                          int groups, int chunks, int HW, float eps) {
  __shared__ double red[2][4];
  const int b = blockIdx.x / groups, g = blockIdx.x % groups;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cpg = C / groups;
  double s = 0.0, q = 0.0;
  const int n = chunks * cpg;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int ch = i / cpg, c = g * cpg + (i % cpg);
    const float* p = partials + (((long long)b * chunks + ch) * C + c) * 2;
    s += (double)p[0];
    q += (double)p[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    q += __shfl_xor(q, o);
  }
  if (lane == 0) { red[0][wave] = s; red[1][wave] = q; }
  __syncthreads();
  s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const double cnt = (double)HW * cpg;
  const double mean = s / cnt;
  double var = q / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  for (int i = threadIdx.x; i < cpg; i += 256) {
    const int c = g * cpg + i;
    const float sc = rstd * gamma[c];
    ss[((long long)b * 2 + 0) * C + c] = sc;
    ss[((long long)b * 2 + 1) * C + c] = beta[c] - (float)mean * sc;
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x,
                                                       const float* __restrict__ ss,
                                                       bf16_t* __restrict__ out, long long nvec,
                                                       int HW, int C, int silu) {
  const int cv = C >> 3;
  const long long per_img = (long long)HW * cv;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * 256) {
    const int b = (int)(i / per_img);
    const int vc = (int)(i % cv);
    const uint4 u = *(const uint4*)(x + i * 8);
    const float* sc = ss + ((long long)b * 2) * C + vc * 8;
    const float* sh = sc + C;
    const float4 s0 = *(const float4*)sc, s1 = *(const float4*)(sc + 4);
    const float4 h0 = *(const float4*)sh, h1 = *(const float4*)(sh + 4);
    float v[8] = {bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y), bflo(u.z), bfhi(u.z), bflo(u.w), bfhi(u.w)};
    const float scs[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float shs[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = v[j] * scs[j] + shs[j];
      if (silu) v[j] = silu_f(v[j]);
    }
    uint4 o;
    o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
    o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
    *(uint4*)(out + i * 8) = o;
  }
}

// One workgroup per (image, group): pass 1 accumulates sum / sum of squares of the group's HW x cpg
// slice (bf16 pairs, strided rows - the slice is tens of KB and L2-resident), a fixed-order block
// reduction gives mean / rstd, pass 2 re-reads the slice, applies scale / shift (+SiLU) and writes it.
__global__ __launch_bounds__(256) void gn_fused_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, bf16_t* __restrict__ out,
                                                       int HW, int C, int groups, int silu, float eps) {
  __shared__ double red[2][4];
  __shared__ float stat[2];
  const int g = blockIdx.x, b = blockIdx.y;
  const int cpg = C / groups, ppr = cpg >> 1;  // bf16 pairs per row of the slice
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bf16_t* xb = x + (long long)b * HW * C + g * cpg;
  bf16_t* ob = out + (long long)b * HW * C + g * cpg;
  const int total = HW * ppr;
  float s = 0.f, q = 0.f;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int r = i / ppr, p = i - r * ppr;
    const uint32_t w = *(const uint32_t*)(xb + (long long)r * C + 2 * p);
    const float a = bflo(w), c = bfhi(w);
    s += a + c;
    q += a * a + c * c;
  }
  double sd = s, qd = q;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sd += __shfl_xor(sd, o);
    qd += __shfl_xor(qd, o);
  }
  if (lane == 0) { red[0][wave] = sd; red[1][wave] = qd; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double S = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const double Q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double cnt = (double)HW * cpg;
    const double mean = S / cnt;
    double var = Q / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    stat[0] = (float)mean;
    stat[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const float mean = stat[0], rstd = stat[1];
  for (int i = threadIdx.x; i < total; i += 256) {
    const int r = i / ppr, p = i - r * ppr;
    const int c = g * cpg + 2 * p;
    const uint32_t w = *(const uint32_t*)(xb + (long long)r * C + 2 * p);
    const float sc0 = rstd * gamma[c], sc1 = rstd * gamma[c + 1];
    float v0 = bflo(w) * sc0 + (beta[c] - mean * sc0);
    float v1 = bfhi(w) * sc1 + (beta[c + 1] - mean * sc1);
    if (silu) { v0 = silu_f(v0); v1 = silu_f(v1); }
    *(uint32_t*)(ob + (long long)r * C + 2 * p) = pack2bf(v0, v1);
  }
}

// LayerNorm over the last dim: one wave per R rows, all R rows' 16-byte loads issued before any
// reduction (with a single row in flight per wave the kernel sits at ~2.7 TB/s: bytes in flight =
// waves x 640 B, Little's law).  C <= 64 * 8 * NV.
template <int NV, int R>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        bf16_t* __restrict__ out, int M, int C,
                                                        float eps) {
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  const int lane = threadIdx.x & 63;
  if (row0 >= M) return;
  const int cv = C >> 3;
  uint4 u[R][NV];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = min(row0 + r, M - 1);
    const bf16_t* xr = x + (long long)row * C;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int vc = lane + k * 64;
      u[r][k] = vc < cv ? *(const uint4*)(xr + vc * 8) : make_uint4(0, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (row0 + r >= M) break;
    float v[NV][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      v[k][0] = bflo(u[r][k].x); v[k][1] = bfhi(u[r][k].x); v[k][2] = bflo(u[r][k].y); v[k][3] = bfhi(u[r][k].y);
      v[k][4] = bflo(u[r][k].z); v[k][5] = bfhi(u[r][k].z); v[k][6] = bflo(u[r][k].w); v[k][7] = bfhi(u[r][k].w);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[k][j];   // padding vectors are zero
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (lane + k * 64 < cv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[k][j] - mean; q += d * d; }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)C + eps);
    bf16_t* orow = out + (long long)(row0 + r) * C;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int vc = lane + k * 64;
      if (vc < cv) {
        const float4 g0 = *(const float4*)(gamma + vc * 8), g1 = *(const float4*)(gamma + vc * 8 + 4);
        const float4 b0 = *(const float4*)(beta + vc * 8), b1 = *(const float4*)(beta + vc * 8 + 4);
        const float gs[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bs[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float rr[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) rr[j] = (v[k][j] - mean) * rstd * gs[j] + bs[j];
        uint4 o;
        o.x = pack2bf(rr[0], rr[1]); o.y = pack2bf(rr[2], rr[3]);
        o.z = pack2bf(rr[4], rr[5]); o.w = pack2bf(rr[6], rr[7]);
        *(uint4*)(orow + vc * 8) = o;
      }
    }
  }
}

}  // namespace

int mg_launch_norm(const mg_op* op, hipStream_t s) {
  switch (op->kind) {
    case MG_OP_GN_STATS: {
      const int B = op->i[0], HW = op->i[1], C = op->i[2], chunks = op->i[3];
      MG_REQUIRE(C % 8 == 0 && C <= 256 * 8 * GN_NV, "gn_stats: unsupported C %d", C);
      MG_REQUIRE(B > 0 && HW > 0 && chunks > 0 && chunks <= HW, "gn_stats: bad dims");
      const int cvv = C / 8, tynn = 256 / (cvv < 256 ? cvv : 256);
      MG_LAUNCH(gn_stats_kernel, dim3(chunks, B), dim3(256), (size_t)tynn * 2 * C * sizeof(float), s,
                         (const bf16_t*)op->p[0], (float*)op->p[1], HW, C, chunks);
      break;
    }
    case MG_OP_GN_FINALIZE: {
      const int B = op->i[0], C = op->i[1], groups = op->i[2], chunks = op->i[3], HW = op->i[4];
      MG_REQUIRE(C % groups == 0, "gn_finalize: C %d not divisible by groups %d", C, groups);
      MG_LAUNCH(gn_finalize_kernel, dim3(B * groups), dim3(256), 0, s,
                         (const float*)op->p[0], (const float*)op->p[1], (const float*)op->p[2],
                         (float*)op->p[3], B, C, groups, chunks, HW, op->f[0]);
      break;
    }
    case MG_OP_GN_APPLY: {
      const int B = op->i[0], HW = op->i[1], C = op->i[2];
      MG_REQUIRE(C % 8 == 0, "gn_apply: C %d must be a multiple of 8", C);
      const long long nvec = (long long)B * HW * (C / 8);
      const int grid = (int)min((nvec + 255) / 256, (long long)256 * 16);
      MG_LAUNCH(gn_apply_kernel, dim3(grid), dim3(256), 0, s, (const bf16_t*)op->p[0],
                         (const float*)op->p[1], (bf16_t*)op->p[2], nvec, HW, C, op->i[3]);
      break;
    }
    case MG_OP_GN_FUSED: {
      const int B = op->i[0], HW = op->i[1], C = op->i[2], groups = op->i[3];
      MG_REQUIRE(B > 0 && HW > 0 && groups > 0 && C % groups == 0 && (C / groups) % 2 == 0,
                 "gn_fused: C %d / groups %d must give an even channel count per group", C, groups);
      MG_LAUNCH(gn_fused_kernel, dim3(groups, B), dim3(256), 0, s, (const bf16_t*)op->p[0], (const float*)op->p[1],
                (const float*)op->p[2], (bf16_t*)op->p[3], HW, C, groups, op->i[4], op->f[0]);
      break;
    }
    case MG_OP_LAYERNORM: {
      const int M = op->i[0], C = op->i[1];
      const int cv = C / 8;
      MG_REQUIRE(C % 8 == 0 && cv <= 256, "layernorm: unsupported C %d (multiple of 8, <= 2048)", C);
#define LN_LAUNCH(NV, R)                                                                                   \
  MG_LAUNCH((layernorm_kernel<NV, R>), dim3((M + 4 * R - 1) / (4 * R)), dim3(256), 0, s, (const bf16_t*)op->p[0], \
            (const float*)op->p[1], (const float*)op->p[2], (bf16_t*)op->p[3], M, C, op->f[0])
      if (cv <= 64) LN_LAUNCH(1, 4);
      else if (cv <= 128) LN_LAUNCH(2, 4);
      else if (cv <= 192) LN_LAUNCH(3, 2);
      else LN_LAUNCH(4, 2);
#undef LN_LAUNCH
      break;
    }
    default: MG_REQUIRE(false, "norm: bad op kind %d", op->kind);
  }
  if (!g_dry_run) MG_CHECK_HIP(hipGetLastError());
  return 0;
}
