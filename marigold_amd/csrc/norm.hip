// GroupNorm (stats partials -> per-(image,channel) scale/shift -> apply [+SiLU]) and LayerNorm
// on bf16 NHWC activations.  All HBM-bound streaming kernels: 16-byte (8 x bf16) accesses,
// fp32 statistics, bit-reproducible (no atomics anywhere).
// Replaces torch group_norm / silu / layer_norm inside diffusers ResnetBlock2D,
// Transformer2DModel and BasicTransformerBlock (reached from
// marigold/marigold_depth_pipeline.py:461-463, 491-492, 512-513).
#include "common.h"

namespace {

constexpr int GN_NV = 4;  // channel vectors per thread: supports C <= 256*8*GN_NV

// grid (chunks, B); block 256.  partials[b][slot][group][2] = (sum, sumsq) of the group's channels THAT THIS SOURCE HOLDS
// over the chunk's rows, slot = source * chunks + chunk (a norm over the un-materialised concat of two tensors gets one
// statistics launch per source; a group that straddles the two gets a contribution from each).
// Threads are laid out txn (channel vectors) x tyn (row lanes); every thread keeps fp32 partials of its rows in
// registers (4 rows in flight), the row lanes and then the channels of a group are combined through LDS in a FIXED
// order - bit-reproducible, no floating-point atomics.
// With ss != nullptr the image's last block to arrive also reduces the table to scale / shift (no finalize launch).
template <bool SC1>   // SC1: the table was written by other workgroups of this launch with sc1 stores - read it past the L1
__device__ __forceinline__ void gn_finalize_image(const float* __restrict__ partials, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float* __restrict__ ss, int b, int C,
                                                  int groups, int slots, int HW, float eps, int tid, int nthreads) {
  // 8 threads per group: thread (g, sub) sums slots sub, sub + 8, ... in fp64, then the 8 are combined by shuffles
  const int cpg = C / groups;
  for (int g0 = 0; g0 < groups; g0 += nthreads / 8) {
    const int g = g0 + tid / 8, sub = tid & 7;
    double sd = 0.0, qd = 0.0;
    if (g < groups) {
      // (sum, sum of squares) pairs are 8-byte aligned: one load each, and GNF of them in flight before any is consumed -
      // the first form waited for every 4-byte load in turn, ~20-40 dependent L2 round trips (10 us of an 18 us launch)
      const unsigned long long* p = (const unsigned long long*)(partials + ((long long)b * slots * groups + g) * 2);
      constexpr int GNF = 8;
      for (int sl0 = sub; sl0 < slots; sl0 += 8 * GNF) {
        unsigned long long v[GNF];
#pragma unroll
        for (int i = 0; i < GNF; ++i) {
          const int sl = sl0 + 8 * i;
          v[i] = 0ull;
          if (sl < slots) {
            if constexpr (SC1) v[i] = __hip_atomic_load(p + (long long)sl * groups, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v[i] = p[(long long)sl * groups];
          }
        }
#pragma unroll
        for (int i = 0; i < GNF; ++i) {   // fixed order: sl0, sl0 + 8, ... (zeros beyond the table add nothing)
          sd += (double)__uint_as_float((unsigned)v[i]);
          qd += (double)__uint_as_float((unsigned)(v[i] >> 32));
        }
      }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
      sd += __shfl_xor(sd, o);
      qd += __shfl_xor(qd, o);
    }
    if (g < groups) {
      const double cnt = (double)HW * cpg;
      const double mean = sd / cnt;
      double var = qd / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      const float rstd = (float)(1.0 / sqrt(var + (double)eps));
      for (int i = sub; i < cpg; i += 8) {
        const int c = g * cpg + i;
        const float sc = rstd * gamma[c];
        ss[((long long)b * 2 + 0) * C + c] = sc;
        ss[((long long)b * 2 + 1) * C + c] = beta[c] - (float)mean * sc;
      }
    }
  }
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x,
                                                       float* __restrict__ partials, int HW, int C,
                                                       int chunks, int Ctot, int coff, int slot0, int slots,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ ss,
                                                       unsigned* __restrict__ counters, int groups, float eps,
                                                       const bf16_t* __restrict__ x1, int C1) {
  extern __shared__ float lds[];  // [tyn][C][2] row-lane partials, then [C][2] channel totals
  int chunk = blockIdx.x;
  const int b = blockIdx.y;
  if (chunk >= chunks) {   // the second source of the concat (blocks [chunks, 2 chunks)): channels [coff + C, +C1), the next slots
    chunk -= chunks;
    x = x1;
    coff += C;
    C = C1;
    slot0 += chunks;
  }
  const int cv = C >> 3;
  const int txn = cv < 256 ? cv : 256;
  const int tyn = 256 / txn;
  const int tx = threadIdx.x % txn, ty = threadIdx.x / txn;
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
      for (; r + 7 * tyn < r1; r += 8 * tyn) {  // 8 independent 16-byte loads in flight
        uint4 u[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = *(const uint4*)(col + (long long)(r + i * tyn) * C);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc8(v, u[i]);
      }
      if (r < r1) {   // the tail as ONE batch (round 3): up to 7 loads in flight from clamped rows, zeroed past the chunk - as a
                      // loop of single loads it was up to 7 serial HBM round trips (of ~2 us) at the end of every workgroup
        uint4 u[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const int ri = r + i * tyn;
          u[i] = *(const uint4*)(col + (long long)(ri < r1 ? ri : r1 - 1) * C);
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          if (r + i * tyn >= r1) u[i] = make_uint4(0, 0, 0, 0);
          acc8(v, u[i]);
        }
      }
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
  float* tot = lds + (long long)tyn * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float t = 0.f;
    for (int y = 0; y < tyn; ++y) t += lds[(long long)y * 2 * C + i];
    tot[i] = t;
  }
  __syncthreads();
  const int cpg = Ctot / groups;
  float* out = partials + (((long long)b * slots + slot0 + chunk) * groups) * 2;
  for (int g = threadIdx.x; g < groups; g += 256) {   // this source's share of every group (zero where it has none)
    const int lo = max(g * cpg, coff) - coff, hi = min((g + 1) * cpg, coff + C) - coff;
    float sg = 0.f, qg = 0.f;
    for (int c = lo; c < hi; ++c) { sg += tot[2 * c]; qg += tot[2 * c + 1]; }
    if (ss) {   // write-through (sc1) stores: visible at agent scope without a per-block L2 write-back fence
      __hip_atomic_store(&out[2 * g], sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&out[2 * g + 1], qg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      out[2 * g] = sg;
      out[2 * g + 1] = qg;
    }
  }
  if (!ss) return;
  // ---- the image's last block to arrive turns the partial table into scale / shift ----
  // hand-off per cdna_hip_programming.md Guideline 16 (write-through form): sc1 stores of the block's row -> every wave
  // drains them -> barrier -> one relaxed agent-scope ticket; the block that draws the last ticket reads the table with
  // agent-scope loads.
  // The reduction order is fixed by the thread index, not by which block happens to be last: bit-reproducible.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    mg_handoff_release();
    const unsigned ticket = __hip_atomic_fetch_add(&counters[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = ticket == (unsigned)(slots - 1);
    lds[0] = last ? 1.f : 0.f;
    if (last) {
      mg_handoff_acquire();
      __hip_atomic_store(&counters[b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (lds[0] == 0.f) return;
  gn_finalize_image<true>(partials, gamma, beta, ss, b, Ctot, groups, slots, HW, eps, threadIdx.x, 256);
}

// one workgroup per image (the stand-alone form of the reduction above)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partials,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          float* __restrict__ ss, int B, int C,
                                                          int groups, int slots, int HW, float eps) {
  gn_finalize_image<false>(partials, gamma, beta, ss, blockIdx.x, C, groups, slots, HW, eps, threadIdx.x, 256);
}

// grid (row chunks, B); threads laid out like gn_stats: txn channel vectors x tyn row lanes.  A thread
// keeps the scale / shift of its 8 channels in registers and streams its rows (2 rows in flight):
// per 16 bytes moved that is 8 fma + 8 fast SiLU + 4 v_cvt_pk instead of 4 extra vector loads and
// ~100 VALU instructions (the first version was VALU-bound at 4.1 TB/s against 6.8 for a copy).
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x,
                                                       const float* __restrict__ ss,
                                                       bf16_t* __restrict__ out, int HW, int C, int silu,
                                                       int chunks, const bf16_t* __restrict__ x1, int C0) {
  // C = C0 + C1 output channels: [0, C0) from x ([B][HW][C0]), [C0, C) from x1 ([B][HW][C - C0]) - the concat of the
  // UNet's up blocks (torch.cat([hidden, skip], dim=1)) happens in this pass' addressing
  const int cv = C >> 3;
  const int txn = cv < 256 ? cv : 256;
  const int tyn = 256 / txn;
  const int tx = threadIdx.x % txn, ty = threadIdx.x / txn;
  if (ty >= tyn) return;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int rpc = (HW + chunks - 1) / chunks;
  const int r0 = chunk * rpc, r1 = min(HW, r0 + rpc);
  bf16_t* ob = out + (long long)b * HW * C;
  auto norm8 = [&](const uint4& u, const float (&sc)[8], const float (&sh)[8]) {
    float v[8] = {bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y), bflo(u.z), bfhi(u.z), bflo(u.w), bfhi(u.w)};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = __builtin_fmaf(v[j], sc[j], sh[j]);
      if (silu) v[j] = silu_fast_f(v[j]);
    }
    uint4 o;
    o.x = cvt_pk_bf16_f32(v[0], v[1]); o.y = cvt_pk_bf16_f32(v[2], v[3]);
    o.z = cvt_pk_bf16_f32(v[4], v[5]); o.w = cvt_pk_bf16_f32(v[6], v[7]);
    return o;
  };
  for (int vc = tx; vc < cv; vc += txn) {
    const float* scp = ss + ((long long)b * 2) * C + vc * 8;
    const float4 s0 = *(const float4*)scp, s1 = *(const float4*)(scp + 4);
    const float4 h0 = *(const float4*)(scp + C), h1 = *(const float4*)(scp + C + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    const bool second = vc * 8 >= C0;
    const int ldx = second ? C - C0 : C0;
    const bf16_t* col = second ? x1 + (long long)b * HW * ldx + (vc * 8 - C0) : x + (long long)b * HW * ldx + vc * 8;
    bf16_t* ocol = ob + vc * 8;
    int r = r0 + ty;
    for (; r + tyn < r1; r += 2 * tyn) {
      const uint4 u0 = *(const uint4*)(col + (long long)r * ldx);
      const uint4 u1 = *(const uint4*)(col + (long long)(r + tyn) * ldx);
      *(uint4*)(ocol + (long long)r * C) = norm8(u0, sc, sh);
      *(uint4*)(ocol + (long long)(r + tyn) * C) = norm8(u1, sc, sh);
    }
    if (r < r1) *(uint4*)(ocol + (long long)r * C) = norm8(*(const uint4*)(col + (long long)r * ldx), sc, sh);
  }
}


// ---- GroupNorm as ONE launch per norm (MG_OP_GN_SLAB; round 3) ---------------------------------------------------------
// The statistics -> ticket -> finalize -> apply chain above costs 16-28 us of serial latency per norm on tensors a
// streaming read covers in 5-10 us (782 + 536 launches, 31 ms per map).  Here a workgroup owns COMPLETE groups of one
// image - the channel window [c0, c0 + cw), cw = lcm(channels per group, 4) - over all H x W rows: no cross-workgroup
// reduction, no partial table, no tickets, and with MAXV > 0 the rows stay in registers between the statistics and the
// normalisation (one read of the tensor, the apply pass disappears).  8-byte vectors (4 channels): thread (tx, ty) owns
// channel vector tx of rows ty, ty + nty, ...; its 4 x (sum, sum of squares) go through LDS in a fixed order (bit-
// reproducible), group totals in fp64.  Two sources = the UNet's skip concat (a vector never straddles the sources).
struct GnSlabArgs {
  const bf16_t* x0;
  const bf16_t* x1;
  bf16_t* y;            // nullptr: statistics only (the consumer applies scale / shift itself: conv_patch's fused fix-up)
  const float* gamma;
  const float* beta;
  float* ss;            // [B][2][C] (scale, shift)
  int HW, C, C0, cpg, cw, vw, nty, silu;
  float eps;
  mg_fastdiv fd_vw, fd_ncol, fd_cpg;
};

template <int NT, int MAXV>
__global__ __launch_bounds__(NT) void gn_slab_kernel(const GnSlabArgs a) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int ty = fdiv(tid, a.fd_vw), tx = tid - ty * a.vw;
  const int b = blockIdx.y;
  const int cwin = blockIdx.x * a.cw;            // first channel of the window
  const int c = cwin + 4 * tx;                   // this thread's 4 channels
  const bool active = ty < a.nty;
  const bool second = c >= a.C0;
  const int ld = second ? a.C - a.C0 : a.C0;
  const bf16_t* src = (second ? a.x1 + (long long)b * a.HW * ld + (c - a.C0) : a.x0 + (long long)b * a.HW * ld + c);
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  auto acc4 = [&](const uint2& u) {
    const float v0 = bflo(u.x), v1 = bfhi(u.x), v2 = bflo(u.y), v3 = bfhi(u.y);
    s[0] += v0; q[0] = __builtin_fmaf(v0, v0, q[0]);
    s[1] += v1; q[1] = __builtin_fmaf(v1, v1, q[1]);
    s[2] += v2; q[2] = __builtin_fmaf(v2, v2, q[2]);
    s[3] += v3; q[3] = __builtin_fmaf(v3, v3, q[3]);
  };
  constexpr int NV = MAXV > 0 ? MAXV : 1;
  uint2 v[NV];
  if (active) {
    if constexpr (MAXV > 0) {   // every row of the thread in flight at once, kept for the normalisation
#pragma unroll
      for (int k = 0; k < MAXV; ++k) {
        const int r = ty + k * a.nty;
        v[k] = r < a.HW ? *(const uint2*)(src + (long long)r * ld) : make_uint2(0u, 0u);
      }
#pragma unroll
      for (int k = 0; k < MAXV; ++k) acc4(v[k]);   // rows beyond HW are zeros: they add nothing
    } else {
      for (int r0 = ty; r0 < a.HW; r0 += 8 * a.nty) {
        uint2 u[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int r = r0 + k * a.nty;
          u[k] = r < a.HW ? *(const uint2*)(src + (long long)r * ld) : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc4(u[k]);
      }
    }
  }
  const int ncol = a.cw * 2;                     // (sum, sum of squares) per channel of the window
  float* red = lds;                              // [nty][ncol]
  float* red2 = red + NT * 8;                    // [parts][ncol]
  float* tot = red2 + NT;                        // [ncol]
  float* gst = tot + 256;                        // [groups of the window][2] (mean, rstd)
  float* lss = gst + 64;                         // [2][cw] (scale, shift)
  if (active) {
    float4* d = (float4*)(red + ((long long)ty * a.vw + tx) * 8);
    d[0] = make_float4(s[0], q[0], s[1], q[1]);
    d[1] = make_float4(s[2], q[2], s[3], q[3]);
  }
  __syncthreads();
  const int parts = min(a.nty, NT / ncol);       // >= 1: ncol <= 256 <= NT
  {
    const int part = fdiv(tid, a.fd_ncol), col = tid - part * ncol;
    if (part < parts) {
      float t = 0.f;
      for (int y = part; y < a.nty; y += parts) t += red[(long long)y * ncol + col];
      red2[part * ncol + col] = t;
    }
  }
  __syncthreads();
  if (tid < ncol) {
    float t = 0.f;
    for (int p = 0; p < parts; ++p) t += red2[p * ncol + tid];
    tot[tid] = t;
  }
  __syncthreads();
  const int gpw = fdiv(a.cw, a.fd_cpg);
  if (tid < gpw) {
    double sd = 0.0, qd = 0.0;
    for (int i = 0; i < a.cpg; ++i) { sd += (double)tot[(tid * a.cpg + i) * 2]; qd += (double)tot[(tid * a.cpg + i) * 2 + 1]; }
    const double cnt = (double)a.HW * a.cpg;
    const double mean = sd / cnt;
    double var = qd / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    gst[2 * tid] = (float)mean;
    gst[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
  }
  __syncthreads();
  if (tid < a.cw) {
    const int g = fdiv(tid, a.fd_cpg);
    const int cg = cwin + tid;
    const float sc = gst[2 * g + 1] * a.gamma[cg];
    const float sh = a.beta[cg] - gst[2 * g] * sc;
    a.ss[((long long)b * 2 + 0) * a.C + cg] = sc;
    a.ss[((long long)b * 2 + 1) * a.C + cg] = sh;
    lss[tid] = sc;
    lss[a.cw + tid] = sh;
  }
  if constexpr (MAXV > 0) {
    __syncthreads();
    if (!active) return;
    const float4 sc = *(const float4*)(lss + 4 * tx), sh = *(const float4*)(lss + a.cw + 4 * tx);
    bf16_t* dst = a.y + (long long)b * a.HW * a.C + c;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int r = ty + k * a.nty;
      if (r >= a.HW) break;
      float o0 = __builtin_fmaf(bflo(v[k].x), sc.x, sh.x), o1 = __builtin_fmaf(bfhi(v[k].x), sc.y, sh.y);
      float o2 = __builtin_fmaf(bflo(v[k].y), sc.z, sh.z), o3 = __builtin_fmaf(bfhi(v[k].y), sc.w, sh.w);
      if (a.silu) { o0 = silu_fast_f(o0); o1 = silu_fast_f(o1); o2 = silu_fast_f(o2); o3 = silu_fast_f(o3); }
      *(uint2*)(dst + (long long)r * a.C) = make_uint2(cvt_pk_bf16_f32(o0, o1), cvt_pk_bf16_f32(o2, o3));
    }
  }
}

template <int NT, int MAXV>
int launch_gn_slab(const GnSlabArgs& a, int B, hipStream_t s) {
  const size_t lds = (size_t)(NT * 9 + 256 + 64 + 256) * sizeof(float);
  MG_LAUNCH((gn_slab_kernel<NT, MAXV>), dim3(a.C / a.cw, B), dim3(NT), lds, s, a);
  return 0;
}

}  // namespace

int mg_launch_norm(const mg_op* op, hipStream_t s) {
  switch (op->kind) {
    case MG_OP_GN_STATS: {
      const int B = op->i[0], HW = op->i[1], C = op->i[2], chunks = op->i[3];
      MG_REQUIRE(C % 8 == 0 && C <= 256 * 8 * GN_NV, "gn_stats: unsupported C %d", C);
      MG_REQUIRE(B > 0 && HW > 0 && chunks > 0 && chunks <= HW, "gn_stats: bad dims");
      const int cvv = C / 8, tynn = 256 / (cvv < 256 ? cvv : 256);
      const int Ctot = op->i[4] > 0 ? op->i[4] : C, coff = op->i[5];
      const int groups = op->i[6], slot0 = op->i[7], slots = op->i[8] > 0 ? op->i[8] : chunks;
      MG_REQUIRE(coff >= 0 && coff + C <= Ctot && coff % 8 == 0, "gn_stats: channel window [%d,+%d) outside %d", coff, C, Ctot);
      MG_REQUIRE(groups > 0 && Ctot % groups == 0, "gn_stats: %d channels not divisible into %d groups", Ctot, groups);
      // p[6] / i[9]: a second source (C1 channels right behind the first's) in the same launch - blocks [chunks, 2 chunks)
      const bf16_t* x1 = (const bf16_t*)op->p[6];
      const int C1 = x1 ? op->i[9] : 0;
      const int nsrc = x1 ? 2 : 1;
      MG_REQUIRE(!x1 || (C1 > 0 && C1 % 8 == 0 && C1 <= 256 * 8 * GN_NV && coff + C + C1 <= Ctot), "gn_stats: second source of %d channels", C1);
      MG_REQUIRE(slot0 >= 0 && slot0 + nsrc * chunks <= slots, "gn_stats: slots [%d,+%d) outside %d", slot0, nsrc * chunks, slots);
      MG_REQUIRE((uintptr_t)op->p[1] % 8 == 0, "gn_stats: the partial table needs 8-byte alignment");
      // optional fused finalize: p[2] gamma p[3] beta p[4] scale_shift [B][2][Ctot] p[5] per-image arrival counters
      float* ssout = (float*)op->p[4];
      if (ssout) MG_REQUIRE(op->p[2] && op->p[3] && op->p[5], "gn_stats: fused finalize needs gamma, beta and counters");
      size_t lds = (size_t)(tynn + 1) * 2 * C * sizeof(float);
      if (x1) {
        const int cv1 = C1 / 8, tyn1 = 256 / (cv1 < 256 ? cv1 : 256);
        lds = max(lds, (size_t)(tyn1 + 1) * 2 * C1 * sizeof(float));
      }
      MG_LAUNCH(gn_stats_kernel, dim3(nsrc * chunks, B), dim3(256), lds, s,
                         (const bf16_t*)op->p[0], (float*)op->p[1], HW, C, chunks, Ctot, coff, slot0, slots,
                         (const float*)op->p[2], (const float*)op->p[3], ssout, (unsigned*)op->p[5], groups, op->f[0], x1, C1);
      break;
    }
    case MG_OP_GN_FINALIZE: {
      const int B = op->i[0], C = op->i[1], groups = op->i[2], slots = op->i[3], HW = op->i[4];
      MG_REQUIRE(groups > 0 && C % groups == 0, "gn_finalize: C %d not divisible by groups %d", C, groups);
      MG_LAUNCH(gn_finalize_kernel, dim3(B), dim3(256), 0, s,
                         (const float*)op->p[0], (const float*)op->p[1], (const float*)op->p[2],
                         (float*)op->p[3], B, C, groups, slots, HW, op->f[0]);
      break;
    }
    case MG_OP_GN_APPLY: {
      const int B = op->i[0], HW = op->i[1], C = op->i[2];
      MG_REQUIRE(C % 8 == 0 && B > 0 && HW > 0, "gn_apply: C %d must be a multiple of 8", C);
      // ~2048 workgroups in total, >= 16 rows per thread row-lane
      const int cvv = C / 8, tynn = 256 / (cvv < 256 ? cvv : 256);
      int chunks = (2048 + B - 1) / B;
      chunks = max(1, min(chunks, HW / max(1, 8 * tynn)));
      const bf16_t* x1 = (const bf16_t*)op->p[3];
      const int C0 = x1 ? op->i[4] : C;
      MG_REQUIRE(C0 > 0 && C0 <= C && C0 % 8 == 0, "gn_apply: first source has %d of %d channels", C0, C);
      MG_LAUNCH(gn_apply_kernel, dim3(chunks, B), dim3(256), 0, s, (const bf16_t*)op->p[0],
                (const float*)op->p[1], (bf16_t*)op->p[2], HW, C, op->i[3], chunks, x1, C0);
      break;
    }
    case MG_OP_GN_SLAB: {
      GnSlabArgs a;
      a.x0 = (const bf16_t*)op->p[0];
      a.x1 = (const bf16_t*)op->p[1];
      a.y = (bf16_t*)op->p[2];
      a.gamma = (const float*)op->p[3];
      a.beta = (const float*)op->p[4];
      a.ss = (float*)op->p[5];
      const int B = op->i[0];
      a.HW = op->i[1]; a.C = op->i[2]; a.C0 = a.x1 ? op->i[3] : a.C;
      const int groups = op->i[4];
      a.silu = op->i[5];
      a.eps = op->f[0];
      MG_REQUIRE(a.x0 && a.gamma && a.beta && a.ss && B > 0 && a.HW > 0 && groups > 0 && a.C % groups == 0, "gn_slab: bad arguments");
      a.cpg = a.C / groups;
      a.cw = a.cpg % 4 == 0 ? a.cpg : (a.cpg % 2 == 0 ? 2 * a.cpg : 4 * a.cpg);   // lcm(cpg, 4)
      MG_REQUIRE(a.cw <= 128 && a.C % a.cw == 0 && a.C0 % 4 == 0 && a.C0 > 0 && a.C0 <= a.C,
                 "gn_slab: %d channels per group need a %d-channel window (<= 128, dividing C = %d; C0 = %d a multiple of 4)", a.cpg, a.cw, a.C, a.C0);
      MG_REQUIRE((uintptr_t)a.x0 % 8 == 0 && (uintptr_t)a.x1 % 8 == 0 && (uintptr_t)a.y % 8 == 0, "gn_slab: 8-byte alignment");
      a.vw = a.cw / 4;
      a.fd_vw = mg_make_fastdiv(a.vw);
      a.fd_ncol = mg_make_fastdiv(a.cw * 2);
      a.fd_cpg = mg_make_fastdiv(a.cpg);
      const long long slab = (long long)a.HW * a.cw * 2;
      const int NT = slab >= 48 * 1024 ? 1024 : 256;
      a.nty = NT / a.vw;
      const int need = (a.HW + a.nty - 1) / a.nty;   // rows per thread
      int rc;
      if (!a.y) rc = NT == 1024 ? launch_gn_slab<1024, 0>(a, B, s) : launch_gn_slab<256, 0>(a, B, s);
      else if (NT == 1024) {
        MG_REQUIRE(need <= 48, "gn_slab: %d rows per thread exceed the register-resident form (H x W %d, window %d)", need, a.HW, a.cw);
        rc = need <= 12 ? launch_gn_slab<1024, 12>(a, B, s) : need <= 24 ? launch_gn_slab<1024, 24>(a, B, s) : launch_gn_slab<1024, 48>(a, B, s);
      } else {
        MG_REQUIRE(need <= 48, "gn_slab: %d rows per thread exceed the register-resident form (H x W %d, window %d)", need, a.HW, a.cw);
        rc = need <= 8 ? launch_gn_slab<256, 8>(a, B, s) : need <= 24 ? launch_gn_slab<256, 24>(a, B, s) : launch_gn_slab<256, 48>(a, B, s);
      }
      if (rc) return rc;
      break;
    }
    default: MG_REQUIRE(false, "norm: bad op kind %d", op->kind);
  }
  if (!g_dry_run) MG_CHECK_HIP(hipGetLastError());
  return 0;
}
