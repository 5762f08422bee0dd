// On-device test-time ensembling (reference: marigold/util/ensemble.py).
//
// Depth (ensemble_depth :39-196).  The reference evaluates, per BFGS cost call, E(E-1)/2
// pairwise RMSE reductions with one .item() sync each (:142-144) plus a median (:147-150).
// Here:
//   * DEPTH_STATS makes ONE pass set over the members and produces per-member min/max/mean
//     and the centred second-moment matrix C (E x E, fp64).  With a_i = s_i d_i + t_i,
//       mean((a_i - a_j)^2) = s_i^2 C_ii + s_j^2 C_jj - 2 s_i s_j C_ij + (s_i m_i + t_i - s_j m_j - t_j)^2
//     so the pairwise term of the cost - and its exact gradient - is O(E^2) host arithmetic.
//   * DEPTH_MEDIAN is the only per-evaluation pixel work: align -> lower-middle median over E
//     (torch.median semantics, :129) -> global min/max (regulariser, :146-150); the same kernel
//     writes the final median / MAD maps (:178-182).
//   * DEPTH_NORM applies the final min/max normalisation (:184-194) from device-resident
//     scalars (no host sync).
// Normals (ensemble_normals :199-249): one fused per-pixel kernel (mean -> normalise -> cosine
// -> arccos mean / pi -> argmax -> gather).
// All HBM-bound streaming kernels: E*HW*4 B read per pass, coalesced over pixels.
#include <stdlib.h>

#include <vector>

#include "common.h"

namespace {

constexpr int EMAX = 32;        // members a thread keeps in registers (selection kernels) / columns per second-moment block
constexpr int EMAX_LDS = 128;   // members of the LDS-resident selection (33 ... 128: any size the reference's users ask for in
                                // practice - it warns above 15, script/depth/run.py:143-144); beyond: depth_median_big_kernel
                                // (bitwise selection straight from memory - the reference accepts ANY ensemble size)
constexpr int ENS_BLOCKS = 512;

// grid (nblk, E, column chunks of 32): block (x, i, jc) accumulates columns [32 jc, 32 jc + 32) of row i of the raw
// second-moment matrix; partial layout [block x][row i][3 + E] (min, max, sum, row of the matrix)
__global__ __launch_bounds__(256) void depth_stats_kernel(const float* __restrict__ d0, double* __restrict__ part,
                                                          int Etot, long long HW) {
  __shared__ double red[4][EMAX + 3];
  const int i = blockIdx.y;
  const int j0 = blockIdx.z * EMAX;
  const int E = min(EMAX, Etot - j0);            // columns of this block
  const float* __restrict__ d = d0 + (long long)j0 * HW;
  const float* __restrict__ di_row = d0 + (long long)i * HW;
  float acc[EMAX];
#pragma unroll
  for (int j = 0; j < EMAX; ++j) acc[j] = 0.f;
  float mn = 3.0e38f, mx = -3.0e38f;
  double sum = 0.0;
  // fp32 partials over short strides, promoted to fp64 every 64 pixels per thread
  double accd[EMAX];
#pragma unroll
  for (int j = 0; j < EMAX; ++j) accd[j] = 0.0;
  int cnt = 0;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long long)gridDim.x * 256) {
    const float di = di_row[p];
    mn = fminf(mn, di);
    mx = fmaxf(mx, di);
    sum += (double)di;
#pragma unroll
    for (int j = 0; j < EMAX; ++j)
      if (j < E) acc[j] += di * d[(long long)j * HW + p];
    if (++cnt == 64) {
#pragma unroll
      for (int j = 0; j < EMAX; ++j) { accd[j] += (double)acc[j]; acc[j] = 0.f; }
      cnt = 0;
    }
  }
#pragma unroll
  for (int j = 0; j < EMAX; ++j) accd[j] += (double)acc[j];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o));
    mx = fmaxf(mx, __shfl_xor(mx, o));
    sum += __shfl_xor(sum, o);
#pragma unroll
    for (int j = 0; j < EMAX; ++j)
      if (j < E) accd[j] += __shfl_xor(accd[j], o);
  }
  if (lane == 0) {
    red[wave][0] = (double)mn; red[wave][1] = (double)mx; red[wave][2] = sum;
#pragma unroll
    for (int j = 0; j < EMAX; ++j)
      if (j < E) red[wave][3 + j] = accd[j];
  }
  __syncthreads();
  double* o = part + ((long long)blockIdx.x * Etot + i) * (Etot + 3);
  for (int k = threadIdx.x; k < E + 3; k += 256) {
    double v;
    if (k == 0) v = fmin(fmin(red[0][0], red[1][0]), fmin(red[2][0], red[3][0]));
    else if (k == 1) v = fmax(fmax(red[0][1], red[1][1]), fmax(red[2][1], red[3][1]));
    else v = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    if (k < 3) { if (j0 == 0) o[k] = v; }       // min / max / sum of row i: once
    else o[3 + j0 + (k - 3)] = v;
  }
}

// one block; out = [min[E], max[E], mean[E], C[E][E]] (C centred)
__global__ __launch_bounds__(256) void depth_stats_final_kernel(const double* __restrict__ part, double* __restrict__ out,
                                                                int E, int nblk, long long HW) {
  // the member means: out[2E ..), written by this (single) block below and read back after the barrier
  double* const mean_s = out + 2 * E;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < E * 3; idx += 256) {
    const int i = idx / 3, k = idx % 3;
    double v = (k == 0) ? 3.0e38 : (k == 1 ? -3.0e38 : 0.0);
    for (int b = 0; b < nblk; ++b) {
      const double x = part[((long long)b * E + i) * (E + 3) + k];
      if (k == 0) v = fmin(v, x);
      else if (k == 1) v = fmax(v, x);
      else v += x;
    }
    if (k == 0) out[i] = v;
    else if (k == 1) out[E + i] = v;
    else mean_s[i] = v / (double)HW;
  }
  __threadfence_block();
  __syncthreads();
  for (int idx = tid; idx < E * E; idx += 256) {
    const int i = idx / E, j = idx % E;
    double v = 0.0;
    for (int b = 0; b < nblk; ++b) v += part[((long long)b * E + i) * (E + 3) + 3 + j];   // same order as before: b ascending
    out[3 * E + idx] = v / (double)HW - mean_s[i] * mean_s[j];
  }
}

template <int E_>
__device__ __forceinline__ float select_rank(const float (&a)[E_], int E, int k) {
  float res = a[0];
#pragma unroll
  for (int e = 0; e < E_; ++e) {
    if (e < E) {
      int rank = 0;
#pragma unroll
      for (int j = 0; j < E_; ++j)
        if (j < E) rank += (a[j] < a[e]) || (a[j] == a[e] && j < e);
      if (rank == k) res = a[e];
    }
  }
  return res;
}

// The benchmark's ensemble size: k-th smallest of <= 10 values through the 29-comparator sorting network (Knuth, TAOCP 3, n = 10;
// 58 min / max against ~300 compare / add of the rank count above).  Entries past E are +inf; the VALUE of rank k under
// (value, index) order is the k-th order statistic, whichever way ties are broken.
__device__ __forceinline__ float select_kth10(const float (&v)[10], int E, int k) {
  float a[10];
#pragma unroll
  for (int e = 0; e < 10; ++e) a[e] = e < E ? v[e] : __builtin_inff();
#define MG_CE(i, j) { const float lo = fminf(a[i], a[j]), hi = fmaxf(a[i], a[j]); a[i] = lo; a[j] = hi; }
  MG_CE(4, 9) MG_CE(3, 8) MG_CE(2, 7) MG_CE(1, 6) MG_CE(0, 5) MG_CE(1, 4) MG_CE(6, 9) MG_CE(0, 3) MG_CE(5, 8) MG_CE(0, 2)
  MG_CE(3, 6) MG_CE(7, 9) MG_CE(0, 1) MG_CE(2, 4) MG_CE(5, 7) MG_CE(8, 9) MG_CE(1, 2) MG_CE(4, 6) MG_CE(7, 8) MG_CE(3, 5)
  MG_CE(2, 5) MG_CE(6, 8) MG_CE(1, 3) MG_CE(4, 7) MG_CE(2, 3) MG_CE(6, 7) MG_CE(3, 4) MG_CE(5, 6) MG_CE(4, 5)
#undef MG_CE
  float r = a[0];
#pragma unroll
  for (int e = 1; e < 10; ++e)
    if (e == k) r = a[e];
  return r;
}
template <int E_>
__device__ __forceinline__ float select_kth(const float (&a)[E_], int E, int k) {
  if constexpr (E_ == 10) return select_kth10(a, E, k);
  else return select_rank<E_>(a, E, k);
}

// E_ = compile-time upper bound of E (registers); st = [s[E], t[E]] fp32; reduction 0 median 1 mean
template <int E_>
__global__ __launch_bounds__(256) void depth_median_kernel(const float* __restrict__ d, const float* __restrict__ st,
                                                           float* __restrict__ med, float* __restrict__ mad,
                                                           float* __restrict__ blockmm, long long* __restrict__ blockpx,
                                                           int E, long long HW,
                                                           int reduction, int has_shift, int aligned) {
  __shared__ float red[8];
  __shared__ long long redp[8];
  long long pmn = 0, pmx = 0;
  float sc[E_], sh[E_];
#pragma unroll
  for (int e = 0; e < E_; ++e) {
    sc[e] = (aligned && e < E) ? st[e] : 1.f;
    sh[e] = (aligned && has_shift && e < E) ? st[E + e] : 0.f;
  }
  float mn = 3.0e38f, mx = -3.0e38f;
  const int k = (E - 1) >> 1;  // torch.median: lower middle
  auto pixel = [&](const float (&raw)[E_], long long p, float& pred, float& unc) {
    float a[E_];
#pragma unroll
    for (int e = 0; e < E_; ++e)   // reference: depth * s + t as two separately rounded fp32 ops (ensemble.py:112)
      a[e] = e < E ? (aligned ? __fadd_rn(__fmul_rn(raw[e], sc[e]), sh[e]) : raw[e]) : 0.f;
    unc = 0.f;
    if (reduction == 0) {
      pred = select_kth<E_>(a, E, k);
      if (mad) {
        float dv[E_];
#pragma unroll
        for (int e = 0; e < E_; ++e) dv[e] = fabsf(__fsub_rn(a[e], pred));
        unc = select_kth<E_>(dv, E, k);
      }
    } else {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < E_; ++e)
        if (e < E) s += a[e];
      pred = s / (float)E;
      if (mad) {
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < E_; ++e)
          if (e < E) { const float dd = a[e] - pred; q += dd * dd; }
        unc = sqrtf(q / (float)(E > 1 ? E - 1 : 1));  // torch.std: unbiased
      }
    }
    if (pred < mn) { mn = pred; pmn = p; }
    if (pred > mx) { mx = pred; pmx = p; }
  };
  // four consecutive pixels per thread and pass (16-byte loads, E of them in flight): the optimiser calls this pass ~100 times
  // per map and as one pixel per thread it was 4-5 dependent HBM round trips long (19 us for 23.6 MB; round 3)
  const long long HW4 = (HW & 3) == 0 && (((uintptr_t)d | (uintptr_t)med | (uintptr_t)mad) & 15) == 0 ? HW : 0;
  for (long long p = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; p < HW4; p += (long long)gridDim.x * 1024) {
    float4 v[E_];
#pragma unroll
    for (int e = 0; e < E_; ++e) v[e] = e < E ? *(const float4*)(d + (long long)e * HW + p) : make_float4(0.f, 0.f, 0.f, 0.f);
    float pr[4], un[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float raw[E_];
#pragma unroll
      for (int e = 0; e < E_; ++e) raw[e] = i == 0 ? v[e].x : (i == 1 ? v[e].y : (i == 2 ? v[e].z : v[e].w));
      pixel(raw, p + i, pr[i], un[i]);
    }
    if (med) *(float4*)(med + p) = make_float4(pr[0], pr[1], pr[2], pr[3]);
    if (mad) *(float4*)(mad + p) = make_float4(un[0], un[1], un[2], un[3]);
  }
  for (long long p = HW4 + (long long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long long)gridDim.x * 256) {
    float raw[E_];
#pragma unroll
    for (int e = 0; e < E_; ++e) raw[e] = e < E ? d[(long long)e * HW + p] : 0.f;
    float pred, unc;
    pixel(raw, p, pred, unc);
    if (med) med[p] = pred;
    if (mad) mad[p] = unc;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float omn = __shfl_xor(mn, o), omx = __shfl_xor(mx, o);
    const long long opmn = __shfl_xor(pmn, o), opmx = __shfl_xor(pmx, o);
    if (omn < mn || (omn == mn && opmn < pmn)) { mn = omn; pmn = opmn; }
    if (omx > mx || (omx == mx && opmx < pmx)) { mx = omx; pmx = opmx; }
  }
  if (lane == 0) { red[wave] = mn; red[4 + wave] = mx; redp[wave] = pmn; redp[4 + wave] = pmx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      if (red[w] < red[0] || (red[w] == red[0] && redp[w] < redp[0])) { red[0] = red[w]; redp[0] = redp[w]; }
      if (red[4 + w] > red[4] || (red[4 + w] == red[4] && redp[4 + w] < redp[4])) { red[4] = red[4 + w]; redp[4] = redp[4 + w]; }
    }
    blockmm[2 * blockIdx.x] = red[0];
    blockmm[2 * blockIdx.x + 1] = red[4];
    blockpx[2 * blockIdx.x] = redp[0];
    blockpx[2 * blockIdx.x + 1] = redp[4];
  }
}

// 33 ... 128 members: the aligned values of a pixel live in LDS ([member][thread]: conflict-free), the lower-middle order
// statistic is found by counting ranks (O(E^2) per pixel - large ensembles are rare and the pass stays HBM-light).
// Same semantics, same outputs as depth_median_kernel.
__global__ __launch_bounds__(256) void depth_median_lds_kernel(const float* __restrict__ d, const float* __restrict__ st,
                                                               float* __restrict__ med, float* __restrict__ mad,
                                                               float* __restrict__ blockmm, long long* __restrict__ blockpx,
                                                               int E, long long HW, int reduction, int has_shift, int aligned) {
  extern __shared__ float lds_a[];   // [E][256] aligned values (128 KB at E = 128), then [2 E] scale / shift
  __shared__ float red[8];
  __shared__ long long redp[8];
  const int tid = threadIdx.x;
  float* a = lds_a + tid;
  float* sc = lds_a + (long long)E * 256;
  for (int e = tid; e < E; e += 256) {
    sc[e] = aligned ? st[e] : 1.f;
    sc[E + e] = (aligned && has_shift) ? st[E + e] : 0.f;
  }
  __syncthreads();
  long long pmn = 0, pmx = 0;
  float mn = 3.0e38f, mx = -3.0e38f;
  const int k = (E - 1) >> 1;
  // element of rank k (ties by member index), as select_rank; DEV: of the absolute deviations |a - centre| (formed on the
  // fly: a second [E][256] array would not fit the 160 KB of LDS)
  auto select = [&](bool dev_, float centre) {
    auto val = [&](int e) { const float x = a[e * 256]; return dev_ ? fabsf(__fsub_rn(x, centre)) : x; };
    float res = val(0);
    for (int e = 0; e < E; ++e) {
      const float ve = val(e);
      int rank = 0;
      for (int j = 0; j < E; ++j) {
        const float vj = val(j);
        rank += (vj < ve) || (vj == ve && j < e);
      }
      if (rank == k) res = ve;
    }
    return res;
  };
  for (long long p = (long long)blockIdx.x * 256 + tid; p < HW; p += (long long)gridDim.x * 256) {
    for (int e = 0; e < E; ++e) {
      const float v = d[(long long)e * HW + p];
      a[e * 256] = aligned ? __fadd_rn(__fmul_rn(v, sc[e]), sc[E + e]) : v;
    }
    float pred, unc = 0.f;
    if (reduction == 0) {
      pred = select(false, 0.f);
      if (mad) unc = select(true, pred);
    } else {
      float s = 0.f;
      for (int e = 0; e < E; ++e) s += a[e * 256];
      pred = s / (float)E;
      if (mad) {
        float q = 0.f;
        for (int e = 0; e < E; ++e) { const float dd = a[e * 256] - pred; q += dd * dd; }
        unc = sqrtf(q / (float)(E > 1 ? E - 1 : 1));
      }
    }
    if (med) med[p] = pred;
    if (mad) mad[p] = unc;
    if (pred < mn) { mn = pred; pmn = p; }
    if (pred > mx) { mx = pred; pmx = p; }
  }
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float omn = __shfl_xor(mn, o), omx = __shfl_xor(mx, o);
    const long long opmn = __shfl_xor(pmn, o), opmx = __shfl_xor(pmx, o);
    if (omn < mn || (omn == mn && opmn < pmn)) { mn = omn; pmn = opmn; }
    if (omx > mx || (omx == mx && opmx < pmx)) { mx = omx; pmx = opmx; }
  }
  if (lane == 0) { red[wave] = mn; red[4 + wave] = mx; redp[wave] = pmn; redp[4 + wave] = pmx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) {
      if (red[w] < red[0] || (red[w] == red[0] && redp[w] < redp[0])) { red[0] = red[w]; redp[0] = redp[w]; }
      if (red[4 + w] > red[4] || (red[4 + w] == red[4] && redp[4 + w] < redp[4])) { red[4] = red[4 + w]; redp[4] = redp[4 + w]; }
    }
    blockmm[2 * blockIdx.x] = red[0];
    blockmm[2 * blockIdx.x + 1] = red[4];
    blockpx[2 * blockIdx.x] = redp[0];
    blockpx[2 * blockIdx.x + 1] = redp[4];
  }
}

// More than 128 members (the reference takes any ensemble size, marigold/util/ensemble.py:39-49): the order statistic of a pixel
// is found by a bitwise selection over the members' values read straight from memory - 32 counting passes over E values per pixel
// (monotone float -> uint32 key, most significant bit first), O(32 E) instead of the rank count's O(E^2), no per-pixel storage.
// Same semantics and outputs as depth_median_kernel: the VALUE of rank (E - 1) / 2 does not depend on how ties are broken.
__global__ __launch_bounds__(256) void depth_median_big_kernel(const float* __restrict__ d, const float* __restrict__ st,
                                                               float* __restrict__ med, float* __restrict__ mad,
                                                               float* __restrict__ blockmm, long long* __restrict__ blockpx,
                                                               int E, long long HW, int reduction, int has_shift, int aligned) {
  __shared__ float red[8];
  __shared__ long long redp[8];
  const int tid = threadIdx.x;
  long long pmn = 0, pmx = 0;
  float mn = 3.0e38f, mx = -3.0e38f;
  const int k = (E - 1) >> 1;
  auto key_of = [](float x) { const unsigned b = __float_as_uint(x); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); };
  auto val_of = [](unsigned key) { return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key); };
  for (long long p = (long long)blockIdx.x * 256 + tid; p < HW; p += (long long)gridDim.x * 256) {
    auto aval = [&](int e) {   // reference: depth * s + t as two separately rounded fp32 ops (ensemble.py:112)
      const float v = d[(long long)e * HW + p];
      return aligned ? __fadd_rn(__fmul_rn(v, st[e]), has_shift ? st[E + e] : 0.f) : v;
    };
    auto select = [&](bool dev_, float centre) {
      unsigned prefix = 0;
      int kk = k;
      for (int bit = 31; bit >= 0; --bit) {
        const unsigned hi_mask = bit == 31 ? 0u : ~((2u << bit) - 1u);   // the bits already decided
        int cnt0 = 0;
        for (int e = 0; e < E; ++e) {
          const float x = aval(e);
          const unsigned key = key_of(dev_ ? fabsf(__fsub_rn(x, centre)) : x);
          cnt0 += ((key & hi_mask) == prefix && !((key >> bit) & 1u)) ? 1 : 0;
        }
        if (kk >= cnt0) { kk -= cnt0; prefix |= 1u << bit; }
      }
      return val_of(prefix);
    };
    float pred, unc = 0.f;
    if (reduction == 0) {
      pred = select(false, 0.f);
      if (mad) unc = select(true, pred);
    } else {
      float s = 0.f;
      for (int e = 0; e < E; ++e) s += aval(e);
      pred = s / (float)E;
      if (mad) {
        float q = 0.f;
        for (int e = 0; e < E; ++e) { const float dd = aval(e) - pred; q += dd * dd; }
        unc = sqrtf(q / (float)(E > 1 ? E - 1 : 1));
      }
    }
    if (med) med[p] = pred;
    if (mad) mad[p] = unc;
    if (pred < mn) { mn = pred; pmn = p; }
    if (pred > mx) { mx = pred; pmx = p; }
  }
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float omn = __shfl_xor(mn, o), omx = __shfl_xor(mx, o);
    const long long opmn = __shfl_xor(pmn, o), opmx = __shfl_xor(pmx, o);
    if (omn < mn || (omn == mn && opmn < pmn)) { mn = omn; pmn = opmn; }
    if (omx > mx || (omx == mx && opmx < pmx)) { mx = omx; pmx = opmx; }
  }
  if (lane == 0) { red[wave] = mn; red[4 + wave] = mx; redp[wave] = pmn; redp[4 + wave] = pmx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) {
      if (red[w] < red[0] || (red[w] == red[0] && redp[w] < redp[0])) { red[0] = red[w]; redp[0] = redp[w]; }
      if (red[4 + w] > red[4] || (red[4 + w] == red[4] && redp[4 + w] < redp[4])) { red[4] = red[4 + w]; redp[4] = redp[4 + w]; }
    }
    blockmm[2 * blockIdx.x] = red[0];
    blockmm[2 * blockIdx.x + 1] = red[4];
    blockpx[2 * blockIdx.x] = redp[0];
    blockpx[2 * blockIdx.x + 1] = redp[4];
  }
}

// out = [min, max, d[0..E)[argmin px], d[0..E)[argmax px]]  (raw member values at the extremal
// pixels of the prediction: the host derives the exact sub-gradient of the regulariser from them)
__global__ __launch_bounds__(64) void minmax_final_kernel(const float* __restrict__ blockmm,
                                                          const long long* __restrict__ blockpx,
                                                          const float* __restrict__ d, float* __restrict__ out,
                                                          int nblk, int E, long long HW) {
  __shared__ long long px[2];
  // one wave: strided scan of the per-block results, then a lane reduction (ties -> lowest pixel)
  const int lane = threadIdx.x;
  float mn = 3.0e38f, mx = -3.0e38f;
  long long pmn = 0x7fffffffffffffffll, pmx = 0x7fffffffffffffffll;
  for (int i = lane; i < nblk; i += 64) {
    const float a = blockmm[2 * i], b = blockmm[2 * i + 1];
    const long long pa = blockpx[2 * i], pb = blockpx[2 * i + 1];
    if (a < mn || (a == mn && pa < pmn)) { mn = a; pmn = pa; }
    if (b > mx || (b == mx && pb < pmx)) { mx = b; pmx = pb; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float omn = __shfl_xor(mn, o), omx = __shfl_xor(mx, o);
    const long long opmn = __shfl_xor(pmn, o), opmx = __shfl_xor(pmx, o);
    if (omn < mn || (omn == mn && opmn < pmn)) { mn = omn; pmn = opmn; }
    if (omx > mx || (omx == mx && opmx < pmx)) { mx = omx; pmx = opmx; }
  }
  if (lane == 0) {
    out[0] = mn; out[1] = mx;
    px[0] = pmn; px[1] = pmx;
  }
  __syncthreads();
  for (int e = lane; e < E; e += 64) {
    out[2 + e] = d[(long long)e * HW + px[0]];
    out[2 + E + e] = d[(long long)e * HW + px[1]];
  }
}

__global__ __launch_bounds__(256) void depth_norm_kernel(float* __restrict__ med, float* __restrict__ unc,
                                                         const float* __restrict__ mm, long long HW, int shift_inv) {
  const float hi = mm[1];
  const float lo = shift_inv ? mm[0] : 0.f;
  const float rng = fmaxf(hi - lo, 1e-6f);
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long long)gridDim.x * 256) {
    med[p] = (med[p] - lo) / rng;
    if (unc) unc[p] = unc[p] / rng;
  }
}

__global__ __launch_bounds__(256) void normals_kernel(const float* __restrict__ n, float* __restrict__ out,
                                                      float* __restrict__ unc, int E, long long HW, int reduction) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long long)gridDim.x * 256) {
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int e = 0; e < E; ++e) {
      const float* q = n + (long long)e * 3 * HW + p;
      sx += q[0]; sy += q[HW]; sz += q[2 * HW];
    }
    float mx_ = sx / (float)E, my = sy / (float)E, mz = sz / (float)E;
    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(mx_, mx_), __fmul_rn(my, my)), __fmul_rn(mz, mz)));
    const float den = fmaxf(nrm, 1e-6f);
    mx_ = mx_ / den; my = my / den; mz = mz / den;
    float best = -2.f, ua = 0.f;
    int bi = 0;
    for (int e = 0; e < E; ++e) {
      const float* q = n + (long long)e * 3 * HW + p;
      float c = __fadd_rn(__fadd_rn(__fmul_rn(mx_, q[0]), __fmul_rn(my, q[HW])), __fmul_rn(mz, q[2 * HW]));
      c = fminf(fmaxf(c, -1.f), 1.f);
      if (c > best) { best = c; bi = e; }
      ua += acosf(c);
    }
    if (unc) unc[p] = ua / (float)E / 3.14159265358979323846f;
    if (reduction == 1) {
      out[p] = mx_; out[HW + p] = my; out[2 * HW + p] = mz;
    } else {
      const float* q = n + (long long)bi * 3 * HW + p;
      out[p] = q[0]; out[HW + p] = q[HW]; out[2 * HW + p] = q[2 * HW];
    }
  }
}

template <int E_>
void launch_median(const mg_op* op, int nblk, hipStream_t s) {
  MG_LAUNCH(depth_median_kernel<E_>, dim3(nblk), dim3(256), 0, s, (const float*)op->p[0],
                     (const float*)op->p[1], (float*)op->p[2], (float*)op->p[3], (float*)op->p[5],
                     (long long*)((char*)op->p[5] + 8 * ENS_BLOCKS), op->i[0], op->l[0], op->i[1], op->i[2],
                     op->p[1] != nullptr);
}

}  // namespace

int mg_launch_ensemble(const mg_op* op, hipStream_t s) {
  switch (op->kind) {
    case MG_OP_ENS_DEPTH_STATS: {
      const int E = op->i[0];
      const long long HW = op->l[0];
      MG_REQUIRE(E >= 1 && E <= 65535, "ens_depth_stats: E %d out of range [1,65535]", E);
      // (the partial table is nblk x E x (E + 3) doubles: fewer pixel blocks for the ensembles beyond the kernels' usual range)
      const int nblk = (int)min((HW + 255) / 256, (long long)(E > 256 ? 32 : 128));
      MG_LAUNCH(depth_stats_kernel, dim3(nblk, E, (E + EMAX - 1) / EMAX), dim3(256), 0, s, (const float*)op->p[0],
                         (double*)op->p[1], E, HW);
      MG_LAUNCH(depth_stats_final_kernel, dim3(1), dim3(256), 0, s, (const double*)op->p[1],
                         (double*)op->p[2], E, nblk, HW);
      break;
    }
    case MG_OP_ENS_DEPTH_MEDIAN: {
      const int E = op->i[0];
      const long long HW = op->l[0];
      MG_REQUIRE(E >= 1, "ens_depth_median: E %d must be >= 1", E);
      MG_REQUIRE(op->p[4] && op->p[5], "ens_depth_median: minmax / scratch missing");
      const int nblk = (int)min((HW + 255) / 256, (long long)ENS_BLOCKS);
      if (E > EMAX_LDS) {
        MG_LAUNCH(depth_median_big_kernel, dim3(nblk), dim3(256), 0, s, (const float*)op->p[0], (const float*)op->p[1],
                  (float*)op->p[2], (float*)op->p[3], (float*)op->p[5], (long long*)((char*)op->p[5] + 8 * ENS_BLOCKS), E,
                  op->l[0], op->i[1], op->i[2], op->p[1] != nullptr);
      } else if (E > EMAX) {
        const size_t lds = ((size_t)E * 256 + 2 * E) * sizeof(float);
        static bool attr = false;
        if (!attr && !g_dry_run) {
          MG_CHECK_HIP(hipFuncSetAttribute((const void*)depth_median_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(((size_t)EMAX_LDS * 256 + 2 * EMAX_LDS) * sizeof(float))));
          attr = true;
        }
        MG_LAUNCH(depth_median_lds_kernel, dim3(nblk), dim3(256), lds, s, (const float*)op->p[0], (const float*)op->p[1],
                  (float*)op->p[2], (float*)op->p[3], (float*)op->p[5], (long long*)((char*)op->p[5] + 8 * ENS_BLOCKS), E,
                  op->l[0], op->i[1], op->i[2], op->p[1] != nullptr);
      } else if (E <= 4) launch_median<4>(op, nblk, s);
      else if (E <= 8) launch_median<8>(op, nblk, s);
      else if (E <= 10) launch_median<10>(op, nblk, s);
      else if (E <= 16) launch_median<16>(op, nblk, s);
      else launch_median<EMAX>(op, nblk, s);
      // (round 3: finishing the reduction in the median kernel's last-arriving block instead - fence + ticket per block - measured
      // SLOWER than this 5 us launch: 40-41 vs 34-36 us per pass, profiles/r3_ab_native_bfgs_alignment.log)
      MG_LAUNCH(minmax_final_kernel, dim3(1), dim3(64), 0, s, (const float*)op->p[5],
                         (const long long*)((char*)op->p[5] + 8 * ENS_BLOCKS), (const float*)op->p[0],
                         (float*)op->p[4], nblk, E, HW);
      break;
    }
    case MG_OP_ENS_DEPTH_NORM: {
      const long long HW = op->l[0];
      const int nblk = (int)min((HW + 255) / 256, (long long)2048);
      MG_LAUNCH(depth_norm_kernel, dim3(nblk), dim3(256), 0, s, (float*)op->p[0], (float*)op->p[1],
                         (const float*)op->p[2], HW, op->i[0]);
      break;
    }
    case MG_OP_ENS_NORMALS: {
      const int E = op->i[0];
      const long long HW = op->l[0];
      MG_REQUIRE(E >= 1, "ens_normals: E must be >= 1");
      const int nblk = (int)min((HW + 255) / 256, (long long)2048);
      MG_LAUNCH(normals_kernel, dim3(nblk), dim3(256), 0, s, (const float*)op->p[0],
                         (float*)op->p[1], (float*)op->p[2], E, HW, op->i[1]);
      break;
    }
    default: MG_REQUIRE(false, "ensemble: bad op kind %d", op->kind);
  }
  if (!g_dry_run) MG_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- host arithmetic of the alignment objective (ensemble.py::DepthAligner.cost_and_grad) ----------------------------
// The reference's optimiser (marigold/util/ensemble.py:154-173: scipy BFGS) calls its cost ~100 times per map; the
// pairwise-RMSE part is a closed form of E x E numbers (ensemble.py, module docstring).  As numpy calls that is 30+
// array operations of ~1 us each on 10 x 10 operands; here it is one C call.  Same operations in the same order as the
// numpy form, INCLUDING numpy's pairwise summation (8 accumulators, blocks of 128), so the values - and with them the
// BFGS iterates - are bit-identical (tests/test_host.py pins that); this file is built with -ffp-contract=off.
namespace {
double np_pairwise_sum(const double* a, int n) {
  if (n < 8) {
    double res = 0.;
    for (int i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    double r[8];
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int k = 0; k < 8; ++k) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}
// np.add.reduce over a contiguous run of n doubles: numpy hands its inner loop at most `bufsize` (8192) elements at a time and
// adds each piece's pairwise sum to the running result (numpy 2.2; checked against ndarray.sum() for n on both sides of 8192) -
// the pair costs of more than 128 members are more than one piece
double np_sum(const double* a, int n) {
  double res = 0.;
  for (int i = 0; i < n; i += 8192) res += np_pairwise_sum(a + i, n - i < 8192 ? n - i : 8192);
  return res;
}
}  // namespace

extern "C" int mg_ens_align_cost_grad(int E, const double* s, const double* t, const double* mean, const double* C,
                                      double* cost, double* gs, double* gt) {
  MG_REQUIRE(E >= 1 && s && t && mean && C && cost && gs && gt, "ens_align_cost_grad: bad arguments");
  // scratch: 3 E + 3 E^2 doubles, kept per thread (the optimiser calls this ~100 times per map); any ensemble size
  static thread_local std::vector<double> scratch;
  if (scratch.size() < (size_t)3 * E + (size_t)3 * E * E) scratch.resize((size_t)3 * E + (size_t)3 * E * E);
  double* const u = scratch.data();
  double* const s2d = u + E;
  double* const sdc = s2d + E;
  double* const r = sdc + E;
  double* const w = r + (size_t)E * E;
  double* const tmp = w + (size_t)E * E;
  for (int i = 0; i < E; ++i) {
    u[i] = s[i] * mean[i] + t[i];
    s2d[i] = (s[i] * s[i]) * C[i * E + i];
    sdc[i] = s[i] * C[i * E + i];
  }
  for (int i = 0; i < E; ++i)
    for (int j = 0; j < E; ++j) {
      const double du = u[i] - u[j];
      double q = ((s2d[i] + s2d[j]) - ((2.0 * (s[i] * s[j])) * C[i * E + j])) + du * du;
      if (!(q > 0.0)) q = (q != q) ? q : 0.0;   // np.maximum(q, 0.0): NaN propagates
      const double rr = sqrt(q);
      r[i * E + j] = rr;
      w[i * E + j] = (rr > 0 && i != j) ? 0.5 / rr : 0.0;
    }
  int n = 0;
  for (int i = 0; i < E; ++i)
    for (int j = i + 1; j < E; ++j) tmp[n++] = r[i * E + j];
  *cost = np_sum(tmp, n);
  for (int i = 0; i < E; ++i) {
    for (int j = 0; j < E; ++j) {
      const double du = u[i] - u[j];
      tmp[j] = w[i * E + j] * (((2.0 * sdc[i]) - ((2.0 * s[j]) * C[i * E + j])) + ((2.0 * du) * mean[i]));
    }
    gs[i] = np_sum(tmp, E);
    for (int j = 0; j < E; ++j) tmp[j] = (w[i * E + j] * 2.0) * (u[i] - u[j]);
    gt[i] = np_sum(tmp, E);
  }
  return 0;
}
