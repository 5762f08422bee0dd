// Image resampling on the device: the resize steps either side of the hot path
// (reference: marigold/util/image_util.py:90-120 `resize_max_res`, marigold_depth_pipeline.py:306-312
// resize-back, marigold/util/ensemble.py:158-161 nearest-exact down-size).  The reference calls
// torchvision.transforms.functional.resize(..., antialias=True), i.e. torch's separable anti-aliased
// bilinear / bicubic interpolation (align_corners = False) - restated here tap for tap:
//   scale = in / out; support = (interp/2) * max(scale, 1); for output i: center = scale * (i + 0.5),
//   xmin = max(int(center - support + 0.5), 0), xsize = min(int(center + support + 0.5), in) - xmin,
//   w_j = filter((j + xmin - center + 0.5) / max(scale, 1)) normalised to sum 1 (fp32 throughout),
// horizontal pass first (into an fp32 temporary), then vertical; uint8 inputs are computed in float,
// rounded half-to-even (clamped to [0, 255] for bicubic) and cast back.  nearest-exact:
// src = floor((i + 0.5) * scale).
#include "common.h"

namespace {

__device__ __forceinline__ float aa_filter(float x, int bicubic) {
  x = fabsf(x);
  if (!bicubic) return x < 1.0f ? 1.0f - x : 0.0f;
  const float a = -0.5f;
  if (x < 1.0f) return ((a + 2.0f) * x - (a + 3.0f)) * x * x + 1.0f;
  if (x < 2.0f) return (((x - 5.0f) * x + 8.0f) * x - 4.0f) * a;
  return 0.0f;
}

__device__ __forceinline__ float ld(const uint8_t* p, long long i) { return (float)p[i]; }
__device__ __forceinline__ float ld(const float* p, long long i) { return p[i]; }
__device__ __forceinline__ void st(float* p, long long i, float v, int) { p[i] = v; }
__device__ __forceinline__ void st(uint8_t* p, long long i, float v, int bicubic) {
  if (bicubic) v = fminf(fmaxf(v, 0.0f), 255.0f);
  p[i] = (uint8_t)rintf(v);  // round half to even, like torch.round()
}

// One output element per thread.  The resampled axis has `in_len` -> `out_len` elements with element
// stride `s_axis`; the other in-plane axis has `other` elements with stride `s_other`; planes are
// contiguous (`plane_in` / `plane_out` elements).  Output is [planes][out rows][out cols] row-major.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void resize_aa_pass_kernel(const TI* __restrict__ src, TO* __restrict__ dst,
                                                             long long total, int in_len, int out_len, int other,
                                                             int horizontal, int bicubic, float scale) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  // output layout: horizontal pass -> [plane][other = rows][out_len = cols]; vertical -> [plane][out_len][other]
  int i, o;
  long long plane;
  if (horizontal) { i = (int)(idx % out_len); o = (int)((idx / out_len) % other); plane = idx / ((long long)out_len * other); }
  else { o = (int)(idx % other); i = (int)((idx / other) % out_len); plane = idx / ((long long)out_len * other); }
  const float support = (bicubic ? 2.0f : 1.0f) * (scale >= 1.0f ? scale : 1.0f);
  const float invscale = scale >= 1.0f ? (float)(1.0 / (double)scale) : 1.0f;
  const float center = scale * ((float)i + 0.5f);
  // torch evaluates `center - support` in fp32 and adds the 0.5 (and scales the filter argument) in fp64
  int xmin = (int)((double)(center - support) + 0.5);
  xmin = xmin > 0 ? xmin : 0;
  int xend = (int)((double)(center + support) + 0.5);
  xend = xend < in_len ? xend : in_len;
  const int xsize = xend - xmin;
  float total_w = 0.0f;
  auto weight = [&](int j) {
    return aa_filter((float)(((double)((float)(j + xmin) - center) + 0.5) * (double)invscale), bicubic);
  };
  for (int j = 0; j < xsize; ++j) total_w += weight(j);
  const long long base = plane * (long long)in_len * other;
  float t = 0.0f;
  for (int j = 0; j < xsize; ++j) {
    float w = weight(j);
    w = total_w != 0.0f ? w / total_w : 0.0f;
    const long long sidx = horizontal ? base + (long long)o * in_len + (j + xmin)
                                      : base + (long long)(j + xmin) * other + o;
    const float v = ld(src, sidx) * w;
    t = j == 0 ? v : t + v;
  }
  st(dst, idx, t, bicubic);
}

template <typename T>
__global__ __launch_bounds__(256) void resize_nearest_exact_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                                   long long total, int Hin, int Win, int Hout,
                                                                   int Wout, float sy, float sx) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int x = (int)(idx % Wout), y = (int)((idx / Wout) % Hout);
  const long long plane = idx / ((long long)Wout * Hout);
  int iy = (int)floorf(((float)y + 0.5f) * sy), ix = (int)floorf(((float)x + 0.5f) * sx);
  iy = iy < Hin - 1 ? iy : Hin - 1;
  ix = ix < Win - 1 ? ix : Win - 1;
  dst[idx] = src[(plane * Hin + iy) * Win + ix];
}

template <typename TI, typename TO>
void launch_pass(const void* src, void* dst, long long total, int in_len, int out_len, int other, int horizontal,
                 int bicubic, hipStream_t s) {
  const float scale = (float)in_len / (float)out_len;
  MG_LAUNCH((resize_aa_pass_kernel<TI, TO>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const TI*)src,
            (TO*)dst, total, in_len, out_len, other, horizontal, bicubic, scale);
}

}  // namespace

// Colour-mapped depth (marigold/util/image_util.py:38-76 colorize_depth_maps + the pipeline's (x * 255).astype(uint8),
// marigold_depth_pipeline.py:318-327): matplotlib's listed / segmented colormaps are 256-entry tables indexed by
// int(x * 256) (x == 1 -> 255); the table arrives as uint8 RGB (built once per colormap on the host from matplotlib
// itself), the output is the HWC uint8 image PIL takes.
__global__ __launch_bounds__(256) void colorize_kernel(const float* __restrict__ depth, const uint8_t* __restrict__ lut,
                                                       uint8_t* __restrict__ out, long long n, float lo, float inv_range) {
  __shared__ uint8_t tab[768];
  for (int i = threadIdx.x; i < 768; i += 256) tab[i] = lut[i];
  __syncthreads();
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float x = (depth[i] - lo) * inv_range;
    x = fminf(fmaxf(x, 0.f), 1.f);
    int k = (int)(x * 256.0f);
    k = k > 255 ? 255 : k;
    out[3 * i + 0] = tab[3 * k + 0];
    out[3 * i + 1] = tab[3 * k + 1];
    out[3 * i + 2] = tab[3 * k + 2];
  }
}

int mg_launch_resize(const mg_op* op, hipStream_t s) {
  if (op->kind == MG_OP_COLORIZE) {
    const long long n = op->l[0];
    MG_REQUIRE(n > 0 && op->p[0] && op->p[1] && op->p[2], "colorize: null pointer / empty map");
    MG_REQUIRE(op->f[1] > op->f[0], "colorize: max_depth must exceed min_depth");
    MG_LAUNCH(colorize_kernel, dim3((unsigned)min((n + 255) / 256, (long long)4096)), dim3(256), 0, s, (const float*)op->p[0],
              (const uint8_t*)op->p[1], (uint8_t*)op->p[2], n, op->f[0], 1.0f / (op->f[1] - op->f[0]));
    if (!g_dry_run) MG_CHECK_HIP(hipGetLastError());
    return 0;
  }
  const long long planes = op->i[0];
  const int Hin = op->i[1], Win = op->i[2], Hout = op->i[3], Wout = op->i[4], mode = op->i[5];
  const int u8 = op->i[6];  // 1: uint8 in and out, 0: fp32
  MG_REQUIRE(planes > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "resize: empty image");
  MG_REQUIRE(mode >= 0 && mode <= 2, "resize: mode must be 0 (bilinear), 1 (bicubic) or 2 (nearest-exact)");
  MG_REQUIRE(op->p[0] && op->p[1], "resize: null pointer");
  if (mode == 2) {
    const long long total = planes * Hout * Wout;
    const float sy = (float)Hin / (float)Hout, sx = (float)Win / (float)Wout;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (u8) MG_LAUNCH(resize_nearest_exact_kernel<uint8_t>, grid, dim3(256), 0, s, (const uint8_t*)op->p[0],
                      (uint8_t*)op->p[1], total, Hin, Win, Hout, Wout, sy, sx);
    else MG_LAUNCH(resize_nearest_exact_kernel<float>, grid, dim3(256), 0, s, (const float*)op->p[0],
                   (float*)op->p[1], total, Hin, Win, Hout, Wout, sy, sx);
  } else {
    const int bicubic = mode == 1;
    const bool do_h = Win != Wout, do_v = Hin != Hout;
    MG_REQUIRE(!(do_h && do_v) || op->p[2], "resize: fp32 temporary [planes][Hin][Wout] missing");
    // horizontal first (rows = Hin), then vertical on the result - torch's order
    if (do_h && do_v) {
      if (u8) launch_pass<uint8_t, float>(op->p[0], op->p[2], planes * Hin * Wout, Win, Wout, Hin, 1, bicubic, s);
      else launch_pass<float, float>(op->p[0], op->p[2], planes * Hin * Wout, Win, Wout, Hin, 1, bicubic, s);
      if (u8) launch_pass<float, uint8_t>(op->p[2], op->p[1], planes * Hout * Wout, Hin, Hout, Wout, 0, bicubic, s);
      else launch_pass<float, float>(op->p[2], op->p[1], planes * Hout * Wout, Hin, Hout, Wout, 0, bicubic, s);
    } else if (do_h) {
      if (u8) launch_pass<uint8_t, uint8_t>(op->p[0], op->p[1], planes * Hin * Wout, Win, Wout, Hin, 1, bicubic, s);
      else launch_pass<float, float>(op->p[0], op->p[1], planes * Hin * Wout, Win, Wout, Hin, 1, bicubic, s);
    } else if (do_v) {
      if (u8) launch_pass<uint8_t, uint8_t>(op->p[0], op->p[1], planes * Hout * Wout, Hin, Hout, Wout, 0, bicubic, s);
      else launch_pass<float, float>(op->p[0], op->p[1], planes * Hout * Wout, Hin, Hout, Wout, 0, bicubic, s);
    } else {
      MG_REQUIRE(false, "resize: sizes are equal (the caller returns the input unchanged)");
    }
  }
  if (!g_dry_run) MG_CHECK_HIP(hipGetLastError());
  return 0;
}
