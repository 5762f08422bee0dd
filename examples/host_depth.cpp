// A host program WITHOUT Python: loads a model image (marigold_amd/image.py::export_model_image) and runs the reference's
// single_infer + ensemble_depth chain (marigold/marigold_depth_pipeline.py:396-477, marigold/util/ensemble.py:39-196) through the
// module-level C ABI of libmarigold_hip.so (include/marigold_hip.h):
//     mg_model_vae_encode -> mg_model_denoise -> mg_model_vae_decode -> mg_ensemble_depth
// Build (gfx950 box):  hipcc -O2 examples/host_depth.cpp -Iinclude -Lmarigold_amd -lmarigold_hip -Wl,-rpath,$PWD/marigold_amd -o host_depth
// Run:                 ./host_depth model.mgimg rgb.f32 noise.f32 depth_out.f32
//   rgb.f32   raw fp32 [1,3,H,W] in [-1,1];  noise.f32  raw fp32 [B,4,h,w] (the initial latents);  depth_out.f32  raw fp32 [H',W']
// (tests/test_gpu_pipeline.py::test_model_image_from_a_c_host builds and runs it and compares with the Python pipeline bit for bit.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "marigold_hip.h"

#define CHECK(x)                                                              \
  do {                                                                        \
    if ((x) != 0) {                                                           \
      fprintf(stderr, "%s failed: %s\n", #x, mg_last_error());                \
      return 1;                                                               \
    }                                                                         \
  } while (0)
#define HIPCHECK(x)                                                           \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));          \
      return 1;                                                               \
    }                                                                         \
  } while (0)

static bool read_file(const char* path, std::vector<float>& v) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  const size_t n = fread(v.data(), sizeof(float), v.size(), f);
  fclose(f);
  return n == v.size();
}

int main(int argc, char** argv) {
  if (argc != 5) {
    fprintf(stderr, "usage: %s model.mgimg rgb.f32 noise.f32 depth_out.f32\n", argv[0]);
    return 2;
  }
  mg_model* m = mg_model_load(argv[1], 0);
  if (!m) {
    fprintf(stderr, "mg_model_load: %s\n", mg_last_error());
    return 1;
  }
  int cfg[16];
  CHECK(mg_model_info(m, cfg));
  const int B = cfg[0], H = cfg[1], W = cfg[2], h = cfg[3], w = cfg[4], steps = cfg[5], C = cfg[6], Ho = cfg[11], Wo = cfg[12];
  if (C != 1 || cfg[8] != 0) {
    fprintf(stderr, "this example runs depth images with a noise-free (DDIM) scheduler\n");
    return 2;
  }
  printf("model image: %d member(s) of %dx%d, latent %dx%d, %d steps, %.1f MB on the device\n", B, H, W, h, w, steps,
         mg_model_device_bytes(m) / 1e6);
  std::vector<float> rgb((size_t)3 * H * W), noise((size_t)B * 4 * h * w), depth((size_t)Ho * Wo);
  if (!read_file(argv[2], rgb) || !read_file(argv[3], noise)) {
    fprintf(stderr, "cannot read the inputs\n");
    return 1;
  }
  float *d_rgb, *d_lat, *d_x, *d_pred, *d_depth;
  HIPCHECK(hipMalloc(&d_rgb, rgb.size() * 4));
  HIPCHECK(hipMalloc(&d_lat, (size_t)4 * h * w * 4));
  HIPCHECK(hipMalloc(&d_x, noise.size() * 4));
  HIPCHECK(hipMalloc(&d_pred, (size_t)B * Ho * Wo * 4));
  HIPCHECK(hipMalloc(&d_depth, depth.size() * 4));
  hipStream_t s;
  HIPCHECK(hipStreamCreate(&s));
  HIPCHECK(hipMemcpy(d_rgb, rgb.data(), rgb.size() * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(d_x, noise.data(), noise.size() * 4, hipMemcpyHostToDevice));
  CHECK(mg_model_vae_encode(m, d_rgb, d_lat, s));          // encode_rgb
  CHECK(mg_model_denoise(m, d_lat, d_x, nullptr, s));      // T x (unet + scheduler.step)
  CHECK(mg_model_vae_decode(m, d_x, d_pred, s));           // decode_depth
  double info[4] = {0, 0, 0, 0};
  if (B > 1) {
    CHECK(mg_ensemble_depth(d_pred, B, Ho, Wo, 1, 1, 0, 0.02, 50, 1e-6, 1024, d_depth, nullptr, info, s));   // the reference's defaults
  } else {
    HIPCHECK(hipMemcpyAsync(d_depth, d_pred, depth.size() * 4, hipMemcpyDeviceToDevice, s));
  }
  HIPCHECK(hipStreamSynchronize(s));
  HIPCHECK(hipMemcpy(depth.data(), d_depth, depth.size() * 4, hipMemcpyDeviceToHost));
  FILE* f = fopen(argv[4], "wb");
  if (!f || fwrite(depth.data(), 4, depth.size(), f) != depth.size()) {
    fprintf(stderr, "cannot write %s\n", argv[4]);
    return 1;
  }
  fclose(f);
  double sum = 0;
  for (float v : depth) sum += v;
  printf("depth %dx%d written, mean %.6f; alignment: cost %.6g after %d evaluations / %d iterations (status %d)\n", Ho, Wo,
         sum / depth.size(), info[0], (int)info[1], (int)info[2], (int)info[3]);
  mg_model_destroy(m);
  return 0;
}
