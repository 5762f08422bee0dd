// Micro-benchmark: HBM write bandwidth of a GEMM-tile-shaped store as a function of the contiguous bytes one wave
// instruction puts on a row (S) - the question behind the short-K GEMM floor (profiles/r2_store_pattern.log).
// Each workgroup (256 threads) writes one 256-row x 256-byte tile of a [M][N] bf16 matrix, 16 bytes per lane per store.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int S>
__global__ __launch_bounds__(256) void store_tile(uint4* out, int ldb /*row bytes*/, int tiles_n) {
  const int t = blockIdx.x, tm = t / tiles_n, tn = t % tiles_n;
  char* base = (char*)out + (long long)tm * 256 * ldb + tn * 256;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int LPR = S / 16, R = 64 / LPR, CB = 256 / S;
  const uint4 v = make_uint4(t, lane, wave, 1);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int b = wave * 16 + i, cb = b % CB, rb = b / CB;
    const int row = rb * R + lane / LPR, colb = cb * S + (lane % LPR) * 16;
    *(uint4*)(base + (long long)row * ldb + colb) = v;
  }
}
// the GEMM epilogue's real order: wave (wm, wn) owns 128 rows x 128 bytes; ni (64 B) -> gp (32 B) -> mi (32 rows)
__global__ __launch_bounds__(256) void store_gemm_order(uint4* out, int ldb, int tiles_n) {
  const int t = blockIdx.x, tm = t / tiles_n, tn = t % tiles_n;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
  char* base = (char*)out + ((long long)tm * 256 + wm * 128) * ldb + tn * 256 + wn * 128;
  const uint4 v = make_uint4(t, lane, wave, 1);
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) *(uint4*)(base + (long long)(mi * 32 + l31) * ldb + ni * 64 + gp * 32 + half * 16) = v;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <class F> float time_it(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 10;
}
int main() {
  const int shapes[][2] = {{92160, 2560}, {92160, 1280}, {368640, 640}, {92160, 320}};
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], ldb = N * 2, tiles_n = ldb / 256, tiles = (M / 256) * tiles_n;
    uint4* out; CK(hipMalloc(&out, (size_t)M * ldb));
    const double mb = (double)M * ldb / 1e6;
    printf("M=%d N=%d (%.0f MB):", M, N, mb);
    float ms;
    ms = time_it([&] { store_gemm_order<<<tiles, 256>>>(out, ldb, tiles_n); }); printf("  gemm-order %.0fus %.2fTB/s |", ms * 1e3, mb / ms / 1e3);
    ms = time_it([&] { store_tile<32><<<tiles, 256>>>(out, ldb, tiles_n); }); printf("  S=32 %.0fus %.2f", ms * 1e3, mb / ms / 1e3);
    ms = time_it([&] { store_tile<64><<<tiles, 256>>>(out, ldb, tiles_n); }); printf("  S=64 %.0fus %.2f", ms * 1e3, mb / ms / 1e3);
    ms = time_it([&] { store_tile<128><<<tiles, 256>>>(out, ldb, tiles_n); }); printf("  S=128 %.0fus %.2f", ms * 1e3, mb / ms / 1e3);
    ms = time_it([&] { store_tile<256><<<tiles, 256>>>(out, ldb, tiles_n); }); printf("  S=256 %.0fus %.2f", ms * 1e3, mb / ms / 1e3);
    ms = time_it([&] { CK(hipMemsetAsync(out, 0, (size_t)M * ldb)); }); printf("  | memset %.0fus %.2f\n", ms * 1e3, mb / ms / 1e3);
    CK(hipFree(out));
  }
  return 0;
}
