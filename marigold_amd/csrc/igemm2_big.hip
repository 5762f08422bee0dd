// Implicit GEMM with a 128 x 128 WAVE tile: 256 x 256 workgroup tile on four waves, one wave per SIMD, the 256 fp32
// accumulators of a lane in the AGPR half of its 512-register file (this file is built WITHOUT -amdgpu-mfma-vgpr-form, see
// the Makefile).  Why (round 3, profiles/r3_pmc_sq_summary.json): the SIMD issues from one wave at a time and every
// instruction costs it ~4-5 cycles, so what bounds the 8-wave tiles is the number of instructions around each MFMA
// (5-9 VALU + LDS + SALU per MFMA against 0.8 for the vendor GEMM); a 128 x 128 wave tile halves the fragment reads per
// MFMA (0.5) and all the per-K-step bookkeeping is amortised over 16 MFMAs per k-substep.  Same body, same epilogues.
#include "igemm2_body.h"

namespace {

// The tile with the K loop placed by hand (igemm2_body.h, LOOP == 3; gen_k4w.py): buffer-load LDS-DMA on SGPR bases,
// whole-tile fragment sets, the next-but-one K tile in flight.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void igemm2_k4w_kernel(const Igemm2Args a) {
  igemm2_body<256, 256, 2, 2, 2, false, false, 3, 0, 64>(a);
}

// ... and its full-width sibling for the N = 320 k layers: 192 x 320 (wave tile 96 x 160)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void igemm2_k4wb_kernel(const Igemm2Args a) {
  igemm2_body<192, 320, 2, 2, 2, false, false, 3, 0, 64>(a);
}

// the same two kernels with their phase stamps compiled in (tuning only: MARIGOLD_IGEMM_STAMPS=1, tools/igemm_phases.py)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void igemm2_k4w_stamped_kernel(const Igemm2Args a) {
  igemm2_body<256, 256, 2, 2, 2, false, false, 3, 0, 64, 0, true>(a);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void igemm2_k4wb_stamped_kernel(const Igemm2Args a) {
  igemm2_body<192, 320, 2, 2, 2, false, false, 3, 0, 64, 0, true>(a);
}

}  // namespace

// which: 2 = the hand-placed 256 x 256 tile, 3 = its 192 x 320 sibling.  The argument struct has the same layout in every
// translation unit (igemm2_body.h); the pointer is launched by igemm2.hip::launch2.
void* mg_igemm2_big_kernel(int which) {
  if (which >= 4) return which == 5 ? (void*)igemm2_k4wb_stamped_kernel : (void*)igemm2_k4w_stamped_kernel;   // + 2: instrumented
  return which == 3 ? (void*)igemm2_k4wb_kernel : (void*)igemm2_k4w_kernel;
}
