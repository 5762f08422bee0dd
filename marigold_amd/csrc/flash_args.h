// Arguments of the head-width-64 flash attention kernels (attention.hip, flash4w.hip).
#pragma once
#include "common.h"

struct FaArgs {
  const bf16_t* Q;
  const bf16_t* K;
  const bf16_t* Vt;
  bf16_t* O;
  const void* zero;
  int B, heads, Ntok, ldq, ldo, ldvt, nqb;
  long long sQ, sK, sVt, sO;
  float scale_log2;
  unsigned long long* dbg;   // tuning only: per workgroup (shader cycles, 100 MHz ticks) of the whole kernel body
  float redo_thr;            // flash4w.hip: row sums at or above this send the workgroup to the running-maximum loop (2^100)
  void* ws;                  // flash4w.hip: workspace of the key-split blocks (tickets + partial results), or null: no split
  long long ws_bytes;
  int split;                 // 0: split the left-over blocks when it pays, 1: always (tests), 2: never
  int m16;                   // flash4w.hip: the key loop on 16x16x32 MFMAs (variant 27) instead of 32x32x16 (26); < 0: by shape
  int n_full, n_rem, n_rem_wg;   // (set by mg_launch_flash4w) whole blocks, split blocks, workgroups over the split blocks
};

constexpr int FA_QB = 128;   // queries per workgroup (4 waves x 32)
constexpr int FA_KB = 64;    // keys per tile
constexpr int FA_STAGE = 2 * FA_KB * 128;  // K tile + V^T tile, bytes
constexpr int FA2_NSTAGE = 3;
constexpr float FA3_THR = 3.0f;         // log2 units: the running max is raised when a row max exceeds it by > 2^3
typedef __attribute__((ext_vector_type(2))) float fa_f32x2;
typedef mg_bf16x2_t fa_bf16x2_t;   // (common.h: bf16, or fp16 in the fp16 build; probabilities need no saturation)
typedef __attribute__((ext_vector_type(2))) float fa_f32x2_t;
__device__ __forceinline__ uint32_t fa_cvt_pk(float lo, float hi) {
  fa_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, fa_bf16x2_t));
}

// flash4w.hip: the hand-placed one-wave-per-SIMD kernel (variant 26); -1 if the shape is not its (the caller falls back)
bool mg_flash4w_ok(const FaArgs& a, bool vt_perm);
int mg_launch_flash4w(const FaArgs& a, hipStream_t s);
