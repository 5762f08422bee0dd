// Flash attention at head width 64, hand-placed (MG_OP_FLASH_ATTN64 variant 26): the UNet's self-attention at 9 216 and
// 2 304 tokens (marigold/marigold_depth_pipeline.py:461-463 -> diffusers Attention).
//
// flash_attn64_v25 (attention.hip) leaves hipcc to order the tile's work and relies on three waves per SIMD to overlap the
// softmax's VALU stream with the MFMAs.  On gfx950 that overlap is only partial (tools/ubench/coissue2.hip: three waves that
// alternate whole phases reach 1.44 x one wave, an instruction-by-instruction interleave in ONE wave 1.57 x), and the compiled
// loop waits lgkmcnt(0) in front of every MFMA.  Here:
//   * 64 queries per wave = two 32-query blocks that share every K / V^T fragment read from LDS (half the LDS traffic per MFMA:
//     at 32 queries per wave the fragment reads alone need the CU's whole 128 B / cycle), four waves per workgroup, TWO
//     workgroups per CU (256 registers per lane, all of them VGPRs: hipcc splits the file 128 / 128 as soon as a function
//     touches AGPRs, so the output accumulators stay in VGPRs and the file is built in the VGPR MFMA form) - one wave alone is bound
//     by its own instruction issue (~1 600 cycles per tile against 1 024 of MFMA time), a second one fills its gaps;
//   * the key loop is ONE asm statement written by gen_fa4w.py (flash4w.inc, where the schedule is described): per 64-key tile
//     32 MFMAs with five VALU instructions of the softmax behind each, the QK^T MFMAs of the next 32 keys and the P V MFMAs of
//     the current ones in the same stream, fragments held in registers for both query blocks and reloaded with counted waits,
//     the LDS-DMA pieces of ring slot t + 3 among them, one barrier per tile;
//   * softmax against a FIXED per-query reference (the row maximum over the first 32 keys, subtracted by the first QK^T MFMA's
//     C operand): no running maximum, no rescaling - 16 v_max3 and the branch per tile are gone and the output accumulators
//     are only ever touched by MFMAs.  Exact as long as no later score tops its reference by more than ~2^100; the row sums
//     tell (>= 2^100 or not finite) and such a workgroup redoes its queries with the running-maximum loop of flash_attn64_v25
//     (flash25_body.h; FaArgs::redo_thr lets the tests force that path).
// Ring slot t of the four 16 KB slots holds what iteration t reads: K rows 64 t + 32 ... 64 t + 95 and V^T tile t; K rows
// 0 ... 31 go to a 4 KB region of their own for the prologue.  Needs Ntok % 256 == 0, at least four key tiles and V^T in the
// accumulator's key order (vt_perm).
#include "flash_args.h"
#include "flash25_body.h"
#include "flash4w.inc"

// the score set Y of the stream (v[160:191]): physical registers, scratch to the compiler
#define F4_CLOBBER_Y "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", \
                     "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191"

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;
constexpr int F4_SLOT = 16384, F4_PRE = 4 * F4_SLOT, F4_LDS = F4_PRE + 4096;

__device__ __forceinline__ void f4_mfma(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void f4_dma(unsigned voff, i32x4 srd, unsigned soff, unsigned m0v) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(srd), "s"(soff), "s"(m0v) : "memory");
}
__device__ __forceinline__ unsigned f4_sgpr(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void flash_attn64_4w_kernel(const FaArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // the ring: LDS address 0 (the stream wraps addresses at 64 KB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int nqb = a.Ntok >> 8;
  const int qb = bid % nqb, bh = bid / nqb;
  const int h = bh % a.heads, b = bh / a.heads;
  const bf16_t* Qb = a.Q + (long long)b * a.sQ + h * 64;
  const bf16_t* Kb = a.K + (long long)b * a.sK + h * 64;
  const bf16_t* Vb = a.Vt + (long long)b * a.sVt + (long long)h * 64 * a.ldvt;

  bf16x8 qf[2][4];
  {
    const float c = a.scale_log2;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int q_row = qb * 256 + wave * 64 + q * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint4 u = *(const uint4*)(Qb + (long long)q_row * a.ldq + ks * 16 + half * 8);
        uint4 w;
        w.x = cvt_pk_bf16_f32(bflo(u.x) * c, bfhi(u.x) * c); w.y = cvt_pk_bf16_f32(bflo(u.y) * c, bfhi(u.y) * c);
        w.z = cvt_pk_bf16_f32(bflo(u.z) * c, bfhi(u.z) * c); w.w = cvt_pk_bf16_f32(bflo(u.w) * c, bfhi(u.w) * c);
        qf[q][ks] = __builtin_bit_cast(bf16x8, w);
      }
    }
  }
  // LDS-DMA: 16-byte element ci = it * 256 + tid of a tile = row ci >> 3, chunk position ci & 7 (source chunk XOR-swizzled)
  unsigned vk[2], vv[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int ci = it * 256 + tid, r = ci >> 3, q = (ci & 7) ^ ((r >> 1) & 7);
    vk[it] = (unsigned)(r * a.ldq * 2 + q * 16);
    vv[it] = (unsigned)(r * a.ldvt * 2 + q * 16);
  }
  auto srd_of = [](const void* p, unsigned bytes) {   // raw buffer: offsets >= bytes read as zero
    const unsigned long long u = (unsigned long long)(uintptr_t)p;
    const i32x4 r = {(int)f4_sgpr((unsigned)u), (int)(f4_sgpr((unsigned)(u >> 32)) & 0xffffu), (int)f4_sgpr(bytes), 0x00020000};
    return r;
  };
  // (the last slot's K piece reaches 32 rows past the sequence - the next tile's half 0, never used: zeros, not a stray read)
  const i32x4 srk = srd_of(Kb, (unsigned)((a.Ntok - 1) * a.ldq * 2 + 128)), srv = srd_of(Vb, 0x80000000u);
  const unsigned kst = f4_sgpr((unsigned)(64 * a.ldq * 2));   // one key tile of K rows, bytes
  const unsigned mw = f4_sgpr((unsigned)(wave * 1024));
  // K rows 0 ... 31 into the prologue's region, then slots 0, 1, 2 (K rows 64 s + 32 ..., V^T tile s)
  f4_dma(vk[0], srk, 0u, f4_sgpr(mw + F4_PRE));
#pragma unroll
  for (int sl = 0; sl < 3; ++sl) {
#pragma unroll
    for (int it = 0; it < 2; ++it) f4_dma(vk[it], srk, f4_sgpr((unsigned)sl * kst + (kst >> 1)), f4_sgpr(mw + sl * F4_SLOT + it * 4096));
#pragma unroll
    for (int it = 0; it < 2; ++it) f4_dma(vv[it], srv, (unsigned)(sl * 128), f4_sgpr(mw + sl * F4_SLOT + 8192 + it * 4096));
  }
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // scores of the first 32 keys (S^T = K Q^T: lane = query, registers = keys) and the queries' reference = their row maximum
  const int sw = (l31 >> 1) & 7;
  unsigned ad[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) ad[ks] = (unsigned)(l31 * 128 + (((2 * ks + half) ^ sw) << 4));
  f32x16 s[2], negm[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[q][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const bf16x8 kf = __builtin_bit_cast(bf16x8, *(const uint4*)(smem + F4_PRE + ad[ks]));
    f4_mfma(s[0], kf, qf[0][ks]);
    f4_mfma(s[1], kf, qf[1][ks]);
  }
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    float mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) mx = fmaxf(mx, fmaxf(s[q][r], s[q][r + 1]));
    float x0, x1;
    half_swap(mx, mx, x0, x1);
    const float ref = fmaxf(x0, x1);
#pragma unroll
    for (int r = 0; r < 16; ++r) { negm[q][r] = -ref; s[q][r] -= ref; }
  }
  f32x16 o[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[q][d][r] = 0.f;
  float l00 = 0.f, l01 = 0.f, l10 = 0.f, l11 = 0.f;
  const int nkt = a.Ntok >> 6;
  unsigned sok = f4_sgpr(3u * kst + (kst >> 1)), sov = f4_sgpr(3u * 128u), mb = f4_sgpr(mw + 3 * F4_SLOT), cnt = f4_sgpr((unsigned)(nkt - 3));
  uint4 kf0, kf1, kf2, kf3, vf0, vf1, vf2, vf3;
  unsigned long long t0 = 0, r0 = 0;
  if (a.dbg) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  asm volatile(FA4W_ASM
               : "+{v[128:143]}"(s[0]), "+{v[144:159]}"(s[1]),
                 [o00] "+v"(o[0][0]), [o01] "+v"(o[0][1]), [o10] "+v"(o[1][0]), [o11] "+v"(o[1][1]),
                 [l00] "+v"(l00), [l01] "+v"(l01), [l10] "+v"(l10), [l11] "+v"(l11),
                 [ad0] "+v"(ad[0]), [ad1] "+v"(ad[1]), [ad2] "+v"(ad[2]), [ad3] "+v"(ad[3]),
                 [kf0] "=&v"(kf0), [kf1] "=&v"(kf1), [kf2] "=&v"(kf2), [kf3] "=&v"(kf3),
                 [vf0] "=&v"(vf0), [vf1] "=&v"(vf1), [vf2] "=&v"(vf2), [vf3] "=&v"(vf3),
                 [sok] "+s"(sok), [sov] "+s"(sov), [mb] "+s"(mb), [cnt] "+s"(cnt)
               : [q00] "v"(qf[0][0]), [q01] "v"(qf[0][1]), [q02] "v"(qf[0][2]), [q03] "v"(qf[0][3]),
                 [q10] "v"(qf[1][0]), [q11] "v"(qf[1][1]), [q12] "v"(qf[1][2]), [q13] "v"(qf[1][3]),
                 [ng0] "v"(negm[0]), [ng1] "v"(negm[1]),
                 [vk0] "v"(vk[0]), [vk1] "v"(vk[1]), [vv0] "v"(vv[0]), [vv1] "v"(vv[1]),
                 [srk] "s"(srk), [srv] "s"(srv), [kst] "s"(kst)
               : "memory", "scc", F4_CLOBBER_Y);
  if (a.dbg) {   // tuning only: shader cycles / 100 MHz ticks of the key loop, per wave
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { a.dbg[(bid * 4 + wave) * 2] = t1 - t0; a.dbg[(bid * 4 + wave) * 2 + 1] = r1 - r0; }
  }
  // ---- finalize: O[q][d] = o^T / l; lane^32 exchange -> 8 consecutive d per lane, 16-byte stores ----
  bool bad = false;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    float la, lb;
    const float l_lane = q == 0 ? l00 + l01 : l10 + l11;
    half_swap(l_lane, l_lane, la, lb);
    const float l_tot = la + lb;
    bad = bad || !(l_tot < a.redo_thr);   // 2^100 (or not finite): a score topped the first tile's maximum by about that much
    const float inv = 1.0f / l_tot;
    const int q_row = qb * 256 + wave * 64 + q * 32 + l31;
    bf16_t* orow = a.O + (long long)b * a.sO + (long long)q_row * a.ldo + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) half_swap(o[q][dt][8 * gp + j] * inv, o[q][dt][8 * gp + 4 + j] * inv, v[j], v[4 + j]);
        uint4 pk;
        pk.x = cvt_pk_bf16_f32(v[0], v[1]); pk.y = cvt_pk_bf16_f32(v[2], v[3]);
        pk.z = cvt_pk_bf16_f32(v[4], v[5]); pk.w = cvt_pk_bf16_f32(v[6], v[7]);
        *(uint4*)(orow + dt * 32 + 16 * gp + 8 * half) = pk;
      }
  }
  // a workgroup whose reference was too low redoes its 256 queries with the running-maximum form (two blocks of 128)
  __syncthreads();   // (no __syncthreads_or: its static LDS word would move the ring off LDS address 0)
  if (lane == 0) ((int*)smem)[wave] = __any(bad) ? 1 : 0;
  __syncthreads();
  const int4 flags = *(const int4*)smem;
  __syncthreads();
  if (flags.x | flags.y | flags.z | flags.w) {
    fa25_body<4, true, 2>(a, smem, 2 * qb, bh);
    __syncthreads();
    fa25_body<4, true, 2>(a, smem, 2 * qb + 1, bh);
  }
}

}  // namespace

bool mg_flash4w_ok(const FaArgs& a, bool vt_perm) {
  const int nkt = a.Ntok / 64;
  return vt_perm && a.Ntok % 256 == 0 && nkt >= 4 && a.ldq % 8 == 0 && a.ldo % 8 == 0 && a.ldvt % 8 == 0 &&
         (uintptr_t)a.Q % 16 == 0 && (uintptr_t)a.K % 16 == 0 && (uintptr_t)a.Vt % 16 == 0 && (uintptr_t)a.O % 16 == 0 &&
         a.sQ % 8 == 0 && a.sK % 8 == 0 && a.sVt % 8 == 0 && a.sO % 8 == 0 &&
         (long long)a.Ntok * a.ldq * 2 < (1ll << 31) && 64ll * a.ldvt * 2 < (1ll << 31);
}

int mg_launch_flash4w(const FaArgs& a, hipStream_t s) {
  MG_REQUIRE(mg_flash4w_ok(a, true), "flash_attn64 (hand-placed form): Ntok %d must be a multiple of 256, "
             "16-byte aligned operands", a.Ntok);
  const int LDS = F4_LDS;
  static bool attr_set = false;
  if (!attr_set && !g_dry_run) {
    hipFuncAttributes fa;
    MG_CHECK_HIP(hipFuncGetAttributes(&fa, (const void*)flash_attn64_4w_kernel));
    MG_REQUIRE(fa.sharedSizeBytes == 0, "flash_attn64 (hand-placed form): the ring must start at LDS address 0 (static LDS %d bytes)", (int)fa.sharedSizeBytes);
    MG_CHECK_HIP(hipFuncSetAttribute((const void*)flash_attn64_4w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  const long long grid = (long long)(a.Ntok / 256) * a.heads * a.B;
  MG_LAUNCH(flash_attn64_4w_kernel, dim3((unsigned)grid), dim3(256), LDS, s, a);
  return 0;
}
