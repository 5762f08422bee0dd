// Body of the hand-placed four-wave patch-resident 3x3 convolution, included once per geometry by conv_patch4w.hip with
//   CP4_NAME   kernel name                      CP4_TH  tile rows (16 pixels wide)      CP4_BN  output channels per tile
//   CP4_MI / CP4_NI  32-row / 32-column fragments of the wave tile (2 x 2 waves)
//   CP4_NPW    1 KB patch pieces allocated per buffer (a multiple of 4: NPW / 4 per wave)
//   CP4_T(x)   the instruction stream CP4A_x / CP4B_x (conv_patch4w.inc)
//   CP4_OUT / CP4_IN  the operand lists of those streams
// Design notes: conv_patch4w.hip, gen_cp4w.py.

template <bool FIX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void CP4_NAME(const ConvPArgs a) {
  typedef __attribute__((ext_vector_type(4))) int i32x4;
  constexpr int MI = CP4_MI, NI = CP4_NI, TH = CP4_TH, BN = CP4_BN;
  constexpr int TW = 16, BM = TH * TW, TM = BM / 2, TN = BN / 2;
  constexpr int ROWB = 128;
  constexpr int NPW = CP4_NPW, NSLOT = NPW / 4;          // patch pieces per buffer / per wave
  constexpr int PATCHB = NPW * 1024, BSTAGE = BN * ROWB, RING = 2 * BSTAGE;
  constexpr int NWP = BN * 8 / 256;                      // weight pieces per wave and K tile
  constexpr int NFIX = NSLOT;                            // fix-up vectors per thread (vector e = it * 256 + tid)
  constexpr unsigned OOB = 0x80000000u;
  static_assert(TM == MI * 32 && TN == NI * 32 && NWP == 2 * NI && ((TH + 2) * (TW + 2) + 7) / 8 <= NPW, "geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const ssl = (float*)(smem + RING + 2 * PATCHB);   // [2][Cin] scale / shift of this image (FIX)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;

  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = bid % a.tiles_n; bid /= a.tiles_n;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; bid /= a.tiles_y;
  const int img = bid % a.B;
  const int z = bid / a.B;                                   // output parity 2a+b in sub-pixel mode, else 0
  const int n0 = tile_n * BN, y0 = ty * TH, x0 = tx * TW;
  const int pad_y = a.subpix ? 1 - (z >> 1) : 1, pad_x = a.subpix ? 1 - (z & 1) : 1;
  const int PW = TW + a.tw - 1, PR = (TH + a.tw - 1) * PW;
  const int T = a.T, chunks = a.chunks, S = chunks * T;

  auto sgpr = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
  auto srd_of = [&](const void* base, unsigned long long byte_off) {
    const unsigned long long b = (unsigned long long)(uintptr_t)base + byte_off;
    const i32x4 r = {(int)sgpr((unsigned)b), (int)(sgpr((unsigned)(b >> 32)) & 0xffffu), (int)OOB, 0x00020000};
    return r;
  };

  // ---- patch staging: piece i = 4 slot + wave covers patch rows 8 i .. 8 i + 7; lane -> row 8 i + (lane >> 3), 16-byte slot
  // lane & 7 holding channel group (lane & 7) ^ ((row >> 1) & 7) (the XOR swizzle rides on the DMA source address) ----
  const int p_j8 = ((lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7)) * 8;   // (row >> 1) & 7 = 4 (wave & 1) + (lane >> 4)
  unsigned vp[NSLOT];
  auto patch_offsets = [&](int lda) {   // (once per source: the pixel indices are not kept in registers)
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const int r = (k * 4 + wave) * 8 + (lane >> 3);
      const int pr = r / PW, pc = r - pr * PW;
      const int iy = y0 - pad_y + pr, ix = x0 - pad_x + pc;
      const bool ok = r < PR && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      vp[k] = ok ? (unsigned)(((img * a.H + iy) * a.W + ix) * lda + p_j8) * 2u : OOB;
    }
  };
  auto patch_srd = [&](int c) {   // channel tile c of the concatenated input
    return c < a.c0t ? srd_of(a.A0, (unsigned long long)c * 128) : srd_of(a.A1, (unsigned long long)(c - a.c0t) * 128);
  };
  // ---- weight staging ----
  const bf16_t* Wb = a.Wt + (long long)z * a.sW;
  unsigned vb[NWP];
#pragma unroll
  for (int it = 0; it < NWP; ++it) {
    const int ci = it * 256 + tid;
    const int r = ci >> 3, p = ci & 7;
    const int n = n0 + r;
    vb[it] = n < a.N ? (unsigned)(n * a.ldw + (p ^ ((r >> 1) & 7)) * 8) * 2u : OOB;
  }
  auto weight_srd = [&](int k) {   // K tile k = (channel tile k / T, tap k % T): weight columns [tap * Cin + c * 64, +64)
    const int c = k / T, t = k - c * T;
    return srd_of(Wb, ((unsigned long long)t * a.Cin + (unsigned long long)c * 64) * 2);
  };
  auto stage_piece = [](unsigned voff, const i32x4& srd, int m0v) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(srd), "s"(m0v) : "memory");
  };

  // ---- fragment read addresses ----
  const int sw = (l31 >> 1) & 7;
  const int hx = half << 4;
  int lb[4], xb[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    lb[ks] = (wn * TN + l31) * ROWB + (((2 * ks + half) ^ sw) << 4);
    xb[ks] = lb[ks] ^ (lb[ks] + BSTAGE);
  }
  int r0[MI], pa[MI][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int p = wm * TM + mi * 32 + l31;
    r0[mi] = (p >> 4) * PW + (p & 15);
  }
  auto tap_off = [&](int t) { const int dy = t / a.tw; return dy * PW + (t - dy * a.tw); };
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {   // tile 0: tap 0 of patch buffer 0 (the streams compute the later tiles' themselves)
    const int r = r0[mi];
    const int swz = ((r << 3) & 0x70) ^ hx;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) pa[mi][ks] = (swz ^ (ks << 5)) + r * ROWB + RING;
  }

  f32x16 acc[NI][MI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
  bf16x8 fa4[4][MI], fb4[4][NI];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
    for (int i = 0; i < MI; ++i) fa4[ks][i] = __builtin_bit_cast(bf16x8, make_uint4(0, 0, 0, 0));
#pragma unroll
    for (int i = 0; i < NI; ++i) fb4[ks][i] = __builtin_bit_cast(bf16x8, make_uint4(0, 0, 0, 0));
  }

  // ---- fix-up state (fused GroupNorm affine + SiLU on the staged patch, in place) ----
  // vector e = it * 256 + tid: patch row e >> 3, slot tid & 7 = channel group (tid & 7) ^ ((row >> 1) & 7) with
  // (row >> 1) & 7 = (tid >> 4) & 7 for every it: one set of 8 scales / shifts per thread and channel tile
  unsigned fmask = 0;
  const int fix_j8 = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;
  float fs[8], fh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { fs[j] = 0.f; fh[j] = 0.f; }
  auto fix_load = [&](int c) {
    const float* scp = ssl + c * 64 + fix_j8;
    const float4 s0 = *(const float4*)scp, s1 = *(const float4*)(scp + 4);
    const float4 h0 = *(const float4*)(scp + a.Cin), h1 = *(const float4*)(scp + a.Cin + 4);
    fs[0] = s0.x; fs[1] = s0.y; fs[2] = s0.z; fs[3] = s0.w; fs[4] = s1.x; fs[5] = s1.y; fs[6] = s1.z; fs[7] = s1.w;
    fh[0] = h0.x; fh[1] = h0.y; fh[2] = h0.z; fh[3] = h0.w; fh[4] = h1.x; fh[5] = h1.y; fh[6] = h1.z; fh[7] = h1.w;
  };
  if constexpr (FIX) {
#pragma unroll
    for (int it = 0; it < NFIX; ++it) {
      const int r = (it * 256 + tid) >> 3;
      const int pr = r / PW, pc = r - pr * PW;
      const int iy = y0 - pad_y + pr, ix = x0 - pad_x + pc;
      if (r < PR && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) fmask |= 1u << it;
    }
    const float* src = a.ss + (long long)img * 2 * a.Cin;
    for (int i = tid; i < 2 * a.Cin; i += 256) ssl[i] = src[i];
  }

  // ---- prologue: patch of channel tile 0, weight tiles 0 and 1 (S >= 4) ----
  patch_offsets(a.lda0);
  {
    const i32x4 sp = patch_srd(0);
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) stage_piece(vp[k], sp, RING + (k * 4 + wave) * 1024);
    const i32x4 w0 = weight_srd(0), w1 = weight_srd(1);
#pragma unroll
    for (int it = 0; it < NWP; ++it) stage_piece(vb[it], w0, it * 4096 + wave * 1024);
#pragma unroll
    for (int it = 0; it < NWP; ++it) stage_piece(vb[it], w1, BSTAGE + it * 4096 + wave * 1024);
  }
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NWP) : "memory");   // patch 0, weight tile 0 (and the scale / shift table)
  __builtin_amdgcn_s_barrier();
  if constexpr (FIX) {
    fix_load(0);
#pragma unroll
    for (int it = 0; it < NFIX; ++it) {
      if (!((fmask >> it) & 1u)) continue;
      uint4* p = (uint4*)(smem + RING + (it * 256 + tid) * 16);
      const uint4 u = *p;
      float v[8] = {bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y), bflo(u.z), bfhi(u.z), bflo(u.w), bfhi(u.w)};
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = silu_fast_f(__builtin_fmaf(v[j], fs[j], fh[j]));
      uint4 o;
      o.x = cvt_pk_bf16_f32(v[0], v[1]); o.y = cvt_pk_bf16_f32(v[2], v[3]);
      o.z = cvt_pk_bf16_f32(v[4], v[5]); o.w = cvt_pk_bf16_f32(v[6], v[7]);
      *p = o;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- the K loop: one hand-placed stream per tile ----
  const int s96 = 96;
  int fxa = 0;
  int stoff = 0, spb = RING, mw = wave * 1024, mp = 0;
  i32x4 swt = weight_srd(2 < S ? 2 : 0), spt = patch_srd(0);
#define CP4_ASM(text) asm volatile(text : CP4_OUT : CP4_IN : "memory", "scc", "vcc", CP4_SCRATCH)
#define CP4_ASM_SEL(text) asm volatile(text : CP4_OUT : CP4_IN, [sel] "s"(sel) : "memory", "scc", "vcc", CP4_SCRATCH)
  CP4_ASM(CP4_T(PROLOGUE));
  int c = 0, t = 0;
  for (int k = 0; k + 2 < S; ++k) {
    // state of tile k + 1 (its pixel-side read addresses are computed inside this tile's stream)
    int c1 = c, t1 = t + 1;
    if (t1 == T) { t1 = 0; ++c1; }
    stoff = tap_off(t1);
    spb = RING + (c1 & 1) * PATCHB;
    swt = weight_srd(k + 2);
    mw = (k & 1) * BSTAGE + wave * 1024;
    const bool next_chunk = c + 1 < chunks;
    int sel = 0;   // stream variant: 0 plain, 1 stages the patch of channel tile c + 1, 2-8 fix-up slices of that patch
    if (t == 0 && next_chunk) {   // (the buffer of channel tile c + 1 was last read in channel tile c - 1)
      if (c + 1 == a.c0t) patch_offsets(a.lda1);
      spt = patch_srd(c + 1);
      mp = RING + ((c + 1) & 1) * PATCHB + wave * 1024;
      sel = 1;
    } else if (FIX && next_chunk && t >= 2 && t <= 8) {
      if (t == 2) {
        fix_load(c + 1);
        fxa = RING + ((c + 1) & 1) * PATCHB + tid * 16;
      }
      sel = t;
    }
    // (scalar operands pinned to SGPRs: hipcc keeps a value merged from several branches in a VGPR and hands THAT to "s")
    sel = __builtin_amdgcn_readfirstlane(sel);
    stoff = __builtin_amdgcn_readfirstlane(stoff);
    spb = __builtin_amdgcn_readfirstlane(spb);
    mw = __builtin_amdgcn_readfirstlane(mw);
    mp = __builtin_amdgcn_readfirstlane(mp);
    CP4_ASM_SEL(CP4_T(LOOP));
    c = c1;
    t = t1;
  }
  {
    int t1 = t + 1;   // tiles S - 2 and S - 1 belong to the last channel tile (T >= 4)
    stoff = __builtin_amdgcn_readfirstlane(tap_off(t1));
    spb = __builtin_amdgcn_readfirstlane(RING + (c & 1) * PATCHB);
    mw = __builtin_amdgcn_readfirstlane(mw);
    mp = __builtin_amdgcn_readfirstlane(mp);
    CP4_ASM(CP4_T(NODMA));
    CP4_ASM(CP4_T(LAST));
  }
#undef CP4_ASM
#undef CP4_ASM_SEL

  // ---------------- epilogue (that of conv_patch.hip with this tile's pixel map) ----------------
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int p = wm * TM + mi * 32 + l31;
    const int oy = y0 + (p >> 4), ox = x0 + (p & 15);
    const bool pix_ok = oy < a.H && ox < a.W;
    long long orow = ((long long)img * a.H + oy) * a.W + ox;
    if (a.subpix) orow = ((long long)img * 2 * a.H + 2 * oy + (z >> 1)) * (2 * a.W) + 2 * ox + (z & 1);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int nb = n0 + wn * TN + ni * 32;
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) half_swap(acc[ni][mi][8 * gp + j], acc[ni][mi][8 * gp + 4 + j], v[j], v[4 + j]);
        const int n = nb + 16 * gp + 8 * half;
        if (pix_ok && n < a.N) {
          if (a.bias) {
            const float4 b0 = *(const float4*)(a.bias + n), b1 = *(const float4*)(a.bias + n + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
          }
          if (a.rowvec) {
            const float* rv = a.rowvec + (long long)img * a.rv_stride + n;
            const float4 q0 = *(const float4*)rv, q1 = *(const float4*)(rv + 4);
            v[0] += q0.x; v[1] += q0.y; v[2] += q0.z; v[3] += q0.w;
            v[4] += q1.x; v[5] += q1.y; v[6] += q1.z; v[7] += q1.w;
          }
          if (a.res) {
            const uint4 r4 = *(const uint4*)(a.res + orow * a.ldr + n);
            v[0] += bflo(r4.x); v[1] += bfhi(r4.x); v[2] += bflo(r4.y); v[3] += bfhi(r4.y);
            v[4] += bflo(r4.z); v[5] += bfhi(r4.z); v[6] += bflo(r4.w); v[7] += bfhi(r4.w);
          }
          uint4 pk;
          pk.x = cvt_pk_bf16_f32(v[0], v[1]); pk.y = cvt_pk_bf16_f32(v[2], v[3]);
          pk.z = cvt_pk_bf16_f32(v[4], v[5]); pk.w = cvt_pk_bf16_f32(v[6], v[7]);
          *(uint4*)(a.out + orow * a.ldo + n) = pk;
        }
      }
    }
  }
}
