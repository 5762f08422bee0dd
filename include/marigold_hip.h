/*
 * marigold_hip.h - C ABI of libmarigold_hip.so, the MI355X (gfx950) engine behind the
 * Marigold inference hot path.
 *
 * The reference has no FFI: its hot path sits behind Python module attributes registered on
 * the pipeline (marigold/marigold_depth_pipeline.py:133-139) whose arithmetic lives in
 * diffusers/torch.  Each entry point below replaces one of those call sites (cited per op).
 * Plain pointers and sizes only: device pointers are raw HIP device addresses, `stream` is a
 * hipStream_t passed as void*.  All functions return 0 on success, non-zero on error
 * (message via mg_last_error()).  Activations are bf16 NHWC ([B][H][W][C] == [B*H*W][C]
 * token-major); latents and decoded maps at the pipeline boundary are fp32 NCHW like the
 * reference's tensors.
 *
 * Every kernel launch is described by one fixed-size `mg_op`; a sequence of them is a
 * *program* (mg_program_*) that the library replays with no host logic in between (and
 * optionally as a captured hipGraph).  The Python host mirrors the reference's
 * unet/vae/scheduler interface by building such programs (marigold_amd/engine.py, modules.py).
 */
#ifndef MARIGOLD_HIP_H
#define MARIGOLD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_ABI_VERSION 4

enum mg_op_kind {
  /* conv3x3 / conv1x1 / Linear / batched GEMM as ONE implicit-GEMM bf16 MFMA kernel.
   * Replaces torch conv2d / linear / matmul inside diffusers UNet2DConditionModel and
   * AutoencoderKL (reference call sites marigold_depth_pipeline.py:461-463, 491-492, 512-513).
   *  p[0] A bf16 [B][H][W][lda>=Cin]   p[1] Wt bf16 [N][ldw>=taps*Cin] (k = (ky*3+kx)*Cin+c)
   *  p[2] out                          p[3] bias f32[N] | NULL
   *  p[4] rowvec f32 [B][N] | NULL (time-embedding add)   p[5] residual bf16 [M][ldr] | NULL
   *  p[6] out2 (transposed section)    p[7] A1 bf16 [B][H][W][lda1] | NULL: second channel source - channels [C0, Cin)
   *  of every tap come from A1, [0, C0) from A (the UNet's skip concat; i[24] = C0, i[25] = lda1)
   *  p[8] ln_out f32 [M][N/32 + 1][2] bytes | NULL: the LayerNorm statistics of the tensor this launch produces (bf16
   *  epilogue, N % 32 == 0, f[1] = eps): every tile writes (sum, sum of squares) of its rows over each 32-column slot into
   *  [M][N/32][2], and the last column tile of a row block to finish reduces them to (mean, rstd) rows [M][2] stored
   *  BEHIND the slots (at ln_out + M * (N/32) * 8 bytes) - that address is the ln_in of the consuming layers.  Launches
   *  that write statistics share one ticket array: they must be stream-ordered with respect to each other.
   *  p[9] ln_in f32 [M][2] (mean, rstd) | NULL, p[10] ln_g f32 [N], p[11] ln_c f32 [N]: LayerNorm FOLDED into this Linear layer
   *  (diffusers BasicTransformerBlock: norm1 -> attn1.to_q/k/v, norm2 -> attn2.to_q, norm3 -> ff.net.0.proj): A holds the
   *  raw rows x, Wt = W * gamma, and out = rstd[m] * (acc - mean[m] * ln_g[n]) + ln_c[n] with ln_g[n] = sum_k Wt[n][k],
   *  ln_c[n] = sum_k beta[k] W[n][k] + bias[n]; (mean, rstd) come from the producer's ln_out.  No separate normalisation
   *  pass, no normalised tensor in HBM.
   *  i[0] B  i[1] H  i[2] W  i[3] Cin  i[4] Ho  i[5] Wo
   *  i[6] N  i[7] taps (1 | 9 | 4 = the sub-pixel form of nearest-2x + conv3x3: batch_z = 4 output parities, see
   *  MG_OP_CONV3X3)  i[8] stride  i[9] pad  i[10] Hu  i[11] Wu (virtual nearest-
   *  upsampled input size, 0 = none)  i[12] epilogue (MG_EPI_*)  i[13] ldo  i[14] trans_from
   *  (columns >= this go to out2 as [img][n-trans_from][ldt] transposed; -1 = none)
   *  i[15] batch_z  i[16] ldr  i[17] lda  i[18] ldt  i[19] tile variant (0 = auto; 20..53 force a tile,
   *  60..63 the 256x256 ping-pong schedule - used by the tuning sweep and the parity tests, see
   *  igemm2.hip::dispatch_tile)  i[20] ldw
   *  i[21] rowvec broadcast (1 = a single [N] row shared by every image)
   *  i[22], i[23] un-padded N, K for FLOP accounting (0 = as launched; ignored by the kernel)
   *  i[26] 1 = the transposed section stores its tokens in accumulator order inside groups of 16 (MG_OP_FLASH_ATTN64 i[7])
   *  i[29], i[30] low / high 32 bits of the device address of the caller's row-block tickets for the p[8] statistics
   *  hand-off (65536 zeroed uint32, one buffer per program / stream; 0 = the library's global buffer: single stream only)
   *  p[12] X0 bf16 [B][H][W][ldx0], p[13] X1 bf16 [B][H][W][ldx1] | NULL, i[32] = Cx, i[33] = Cx0, i[34] ldx0, i[35] ldx1: a 1x1 convolution
   *  of a SECOND tensor folded in as extra K (diffusers ResnetBlock2D: conv2(h) + conv_shortcut(x) in ONE launch - no shortcut
   *  launch, no residual round trip): K = taps * Cin + Cx, weight row n = [conv weights (taps * Cin) | shortcut weights (Cx)],
   *  the extra K tiles read pixel (y, x) of X0 (channels [0, Cx0)) and X1 ([Cx0, Cx): the UNet's skip concat); taps = 9, stride 1,
   *  pad 1 only; Cx, Cx0 multiples of 64; the bias is the two layers' sum
   *  i[31] split-K: 0 = automatic (few output tiles x long K: fp32 partials + a fixed-order reduce launch), n >= 1 = exactly n
   *  K ranges per tile (1 = none) - bf16 epilogue without row statistics / folded LayerNorm / batching only
   *  p[14] split-K workspace of the caller (64 MiB, 16-byte aligned) | NULL = the library's own, which programs on ONE stream may
   *  share (stream-ordered reuse); programs that run concurrently on several streams each bring their own
   *  l[0..3] z-strides (elements) of A, Wt, out, residual      f[0] scale on the accumulator */
  MG_OP_IGEMM = 1,
  /* GroupNorm, 3 launches (stats partials -> per-(b,c) scale/shift -> apply [+SiLU]).
   * Replaces torch group_norm + silu in every ResNet block / Transformer2D input norm.
   *  STATS:    p[0] x bf16 [B][HW][C]  p[1] partials f32 [B][slots][groups][2] ; i: B,HW,C,chunks, Ctot (0 = C), coff,
   *            groups, slot0, slots (0 = chunks) - x holds channels [coff, coff+C) of a Ctot-channel norm (the UNet's skip
   *            concat torch.cat([hidden, skip]) is normalised source by source, never materialised); block (chunk, b)
   *            writes slot slot0 + chunk.  With p[4] != NULL the image's last-arriving block also does FINALIZE's job
   *            (no finalize launch): p[2] gamma p[3] beta p[4] scale_shift [B][2][Ctot] p[5] uint32 [B] arrival counters
   *            (zero before the first use; left zero) ; f[0] eps.  p[6] x1 bf16 [B][HW][C1] | NULL, i[9] = C1: a second
   *            source (channels [coff+C, +C1)) in the same launch - blocks [chunks, 2 chunks) write slots slot0 + chunks + ...
   *  FINALIZE: p[0] partials p[1] gamma f32 p[2] beta f32 p[3] scale_shift f32 [B][2][C];
   *            i: B,C,groups,slots,HW ; f[0] eps
   *  APPLY:    p[0] x  p[1] scale_shift  p[2] out bf16 [B][HW][C]  p[3] x1 | NULL ; i: B,HW,C,silu, C0 - with x1 the
   *            output channels [0,C0) come from x ([B][HW][C0]) and [C0,C) from x1 ([B][HW][C-C0]) */
  MG_OP_GN_STATS = 2,
  MG_OP_GN_FINALIZE = 3,
  MG_OP_GN_APPLY = 4,
  /* GroupNorm as ONE launch (statistics + scale/shift [+ normalised output]): a workgroup owns whole groups of one image
   * (channel window lcm(C / groups, 4) <= 128) over all H x W rows - no partial table, no tickets, one read of the tensor.
   *  p[0] x0 bf16 [B][HW][C0]  p[1] x1 bf16 [B][HW][C - C0] | NULL (second channel source: the UNet's skip concat)
   *  p[2] out bf16 [B][HW][C] | NULL (statistics only)  p[3] gamma f32 [C]  p[4] beta f32 [C]  p[5] scale_shift f32 [B][2][C]
   *  i: B, HW, C, C0 (with x1), groups, silu ; f[0] eps.  The normalised form keeps the rows in registers: H x W x window
   *  <= 48 rows per thread of a 1024-thread workgroup (the UNet's 96^2 ... 12^2 levels at any width). */
  MG_OP_GN_SLAB = 9,
  /* Row-resident GEMM for the token-local Linear layers at K = 320 / 640 (the two widest transformer levels): out[M][N] =
   * epilogue(x[M][K] W[N][K]^T).  A wave keeps 32 whole rows of x in registers for the launch and the weights stream past
   * it in 64-column stages, pre-packed in MFMA fragment order with a per-stage trailer of per-channel constants
   * (marigold_amd/weights.py::pack_rowgemm; csrc/rowgemm.hip).  M % 32 == 0, N % 64 == 0, N >= 128; K = 640 runs 8 waves
   * per workgroup (i[10] = 0 | 8), K = 320 4 / 8 / 12.
   *  p[0] x bf16 [M][ldx]  p[1] packed weights  p[2] out bf16 [M][ldo]  p[3] residual bf16 [M][ldr] | NULL (may alias out)
   *  p[4] (mean, rstd) f32 [M][2] of the rows of x | NULL: LayerNorm folded (the packed trailer holds its g and c vectors)
   *  p[5] (mean, rstd) f32 [M][2] of the OUTPUT rows | NULL  p[6] V^T bf16 [B][N - i[9]][ldt] (QKV form)
   *  p[7] GroupNorm scale/shift f32 [B][2][K] | NULL: x is normalised while it is loaded (bf16(x * scale + shift))
   *  i: M, K, N, ldx, ldo, ldr, form (0 bias [+ residual] [+ row statistics], 1 GEGLU: stage = 32 value + 32 gate
   *  channels, out [M][N/2], 2 QKV: columns >= i[9] go to V^T in MG_OP_FLASH_ATTN64's permuted key order), i[7] tokens per
   *  image (forms with p[6] / p[7]; % 32 == 0), i[8] ldt, i[9] first V column (% 64 == 0), i[10] waves per workgroup
   *  (0 = 12; 4 / 8 / 12), i[12] column split (0 / 1 = none; n: the N / 64 stages are shared out over n workgroups per row
   *  block - few rows, many columns; not with p[5]) ; f[0] LayerNorm eps of p[5].
   *  form 3: the collapsed 2-token cross-attention (as MG_EPI_XATTN2) in place on the residual stream: N = 64 score
   *  columns, i[11] = 2 x heads of them live, f[1] softmax scale; p[1] = weights.pack_rowgemm_xattn (scores stage + VO^T
   *  fragments + bias), p[4] required, out[M][K] = P VO^T + bias + x, p[5] its row statistics; out may alias x.  K = 640 /
   *  1280 (the deeper levels): the K-split kernel - 32-row workgroups whose four waves split K and the output channels,
   *  p[1] = weights.pack_rowgemm_xattn_ksplit.  p[8] (tuning only): per-wave phase cycle stamps | NULL.
   *  form 1 with p[9] (round 6; K = 320, no column split): the collapsed cross-attention as the PROLOGUE of the GEGLU launch
   *  (BasicTransformerBlock: x += attn2(norm2(x)); ff(norm3(x))) - p[9] = a weights.pack_rowgemm_xattn image, p[4] = (mean, rstd) of
   *  the rows AS LOADED (norm2's), i[11] = 2 x heads, f[1] softmax scale; the rows are updated in registers, written once to
   *  p[10] bf16 [M][ldx] (may alias x: ff.out's residual) and the GEGLU projection's folded LayerNorm (norm3, f[0] eps) takes its
   *  statistics from the wave's own sums.  Bit-identical to the form-3 launch followed by the plain form-1 launch. */
  MG_OP_ROWGEMM = 10,
  /* Self-attention core, head dim 64, bf16 MFMA flash attention with LDS-staged K / V^T
   * tiles (replaces diffusers Attention / SDPA / xformers, run.py:217-220).
   *  p[0] Q bf16 (row stride ldq)  p[1] K (row stride ldq)  p[2] Vt bf16 [B][heads*64][ldvt]
   *  p[3] O bf16 (row stride ldo); i: B, heads, Ntok, ldq, ldo, ldvt, variant (0 = current
   *  kernel, 1 = generation-1 kernel kept for A/B runs), i[7] 1 = Vt's keys are in the order [0-3, 8-11, 4-7, 12-15]
   *  inside every group of 16 (as written by MG_OP_IGEMM i[26]; Ntok % 16 == 0) ; p[4] tuning only: cycle stamps | NULL ;
   *  l[0] q batch stride l[1] k batch stride l[2] vt batch stride l[3] o batch stride;
   *  f[0] softmax scale.
   *  With i[7] = 1 and Ntok % 256 == 0 (>= 256) the current kernel is the hand-placed one (flash4w.hip; variant 26 forces
   *  its 32x32x16-MFMA stream, 27 the 16x16x32 one with the row sums on the matrix pipe - chosen by itself at >= 4 096 tokens
   *  from two blocks of 256 queries per CU): softmax against a fixed per-query reference, exact, with an in-kernel running-maximum fallback for rows whose sums
   *  reach f[1] (0 = 2^100; tests force the fallback with a tiny value).  Optional p[5]: workspace (16-byte aligned, i[8] KB,
   *  ZEROED once by the caller, then owned by this op's launches on ONE stream - tickets return to zero): the blocks of 256
   *  queries left over beyond a multiple of the CU count are then split along the keys over the chip (bit-reproducible);
   *  4 KB + 69 632 bytes x 4 x (blocks % CUs) suffice; i[9]: 0 = split when it pays, 1 = always, 2 = never. */
  MG_OP_FLASH_ATTN64 = 6,
  /* Self-attention core of ONE head of width 512 (the mid-block attention of AutoencoderKL: diffusers Attention in
   * UNetMidBlock2D, marigold_depth_pipeline.py:491-492, 512-513), flash form: the scores stay in registers.
   *  p[0] Q bf16 (row stride ldq)  p[1] K (row stride ldq)  p[2] Vt bf16 [B][512][ldvt] (natural key order, ldvt >= Ntok rounded
   *  up to 32, pad columns zero)  p[3] O bf16 (row stride ldo); i: B, Ntok, ldq, ldo, ldvt ;
   *  l[0] q batch stride l[1] k batch stride l[2] vt batch stride l[3] o batch stride ; f[0] softmax scale */
  MG_OP_FLASH_ATTN512 = 11,
  /* Row softmax fp32 -> bf16 (single-head attention of a width other than 512, materialised scores; the collapsed 2-token
   * cross-attention of marigold_depth_pipeline.py:381-394, 438-442 needs no softmax op: scores = LN(x) Wqk^T with
   * Wqk[(h,j)] = Wq_h^T k_{j,h}, p = softmax over the key pair, out = p VO + bias + x with VO[(h,j)] = Wo[:,h] v_{j,h} run as
   * MG_EPI_XATTN2 / MG_OP_ROWGEMM form 3).
   *  p[0] S f32 [R][lds] p[1] P bf16 [R][ldp] ; i: R, ncols, lds, ldp (pad cols zeroed) */
  MG_OP_SOFTMAX_ROWS = 7,
  /* Scheduler update (DDIM / LCM, diffusers *.step at marigold_depth_pipeline.py:466-468):
   * out = f[0]*x + f[1]*model_out + f[2]*noise.  p[0] x f32 p[1] model_out f32
   * p[2] noise f32 | NULL  p[3] out f32 ; l[0] n elements */
  MG_OP_SCHED_STEP = 12,
  /* Small-M dense layer in fp32 (time-embedding MLP and per-ResNet projections):
   * out[m][n] = act_out(sum_k act_in(in[m][k]) * W[n][k] + b[n]).
   *  p[0] in f32 [M][K] p[1] W f32 [N][K] p[2] b f32|NULL p[3] out f32 [M][ldo];
   *  i: M,N,K,act_in,act_out(0 none,1 silu),ldo */
  MG_OP_LINEAR_SMALL_M = 13,
  /* 1x1 conv on fp32 NCHW latents with input scale (post_quant_conv after /0.18215,
   * marigold_depth_pipeline.py:510-512).  p[0] in f32 [B][Ci][HW] p[1] W f32 [Co][Ci]
   * p[2] b f32 p[3] out f32 [B][Co][HW] ; i: B,Ci,Co,HW ; f[0] input scale */
  MG_OP_LATENT_1X1 = 14,
  /* Pointwise tail of a small-Cout convolution computed by MG_OP_IGEMM into a padded fp32 buffer:
   * out NCHW = post(in[m][0..Cout) * f[0]): MG_POST_DEPTH = mean over channels, clip, (x+1)/2 (marigold_depth_
   * pipeline.py:515,473-475); MG_POST_NORMALS = clip, L2 normalise (marigold_normals_pipeline.py:438-440); MG_POST_UNIT;
   * MG_POST_SCHED = the DDIM / LCM update of MG_OP_SCHED_STEP applied to conv_out's result in place of storing it:
   * out <- f[1]*out + f[2]*in + f[3]*noise (out = the latent x_t, NCHW; marigold_depth_pipeline.py:466-468).
   *  p[0] in f32 [B*HW][ldi]  p[1] out f32 NCHW  p[2] noise f32 NCHW | NULL (MG_POST_SCHED) ;
   *  i: B, HW, Cout, ldi, post ; f[0] scale, f[1..3] cx, cm, cn */
  MG_OP_POST_NCHW = 15,
  /* im2col of a 3x3 / pad 1 neighbourhood for the <= 8-channel convolutions at the latent / image
   * boundary (conv_in of the UNet incl. the torch.cat of marigold_depth_pipeline.py:456-458, of the
   * VAE encoder and decoder): fp32 NCHW (two sources) -> bf16 [B*H*W][Kp], k = tap*(C0+C1) + c,
   * columns >= 9*(C0+C1) zero.  The convolution itself is then an MG_OP_IGEMM with K = Kp.
   *  p[0] src0 f32 [B|1][C0][H][W]  p[1] src1 f32 [B][C1][H][W] | NULL  p[2] out bf16 [B*H*W][Kp];
   *  i: B,H,W,C0,C1,Kp, src0_broadcast */
  MG_OP_IM2COL_SMALL = 16,
  /* Patch-resident conv3x3 (stride 1, pad 1) with the ResNet block's GroupNorm + SiLU fused into the operand staging
   * (diffusers ResnetBlock2D: norm1 -> silu -> conv1, norm2 -> silu -> conv2), the UNet's skip concat folded into the
   * channel loop (torch.cat([hidden, skip]) in the up blocks) and Upsample2D's nearest-2x + conv in sub-pixel form.
   *  p[0] A0 bf16 [B][H][W][lda0 >= C0]  p[1] Wt bf16 [N][ldw >= 9*Cin], k = (ky*3+kx)*Cin + c, Cin = C0 + C1
   *       (sub-pixel mode: [4][N][4*Cin], parity z = 2a+b, k = (ty*2+tx)*Cin + c - weights.py::pack_conv3x3_subpix)
   *  p[2] out bf16 [B][H][W][ldo] (sub-pixel: [B][2H][2W][ldo])  p[3] bias f32 [N] | NULL
   *  p[4] rowvec f32 [B][N] | NULL  p[5] residual bf16 (out's shape, row stride ldr) | NULL
   *  p[6] A1 bf16 [B][H][W][lda1 >= C1] | NULL (second channel source)
   *  p[7] scale_shift f32 [B][2][Cin] | NULL: input = silu?(x * scale + shift) for in-image pixels (MG_OP_GN_FINALIZE's
   *       output; zero padding stays zero)
   *  i[0] B  i[1] H  i[2] W  i[3] C0  i[4] C1  i[5] N  i[6] sub-pixel 2x mode  i[7] silu  i[8] lda0  i[9] lda1
   *  i[10] ldo  i[11] ldr  i[12] ldw  i[13] rowvec broadcast  i[14] tile variant (0 = auto)
   *  l[0] parity stride of Wt in elements (sub-pixel mode)
   *  p[8] (optional) f32 [B][i[16]][N / i[15]][2]: (sum, sum of squares) of every group of i[15] (4 | 8 | 16 | 32) output
   *  channels over each tile's pixels, of the values as stored - the partial table MG_OP_GN_FINALIZE reduces (slots = i[16] =
   *  mg_conv3x3_gn_slots(op), HW = H W or 4 H W): the next GroupNorm's statistics without a pass over the tensor. */
  MG_OP_CONV3X3 = 17,
  /* The output heads: GroupNorm apply [+ SiLU] + conv3x3 (pad 1) to <= 4 channels in one launch (conv_norm_out -> conv_act ->
   * conv_out of the UNet and of the VAE decoder - the tail of the modules the reference calls at marigold_depth_pipeline.py:461-463
   * and :498-516; csrc/head_conv.hip: the raw input patch of a pixel tile is normalised on its
   * way into LDS, the taps are packed bf16 dot products - no MFMA work at <= 4 output channels, one HBM read of the input).
   *  p[0] x bf16 [B][H][W][C]  p[1] scale_shift f32 [B][2][C] | NULL  p[2] Wt bf16 [>= Cout][9 C], k = (ky*3+kx)*C + c
   *  p[3] bias f32 | NULL  p[4] out f32 [B H W][ldo] (columns [0, Cout): what MG_OP_POST_NCHW reads)
   *  i[0] B  i[1] H  i[2] W  i[3] C (% 32 == 0)  i[4] Cout (1..4)  i[5] ldo (0 = Cout)  i[6] silu */
  MG_OP_CONV3X3_HEAD = 18,
  /* Test-time ensembling (marigold/util/ensemble.py).
   * DEPTH_STATS : one pass over [E][HW]: per-member min,max,mean and the centred E x E
   *               second-moment matrix (closed form of the pairwise-RMSE cost, :138-145).
   *   p[0] d f32 [E][HW] p[1] blocks f64 scratch p[2] out f64 [3E + E*E] ; i: E ; l[0] HW
   * DEPTH_MEDIAN: aligned = s*d+t; lower-middle median over E (+MAD); block min/max.
   *   p[0] d f32 [E][HW] p[1] st f32 [s[E], t[E]] | NULL (no alignment) p[2] med f32 [HW]|NULL
   *   p[3] mad f32 [HW]|NULL  p[4] out f32 [2+2E] = min, max of the prediction and the raw member
   *   values d[.][argmin px], d[.][argmax px] (exact sub-gradient of the regulariser on the
   *   host)  p[5] scratch (>= 12288 B) ; i: E, reduction(0 median,1 mean), has_shift ; l[0] HW
   * (DEPTH_STATS scratch: >= nblk*E*(E+3) doubles, nblk = min(ceil(HW / 256), E > 256 ? 32 : 128) - 128*E*(E+3) always suffices.  Any E >= 1: <= 32 members are selected in registers, <= 128 in LDS,
   * larger ensembles by a bitwise selection over the members in memory - the reference has no limit, ensemble.py:39-49)
   * DEPTH_NORM  : out = (med - lo)/range ; unc /= range.  p[0] med p[1] mad|NULL p[2] minmax
   *   ; i[0] shift_invariant ; l[0] HW
   * NORMALS     : p[0] n f32 [E][3][HW] p[1] out f32 [3][HW] p[2] unc f32 [HW]|NULL ;
   *   i: E, reduction(0 closest,1 mean) ; l[0] HW */
  MG_OP_ENS_DEPTH_STATS = 20,
  MG_OP_ENS_DEPTH_MEDIAN = 21,
  MG_OP_ENS_DEPTH_NORM = 22,
  MG_OP_ENS_NORMALS = 23,
  /* Image resampling either side of the path (marigold/util/image_util.py:90-120, marigold_depth_
   * pipeline.py:306-312, ensemble.py:158-161): torchvision resize(..., antialias=True) semantics.
   *  p[0] src  p[1] dst  p[2] f32 temporary [planes][Hin][Wout] (needed when both sizes change) ;
   *  i: planes (= B*C), Hin, Win, Hout, Wout, mode (0 bilinear, 1 bicubic, 2 nearest-exact),
   *  dtype (1: uint8 in/out - computed in float, rounded half-to-even; 0: fp32) */
  MG_OP_RESIZE = 24,
  /* Colour-mapped depth image (marigold/util/image_util.py:38-76 colorize_depth_maps followed by the pipeline's
   * (x * 255).astype(uint8), marigold_depth_pipeline.py:318-327): out[px] = LUT[min(int(clip((d-f[0])/(f[1]-f[0]),0,1)*256),255)].
   *  p[0] depth f32 [n]  p[1] LUT uint8 [256][3] (matplotlib's table)  p[2] out uint8 [n][3] (HWC) ; l[0] n ;
   *  f[0] min_depth f[1] max_depth */
  MG_OP_COLORIZE = 25,
  MG_OP_MEMSET = 30, /* p[0] dst ; i[0] byte value ; l[0] bytes */
  MG_OP_COPY = 31    /* p[0] src p[1] dst ; l[0] bytes (device to device) */
};

enum { MG_EPI_BF16 = 0, MG_EPI_GEGLU = 1, MG_EPI_F32 = 2,
       MG_EPI_SOFTMAX2 = 3 /* bf16 out = softmax over column pairs (2h, 2h+1) of f[2] * acc; i[27] = real columns, the rest -> 0:
                              the collapsed 2-token cross-attention's probabilities straight from the scores GEMM */,
       MG_EPI_XATTN2 = 4   /* the whole collapsed cross-attention in one launch (diffusers Attention with a 2-token context,
                              BasicTransformerBlock.attn2): N = 64 score columns as in MG_EPI_SOFTMAX2; the probabilities stay in
                              registers as the operand of a second MFMA stage against p[6] = W2 bf16 [i[28]][64] (the context's
                              values pushed through to_out), out[M][i[28]] = P W2^T + bias (p[3], of the second stage) + residual
                              (p[5]); p[8] = f32 [M][2] (mean, rstd) of the OUTPUT rows (one wave owns whole rows: no slots, no
                              ticket).  out may alias A and the residual (a row block belongs to one workgroup). */ };
enum { MG_POST_NONE = 0, MG_POST_DEPTH = 1, MG_POST_NORMALS = 2, MG_POST_UNIT = 3 /* IID: clip, (x+1)/2 */,
       MG_POST_SCHED = 4 /* scheduler update in place of the store, see MG_OP_POST_NCHW */ };

typedef struct mg_op {
  int32_t kind;
  int32_t i[40];
  float f[8];
  void* p[16];
  int64_t l[4];
} mg_op;

typedef struct mg_program mg_program;

/* Library / device */
int mg_abi_version(void);
/* The 16-bit operand type every "bf16" buffer of this build holds: 0 = bf16 (libmarigold_hip.so), 1 = IEEE fp16 (libmarigold_hip_f16.so,
 * the same sources built with OPERAND_F16=1 - the reference's `--fp16` arithmetic, script/depth/run.py:203-211).  Same ABI, same ops. */
int mg_operand_bits(void);
const char* mg_last_error(void);
int mg_init(int device);                 /* idempotent; allocates the zero page */
/* GEGLU weight-row interleave the host must pack ff.net.0.proj with (32: 16 u rows, then their 16 gate rows). */
int mg_geglu_interleave(void);
int mg_device_info(int* cu_count, int* lds_bytes, int64_t* hbm_bytes, char* arch, int arch_len);

/* One launch (also the body of mg_program_run) */
int mg_launch(const mg_op* op, void* stream);

/* Programs: replace the per-step Python loop of single_infer
 * (marigold_depth_pipeline.py:455-468) and the module forwards it calls. */
mg_program* mg_program_create(const mg_op* ops, int n_ops);
int mg_program_num_ops(const mg_program* prog);
int mg_program_run(mg_program* prog, void* stream);
/* Check every op of the program against its kernel's shape / alignment contract WITHOUT touching
 * the device (no GPU needed): 0 = launchable, else the first violation in mg_last_error(). */
int mg_program_validate(mg_program* prog);
int mg_program_run_range(mg_program* prog, int first, int count, void* stream);
/* Capture the program into a hipGraph on `stream` and replay that on later runs. */
int mg_program_capture(mg_program* prog, void* stream);
/* Time every op with HIP events on `stream` (ms per op written to `ms`, length n_ops). */
int mg_program_profile(mg_program* prog, void* stream, float* ms);
void mg_program_destroy(mg_program* prog);

/* Module-level entry points over a MODEL IMAGE: the pipeline's three native programs for one problem shape (image size,
 * members per call, scheduler steps), their kernel-ready weights and a memory plan, written once by
 * marigold_amd/image.py::export_model_image and run from any host language without Python.  They stand where single_infer's
 * calls stand (marigold/marigold_depth_pipeline.py:396-477): encode_rgb (:479-496), the T-step unet + scheduler.step loop
 * (:455-468), decode_depth / decode_normals (:498-516, :473-475).  All device pointers are the caller's (fp32, NCHW, contiguous);
 * the model owns its weights and workspace (one device allocation, mg_model_device_bytes).  One stream at a time per model.
 *  mg_model_load(path, device): device >= 0 binds the library to that GPU (mg_init) and uploads; device < 0 = host-only: the
 *  image is parsed and relocated against fake addresses so that mg_model_validate can check every op's contract without a GPU.
 *  mg_model_info: cfg16 = B, H, W, latent h, latent w, steps, prediction channels, MG_POST_*, step-noise tensors, sizeof(mg_op),
 *  modalities, decoded H, decoded W.
 *  mg_model_vae_encode: rgb [1,3,H,W] in [-1,1] -> latent [1,4,h,w] (x 0.18215, posterior mean).
 *  mg_model_denoise: rgb_latent [1,4,h,w], x [B,C,h,w] in / out (the initial noise -> the denoised latent), step_noise
 *  [n][B,C,h,w] for the LCM scheduler's n noisy steps (NULL for DDIM).
 *  mg_model_vae_decode: latent [B*modalities,4,h,w] -> pred [B, channels, Hout, Wout] with the pipeline's pointwise tail. */
typedef struct mg_model mg_model;
mg_model* mg_model_load(const char* path, int device);
void mg_model_destroy(mg_model* m);
int mg_model_info(const mg_model* m, int* cfg16);
long long mg_model_device_bytes(const mg_model* m);
int mg_model_validate(mg_model* m);
int mg_model_vae_encode(mg_model* m, const float* rgb, float* latent, void* stream);
int mg_model_denoise(mg_model* m, const float* rgb_latent, float* x, const float* step_noise, void* stream);
int mg_model_vae_decode(mg_model* m, const float* latent, float* pred, void* stream);

/* ensemble_depth(depth[E,1,H,W], scale_invariant, shift_invariant, output_uncertainty, reduction, regularizer_strength, max_iter,
 * tol, max_res) of marigold/util/ensemble.py:39-196 as ONE call on device pointers: member statistics, init_param, the native
 * scipy-BFGS alignment (mg_ens_align_minimize), align -> median (+ MAD) | mean (+ std) -> min / max normalisation.  preds fp32
 * [E][H*W]; depth_out [H*W]; unc_out [H*W] | NULL; reduction 0 median / 1 mean; max_res <= 0 = no down-sampling for the alignment;
 * info4 (optional) = achieved cost, cost evaluations, BFGS iterations, scipy's status (0 converged, 1 maxiter, 2 precision loss,
 * 3 NaN -> the alignment falls back to its starting point).  Same results as marigold_amd.ensemble.ensemble_depth, bit for bit
 * (tests/test_gpu_pipeline.py).  The reference's ValueErrors come back as error messages with the reference's texts.
 * Synchronises the stream. */
int mg_ensemble_depth(const float* preds, int E, int H, int W, int scale_invariant, int shift_invariant, int reduction,
                      double regularizer_strength, int max_iter, double tol, int max_res, float* depth_out, float* unc_out,
                      double* info4, void* stream);

/* Named wrappers - what a binding for the reference's seams would call directly. */
int mg_conv2d_igemm(const mg_op* conv_desc, void* stream);   /* kind must be MG_OP_IGEMM */
/* Host-only test hook (no device work): how MG_OP_FLASH_ATTN64's hand-placed kernel would share out B x heads sequences of Ntok
 * tokens over a chip of n_cu CUs given a workspace of ws_bytes and the split mode (op i[9]): out[0] whole blocks of 256 queries,
 * out[1] blocks split along the keys, out[2] workgroups over the split blocks; bounds[0 .. out[2]] (or NULL): the workgroups'
 * piece boundaries in 64-key tiles over the concatenated split blocks. */
int mg_flash4w_plan_test(int B, int heads, int Ntok, int n_cu, long long ws_bytes, int split, int* out, unsigned* bounds);
int mg_conv3x3(const mg_op* conv_desc, void* stream);        /* kind must be MG_OP_CONV3X3 (ResnetBlock2D norm+silu+conv) */
/* Slots per image of the partial table this MG_OP_CONV3X3 can fill with the GroupNorm statistics of its OUTPUT (p[8], see the
 * op), or 0 when the tile variant it runs on does not produce them (then leave p[8] NULL and use MG_OP_GN_STATS). */
int mg_conv3x3_gn_slots(const mg_op* conv_desc);
int mg_sched_step(const float* x, const float* model_out, const float* noise, float* out,
                  int64_t n, float cx, float cm, float cn, void* stream);
int mg_ensemble_normals(const float* normals, float* out, float* unc, int E, int64_t hw,
                        int reduction, void* stream);

/* Host arithmetic of ensemble_depth's alignment objective (marigold/util/ensemble.py:129-152, as the closed form of
 * marigold_amd/ensemble.py): pairwise-RMSE cost of the aligned members and its gradient w.r.t. scales s[E] / shifts t[E],
 * from the per-member means mean[E] and the centred second-moment matrix C[E*E] gathered by MG_OP_ENS_DEPTH_STATS.
 * No device work; fp64; summation order = numpy's (bit-identical to the numpy form it replaces). */
int mg_ens_align_cost_grad(int E, const double* s, const double* t, const double* mean, const double* C,
                           double* cost, double* gs, double* gt);
/* scipy.optimize.minimize(fn, x, jac=True, method="BFGS", tol=gtol, options={"maxiter": maxiter}) restated natively
 * (scipy 1.15: _minimize_bfgs, DCSRCH line search with the Wolfe-2 fall-back, one objective evaluation per distinct point;
 * csrc/bfgs.hip).  fn(user, n, x, &f, g) returns 0.  x in / out; status = scipy's warnflag (0 converged, 1 maxiter,
 * 2 precision loss, 3 NaN).  Any n. */
int mg_bfgs_minimize(int (*fn)(void* user, int n, const double* x, double* f, double* g), void* user, int n, double* x,
                     double gtol, int maxiter, double* fval, int* nit, int* nfev, int* status);
/* The whole alignment of ensemble_depth (marigold/util/ensemble.py:154-173: compute_param + scipy BFGS) as one call: the
 * objective of mg_ens_align_cost_grad + the regulariser from one device pass per evaluation - reg_op, an
 * MG_OP_ENS_DEPTH_MEDIAN op whose scale / shift input st_host [2E] and (min, max, member values) output mm_host [2 + 2E]
 * live in host-mapped memory - times the forward-difference survival factor of the reference's fp32 parameter cast.
 * affine: scale + shift (n = 2E) or scale only; reduction 0 median / 1 mean; lam = regulariser strength. */
int mg_ens_align_minimize(const mg_op* reg_op, void* stream, int E, int affine, int reduction, double lam, const double* mean,
                          const double* C, float* st_host, const float* mm_host, double* x, double gtol, int maxiter,
                          double* fval, int* nit, int* nfev, int* status);

/* Shader clock under matrix-core load, for bench.py's calibration block (the sysfs sensors do not answer on every box): one
 * workgroup per CU (one wave per SIMD) runs a fixed chain of v_mfma_f32_32x32x16_bf16 on random (zero_operands = 0) or zero
 * operands for ~2 ms and times it with s_memtime (shader cycles) against s_memrealtime (100 MHz).  mhz = mean over the
 * workgroups; tflops = the chain's rate (the clock-limited MFMA roof of this box on this data).  Synchronises the stream. */
int mg_clock_probe(void* stream, int zero_operands, double* mhz, double* tflops);

/* Tuning only (MARIGOLD_TUNING=1 MARIGOLD_IGEMM_STAMPS=1): the per-workgroup phase stamps of the last forced-tile MG_OP_IGEMM
 * launch (8 x uint64 of the 100 MHz s_memrealtime per workgroup, in the split-K workspace) -> host.  tools/igemm_phases.py. */
int mg_debug_read_workspace(void* host_dst, long long bytes);

/* HIP-event timing helpers for bench.py (the kernels run on the caller's stream). */
void* mg_event_create(void);
int mg_event_record(void* ev, void* stream);
int mg_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on stop */
void mg_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* MARIGOLD_HIP_H */
