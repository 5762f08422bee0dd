#!/usr/bin/env python3
"""Generator of flash4w.inc: the whole key loop of flash attention at head width 64 as ONE hand-placed instruction stream
(flash4w.hip: 64 queries per wave, 64-key tiles, two waves per SIMD).

Why a generator: round 4's micro-benchmark (tools/ubench/coissue2.hip, profiles/r4_ubench_flash_like_waves.log) shows that on
gfx950 the softmax's VALU work and the MFMAs only overlap when they are interleaved instruction by instruction - three waves per
SIMD that alternate whole phases (what hipcc emits for flash_attn64_v25) reach 75 % of the interleaved stream's rate, and the
compiled kernel, with its `s_waitcnt lgkmcnt(0)` in front of every MFMA and `s_nop 10` behind every score tile, about half.
A single wave, though, is bound by its own issue (one instruction at a time: ~9 cycles per v_exp_f32 / v_cvt_pk_bf16_f32, ~5 per
v_add_f32, ~8 per MFMA: 1 600-1 680 cycles per tile against 1 024 of MFMA time, the first form of this file), so the stream is
kept to 192 VGPRs + 64 AGPRs and two workgroups share a CU.

The wave's 64 queries are two blocks q of 32; a key tile is two halves of 32 keys = four groups g of 16.  The scores live in
two register sets of 2 x 16: X = v[128:159] (half 0 of a tile), Y = v[160:191] (half 1) - physical registers, single ones are
VALU operands.  One iteration t = one tile = 32 MFMAs in eight groups SG0..SG7 of four:
    SG0-3 (half 0):  two QK^T MFMAs of tile t's half 1 into Y (q0: SG0 / SG1, q1: SG2 / SG3), two P V MFMAs of X's groups
    SG4-7 (half 1):  two QK^T MFMAs of tile t + 1's half 0 into X,                              two P V MFMAs of Y's groups
  * a QK^T chain's first k-step takes -reference as its C operand: p = exp2(s) needs no subtraction;
  * P(q, g) - eight probabilities per lane - is packed IN PLACE into the first four registers of its own eight scores, by the
    twenty VALU instructions (8 v_exp_f32, 8 v_add_f32 for the row sums, 4 v_cvt_pk_bf16_f32) placed five behind each MFMA of
    the group BEFORE the one whose P V MFMAs read it (a VALU result needs wait states before an MFMA may read it as an operand:
    nothing inserts them in an asm stream; the QK^T pair in front of every P V pair provides them);
  * the four K fragments of a half and the four V^T fragments of two groups stay in registers for both query blocks (half the
    LDS traffic per MFMA of a 32-query wave), reloaded by ds_read_b128 as soon as their last MFMA has issued, every wait a
    counted lgkmcnt;
  * ring slot t of four 16 KB slots holds what iteration t reads: K rows 64 t + 32 ... 64 t + 95 (tile t's half 1, tile t + 1's
    half 0) and V^T tile t; the LDS-DMA pieces of slot t + 3 go out behind the first MFMAs of SG1-SG4; one barrier per tile,
    behind SG6: slot t + 1 has landed for every wave and nobody reads slot t - 1 any more.

Stream = ENTRY, LOOP x cnt {FULL}, NODMA(vmcnt 4), NODMA(vmcnt 0), LAST        (cnt = nkt - 3 >= 1)
"""
import sys

X = {0: 128, 1: 144}      # score set of a tile's half 0: query block -> first register
Y = {0: 160, 1: 176}
MFMA = "v_mfma_f32_32x32x16_bf16"


def vt(b, n):
    return f"v[{b}:{b + n - 1}]"


def vgroup(R, la, lb):
    """exp2, row sums and packing of the eight scores in v[R : R + 7]; P ends up in v[R : R + 3].  A transcendental's result is
    never read by the next instruction (gfx940-family TRANS -> VALU hazard: one wait state, nothing inserts it here)."""
    r = [f"v{R + j}" for j in range(8)]
    return [
        f"v_exp_f32 {r[0]}, {r[0]}",
        f"v_exp_f32 {r[1]}, {r[1]}",
        f"v_add_f32 {la}, {la}, {r[0]}",
        f"v_exp_f32 {r[2]}, {r[2]}",
        f"v_add_f32 {lb}, {lb}, {r[1]}",
        f"v_exp_f32 {r[3]}, {r[3]}",
        f"v_add_f32 {la}, {la}, {r[2]}",
        f"v_cvt_pk_bf16_f32 {r[0]}, {r[0]}, {r[1]}",
        f"v_exp_f32 {r[4]}, {r[4]}",
        f"v_add_f32 {lb}, {lb}, {r[3]}",
        f"v_cvt_pk_bf16_f32 {r[1]}, {r[2]}, {r[3]}",
        f"v_exp_f32 {r[5]}, {r[5]}",
        f"v_add_f32 {la}, {la}, {r[4]}",
        f"v_exp_f32 {r[6]}, {r[6]}",
        f"v_add_f32 {lb}, {lb}, {r[5]}",
        f"v_exp_f32 {r[7]}, {r[7]}",
        f"v_cvt_pk_bf16_f32 {r[2]}, {r[4]}, {r[5]}",
        f"v_add_f32 {la}, {la}, {r[6]}",
        f"v_add_f32 {lb}, {lb}, {r[7]}",
        f"v_cvt_pk_bf16_f32 {r[3]}, {r[6]}, {r[7]}",
    ]


class Stream:
    def __init__(self, queue):
        self.out = []
        self.queue = list(queue)     # LDS reads in flight, oldest first (fragment buffer names)

    def op(self, text):
        self.out.append(text)

    def read(self, buf, ad, off):
        self.out.append(f"ds_read_b128 %[{buf}], %[ad{ad}] offset:{off}")   # (ad0..ad3: the 32x32x16 stream; adk0..3 / adv0,1: the 16x16x32 one)
        assert buf not in self.queue, buf
        self.queue.append(buf)
        assert len(self.queue) <= 15

    def need(self, buf):
        if buf in self.queue:
            p = self.queue.index(buf)
            self.out.append(f"s_waitcnt lgkmcnt({len(self.queue) - 1 - p})")
            self.queue = self.queue[p + 1:]

    def drain(self):
        if self.queue:
            self.out.append("s_waitcnt lgkmcnt(0)")
            self.queue = []


KOFF0, KOFF1, VOFF = 0, 4096, 8192     # inside a slot: K of this tile's half 1, K of the next tile's half 0, V^T


def entry_reads(st):
    """The fragments SG0 of an iteration starts from, out of the slot the address registers point at."""
    st.read("kf0", 0, KOFF0)
    st.read("kf1", 1, KOFF0)
    st.read("vf0", 0, VOFF)
    st.read("vf1", 0, VOFF + 4096)


def entry_reads2(st):
    st.read("kf2", 2, KOFF0)
    st.read("kf3", 3, KOFF0)
    st.read("vf2", 1, VOFF)
    st.read("vf3", 1, VOFF + 4096)


def block(kind, queue):
    """kind: 'full' (LDS-DMA of slot t + 3) | 'nodma4' | 'nodma0' (no DMA; tail waits vmcnt(4) / vmcnt(0)) | 'last'."""
    st = Stream(queue)
    last = kind == "last"
    dma = {}
    if kind == "full":
        dma = {1: ("vk0", "srk", "sok", 0), 2: ("vk1", "srk", "sok", 4096), 3: ("vv0", "srv", "sov", 8192), 4: ("vv1", "srv", "sov", 12288)}
    for i in range(8):
        half = i >> 2
        cur = X if half == 0 else Y          # scores being consumed
        nxt = Y if half == 0 else X          # scores being produced (this tile's half 1 / the next tile's half 0)
        q = (i >> 1) & 1                     # SG0,1 / SG4,5: query block 0;  SG2,3 / SG6,7: query block 1
        pair = i & 1                         # k-steps (2 pair, 2 pair + 1) of the QK^T chain; group 2 half + pair of P V
        qk, pv = [], []
        if not (last and half == 1):
            for ks in (2 * pair, 2 * pair + 1):
                D = vt(nxt[q], 16)
                C = f"%[ng{q}]" if ks == 0 else D
                qk.append((f"{MFMA} {D}, %[kf{ks}], %[q{q}{ks}], {C}", f"kf{ks}"))
        P = vt(cur[q] + 8 * pair, 4)
        vb = 2 * pair                        # V^T buffers: vf0, vf1 for a half's first group, vf2, vf3 for its second
        for d in (0, 1):
            pv.append((f"{MFMA} %[o{q}{d}], %[vf{vb + d}], {P}, %[o{q}{d}]", f"vf{vb + d}"))
        # QK, PV, QK, PV: the two k-steps of one score accumulator are not back to back, every P V MFMA has an MFMA and five
        # VALU instructions between the packing of its P and itself
        mf = [qk[0], pv[0], qk[1], pv[1]] if qk else pv
        # the VALU stream: P of the group whose P V MFMAs come in the NEXT group of MFMAs
        j = i + 1
        if j < 8:
            jh, jq, jp = j >> 2, (j >> 1) & 1, j & 1
            va = vgroup((X if jh == 0 else Y)[jq] + 8 * jp, f"%[l{jq}0]", f"%[l{jq}1]")
        elif not last:
            va = vgroup(X[0], "%[l00]", "%[l01]")       # (q0, first group) of the next tile
        else:
            va = []
        n = len(mf)
        per = -(-len(va) // n) if va else 0
        for k, (text, buf) in enumerate(mf):
            first = k == 0
            if first and i in dma:
                st.op(f"s_add_u32 m0, %[mb], {dma[i][3]}")
            if last and half == 1 and first:
                st.op("s_nop 4")      # (no QK^T pair in front of this P V pair: the wait states behind the VALU that packed its P)
            st.need(buf)
            st.op(text)
            if first and i in dma:
                vo, srd, so, _ = dma[i]
                st.op(f"buffer_load_dwordx4 %[{vo}], %[{srd}], %[{so}] offen lds")
                if i == 2:
                    st.op("s_add_u32 %[sok], %[sok], %[kst]")
                if i == 4:
                    st.op("s_add_u32 %[sov], %[sov], 128")
                    st.op("s_add_u32 %[mb], %[mb], 0x4000")
                    st.op("s_and_b32 %[mb], %[mb], 0xffff")
            for text2 in va[k * per:(k + 1) * per]:
                st.op(text2)
            qk_done = (not (last and half == 1)) and k == 2       # behind the group's second QK^T MFMA
            pv_done = k == n - 1
            # fragment reloads, as soon as the buffer's last MFMA has issued (q1's groups: SG2, SG3 / SG6, SG7)
            if i == 2 and qk_done and not last:
                st.read("kf0", 0, KOFF1)
                st.read("kf1", 1, KOFF1)
            if i == 3 and qk_done and not last:
                st.read("kf2", 2, KOFF1)
                st.read("kf3", 3, KOFF1)
            if i == 2 and pv_done:
                st.read("vf0", 2, VOFF)
                st.read("vf1", 2, VOFF + 4096)
            if i == 3 and pv_done:
                st.read("vf2", 3, VOFF)
                st.read("vf3", 3, VOFF + 4096)
        if i == 6 and not last:
            # slot t + 1 has landed for everybody, slot t - 1 is free; the address registers move on
            st.drain()
            st.op("s_waitcnt vmcnt(%d)" % {"full": 8, "nodma4": 4, "nodma0": 0}[kind])
            st.op("s_barrier")
            for jj in range(4):
                st.op(f"v_add_u32 %[ad{jj}], 0x4000, %[ad{jj}]")
                st.op(f"v_and_b32 %[ad{jj}], 0xffff, %[ad{jj}]")
            entry_reads(st)
        if i == 7 and not last:
            entry_reads2(st)
    if last:
        st.drain()
        st.op("s_nop 7")
        st.op("s_nop 7")
        st.op("s_nop 7")
    return st.out, st.queue



# ---------------------------------------------------------------------------------------------------------------------------
# The same loop on v_mfma_f32_16x16x32 (round 6, FA4W16_ASM; flash4w.hip, variant 27).  Why: the chip is power-limited under
# matrix load and this kernel's operands are on the CU (registers / LDS) - the regime in which a wave-tile step on 16x16x32
# MFMAs ran 14 % faster than on 32x32x16 (tools/ubench/mfma_shape_asm.hip, profiles/r6_mfma_shape.log).
#
# The wave's 64 queries are FOUR blocks q of 16; a half (32 keys) is ONE k-step of the P V product and two key blocks kb of the
# scores.  Score sets: X16 = v[128:159], Y16 = v[160:191], eight registers per query block: [kb 0: r0-r3 | kb 1: r0-r3] - lane
# (query l15, key group kq = lane >> 4), register r of block kb = the score of MFMA row 4 kq + r, whose key the K fragment's
# row map chooses (flash4w.hip: position 8 kq + 4 kb + r of the half in the permuted V^T order), so that the eight packed
# probabilities of a lane ARE its 8 consecutive k values of the P V MFMA's B operand.  One iteration = 72 MFMAs in eight groups
# SG0..SG7 of eight (ten); SG s of a half = (query pair qp = s >> 1, pair = s & 1):
#     four QK^T MFMAs:  queries 2 qp, 2 qp + 1  x  d-steps 0, 1  of key block kb = pair          (K fragments kf[2 pair + ds])
#     four P V MFMAs:   queries 2 qp, 2 qp + 1  x  d blocks 2 pair, 2 pair + 1                    (V^T fragments vf[2 pair + j])
#   so the fragments of `pair` are used in SG `pair` and SG 2 + `pair` of a half and reloaded behind the second use, as in the
#   32x32x16 stream; order QK, PV, QK, PV ...: the two d-steps of a score block are four MFMAs apart, a query block's first P V MFMA
#   has at least one MFMA between itself and the packing of its P;
#   VALU (12 instructions per query block: 8 v_exp_f32, 4 v_cvt_pk; Bresenham over the group's MFMAs): SG0 / SG1 of a half pack P of
#   its queries 2 / 3, SG2 / SG3 P of queries 0 / 1 of the NEXT half - whose scores SG0 / SG1 have just finished;
#   the `pair` = 0 groups carry two more MFMAs: the row sums of their two query blocks (below).
X16 = {q: 128 + 8 * q for q in range(4)}
Y16 = {q: 160 + 8 * q for q in range(4)}
MFMA16 = "v_mfma_f32_16x16x32_bf16"


# Row sums on the matrix pipe: one MFMA of P against ones per (query block, half) - every row of the 16 x 16 result is the sum over
# the half's 32 keys of the lane's query - instead of eight v_add_f32 per query block.  With the adds on the VALU this stream was a tie
# with the 32x32x16 one (2 936 cycles per wave-tile at 1.74-1.84 GHz against 2 408 at 1.6: a 16-cycle MFMA costs the wave the same ~8
# issue cycles as a 32-cycle one, and the kernel is issue-bound); without them 2 589 cycles at 1.8 GHz, the matrix pipe 89 % busy
# (72 MFMAs per wave-tile): 1 201 / 1 278 TFLOP/s (whole blocks / key-split) against 1 130 / 1 193 at E = 10, 9 216 tokens
# (profiles/r6_flash_mfma16.log).  The same trade LOST in the 32x32x16 stream (a third 32-cycle MFMA per pair, round 4).
def vgroup_noadd(R):
    """exp2 and packing only (the row sums are an MFMA against ones); a transcendental's result is not read by the next instruction"""
    r = [f"v{R + j}" for j in range(8)]
    return [f"v_exp_f32 {r[0]}, {r[0]}", f"v_exp_f32 {r[1]}, {r[1]}", f"v_exp_f32 {r[2]}, {r[2]}",
            f"v_cvt_pk_bf16_f32 {r[0]}, {r[0]}, {r[1]}", f"v_exp_f32 {r[3]}, {r[3]}", f"v_exp_f32 {r[4]}, {r[4]}",
            f"v_cvt_pk_bf16_f32 {r[1]}, {r[2]}, {r[3]}", f"v_exp_f32 {r[5]}, {r[5]}", f"v_exp_f32 {r[6]}, {r[6]}",
            f"v_exp_f32 {r[7]}, {r[7]}", f"v_cvt_pk_bf16_f32 {r[2]}, {r[4]}, {r[5]}", f"v_cvt_pk_bf16_f32 {r[3]}, {r[6]}, {r[7]}"]


def vg16(R, q):
    return vgroup_noadd(R)


def entry_reads16(st):
    st.read("kf0", "k0", KOFF0)
    st.read("kf1", "k1", KOFF0)
    st.read("vf0", "v0", VOFF)
    st.read("vf1", "v0", VOFF + 2048)


def entry_reads16b(st):
    st.read("kf2", "k2", KOFF0)
    st.read("kf3", "k3", KOFF0)
    st.read("vf2", "v0", VOFF + 4096)
    st.read("vf3", "v0", VOFF + 6144)


def block16(kind, queue):
    st = Stream(queue)
    last = kind == "last"
    dma = {}
    if kind == "full":
        dma = {1: ("vk0", "srk", "sok", 0), 2: ("vk1", "srk", "sok", 4096), 3: ("vv0", "srv", "sov", 8192), 4: ("vv1", "srv", "sov", 12288)}
    for i in range(8):
        half, s = i >> 2, i & 3
        qp, pair = s >> 1, s & 1
        cur = X16 if half == 0 else Y16
        nxt = Y16 if half == 0 else X16
        qs = (2 * qp, 2 * qp + 1)
        qk, pv = [], []
        if not (last and half == 1):
            for ds in (0, 1):
                for q in qs:
                    D = vt(nxt[q] + 4 * pair, 4)
                    C = f"%[ng{q}]" if ds == 0 else D
                    qk.append((f"{MFMA16} {D}, %[kf{2 * pair + ds}], %[q{q}{ds}], {C}", f"kf{2 * pair + ds}"))
        for q in qs:
            for j in (0, 1):
                db = 2 * pair + j
                pv.append((f"{MFMA16} %[o{q}{db}], %[vf{db}], {vt(cur[q], 4)}, %[o{q}{db}]", f"vf{db}"))
        mf = [x for pr in zip(qk, pv) for x in pr] if qk else pv
        if pair == 0:   # the row sums of the pair's two query blocks: P against ones (every row of the block = the sum)
            mf += [(f"{MFMA16} %[ls{q}], %[ones], {vt(cur[q], 4)}, %[ls{q}]", None) for q in qs]
        # the VALU stream of this group
        if s < 2:
            va = vg16(cur[2 + s], 2 + s)
        elif not (last and half == 1):
            va = vg16(nxt[s - 2], s - 2)
        else:
            va = []
        n = len(mf)
        n_qk_seen = 0
        for k, (text, buf) in enumerate(mf):
            first = k == 0
            if first and i in dma:
                st.op(f"s_add_u32 m0, %[mb], {dma[i][3]}")
            if last and half == 1 and first:
                st.op("s_nop 4")      # (no QK^T MFMA in front of this P V MFMA: the wait states behind the VALU that packed its P)
            if buf is not None:
                st.need(buf)
            st.op(text)
            if first and i in dma:
                vo, srd, so, _ = dma[i]
                st.op(f"buffer_load_dwordx4 %[{vo}], %[{srd}], %[{so}] offen lds")
                if i == 2:
                    st.op("s_add_u32 %[sok], %[sok], %[kst]")
                if i == 4:
                    st.op("s_add_u32 %[sov], %[sov], 128")
                    st.op("s_add_u32 %[mb], %[mb], 0x4000")
                    st.op("s_and_b32 %[mb], %[mb], 0xffff")
            for text2 in va[(len(va) * k) // n:(len(va) * (k + 1)) // n]:
                st.op(text2)
            is_qk = bool(qk) and k < 2 * len(qk) and k % 2 == 0
            if is_qk:
                n_qk_seen += 1
            qk_done = bool(qk) and is_qk and n_qk_seen == len(qk)     # behind the group's last QK^T MFMA
            pv_done = buf is not None and buf.startswith("vf") and (k == n - 1 or mf[k + 1][1] is None)
            # fragment reloads for the tile's second half, as soon as the buffer's last MFMA has issued (SG2: pair 0, SG3: pair 1)
            if i == 2 and qk_done and not last:
                st.read("kf0", "k0", KOFF1)
                st.read("kf1", "k1", KOFF1)
            if i == 3 and qk_done and not last:
                st.read("kf2", "k2", KOFF1)
                st.read("kf3", "k3", KOFF1)
            if i == 2 and pv_done:
                st.read("vf0", "v1", VOFF)
                st.read("vf1", "v1", VOFF + 2048)
            if i == 3 and pv_done:
                st.read("vf2", "v1", VOFF + 4096)
                st.read("vf3", "v1", VOFF + 6144)
        if i == 6 and not last:
            # slot t + 1 has landed for everybody, slot t - 1 is free; the address registers move on
            st.drain()
            st.op("s_waitcnt vmcnt(%d)" % {"full": 8, "nodma4": 4, "nodma0": 0}[kind])
            st.op("s_barrier")
            for jj in ("k0", "k1", "k2", "k3", "v0", "v1"):
                st.op(f"v_add_u32 %[ad{jj}], 0x4000, %[ad{jj}]")
                st.op(f"v_and_b32 %[ad{jj}], 0xffff, %[ad{jj}]")
            entry_reads16(st)
        if i == 7 and not last:
            entry_reads16b(st)
    if last:
        st.drain()
        st.op("s_nop 7")
        st.op("s_nop 7")
        st.op("s_nop 7")
    return st.out, st.queue


def stream16():
    lines = ["s_waitcnt lgkmcnt(0)"]
    lines += vg16(X16[0], 0) + vg16(X16[1], 1)
    st = Stream([])
    st.op("s_waitcnt vmcnt(8)")
    st.op("s_barrier")
    entry_reads16(st)
    entry_reads16b(st)
    lines += st.out
    q0 = st.queue
    full, qa = block16("full", q0)
    assert qa == q0, (q0, qa)
    lines += [".Lfa4w16_loop%=:"]
    lines += full
    lines += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 .Lfa4w16_loop%="]
    n4, q1 = block16("nodma4", q0)
    n0, q2 = block16("nodma0", q1)
    lb, q3 = block16("last", q2)
    assert q3 == []
    lines += n4 + n0 + lb
    return lines, full


def c_literal(ln):
    """One instruction as a C string literal; the operand-type mnemonics come from common.h (MG_MFMA32_ASM, MG_CVT_PK_ASM: bf16 in the
    product build, fp16 in the fp16 build) as adjacent literals."""
    for mnem, macro in (("v_mfma_f32_32x32x16_bf16", "MG_MFMA32_ASM"), ("v_mfma_f32_16x16x32_bf16", "MG_MFMA16_ASM"),
                        ("v_cvt_pk_bf16_f32", "MG_CVT_PK_ASM")):
        if ln.startswith(mnem + " "):
            return macro + ' "' + ln[len(mnem):]
    return '"' + ln


def main():
    lines = ["s_waitcnt lgkmcnt(0)"]
    lines += vgroup(X[0], "%[l00]", "%[l01]")
    st = Stream([])
    st.op("s_waitcnt vmcnt(8)")
    st.op("s_barrier")
    entry_reads(st)
    entry_reads2(st)
    lines += st.out
    q0 = st.queue
    full, qa = block("full", q0)
    assert qa == q0, (q0, qa)
    lines += [".Lfa4w_loop%=:"]
    lines += full
    lines += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 .Lfa4w_loop%="]
    n4, q1 = block("nodma4", q0)
    n0, q2 = block("nodma0", q1)
    lb, q3 = block("last", q2)
    assert q3 == []
    lines += n4 + n0 + lb
    n_mfma = sum(1 for x in full if x.startswith(MFMA))
    n_valu = sum(1 for x in full if x.startswith("v_") and not x.startswith(MFMA))
    n_lds = sum(1 for x in full if x.startswith("ds_"))
    with open(sys.argv[1] if len(sys.argv) > 1 else "flash4w.inc", "w") as f:
        f.write("// Generated by gen_fa4w.py - do not edit.  One FULL iteration: %d MFMA, %d VALU, %d ds_read_b128, %d lines.\n"
                % (n_mfma, n_valu, n_lds, len(full)))
        f.write("#define FA4W_ASM \\\n")
        for x in lines:
            f.write('  %s\\n\\t" \\\n' % c_literal(x))
        f.write('  ""\n')
        lines16, full16 = stream16()
        f.write("// The 16x16x32 form.  One FULL iteration: %d MFMA, %d VALU, %d ds_read_b128, %d lines.\n"
                % (sum(1 for x in full16 if x.startswith(MFMA16)), sum(1 for x in full16 if x.startswith("v_") and not x.startswith(MFMA16)),
                   sum(1 for x in full16 if x.startswith("ds_")), len(full16)))
        f.write("#define FA4W16_ASM \\\n")
        for x in lines16:
            f.write('  %s\\n\\t" \\\n' % c_literal(x))
        f.write('  ""\n')


if __name__ == "__main__":
    main()
