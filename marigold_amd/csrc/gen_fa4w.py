#!/usr/bin/env python3
"""Generator of flash4w.inc: the whole key loop of flash attention at head width 64 as ONE hand-placed instruction stream
(flash4w.hip: one wave per SIMD, 64 queries per wave, 64-key tiles).

Why a generator: round 4's micro-benchmark (tools/ubench/coissue2.hip, profiles/r4_ubench_flash_like_waves.log) shows that on
gfx950 the softmax's VALU work and the MFMAs only overlap when they are interleaved instruction by instruction - three waves per
SIMD that alternate whole phases (what hipcc emits for flash_attn64_v25) reach 75 % of the interleaved stream's rate, and the
compiled kernel, with its `s_waitcnt lgkmcnt(0)` in front of every MFMA and `s_nop 10` behind every score tile, about half.

One iteration t of the stream (one 64-key tile; the wave's 64 queries are two blocks q of 32) is 40 MFMAs in eight groups
SG0..SG7:
    SG i:  QK(kb, ks, q0)  QK(kb, ks, q1)  PV(g, d0, q)  PV(g, d1, q)       g = i >> 1, q = i & 1, kb = i >> 2, ks = i & 3
  * QK(kb, ks, q) accumulates the scores of tile t + 1 (key block kb, 16 channels ks) into the NEXT score registers; its first
    k-step takes -reference as its C operand, so that p = exp2(s) needs no subtraction;
  * PV(g, d, q) multiplies V^T (16 keys g, 32 channels d) of tile t by P(q, g), eight probabilities per lane packed IN PLACE
    into the first four registers of their own eight scores;
  * a fifth MFMA per group, ones x P(q, g), accumulates the row sums (every element of its accumulator = the query's sum);
  * behind each MFMA come two or three of the twelve VALU instructions (8 v_exp_f32, 4 v_cvt_pk_bf16_f32) that turn the
    eight scores of group i + 1 into P(i + 1); SG7 does group 0 of tile t + 1.
  * K / V^T fragments are read (ds_read_b128) three to seven MFMAs ahead into two K and four V buffers, every wait is a counted
    lgkmcnt; the LDS-DMA pieces of ring slot t + 3 (K tile t + 3, V^T tile t + 2) go out behind MFMAs 8 / 12 / 16 / 20;
  * one barrier per tile, behind MFMA 27: slot t + 2 has landed for every wave and nobody reads slot t - 1 any more; the
    fragments of the next iteration's first MFMAs are requested right behind it, under MFMAs 28-31.
The score registers ping-pong between v[128:191] and v[192:255] (physical: single registers of the score tuples are VALU
operands), so the stream is emitted for both parities; the loop body is two iterations.

Stream = ENTRY, LOOP x cnt {FULL_A, FULL_B}, FULL_A, VONLY_B, NODMA_A, LAST_B   (nkt even, >= 4; cnt = (nkt - 4) / 2)
"""
import sys

SA = {(0, 0): 128, (0, 1): 144, (1, 0): 160, (1, 1): 176}
SB = {(0, 0): 192, (0, 1): 208, (1, 0): 224, (1, 1): 240}
MFMA = "v_mfma_f32_32x32x16_bf16"


def vt(b, n):
    return f"v[{b}:{b + n - 1}]"


def vgroup(R):
    """exp2 and packing of the eight scores in v[R : R + 7]; P ends up in v[R : R + 3] (the row sums come off the matrix pipe:
    a P x ones MFMA per group - in one wave a v_add_f32 costs 5 issue cycles, 64 of them a fifth of the iteration).  A
    transcendental's result is never read by the next instruction (gfx940-family TRANS -> VALU hazard: one wait state,
    nothing inserts it here)."""
    r = [f"v{R + j}" for j in range(8)]
    return [
        f"v_exp_f32 {r[0]}, {r[0]}",
        f"v_exp_f32 {r[1]}, {r[1]}",
        f"v_exp_f32 {r[2]}, {r[2]}",
        f"v_exp_f32 {r[3]}, {r[3]}",
        f"v_cvt_pk_bf16_f32 {r[0]}, {r[0]}, {r[1]}",
        f"v_exp_f32 {r[4]}, {r[4]}",
        f"v_cvt_pk_bf16_f32 {r[1]}, {r[2]}, {r[3]}",
        f"v_exp_f32 {r[5]}, {r[5]}",
        f"v_exp_f32 {r[6]}, {r[6]}",
        f"v_exp_f32 {r[7]}, {r[7]}",
        f"v_cvt_pk_bf16_f32 {r[2]}, {r[4]}, {r[5]}",
        f"v_cvt_pk_bf16_f32 {r[3]}, {r[6]}, {r[7]}",
    ]


class Stream:
    def __init__(self, queue):
        self.out = []
        self.queue = list(queue)     # LDS reads in flight, oldest first (fragment buffer names)

    def op(self, text):
        self.out.append(text)

    def read(self, buf, ad, off):
        self.out.append(f"ds_read_b128 %[{buf}], %[ad{ad}] offset:{off}")
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


def rotate(st, j):
    st.op(f"v_add_u32 %[ad{j}], 0x4000, %[ad{j}]")
    st.op(f"v_and_b32 %[ad{j}], 0xffff, %[ad{j}]")


def tail(st, vmcnt, kreads):
    """Slot t + 2 has landed for everybody, slot t - 1 is free; the first fragments of the next iteration."""
    st.drain()
    st.op(f"s_waitcnt vmcnt({vmcnt})")
    st.op("s_barrier")
    rotate(st, 0)
    if kreads:
        st.read("ka", 0, 0)
    st.read("va0", 0, 8192)
    st.read("va1", 0, 8192 + 4096)
    for j in (1, 2, 3):
        rotate(st, j)


def block(par, kind, queue):
    """kind: 'full' (K + V^T pieces of slot t + 3) | 'vonly' (V^T of the last tile only) | 'nodma' | 'last' (no next tile)."""
    cur, nxt = (SA, SB) if par == 0 else (SB, SA)
    st = Stream(queue)
    last = kind == "last"
    if not last:
        st.read("kb", 1, 0)                       # K(j = 1): kb = 0, ks = 1
    # fragment reads issued behind MFMA k of group i: (buffer, address register, offset)
    rd = {(0, 1): [("vb0", 1, 8192), ("vb1", 1, 12288)],
          (1, 3): [("va0", 2, 8192), ("va1", 2, 12288)],
          (3, 3): [("vb0", 3, 8192), ("vb1", 3, 12288)]}
    if not last:
        for i, j in ((0, 2), (1, 3), (2, 4), (3, 5), (4, 6), (5, 7)):   # K(j) into the buffer group i has just released
            rd.setdefault((i, 1), []).insert(0, ("ka" if j % 2 == 0 else "kb", j & 3, (j >> 2) * 4096))
    dma = {}
    if kind == "full":
        dma = {(2, 0): ("vk0", "srk", "sok", 0), (3, 0): ("vk1", "srk", "sok", 4096),
               (4, 0): ("vv0", "srv", "sov", 8192), (5, 0): ("vv1", "srv", "sov", 12288)}
    elif kind == "vonly":
        dma = {(4, 0): ("vv0", "srv", "sov", 8192), (5, 0): ("vv1", "srv", "sov", 12288)}
    for i in range(8):
        g, q = i >> 1, i & 1
        kbj, ksj = i >> 2, i & 3
        vbuf = "va" if g % 2 == 0 else "vb"
        P = vt(cur[(q, g >> 1)] + 8 * (g & 1), 4)
        mf = []
        if not last:
            kbuf = "ka" if i % 2 == 0 else "kb"
            for qq in (0, 1):
                D = vt(nxt[(qq, kbj)], 16)
                C = f"%[ng{qq}]" if ksj == 0 else D
                mf.append((f"{MFMA} {D}, %[{kbuf}], %[q{qq}{ksj}], {C}", kbuf))
        for d in (0, 1):     # behind the QK^T pair: the P of this group was packed by the last VALU instructions of the
            mf.append((f"{MFMA} %[o{q}{d}], %[{vbuf}{d}], {P}, %[o{q}{d}]", f"{vbuf}{d}"))   # previous group (VALU write ->
            #                                                      MFMA operand read needs wait states nothing inserts here)
        mf.append((f"{MFMA} %[rs{q}], %[one], {P}, %[rs{q}]", None))      # row sums: every element = the query's sum over the group
        # VALU of this group of MFMAs: scores of group i + 1 (SG7: group 0 of the next tile)
        if i < 7:
            gi = i + 1
            gq, gg = gi & 1, gi >> 1
            va = vgroup(cur[(gq, gg >> 1)] + 8 * (gg & 1))
        elif not last:
            va = vgroup(nxt[(0, 0)])
        else:
            va = []
        split = ([3, 3, 2, 2, 2] if not last else [5, 5, 2]) if va else [0] * len(mf)
        for k, (text, buf) in enumerate(mf):
            kk = k if not last else 2 * k + 1      # (the last tile's two PV MFMAs stand for positions 1 and 3 of the full group)
            if last and k == 0:                    # (... whose P was packed just before: the wait states a QK^T pair gives elsewhere)
                st.op("s_nop 4")
            d_ = dma.get((i, k))
            if d_:
                st.op(f"s_add_u32 m0, %[mb], {d_[3]}")
            if buf:
                st.need(buf)
            st.op(text)
            if d_:
                vo, srd, so, _ = d_
                st.op(f"buffer_load_dwordx4 %[{vo}], %[{srd}], %[{so}] offen lds")
                if (i, k) == (3, 0):
                    st.op("s_add_u32 %[sok], %[sok], %[kst]")
                if (i, k) == (5, 0):
                    st.op("s_add_u32 %[sov], %[sov], 128")
                    st.op("s_add_u32 %[mb], %[mb], 0x4000")
                    st.op("s_and_b32 %[mb], %[mb], 0xffff")
            for text2 in va[sum(split[:k]):sum(split[:k + 1])]:
                st.op(text2)
            for (b, ad, off) in rd.get((i, kk), []):
                st.read(b, ad, off)
        if i == 6 and not last:
            tail(st, {"full": 4, "vonly": 2, "nodma": 0}[kind], kind != "nodma")
    if last:
        st.drain()
        st.op("s_nop 7")
        st.op("s_nop 7")
        st.op("s_nop 7")
    return st.out, st.queue


def main():
    lines = ["s_waitcnt lgkmcnt(0)"]
    lines += vgroup(SA[(0, 0)])
    st = Stream([])
    st.op("s_waitcnt vmcnt(4)")
    st.op("s_barrier")
    st.read("ka", 0, 0)
    st.read("va0", 0, 8192)
    st.read("va1", 0, 8192 + 4096)
    lines += st.out
    q0 = st.queue
    fa, qa = block(0, "full", q0)
    fb, qb = block(1, "full", qa)
    assert qa == q0 and qb == q0, (q0, qa, qb)
    lines += ["s_cmp_eq_u32 %[cnt], 0", "s_cbranch_scc1 .Lfa4w_after%=", ".Lfa4w_loop%=:"]
    lines += fa + fb
    lines += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 .Lfa4w_loop%=", ".Lfa4w_after%=:"]
    lines += fa
    vb, q1 = block(1, "vonly", qa)
    lines += vb
    na, q2 = block(0, "nodma", q1)
    lines += na
    lb, q3 = block(1, "last", q2)
    assert q3 == []
    lines += lb
    n_mfma = sum(1 for x in fa if x.startswith(MFMA))
    n_valu = sum(1 for x in fa if x.startswith("v_") and not x.startswith(MFMA))
    n_lds = sum(1 for x in fa if x.startswith("ds_"))
    with open(sys.argv[1] if len(sys.argv) > 1 else "flash4w.inc", "w") as f:
        f.write("// Generated by gen_fa4w.py - do not edit.  One FULL iteration: %d MFMA, %d VALU, %d ds_read_b128, %d lines.\n"
                % (n_mfma, n_valu, n_lds, len(fa)))
        f.write("#define FA4W_ASM \\\n")
        for x in lines:
            f.write('  "%s\\n\\t" \\\n' % x)
        f.write('  ""\n')
        f.write("#define FA4W_CLOBBERS " + ", ".join('"v%d"' % r for r in range(192, 256)) + "\n")


if __name__ == "__main__":
    main()
