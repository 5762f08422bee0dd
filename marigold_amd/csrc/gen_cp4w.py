#!/usr/bin/env python3
"""Writes conv_patch4w.inc: the hand-placed instruction streams of the four-wave patch-resident 3x3 convolution
(conv_patch4w.hip; one wave per SIMD, accumulators in AGPRs) - the schedule of gen_k4w.py (igemm2 variant 72) carried over to
the kernel that keeps the input patch of a channel tile resident in LDS.

Geometries (PREFIX_...):  CP4A: 16 x 16 pixels x 256 channels, wave tile 128 x 128 (MI = NI = 4, 64 MFMAs per K tile);
                          CP4B: 12 x 16 pixels x 320 channels, wave tile  96 x 160 (MI = 3, NI = 5, 60 MFMAs per K tile).
One K tile = (channel tile c, tap t): 64 input channels of one tap.  Per tile k (weight stage k & 1, patch buffer c & 1):

  gap 0..7     ds_read fragment sets F[2], F[3] of tile k (pixel side: the patch at this tap's shift; weight side: the stage)
  gap R1       s_waitcnt lgkmcnt(0) ; s_barrier          every wave has read all of weight stage k & 1 -> it is free
  then         [patch pieces of channel tile c + 1 (taps 0-2)] ; the weight pieces of tile k + 2 -> stage k & 1   (LDS-DMA,
               buffer loads on SGPR bases, one piece every DSTEP gaps, each with its M0 write one MFMA earlier)
               VALU: the 4 x MI pixel-side read addresses of tile k + 1 (tap shift + XOR swizzle: 9 instructions per mi)
               [fix-up slice (taps 4-7, fused GroupNorm + SiLU on patch c + 1, in place): ds_read_b128 -> 8 x (unpack, fma,
               silu) -> 4 x cvt_pk -> select (padding rows stay zero) -> ds_write_b128, spread one instruction per slot]
  gap R2       s_waitcnt vmcnt(n) lgkmcnt(0) ; s_barrier  the weights of tile k + 1 (and, by then, every older piece) landed
  gap N0..     ds_read F[0], F[1] of tile k + 1

Counted waits come from queue models of the LDS and VMEM streams (an item's wait = the number of later-issued items of its
queue), so a schedule edit cannot silently leave a wait short.  Scratch registers of the VALU sequences are the PHYSICAL
registers v228-v255 (clobbered): an asm operand cannot name the dwords of a 128-bit tuple.

Operands (bound in conv_patch4w.hip): c<ni><mi> accumulators; a<ks><mi> / b<ks><ni> fragments; pa<mi><ks> pixel-side read
addresses (absolute LDS bytes); lb<ks> weight-side read addresses, xb<ks> their stage toggles (XOR); r0<mi> the lane's patch row
of fragment mi at tap (0, 0); hx = (lane >> 5) << 4; vb<i> / vp<j> byte offsets of the weight / patch DMA pieces; sw / sp buffer
resources; mw / mp LDS addresses of this wave's first weight / patch piece; stoff, spb: tap row offset and patch LDS base of
tile k + 1; s96 = 96; fxa, fmask, fs<j>, fh<j>: fix-up address (patch base + tid * 16), row mask, scale / shift of the lane's 8
channels.
"""
import sys

STATS = []


class Geo:
    def __init__(self, prefix, mi, ni):
        self.prefix, self.MI, self.NI = prefix, mi, ni
        self.GS = mi * ni            # MFMAs per k-step
        self.NM = 4 * self.GS        # MFMAs per K tile
        self.NW = ni * 2             # weight pieces per wave (BN * 8 / 256 = NI * 2 * 32 * 8 / 256)


def mfma(G, g):
    ks, ni, mi = g // G.GS, (g % G.GS) // G.MI, g % G.MI
    return f"v_mfma_f32_32x32x16_bf16 %[c{ni}{mi}], %[b{ks}{ni}], %[a{ks}{mi}], %[c{ni}{mi}]"


def reads(G, ks):
    """fragment set F[ks]: (text, queue tag)"""
    out = [(f"ds_read_b128 %[a{ks}{i}], %[pa{i}{ks}]", f"F{ks}") for i in range(G.MI)]
    out += [(f"ds_read_b128 %[b{ks}{i}], %[lb{ks}] offset:{i * 4096}", f"F{ks}") for i in range(G.NI)]
    return out


def toggles(ks):
    return [f"v_xor_b32 %[lb{ks}], %[xb{ks}], %[lb{ks}]"]


def addr_ops(G):
    """pixel-side read addresses of the next tile: r = r0 + toff; swz = ((r << 3) & 0x70) ^ hx; base = r * 128 + patch base;
    pa[ks] = (swz ^ (ks << 5)) + base.  Scratch v228 (r), v229 (swz), v230 (base)."""
    ops = []
    for mi in range(G.MI):
        ops += [f"v_add_u32 v228, %[stoff], %[r0{mi}]",
                "v_lshlrev_b32 v229, 3, v228",
                "v_and_b32 v229, 0x70, v229",
                f"v_xor_b32 v229, %[hx], v229",
                f"v_lshl_add_u32 v230, v228, 7, %[spb]",
                f"v_xad_u32 %[pa{mi}0], v229, 0, v230",
                f"v_xad_u32 %[pa{mi}1], v229, 32, v230",
                f"v_xad_u32 %[pa{mi}2], v229, 64, v230",
                f"v_xad_u32 %[pa{mi}3], v229, %[s96], v230"]
    return ops


def fix_vec(it, useq):
    """GroupNorm affine + SiLU on one staged 16-byte vector (8 channels of one patch row), in place; the arithmetic of
    conv_patch.hip::fix_vec (fma, x * rcp(1 + exp2(-log2e x)), v_cvt_pk_bf16_f32).  -> (the vector's ds_read, the rest); the
    loaded vector lives in v224-v227 / v232-v235 alternately so that the next vector's read can be issued ahead."""
    U = 224 if useq % 2 else 232
    X, T, R = 236, 244, 252
    rd = (f"ds_read_b128 v[{U}:{U + 3}], %[fxa] offset:{it * 4096}", "lds", f"fixr{it}")
    ops = []
    first = True
    for j in range(4):     # the two halves of a dword interleaved: a transcendental's result is read two instructions
        w = U + j          # later (gfx940-family TRANS -> VALU forwarding hazard: one wait state, which nothing inserts here)
        xa, xb, ta, tb = X + 2 * j, X + 2 * j + 1, T + 2 * j, T + 2 * j + 1
        seq = [f"v_lshlrev_b32 v{xa}, 16, v{w}",
               f"v_and_b32 v{xb}, 0xffff0000, v{w}",
               f"v_fma_f32 v{xa}, v{xa}, %[fs{2 * j}], %[fh{2 * j}]",
               f"v_fma_f32 v{xb}, v{xb}, %[fs{2 * j + 1}], %[fh{2 * j + 1}]",
               f"v_mul_f32 v{ta}, 0xbfb8aa3b, v{xa}",
               f"v_mul_f32 v{tb}, 0xbfb8aa3b, v{xb}",
               f"v_exp_f32 v{ta}, v{ta}",
               f"v_exp_f32 v{tb}, v{tb}",
               f"v_add_f32 v{ta}, 1.0, v{ta}",
               f"v_add_f32 v{tb}, 1.0, v{tb}",
               f"v_rcp_f32 v{ta}, v{ta}",
               f"v_rcp_f32 v{tb}, v{tb}",
               f"v_mul_f32 v{xa}, v{xa}, v{ta}",
               f"v_mul_f32 v{xb}, v{xb}, v{tb}"]
        for k, sq in enumerate(seq):
            ops.append((sq, "waitfix" if (first and k == 0) else "op", f"fixr{it}"))
        first = False
    for j in range(4):
        ops.append((f"v_cvt_pk_bf16_f32 v{R + j}, v{X + 2 * j}, v{X + 2 * j + 1}", "op", None))
    ops.append((f"v_and_b32 v{T}, {1 << it}, %[fmask]", "op", None))
    ops.append((f"v_cmp_ne_u32 vcc, 0, v{T}", "op", None))
    for j in range(4):
        ops.append((f"v_cndmask_b32 v{R + j}, v{U + j}, v{R + j}, vcc", "op", None))
    ops.append((f"ds_write_b128 %[fxa], v[{R}:{R + 3}] offset:{it * 4096}", "lds", f"fixw{it}"))
    return rd, ops


def block(G, mode, npatch, fix_its, p):
    """mode: 'full' | 'nodma' | 'last'.  npatch: patch pieces (operands vp0..) staged in this tile; fix_its: fix-up vectors."""
    NM = G.NM
    slots = [[] for _ in range(2 * NM + 1)]      # slot 2g: in front of MFMA g, slot 2g + 1: behind it, slot 2 NM: tail
    def put(slot, text, kind="op", tag=None):
        slots[slot].append((text, kind, tag))
    nF = G.MI + G.NI
    put(0, None, "wait_lds", "F0")
    # F[2], F[3] of this tile, two per gap
    cur = reads(G, 2) + reads(G, 3)
    g = 0
    while cur:
        for _ in range(2):
            if cur:
                t, tag = cur.pop(0)
                put(2 * g + 1, t, "lds", tag)
        g += 1
    g_reads_end = g
    for t in toggles(2):
        put(2 * g_reads_end + 1, t)
    for t in toggles(3):
        put(2 * (g_reads_end + 1) + 1, t)
    r1, r2, n0 = p["r1"], NM - 2 * nF - 2 + p["r2off"], NM - 2 * nF - 1 + p["r2off"]
    assert r1 > g_reads_end and n0 + 2 * nF <= NM
    if mode == "last":
        put(2 * r1, None, "wait_lds", "ALL")
        return finish(G, slots, mode, [], [])
    put(2 * r1, None, "wait_lds", "ALL")
    put(2 * r1, "s_barrier")
    # ---- LDS-DMA: patch pieces first, then the weight pieces of tile k + 2 ----
    dma = []
    if mode == "full":
        for j in range(npatch):
            dma.append((f"s_add_u32 m0, %[mp], {j * 4096}", f"buffer_load_dwordx4 %[vp{j}], %[sp], 0 offen lds", "P"))
        for i in range(G.NW):
            dma.append((f"s_add_u32 m0, %[mw], {i * 4096}", f"buffer_load_dwordx4 %[vb{i}], %[sw], 0 offen lds", "W2"))
    g = r1 + 1
    for (a, b, tag) in dma:
        put(2 * g, a)
        put(2 * g + 1, b, "vmem", tag)
        g += p["dstep"]
    assert g - p["dstep"] < NM, (g, NM)
    # ---- VALU fillers: next tile's pixel-side addresses, then the fix-up slice ----
    # pa<mi><2,3> are read by this tile's F[2] / F[3] reads (gaps < g_reads_end): fillers start behind them.  Everything -
    # the addresses the F[0] / F[1] reads of tile k + 1 use, and every LDS operation of the fix-up (the next stream's entry
    # wait counts only fragment reads) - sits in front of n0.  A fix-up vector's ds_read is issued LEAD fillers ahead of its
    # first use (LDS latency under the MFMAs, not in front of them).
    LEAD = 14
    fill = [(t, "op", None) for t in addr_ops(G)]
    for q, it in enumerate(fix_its):
        rd, ops = fix_vec(it, q)
        pos = max(0, len(fill) - LEAD)
        fill.insert(pos, rd)
        fill += ops
    s0 = 2 * (g_reads_end + 2)
    free = [sl for sl in range(s0, 2 * n0) if not any(x[1].startswith("wait") or x[0] == "s_barrier" for x in slots[sl])]
    per = max(1, -(-len(fill) // max(1, len(free))))
    assert per <= p["max_per_slot"], f"{G.prefix} {mode} fix {fix_its}: {len(fill)} fillers in {len(free)} slots"
    # spread evenly over the free slots (Bresenham): a dense front would starve the matrix pipe early and idle late
    fi = 0
    for n, sl in enumerate(free):
        want = (len(fill) * (n + 1)) // len(free)
        while fi < want:
            t, kind, tag = fill[fi]
            fi += 1
            if kind == "waitfix":
                put(sl, None, "wait_lds", tag)
                kind = "op"
            put(sl, t, kind, tag)
    STATS.append((G.prefix, mode, npatch, fix_its, len(fill), len(free)))
    assert fi == len(fill)
    # ---- tile k + 1 has landed: its first two fragment sets ----
    put(2 * r2, None, "wait_vm", "W1")
    put(2 * r2, None, "wait_lds_before_reads", None)
    put(2 * r2, "s_barrier")
    nxt = reads(G, 0) + reads(G, 1)
    g = n0
    while nxt:
        t, tag = nxt.pop(0)
        put(2 * g + 1, t, "lds", tag + "n")
        g += 1
    for t in toggles(0) + toggles(1):
        put(2 * NM, t)
    return finish(G, slots, mode, ["W1"] * G.NW, [])


def finish(G, slots, mode, vm_queue0, _):
    """linearise, resolve the counted waits"""
    NM = G.NM
    nF = G.MI + G.NI
    lin = []
    for g in range(NM):
        lin += slots[2 * g]
        lin.append((mfma(G, g), "mfma", None))
        lin += slots[2 * g + 1]
    lin += slots[2 * NM]
    # queues at tile entry: LDS = F[0] then F[1] reads of this tile (issued at the end of the previous stream);
    # VMEM = the weight pieces of tile k + 1
    ldsq = ["F0"] * nF + ["F1"] * nF
    vmq = list(vm_queue0)
    out = []
    for (t, kind, tag) in lin:
        if kind == "lds":
            ldsq.append(tag)
            out.append(t)
        elif kind == "vmem":
            vmq.append(tag)
            out.append(t)
        elif kind == "wait_lds":
            if tag == "ALL":
                n = 0
            else:
                idx = max(i for i, x in enumerate(ldsq) if x == tag)
                n = len(ldsq) - 1 - idx
            assert n <= 15
            out.append(f"s_waitcnt lgkmcnt({n})")
        elif kind == "wait_lds_before_reads":
            # the fix-up's LDS traffic of this tile is retired before the next tile's fragment reads are queued: the next
            # stream's entry wait counts only those reads
            out.append("s_waitcnt lgkmcnt(0)")
        elif kind == "wait_vm":
            idxs = [i for i, x in enumerate(vmq) if x == tag]
            n = len(vmq) - 1 - max(idxs) if idxs else len(vmq)
            out.append(f"s_waitcnt vmcnt({n})")
        else:
            out.append(t)
    if mode == "last":
        out += ["s_nop 7", "s_nop 7", "s_nop 7"]   # MFMA results -> the epilogue's v_accvgpr_read (see gen_k4w.py)
    return out


def prologue(G):
    out = [t for (t, _) in reads(G, 0) + reads(G, 1)]
    return out + toggles(0) + toggles(1)



def c_literal(ln):
    """One instruction as a C string literal; the operand-type mnemonics come from common.h (MG_MFMA32_ASM, MG_CVT_PK_ASM: bf16 in the
    product build, fp16 in the fp16 build) as adjacent literals."""
    for mnem, macro in (("v_mfma_f32_32x32x16_bf16", "MG_MFMA32_ASM"), ("v_cvt_pk_bf16_f32", "MG_CVT_PK_ASM")):
        if ln.startswith(mnem + " "):
            return macro + ' "' + ln[len(mnem):]
    return '"' + ln


def emit(name, lines):
    out = [f"#define {name} \\"]
    for ln in lines:
        out.append(f'  {c_literal(ln)}\\n" \\')
    out.append('  ""')
    return "\n".join(out)


def main():
    p = dict(r1=12, dstep=2, r2off=0, max_per_slot=5)
    for a in sys.argv[1:]:
        k, v = a.split("=")
        p[k] = int(v)
    txt = ["// GENERATED by gen_cp4w.py " + " ".join(f"{k}={v}" for k, v in p.items()) + " - do not edit; see the generator for the schedule."]
    # patch pieces per wave: CP4A 11, CP4B 8, all staged during tap 0 (landed for everyone at tap 1's second barrier);
    # fix-up vectors per thread: 11 / 8 over taps 2-8
    for G, patches, fixsets in ((Geo("CP4A", 4, 4), (11,), ((0, 1), (2, 3), (4, 5), (6, 7), (8,), (9,), (10,))),
                                (Geo("CP4B", 3, 5), (8,), ((0, 1), (2,), (3,), (4,), (5,), (6,), (7,)))):
        P = G.prefix
        # ONE statement for the steady state: the stream variants side by side behind a scalar dispatch on %[sel] (0 plain,
        # 1 patch staging, 2.. fix-up slices).  As separate asm statements in a switch hipcc spilled the accumulators and
        # fragment sets around every branch (100+ scratch_store_dwordx4 per tile).
        variants = [block(G, "full", 0, (), p)] + [block(G, "full", n, (), p) for n in patches] + \
                   [block(G, "full", 0, its, p) for its in fixsets]
        loop = []
        for i in range(1, len(variants)):
            loop += [f"s_cmp_eq_u32 %[sel], {i}", f"s_cbranch_scc1 .Lcp4_%=_{i}"]
        for i, v in enumerate(variants):
            if i:
                loop.append(f".Lcp4_%=_{i}:")
            loop += v
            if i + 1 < len(variants):
                loop.append(f"s_branch .Lcp4_%=_end")
        loop.append(".Lcp4_%=_end:")
        txt += [emit(f"{P}_LOOP", loop), ""]
        txt += [emit(f"{P}_NODMA", block(G, "nodma", 0, (), p)), ""]
        txt += [emit(f"{P}_LAST", block(G, "last", 0, (), p)), ""]
        txt += [emit(f"{P}_PROLOGUE", prologue(G)), ""]
    print("\n".join(txt))
    for st in STATS:
        print("stats", st, file=sys.stderr)


if __name__ == "__main__":
    main()
