#!/usr/bin/env python3
"""Writes igemm2_k4w.inc: the hand-placed K-tile instruction streams of the 256 x 256 / four-wave implicit GEMM
(igemm2_body.h, LOOP == 3; one wave per SIMD, 128 x 128 wave tile, accumulators in AGPRs).

One K tile (64 deep) of a wave = 64 v_mfma_f32_32x32x16_bf16 (4 k-steps x 4 x 4 fragments) with every other instruction of the
tile placed BY HAND in the gaps between them - hipcc's scheduler groups the fragment reads in front of the MFMAs, which on a
one-wave-per-SIMD kernel leaves the matrix pipe idle for every LDS round trip (profiles/r3_sweep_big_wave_tile.log: 1.0 PF/s).
The stream, per K tile t (LDS buffer b = t & 1; fragment set F[ks] = 4 pixel-side + 4 weight-side fragments of k-step ks):

  gap  0.. 7   ds_read F[2], F[3] of tile t            (two per gap; F[0], F[1] were read during tile t-1)
  gap  R1      s_waitcnt lgkmcnt(0) ; s_barrier         every wave has read ALL of buffer b -> it is free
  gap  D0..    16 x { s_add m0 ; buffer_load_dwordx4 .. lds }   tile t+2 -> buffer b, one piece every other gap
  gap  R2      s_waitcnt vmcnt(16) ; s_barrier          tile t+1 (issued during tile t-1) has landed for everyone
  gap  N0..    ds_read F[0], F[1] of tile t+1           (one per gap)

so a piece has 66-96 MFMA gaps (2.1-3.1 k cycles) to land, the fragments of a k-step are in registers 16+ gaps before their
first MFMA, and a gap carries at most two non-MFMA issues (MI355X_MICROARCH.md: <= 5 hide behind a 32 x 32 x 16 MFMA).
Waits are counted (vmcnt retires in order; the LDS-DMA is only ordered against ds_read by vmcnt + barrier).

Operand names (bound in igemm2_body.h): c<ni><mi> accumulators, a<ks><mi> / b<ks><ni> fragments (pixel / weight side),
la<ks> / lb<ks> LDS byte addresses of the fragment reads, xa<ks> / xb<ks> their buffer toggles (XOR), va<i> / vb<i> per-lane byte
offsets of the DMA pieces, sa / sb buffer resources, ma / mb = LDS addresses of this wave's first pixel / weight piece in the
buffers being filled.
"""
import sys

# Geometry (set by main): MI x NI 32 x 32 fragments per wave (2 x 2 waves): 4 x 4 = the 256 x 256 tile (prefix K4W), 3 x 5 = the
# 192 x 320 tile (prefix K4WB: full-width tiles for the N = 320 k layers).  LDS: pixel-row buffers, then weight-row buffers;
# a read address toggles between an operand's two buffers by XOR with a per-register constant (xa<ks> / xb<ks>), the DMA bases
# (ma / mb) are toggled by the caller.
MI, NI = 4, 4


def GS():
    return MI * NI


def mfma(g):
    ks, ni, mi = g // GS(), (g % GS()) // MI, g % MI
    return f"v_mfma_f32_32x32x16_bf16 %[c{ni}{mi}], %[b{ks}{ni}], %[a{ks}{mi}], %[c{ni}{mi}]"


def reads(ks):
    """the fragment reads of k-step ks, pixel side first"""
    return ([f"ds_read_b128 %[a{ks}{i}], %[la{ks}] offset:{i * 4096}" for i in range(MI)] +
            [f"ds_read_b128 %[b{ks}{i}], %[lb{ks}] offset:{i * 4096}" for i in range(NI)])


def toggles(ks):
    return [f"v_xor_b32 %[la{ks}], %[xa{ks}], %[la{ks}]", f"v_xor_b32 %[lb{ks}], %[xb{ks}], %[lb{ks}]"]


def dma(i):
    """piece i of the next-but-one tile: (instruction in front of the MFMA, instruction behind it) - an SALU write of M0 needs
    one instruction before the LDS-DMA that reads it"""
    if i < 2 * MI:
        return (f"s_add_u32 m0, %[ma], {i * 4096}", f"buffer_load_dwordx4 %[va{i}], %[sa], 0 offen lds")
    j = i - 2 * MI
    return (f"s_add_u32 m0, %[mb], {j * 4096}", f"buffer_load_dwordx4 %[vb{j}], %[sb], 0 offen lds")


def block(mode, p):
    """mode: 'full' (tiles t+1 and t+2 exist), 'nodma' (t+1 exists), 'last'"""
    NM, nF, ND = 4 * GS(), MI + NI, 2 * (MI + NI)
    pre = {g: [] for g in range(NM + 1)}     # instructions in front of MFMA g (g = NM: behind the last)
    post = {g: [] for g in range(NM)}        # instructions right behind MFMA g
    pre[0].append(f"s_waitcnt lgkmcnt({nF})")
    # --- F[2], F[3] of this tile
    cur = reads(2) + reads(3)
    g = 0
    while cur:
        for _ in range(p["rd_per_gap"]):
            if cur:
                post[g].append(cur.pop(0))
        g += 1
    last_read_gap = g - 1
    post[last_read_gap + 1] += toggles(2)
    post[last_read_gap + 2] += toggles(3)
    r2 = p["r2"] if NM == 64 else NM - 2 * nF - 2
    n0 = p["n0"] if NM == 64 else NM - 2 * nF - 1
    if mode == "last":
        pre[p["r1"]].append("s_waitcnt lgkmcnt(0)")
    else:
        pre[p["r1"]] += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
        issued = 0
        if mode == "full":
            for i in range(ND):
                gg = p["d0"] + i * p["dstep"]
                a, b = dma(i)
                pre[gg].append(a)
                post[gg].append(b)
                if gg < r2:
                    issued += 1
        pre[r2] += [f"s_waitcnt vmcnt({issued})", "s_barrier"]
        nxt = reads(0) + reads(1)
        g = n0
        while nxt:
            post[g].append(nxt.pop(0))
            g += 1
        assert g <= NM, g
        pre[NM] += toggles(0) + toggles(1)
    lines = []
    for g in range(NM):
        lines += pre[g]
        lines.append(mfma(g))
        lines += post[g]
    lines += pre[NM]
    if mode == "last":
        # the epilogue's v_accvgpr_read follow in compiler code, whose hazard recogniser does not see the MFMAs in here: an
        # 8-pass MFMA's result may be read 11 wait states after its issue at the earliest (round 4, first run: the LAST
        # accumulator of every wave held its value from before the final MFMA)
        lines += ["s_nop 7", "s_nop 7", "s_nop 7"]
    return lines


def prologue():
    """fragment sets F[0], F[1] of tile 0 (buffer 0), then la0 / la1 -> buffer 1"""
    return reads(0) + reads(1) + toggles(0) + toggles(1)



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
    # schedule parameters: reads per gap in the first segment, gap of the buffer-release barrier, first DMA gap and DMA stride,
    # gap of the landed barrier, first gap of the next tile's fragment reads
    p = dict(rd_per_gap=2, r1=12, d0=13, dstep=2, r2=46, n0=47)
    for a in sys.argv[1:]:
        k, v = a.split("=")
        p[k] = int(v)
    assert p["d0"] > p["r1"] and p["n0"] >= p["r2"] and p["n0"] + 16 <= 64 + 0
    global MI, NI
    txt = ["// GENERATED by gen_k4w.py " + " ".join(f"{k}={v}" for k, v in p.items()) + " - do not edit; see the generator for the schedule."]
    for prefix, mi, ni in (("K4W", 4, 4), ("K4WB", 3, 5)):
        MI, NI = mi, ni
        txt += [emit(f"{prefix}_ASM_FULL", block("full", p)), "", emit(f"{prefix}_ASM_NODMA", block("nodma", p)), "",
                emit(f"{prefix}_ASM_LAST", block("last", p)), "", emit(f"{prefix}_ASM_PROLOGUE", prologue()), ""]
    print("\n".join(txt))


if __name__ == "__main__":
    main()
