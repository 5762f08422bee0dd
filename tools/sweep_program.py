#!/usr/bin/env python
"""Program-driven tile sweep (run on the MI355X; tuning, not the product path): builds the real denoising / VAE-decode programs
for the given ensemble sizes, takes every DISTINCT MG_OP_IGEMM launch of one UNet forward (real buffers, epilogues, folded
LayerNorm, second sources ...), and times it under each candidate tile variant x split-K count (op i[19], i[31]).  Prints one
line per layer with the automatic choice's time and the best candidates, writes gpurun_out/sweep_program_E<e>.tsv.  The rules of
mg_igemm_auto_variant / mg_igemm_auto_split are set from these tables (profiles/r5_sweep_program_*.tsv).

    python tools/sweep_program.py --ensembles 1,2,3,5,10 [--variants 23,35,...] [--splits 1,2,3,4,6,8] [--vae]
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import marigold_amd as M  # noqa: E402
from marigold_amd import _lib as L, ops as O, tuning  # noqa: E402
from marigold_amd.schedulers import DDIMScheduler  # noqa: E402


def clone(op, variant=None, splits=None):
    c = L.MgOp()
    ctypes.memmove(ctypes.addressof(c), ctypes.addressof(op), ctypes.sizeof(L.MgOp))
    if variant is not None:
        c.i[19] = variant
    if splits is not None:
        c.i[31] = splits
    return c


def time_op(op, iters, stream):
    lib = L.load()
    for _ in range(2):
        if lib.mg_launch(ctypes.byref(op), stream) != 0:
            return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.mg_launch(ctypes.byref(op), stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def key_of(op):
    return (op.kind,) + tuple(op.i[j] for j in range(36) if j not in (19, 31)) + tuple(bool(op.p[j]) for j in range(16))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ensembles", default="1,2,3,5,10")
    ap.add_argument("--variants", default="23,24,25,26,35,32,22,36,46,51,62,72,73")
    ap.add_argument("--splits", default="1,2,3,4,6,8")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--vae", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--rounds", type=int, default=1, help="> 1: every candidate is timed this many times, interleaved; the median counts")
    ap.add_argument("--emit-db", default="", help="write marigold_amd/tuning-style entries (candidates >= --min-gain ahead of the heuristic) to this JSON file")
    ap.add_argument("--min-gain", type=float, default=0.06)
    args = ap.parse_args()
    variants = [int(v) for v in args.variants.split(",")]
    splits = [int(v) for v in args.splits.split(",")]
    dev = torch.device("cuda:0")
    pipe = M.build_synthetic_pipeline("depth", default_processing_resolution=0).to(dev)
    pipe.unet.set_context(pipe.empty_text_embed)
    stream = O.current_stream_handle()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    db = {}
    for E in [int(e) for e in args.ensembles.split(",")]:
        prog = pipe.unet.denoise_program(E, 96, 96, DDIMScheduler(), 10, rgb_broadcast=True)
        seq = prog.seq
        prog.x.normal_()
        prog.rgb_latent.normal_()
        seq.run()   # every buffer holds plausible values
        todo = list(zip(seq.ops[prog.n_prologue_ops:prog.n_prologue_ops + prog.n_fwd_ops],
                        seq.labels[prog.n_prologue_ops:prog.n_prologue_ops + prog.n_fwd_ops]))
        if args.vae:
            vseq, _, _ = pipe.vae._program("decode", E, 96, 96, L.POST_DEPTH)
            vseq.run()
            todo += list(zip(vseq.ops, ["vae." + l for l in vseq.labels]))
        seen = {}
        rows = []
        for op, label in todo:
            if op.kind != L.OP_IGEMM or (args.only and args.only not in label):
                continue
            k = key_of(op)
            if k in seen:
                seen[k][1] += 1
                continue
            seen[k] = [label, 1]
            B, H, W, Cin, Ho, Wo, N, taps = (op.i[j] for j in range(8))
            Mrows, K = B * Ho * Wo, taps * Cin + (op.i[32] if op.p[12] else 0)
            epi = op.i[12]
            can_split = epi == L.EPI_BF16 and op.i[14] < 0 and op.i[15] <= 1 and not op.p[8] and not op.p[9]
            op = clone(op, 0, 0)   # the heuristic's choice is the baseline (whatever table the engine applied)
            cands = [(v, sp) for v in [0] + variants for sp in ([0] + splits if can_split else [0])]
            samples = {c: [] for c in cands}
            for _ in range(args.rounds):
                for c in cands:
                    if samples[c] is None:
                        continue
                    t = time_op(clone(op, c[0], c[1]), args.iters, stream)
                    if t is None:
                        samples[c] = None
                    else:
                        samples[c].append(t)
            med = {c: sorted(ts)[len(ts) // 2] for c, ts in samples.items() if ts}
            t_auto = med[(0, 0)]
            res = {c: t for c, t in med.items() if c != (0, 0)}
            if args.emit_db and res:
                (bv, bs), bt = min(((c, t) for c, t in res.items() if c[0] != 0), key=lambda kv: kv[1], default=((0, 0), 1e30))
                # a forced tile without a split count runs unsplit: name the count explicitly
                if bt < (1.0 - args.min_gain) * t_auto and (t_auto - bt) >= 1.0:
                    db[tuning.key_of(op)] = [bv, max(1, bs), round(t_auto, 1), round(bt, 1), f"E={E} {label}"]
            best = sorted(res.items(), key=lambda kv: kv[1])[:4]
            flops = 2.0 * Mrows * N * K * max(1, op.i[15])
            rows.append((label, Mrows, N, K, taps, epi, int(bool(op.p[5])), int(bool(op.p[8])), int(bool(op.p[9])), t_auto, best, res))
            print(f"E={E} {label[-52:]:52s} M={Mrows:6d} N={N:5d} K={K:6d} epi={epi} auto {t_auto:7.1f}us ({flops / t_auto / 1e6:6.0f} TF) | " +
                  "  ".join(f"v{v}/s{sp}:{t:6.1f}" for (v, sp), t in best), flush=True)
        with open(os.path.join(ROOT, "gpurun_out", f"sweep_program_E{E}.tsv"), "w") as f:
            f.write("label\tcount\tM\tN\tK\ttaps\tepi\tres\tln_out\tln_in\tauto_us\tbest\tall\n")
            for (label, Mrows, N, K, taps, epi, r_, lo, li, t_auto, best, res) in rows:
                cnt = [c for (lab, c) in seen.values() if lab == label][0]
                f.write(f"{label}\t{cnt}\t{Mrows}\t{N}\t{K}\t{taps}\t{epi}\t{r_}\t{lo}\t{li}\t{t_auto:.1f}\t" +
                        ";".join(f"v{v}s{sp}={t:.1f}" for (v, sp), t in best) + "\t" +
                        ";".join(f"v{v}s{sp}={t:.1f}" for (v, sp), t in sorted(res.items())) + "\n")
        saved = sum((r[9] - min(r[9], r[10][0][1])) * [c for (lab, c) in seen.values() if lab == r[0]][0] for r in rows if r[10])
        print(f"E={E}: per UNet forward the best candidates would save {saved:.0f} us over the automatic choice "
              f"({saved * 10 / 1e3:.1f} ms per 10-step map)", flush=True)
        del prog
        pipe.unet._programs.clear()
        torch.cuda.empty_cache()
    if args.emit_db:
        import json
        with open(args.emit_db, "w") as f:
            json.dump({"device": torch.cuda.get_device_name(0), "min_gain": args.min_gain, "rounds": args.rounds, "igemm": db}, f, indent=0)
        print(f"{len(db)} table entries -> {args.emit_db}")


if __name__ == "__main__":
    main()
