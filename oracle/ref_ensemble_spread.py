"""TEST INFRASTRUCTURE (never imported by the product path): how reproducible is the REFERENCE's own ``ensemble_depth``?

The reference aligns the members with scipy BFGS on finite differences of a cost it evaluates in fp32 torch reductions
(/root/reference/marigold/util/ensemble.py:96-173).  Those reductions change their summation order with the CPU thread count, and
the optimiser stops where that noise swamps its differences - so the reference's output depends on ``torch.get_num_threads()``.
This script runs the reference itself (imported from /root/reference, as oracle/make_golden.py does) on the members of the metric
configuration's golden (E = 10, 768 x 768, seed 51) with several thread counts and stores the outputs next to the committed
8-thread golden: ``tests/golden/ensemble_ref_768_threads.npz``.  tests/test_gpu_pipeline.py::
test_ensemble_depth_metric_config_vs_reference holds the engine's deviation from the reference against this spread.

    python -m oracle.ref_ensemble_spread [768] [threads,threads,...]        (minutes of CPU per thread count)

Measured in the build container (8 cores; profiles/r5_reference_ensemble_thread_spread.log): the 8-thread run reproduces the
committed golden bit for bit; 4 vs 8 vs 16 threads differ by max 1.2-1.6e-2, mean 4.3-4.9e-3, delta1 0.9959-0.9976.
"""
import os
import sys
import time

import numpy as np
import torch

from oracle.make_golden import GOLD, _import_reference_ensemble, synth_realistic_depth_members


def main():
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 768
    counts = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "4,8,16").split(","))
    ref = _import_reference_ensemble()
    x = synth_realistic_depth_members(10, H, H, 51)
    outs = {}
    for nt in counts:
        torch.set_num_threads(nt)
        t0 = time.time()
        d, _ = ref.ensemble_depth(x.clone(), scale_invariant=True, shift_invariant=True, output_uncertainty=True)
        outs[nt] = d.numpy()[0, 0]
        print(f"threads {nt}: {time.time() - t0:.1f} s", flush=True)
    keys = list(outs)
    for i in range(len(keys)):
        for j in range(i + 1, len(keys)):
            a, b = outs[keys[i]], outs[keys[j]]
            diff = np.abs(a - b)
            d1 = np.mean(np.maximum(a / np.maximum(b, 1e-6), b / np.maximum(a, 1e-6)) < 1.25)
            print(f"reference({keys[i]} threads) vs reference({keys[j]} threads): max {diff.max():.4e} mean {diff.mean():.4e} delta1 {d1:.5f}")
    if H == 768:
        np.savez_compressed(os.path.join(GOLD, "ensemble_ref_768_threads.npz"),
                            **{f"d_real_e10_768_out_t{k}": v.astype(np.float16) for k, v in outs.items() if k != 8})


if __name__ == "__main__":
    main()
