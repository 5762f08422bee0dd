"""Inference batch-size policy (reference: marigold/util/batchsize.py:62-90).

The reference keys a VRAM table measured on A100 / 3090 / 1080Ti and has no bf16 rows, so bf16
silently degrades to batch 1.  On MI355X one GPU has 288 GB of HBM3E:
the whole ensemble is one batch (bigger GEMM M-dimension fills the 256 CUs at the deep, small
UNet levels).  Same signature and the same "return 1 without a GPU" behaviour.
"""
import math

import torch

# measured-feasible members per batch on MI355X (bf16 engine); activations of the VAE decoder at
# 768^2 peak at ~1.2 GB per member, far below 288 GB - the cap keeps headroom for res 2048+.
# (the engine's two operand types - bf16, and fp16 for the --fp16 build - have the same footprint: one row set each, like the
# reference's per-dtype rows)
bs_search_table = [
    {"res": res, "total_vram": 250, "bs": bs, "dtype": dt}
    for dt in (torch.bfloat16, torch.float16) for res, bs in ((768, 64), (1024, 32), (2048, 8))
]


def find_batch_size(ensemble_size: int, input_res: int, dtype: torch.dtype) -> int:
    if not torch.cuda.is_available():
        return 1
    total_vram = torch.cuda.mem_get_info()[1] / 1024.0 ** 3
    for settings in sorted((s for s in bs_search_table if s["dtype"] == dtype),
                           key=lambda k: (k["res"], -k["total_vram"])):
        if input_res <= settings["res"] and total_vram >= settings["total_vram"]:
            bs = settings["bs"]
            if bs > ensemble_size:
                bs = ensemble_size
            elif bs > math.ceil(ensemble_size / 2) and bs < ensemble_size:
                bs = math.ceil(ensemble_size / 2)
            return bs
    return 1
