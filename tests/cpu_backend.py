"""torch-CPU stand-in for the two HIP passes behind ``marigold_amd.ensemble.DepthAligner`` (test
infrastructure: lets the host-side optimiser logic be checked against the reference's goldens
without a GPU; the kernels themselves are checked against the same maths in tests/test_gpu_*)."""
import numpy as np
import torch


class TorchStatsBackend:
    def __init__(self, d, reduction, affine):
        self.d = d.reshape(d.shape[0], -1).float()
        self.E = self.d.shape[0]
        self.red, self.affine = reduction, affine

    def stats(self):
        x = self.d.double()
        mean = x.mean(dim=1)
        xc = x - mean[:, None]
        C = (xc @ xc.t()) / x.shape[1]
        return (x.min(dim=1).values.numpy(), x.max(dim=1).values.numpy(), mean.numpy(), C.numpy())

    def regulariser(self, s32, t32):
        s = torch.from_numpy(np.asarray(s32, dtype=np.float32))[:, None]
        t = torch.from_numpy(np.asarray(t32, dtype=np.float32))[:, None]
        a = self.d * s + t if self.affine else self.d * s
        pred = torch.median(a, dim=0).values if self.red == 0 else a.mean(dim=0)
        imn, imx = int(pred.argmin()), int(pred.argmax())
        return (float(pred[imn]), float(pred[imx]), self.d[:, imn].double().numpy(),
                self.d[:, imx].double().numpy())
