"""The oracle's ensembling restatement vs outputs of the reference's own
marigold/util/ensemble.py (tests/golden/ensemble_ref.npz, made by oracle/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import ensemble as oens


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "ensemble_ref.npz"))


@pytest.mark.parametrize("name", ["d_e4", "d_e10", "d_e3"])
def test_depth_affine_median(gold, name):
    x = torch.from_numpy(gold[f"{name}_in"])
    d, u = oens.ensemble_depth(x.clone(), True, True, output_uncertainty=True)
    # BFGS trajectory is identical code on identical scipy -> tight tolerance
    np.testing.assert_allclose(d.numpy(), gold[f"{name}_out"], atol=2e-5)
    np.testing.assert_allclose(u.numpy(), gold[f"{name}_unc"], atol=2e-5)


def test_depth_scale_only_mean(gold):
    x = torch.from_numpy(gold["d_scale_mean_in"])
    d, u = oens.ensemble_depth(x.clone(), True, False, output_uncertainty=True, reduction="mean")
    np.testing.assert_allclose(d.numpy(), gold["d_scale_mean_out"], atol=2e-5)
    np.testing.assert_allclose(u.numpy(), gold["d_scale_mean_unc"], atol=2e-5)


def test_depth_error_behaviour(gold):
    assert int(gold["abs_raises"]) == 1  # the reference raises for (False, False)
    x = torch.rand(3, 1, 8, 8)
    with pytest.raises(ValueError):
        oens.ensemble_depth(x, False, False)
    with pytest.raises(ValueError):
        oens.ensemble_depth(x, False, True)
    with pytest.raises(ValueError):
        oens.ensemble_depth(x[:, 0], True, True)
    with pytest.raises(ValueError):
        oens.ensemble_depth(x, True, True, reduction="max")


@pytest.mark.parametrize("name", ["n_e4", "n_e10"])
def test_normals(gold, name):
    x = torch.from_numpy(gold[f"{name}_in"])
    n, u = oens.ensemble_normals(x.clone(), output_uncertainty=True)
    m, _ = oens.ensemble_normals(x.clone(), reduction="mean")
    np.testing.assert_array_equal(n.numpy(), gold[f"{name}_closest"])
    np.testing.assert_allclose(u.numpy(), gold[f"{name}_unc"], atol=1e-6)
    np.testing.assert_allclose(m.numpy(), gold[f"{name}_mean"], atol=1e-6)
    with pytest.raises(ValueError):
        oens.ensemble_normals(x[:, :2])
    with pytest.raises(ValueError):
        oens.ensemble_normals(x, reduction="median")


def test_median_is_lower_middle():
    a = torch.tensor([[1.0], [4.0], [2.0], [3.0]]).view(4, 1, 1, 1)
    p, _ = oens.depth_reduce(a, "median", False)
    assert p.item() == 2.0


@pytest.mark.parametrize("red", ["median", "mean"])
def test_iid(gold, red):
    x = torch.from_numpy(gold["iid_in"])
    p, u = oens.ensemble_iid(x.clone(), output_uncertainty=True, reduction=red)
    np.testing.assert_array_equal(p.numpy(), gold[f"iid_{red}_pred"])
    np.testing.assert_allclose(u.numpy(), gold[f"iid_{red}_unc"], atol=1e-7)
    with pytest.raises(ValueError):
        oens.ensemble_iid(x, reduction="max")
