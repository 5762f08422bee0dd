"""N3: dataset readers, alignment, metrics and the infer/eval programs, against vectors
produced by the reference's own src/util/{metric,alignment}.py and src/dataset/* (oracle/make_eval_golden.py
-> tests/golden/eval_ref.npz).  Everything here is host code and runs without a GPU."""
import os

import numpy as np
import pytest
import torch

from marigold_amd import evaluation as E
from marigold_amd.evaluation import datasets as D, harness as H, metrics as M
from oracle.make_eval_golden import eval_inputs, write_synthetic_datasets


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "eval_ref.npz"))


@pytest.fixture(scope="module")
def cases():
    return eval_inputs()


@pytest.mark.parametrize("key", ["depth_a", "depth_b", "depth_big"])
def test_depth_metrics_match_reference(gold, cases, key):
    c = cases[key]
    got = [getattr(M, n)(c["pred"], c["gt"], c["mask"]) for n in M.DEPTH_METRICS]
    np.testing.assert_allclose(got, gold[f"{key}/metrics_masked"], rtol=2e-6, atol=1e-7)
    names = [n for n in M.DEPTH_METRICS if not n.startswith("delta")]
    got = [getattr(M, n)(c["pred"], c["gt"], None) for n in names]
    np.testing.assert_allclose(got, gold[f"{key}/metrics_nomask"], rtol=2e-6, atol=1e-7)
    # garbage outside the mask (zeros / NaN in the ground truth) must not leak into the scores
    gt = c["gt"].copy()
    gt[~c["mask"]] = 0
    pred = c["pred"].copy()
    pred[~c["mask"]] = np.nan
    got2 = [getattr(M, n)(pred, gt, c["mask"]) for n in M.DEPTH_METRICS]
    np.testing.assert_allclose(got2, gold[f"{key}/metrics_masked"], rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("key", ["depth_a", "depth_b", "depth_big"])
def test_least_squares_alignment_matches_reference(gold, cases, key):
    c = cases[key]
    for res in (None, 64):
        aligned, s, t = E.align_depth_least_square(c["gt"], c["rel"], c["mask"], True, res)
        # the reference solves the fp32 system with lstsq; the normal equations in fp64 agree to fp32 rounding
        np.testing.assert_allclose([float(s[0]), float(t[0])], gold[f"{key}/ls_{res}"], rtol=2e-4)
        np.testing.assert_allclose(aligned, gold[f"{key}/ls_{res}_aligned"], rtol=0, atol=2e-3)
        assert aligned.shape == c["rel"].shape
    disp, pos = E.depth2disparity(c["gt_holes"], return_mask=True)
    assert np.array_equal(disp, gold[f"{key}/disparity"]) and np.array_equal(pos, c["gt_holes"] > 0)
    ok = c["mask"] & pos & (c["rel"] > 0)
    _, s, t = E.align_depth_least_square(disp, c["rel"], ok, True, None)
    np.testing.assert_allclose([float(s[0]), float(t[0])], gold[f"{key}/ls_disp"], rtol=2e-4)


def test_alignment_downscale_is_torch_nearest():
    from marigold_amd.evaluation.alignment import _nearest_downscale
    a = np.arange(37 * 53, dtype=np.float32).reshape(37, 53)
    for f in (64 / 53, 0.5, 0.37, 20 / 53):
        if f >= 1:
            continue
        want = torch.nn.Upsample(scale_factor=f, mode="nearest")(torch.from_numpy(a)[None])[0].numpy()
        assert want.shape[0] == 37 and np.array_equal(_nearest_downscale(a, f), want)   # rows are all kept


def test_normals_metrics_match_reference(gold, cases):
    c = cases["normals"]
    for masked in (False, True):
        err = M.compute_cosine_error(c["pred"][None], c["gt"][None], masked)
        ref = gold[f"normals/err_masked{int(masked)}"]
        assert err.shape == ref.shape
        np.testing.assert_allclose(err, ref, atol=2e-2)   # acos near 0 deg amplifies fp32 rounding of the cosine
        names = ("mean_angular_error", "median_angular_error", "rmse_angular_error", "sub5_error", "sub7_5_error",
                 "sub11_25_error", "sub22_5_error", "sub30_error")
        np.testing.assert_allclose([getattr(M, n)(ref) for n in names], gold[f"normals/metrics_masked{int(masked)}"],
                                   rtol=1e-6)
        np.testing.assert_allclose([getattr(M, n)(err) for n in names], gold[f"normals/metrics_masked{int(masked)}"],
                                   atol=0.1)


def test_iid_metric_helpers_match_reference(gold, cases):
    c = cases["iid"]
    np.testing.assert_allclose(M.compute_alignment_scale(c["pred"], c["gt"]), gold["iid/scale_nomask"], rtol=1e-5)
    np.testing.assert_allclose(M.compute_alignment_scale(c["pred"], c["gt"], c["mask"]), gold["iid/scale_masked"],
                               rtol=1e-5)
    for tag, m in (("nomask", None), ("masked", c["mask"])):
        p, g = M.quantile_map(c["pred"], c["gt"], m)
        np.testing.assert_allclose(p, gold[f"iid/qmap_pred_{tag}"], atol=2e-6)
        np.testing.assert_allclose(g, gold[f"iid/qmap_gt_{tag}"], atol=2e-6)
        for target in ("albedo", "shading"):
            got = M.compute_iid_metric(c["pred"].copy(), c["gt"].copy(), target, "psnr", m)
            np.testing.assert_allclose(got, gold[f"iid/psnr_{target}_{tag}"], rtol=1e-5)
    # SSIM: identical images -> 1, symmetric, drops with noise (no reference implementation available offline)
    x = c["gt"][None]
    assert abs(M.ssim(x, x) - 1) < 1e-12 and abs(M.ssim(x, c["pred"][None]) - M.ssim(c["pred"][None], x)) < 1e-12
    assert M.ssim(x, c["pred"][None]) < 0.99
    with pytest.raises(NotImplementedError):
        M.compute_iid_metric(c["pred"], c["gt"], "albedo", "lpips")


def test_pred_names(gold):
    names = {"rgb_id": "rgb_0012.png", "i_d_rgb": "3_17_rgb.jpg", "id": "0000000005.png",
             "rgb_i_d": "rgb_cam_00_fr0001.png"}
    got = [E.get_pred_name(v, D.PredNameMode[k], suffix=".npy") for k, v in names.items()]
    assert got == list(gold["pred_names"])


def _summary(v):
    flat = v.reshape(-1).astype(np.float64)
    return np.concatenate([[float(x) for x in v.shape], [np.nansum(flat), np.nansum(flat * flat)],
                           flat[::max(1, flat.size // 257)]])


@pytest.mark.parametrize("as_tar", [(), ("nyu", "diode", "nyu_normals", "kitti", "hypersim_iid")])
def test_dataset_readers_match_reference(gold, tmp_path, as_tar):
    """Every on-disk format (16-bit PNG / .npy / raw masks, folder and tar) decodes to what the reference's
    dataset classes return: same keys, shapes, values, masks, crops and file filtering."""
    cfgs = write_synthetic_datasets(str(tmp_path), as_tar=as_tar)
    checked = 0
    for key, cfg in cfgs.items():
        for mode in (D.DatasetMode.RGB_ONLY, D.DatasetMode.EVAL):
            ds = E.get_dataset(cfg, str(tmp_path), mode)
            assert len(ds) == int(gold[f"ds/{key}/{mode.value}/len"])
            if f"ds/{key}/range" in gold.files:
                assert [ds.min_depth, float(ds.max_depth)] == list(gold[f"ds/{key}/range"])
            for i, item in enumerate(ds):
                prefix = f"ds/{key}/{mode.value}/{i}/"
                want_keys = {k[len(prefix):] for k in gold.files if k.startswith(prefix)}
                have = set(item) - {"index"}
                if mode == D.DatasetMode.RGB_ONLY and ds.spec.kind != "depth":
                    # the reference's normals / iid readers compare the mode against a *different* Enum class
                    # (base_normals_dataset.py:44,111; base_iid_dataset.py:54,118), so they decode the ground truth
                    # even for inference; here RGB_ONLY reads the image only
                    assert have < want_keys and not any(k in have for k in ("normals", "albedo"))
                else:
                    assert want_keys == have, (key, mode, want_keys ^ have)
                for k in have:
                    ref, v = gold[prefix + k], item[k]
                    if k == "rgb_relative_path":
                        assert str(ref) == v
                        continue
                    v = _summary(v) if v.size > 4096 else v
                    assert v.shape == ref.shape, (key, k, v.shape, ref.shape)
                    if ref.dtype == bool or np.issubdtype(np.asarray(item[k]).dtype, np.integer):
                        assert np.array_equal(v, ref), (key, k)
                    else:
                        np.testing.assert_allclose(v, ref, rtol=2e-6, atol=1e-6, err_msg=f"{key}/{k}")
                    checked += 1
    assert checked > 60


def test_dataset_config_and_errors(tmp_path):
    cfgs = write_synthetic_datasets(str(tmp_path))
    import yaml
    p = tmp_path / "data_nyu.yaml"
    p.write_text(yaml.safe_dump(cfgs["nyu"]))
    cfg = E.load_dataset_config(str(p))
    assert cfg["name"] == "nyu_depth" and cfg["eigen_valid_mask"] is True
    with pytest.raises(NotImplementedError):
        E.get_dataset(dict(cfg, name="cityscapes_depth"), str(tmp_path), D.DatasetMode.EVAL)
    with pytest.raises(AssertionError, match="mixed"):
        E.get_dataset(dict(cfg, name="mixed"), str(tmp_path), D.DatasetMode.EVAL)
    with pytest.raises(AssertionError, match="does not exist"):
        E.get_dataset(dict(cfg, dir="nowhere"), str(tmp_path), D.DatasetMode.EVAL)
    (tmp_path / "bad.yaml").write_text("- 1\n- 2\n")
    with pytest.raises(ValueError):
        E.load_dataset_config(str(tmp_path / "bad.yaml"))
    # ETH3D raw float32 rasters with +inf holes
    eth = tmp_path / "eth3d" / "s"
    eth.mkdir(parents=True)
    raw = np.full(D.ETH3D_HW, np.inf, np.float32)
    raw[:10, :10] = 2.5
    raw.tofile(eth / "d.bin")
    from PIL import Image
    Image.fromarray(np.zeros((8, 12, 3), np.uint8)).save(eth / "i.png")
    (tmp_path / "eth.txt").write_text("s/i.png s/d.bin\n")
    ds = E.get_dataset(dict(name="eth3d_depth", disp_name="e", dir="eth3d", filenames=str(tmp_path / "eth.txt")),
                       str(tmp_path), D.DatasetMode.EVAL)
    it = ds[0]
    assert it["depth_raw_linear"].shape == (1,) + D.ETH3D_HW and it["valid_mask_raw"].sum() == 100
    assert it["depth_raw_linear"].max() == 2.5


class _FakeDepthOut:
    def __init__(self, d):
        self.depth_np = d


def test_depth_infer_then_eval_end_to_end(tmp_path, gold):
    """infer_main -> <out>/<scene>/pred_<id>.npy -> eval_main with least-squares alignment: a stand-in pipeline
    that returns an affine transform of the ground truth must score (almost) perfectly, the csv / txt files
    have the reference's layout (script/depth/eval.py:139-245)."""
    import yaml
    cfgs = write_synthetic_datasets(str(tmp_path), as_tar=("nyu",))
    cfg_path = tmp_path / "nyu.yaml"
    cfg_path.write_text(yaml.safe_dump(cfgs["nyu"]))
    gt_ds = E.get_dataset(cfgs["nyu"], str(tmp_path), D.DatasetMode.EVAL)
    gts = iter([s["depth_raw_linear"][0] for s in gt_ds])
    calls = []

    class FakePipe:
        device = "cpu"

        def __call__(self, image, **kw):
            calls.append(kw)
            assert image.size == (640, 480) and kw["color_map"] is None and kw["batch_size"] == 0
            return _FakeDepthOut((1.0 - next(gts) / 12.0).astype(np.float32) * 0.8 + 0.1)

    out = tmp_path / "pred"
    argv = ["--dataset_config", str(cfg_path), "--base_data_dir", str(tmp_path), "--output_dir", str(out),
            "--denoise_steps", "4", "--processing_res", "0", "--ensemble_size", "2", "--seed", "7"]
    assert H.infer_main("depth", argv, pipeline=FakePipe()) == 0
    assert len(calls) == 2 and calls[0]["denoising_steps"] == 4 and calls[0]["ensemble_size"] == 2
    assert calls[0]["generator"].initial_seed() == 7 and calls[0]["match_input_res"] is True
    assert sorted(os.listdir(out / "test" / "kitchen")) == ["pred_0003.npy", "pred_0012.npy"]
    ev = tmp_path / "eval"
    eargv = ["--prediction_dir", str(out), "--dataset_config", str(cfg_path), "--base_data_dir", str(tmp_path),
             "--output_dir", str(ev), "--alignment", "least_square"]
    assert H.eval_main("depth", eargv) == 0
    rows = (ev / "per_sample_metrics.csv").read_text().strip().split("\n")
    assert rows[0] == "filename," + ",".join(M.DEPTH_METRICS) and len(rows) == 3
    assert rows[1].startswith("test/kitchen/pred_0003.npy,")
    vals = np.array([float(x) for x in rows[1].split(",")[1:]])
    assert vals[0] < 1e-4 and vals[5] == 1.0    # abs_rel ~ 0, delta1 = 1 after the affine alignment
    txt = (ev / "eval_metrics-least_square.txt").read_text()
    assert "on dataset: nyu_synth" in txt and "min_depth = 0.001" in txt and "abs_relative_difference" in txt
    # without alignment the affine-distorted prediction is bad, and the file name has no suffix
    assert H.eval_main("depth", eargv[:-2]) == 0
    assert (ev / "eval_metrics.txt").exists()
    first = (ev / "per_sample_metrics.csv").read_text().strip().split("\n")[1].split(",")
    assert float(first[1]) > 0.1
    # disparity alignment path runs and clips
    assert H.eval_main("depth", eargv[:-1] + ["least_square_disparity", "--alignment_max_res", "320"]) == 0
    assert (ev / "eval_metrics-least_square_disparity.txt").exists()
    # a second inference into the same folder asks first; --yes skips the question
    gts = iter([s["depth_raw_linear"][0] for s in gt_ds])
    assert H.infer_main("depth", argv + ["--yes"], pipeline=FakePipe()) == 0


def test_normals_and_iid_infer_eval(tmp_path):
    import yaml
    import marigold_amd as MA
    cfgs = write_synthetic_datasets(str(tmp_path))
    # normals
    cfg_path = tmp_path / "n.yaml"
    cfg_path.write_text(yaml.safe_dump(cfgs["nyu_normals"]))
    gt = E.get_dataset(cfgs["nyu_normals"], str(tmp_path), D.DatasetMode.EVAL)[0]["normals"]

    class FakeNormals:
        device = "cpu"

        def __call__(self, image, **kw):
            assert "color_map" not in kw
            n = gt / np.maximum(np.linalg.norm(gt, axis=0, keepdims=True), 1e-6)
            return MA.MarigoldNormalsOutput(n.astype(np.float32), None, None)

    out = tmp_path / "pn"
    base = ["--dataset_config", str(cfg_path), "--base_data_dir", str(tmp_path)]
    assert H.infer_main("normals", base + ["--output_dir", str(out), "--denoise_steps", "4", "--processing_res", "0",
                                           "--ensemble_size", "1"], pipeline=FakeNormals()) == 0
    assert (out / "x" / "img.npy").exists()
    ev = tmp_path / "en"
    assert H.eval_main("normals", base + ["--prediction_dir", str(out), "--output_dir", str(ev)]) == 0
    row = (ev / "per_sample_metrics.csv").read_text().strip().split("\n")
    assert row[0] == "filename," + ",".join(M.NORMALS_METRICS) and row[1].startswith("x/img.png,")
    vals = [float(v) for v in row[1].split(",")[1:]]
    assert vals[0] < 0.1 and vals[-1] == 100.0
    with pytest.raises(AssertionError, match="is a normals dataset"):
        H.infer_main("depth", base + ["--output_dir", str(tmp_path / "zz"), "--denoise_steps", "1",
                                      "--processing_res", "0", "--ensemble_size", "1"], pipeline=FakeNormals())
    # iid (Hypersim lighting: albedo / shading / residual)
    cfg_path = tmp_path / "i.yaml"
    cfg_path.write_text(yaml.safe_dump(cfgs["hypersim_iid"]))
    sample = E.get_dataset(cfgs["hypersim_iid"], str(tmp_path), D.DatasetMode.EVAL)[0]
    props = {"target_names": ["albedo", "shading", "residual"], "albedo": {"prediction_space": "linear"},
             "shading": {"prediction_space": "linear", "up_to_scale": True},
             "residual": {"prediction_space": "linear", "up_to_scale": True}}

    class FakeIID:
        device = "cpu"
        target_names = props["target_names"]

        def __call__(self, image, **kw):
            o = MA.MarigoldIIDOutput(self.target_names)
            for t in self.target_names:
                scale = 0.5 if t != "albedo" else 1.0    # shading / residual are only defined up to scale
                o.fill_entry(t, torch.from_numpy(np.nan_to_num(sample[t]) * scale)[None], None, props)
            return o

    out = tmp_path / "pi"
    base = ["--dataset_config", str(cfg_path), "--base_data_dir", str(tmp_path)]
    assert H.infer_main("iid", base + ["--output_dir", str(out), "--denoise_steps", "4", "--processing_res", "0",
                                       "--ensemble_size", "1"], pipeline=FakeIID()) == 0
    assert sorted(os.listdir(out / "ai")) == ["rgb_cam_00_fr0000_albedo.npy", "rgb_cam_00_fr0000_residual.npy",
                                              "rgb_cam_00_fr0000_shading.npy"]
    ev = tmp_path / "ei"
    assert H.eval_main("iid", base + ["--prediction_dir", str(out), "--output_dir", str(ev), "--use_mask",
                                      "--target_names", "albedo", "shading", "residual"]) == 0
    rows = (ev / "per_sample_metrics.csv").read_text().strip().split("\n")
    assert rows[0] == "filename,psnr_albedo,ssim_albedo,psnr_shading,ssim_shading,psnr_residual,ssim_residual"
    vals = [float(v) for v in rows[1].split(",")[1:]]
    assert vals[2] > 60 and vals[3] > 0.999 and vals[4] > 60    # exact up to scale -> aligned away
    with pytest.raises(ValueError, match="does not belong"):
        H.eval_main("iid", base + ["--prediction_dir", str(out), "--output_dir", str(ev),
                                   "--targets_to_eval_in_linear_space", "shading"])
