"""Host logic around the path (CPU, -m "not gpu"): image/metric helpers vs the reference's pinned outputs,
analytic complexity counters vs model_summary's numbers, results.txt layout, round-robin sharding and the
gathered run() under a 2-process gloo group (the oracle port stands in for the model on CPU)."""
import json
import logging
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import GOLD, REPO, load_sd_torch
from ntire2022_esr_amd import dist as D
from ntire2022_esr_amd import harness as H
from ntire2022_esr_amd import image_util as util


def test_image_util_matches_reference_pins():
    m = np.load(os.path.join(GOLD, "metrics.npz"))
    mj = json.load(open(os.path.join(GOLD, "metrics.json")))
    for dr in (1, 255):
        assert np.array_equal(util.uint2tensor4(m["a"], float(dr)).numpy(), m[f"u2t_{dr}"])
        x = torch.from_numpy(m[f"t2u_in_{dr}"].copy())
        x0 = x.clone()
        assert np.array_equal(util.tensor2uint(x, float(dr)), m[f"t2u_out_{dr}"])    # incl. .5 ties (half-to-even)
        assert torch.equal(x, x0), "tensor2uint must not clamp its argument in place"
    assert util.calculate_psnr(m["a"], m["b"], border=4) == mj["psnr_ab_border4"]
    assert util.calculate_psnr(m["a"], m["b"], border=0) == mj["psnr_ab_border0"]
    assert util.calculate_psnr(m["p1"], m["p2"], border=4) == mj["psnr_p1p2_border4"]
    assert util.calculate_psnr(m["a"], m["a"], border=4) == float("inf")
    with pytest.raises(ValueError):
        util.calculate_psnr(m["a"], m["p1"])
    assert list(util.modcrop(np.zeros((1357, 2041, 3), np.uint8), 4).shape) == mj["modcrop_1357x2041x3"]
    assert list(util.modcrop(np.zeros((30, 30), np.uint8), 4).shape) == mj["modcrop_30x30"]
    with pytest.raises(ValueError):
        util.modcrop(np.zeros((2, 2, 2, 2)), 4)


def test_imread_uint_cv2_edge_cases(tmp_path):
    """imread_uint's restatement of cv2.imread(IMREAD_UNCHANGED) + cvtColor (utils/utils_image.py:122-134) on the inputs a DIV2K-style
    folder can contain besides 8-bit RGB: palette, alpha, gray + alpha, 1-bit, 16-bit gray (depth kept), 16-bit RGB (refused)."""
    from PIL import Image
    from ntire2022_esr_amd import image_util as util
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (9, 11, 3), dtype=np.uint8)
    # palette (with and without tRNS): expanded to RGB, transparency ignored
    pal = Image.fromarray(rgb).quantize(16)
    pal.save(tmp_path / "p.png")
    assert np.array_equal(util.imread_uint(str(tmp_path / "p.png")), np.array(pal.convert("RGB")))
    pal.save(tmp_path / "pt.png", transparency=3)
    assert np.array_equal(util.imread_uint(str(tmp_path / "pt.png")), np.array(pal.convert("RGB")))
    # RGBA: alpha dropped, colours NOT composited
    a = rng.integers(0, 256, (9, 11, 1), dtype=np.uint8)
    Image.fromarray(np.concatenate([rgb, a], axis=2), "RGBA").save(tmp_path / "rgba.png")
    assert np.array_equal(util.imread_uint(str(tmp_path / "rgba.png")), rgb)
    # gray + alpha: GGG
    g = rng.integers(0, 256, (9, 11), dtype=np.uint8)
    Image.fromarray(np.stack([g, a[..., 0]], axis=2), "LA").save(tmp_path / "la.png")
    assert np.array_equal(util.imread_uint(str(tmp_path / "la.png")), np.stack([g, g, g], axis=2))
    # 1-bit: 0 / 255
    b = (g > 127)
    Image.fromarray(b).save(tmp_path / "b.png")
    assert Image.open(tmp_path / "b.png").mode == "1"
    assert np.array_equal(util.imread_uint(str(tmp_path / "b.png")), np.stack([b * np.uint8(255)] * 3, axis=2))
    # 16-bit gray: uint16 kept (IMREAD_UNCHANGED), GGG
    g16 = rng.integers(0, 65536, (9, 11), dtype=np.uint16)
    Image.fromarray(g16).save(tmp_path / "g16.png")
    r = util.imread_uint(str(tmp_path / "g16.png"))
    assert r.dtype == np.uint16 and np.array_equal(r, np.stack([g16, g16, g16], axis=2))
    assert util.imread_uint(str(tmp_path / "g16.png"), 1).shape == (9, 11, 1)
    # 16-bit RGB: cv2 would return uint16; refused rather than truncated (a hand-made 2 x 1 PNG: PIL cannot write the format)
    import struct, zlib
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    raw = b"".join(b"\x00" + struct.pack(">6H", 1000, 2000, 3000, 40000, 50000, 60000) for _ in range(1))
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 2, 1, 16, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
    (tmp_path / "rgb16.png").write_bytes(png)
    with pytest.raises(NotImplementedError):
        util.imread_uint(str(tmp_path / "rgb16.png"))
    with pytest.raises(ValueError):
        util.imread_uint(str(tmp_path / "p.png"), 4)
    # imsave: squeezes HxWx1, writes what imread_uint reads back (utils_image.py:137-141)
    util.imsave(g[..., None], str(tmp_path / "sub" / "g.png"))
    assert np.array_equal(util.imread_uint(str(tmp_path / "sub" / "g.png"), 1)[..., 0], g)


def test_imread_imsave_roundtrip(tmp_path):
    img = np.random.RandomState(0).randint(0, 256, (13, 17, 3)).astype(np.uint8)
    p = str(tmp_path / "sub" / "x.png")
    util.imsave(img, p)
    assert np.array_equal(util.imread_uint(p, 3), img)
    from PIL import Image
    Image.fromarray(img[..., 0]).save(str(tmp_path / "g.png"))
    g = util.imread_uint(str(tmp_path / "g.png"), 3)
    assert g.shape == (13, 17, 3) and np.array_equal(g[..., 0], g[..., 2])          # gray -> GGG
    assert util.imread_uint(str(tmp_path / "g.png"), 1).shape == (13, 17, 1)


@pytest.mark.parametrize("name", ["imdn_baseline", "rfdn_baseline", "team04_rlfn", "team18_bsrn"])
def test_complexity_counters_match_model_summary(name):
    """analytic counters == what the reference's hook-based model_summary printed for the same network
    (tests/golden/summary.json, generated by tools/gen_golden.py), incl. BSRN's Linear-hook quirk"""
    import contextlib
    import io
    from ntire2022_esr_amd import BSRN, IMDN, RFDN, RLFN_cut
    from ntire2022_esr_amd.summary import model_complexity
    want = json.load(open(os.path.join(GOLD, "summary.json")))[name]
    with contextlib.redirect_stdout(io.StringIO()):
        m = {"imdn_baseline": IMDN, "rfdn_baseline": RFDN, "team04_rlfn": lambda: RLFN_cut(in_nc=3, out_nc=3),
             "team18_bsrn": lambda: BSRN(num_in_ch=3, num_feat=48, num_block=5, num_out_ch=3, upscale=4, conv="BSConvU",
                                         upsampler="pixelshuffledirect")}[name]()
    assert model_complexity(m, (3, 256, 256)) == want
    if name == "imdn_baseline":   # published table (figs/results.png): 0.894 M / 58.53 G / 154.14 M / 43
        assert round(want["flops"] / 1e9, 2) == 58.53 and round(want["activations"] / 1e6, 2) == 154.14


def test_registry_names_and_ranges_match_reference():
    """results.json keys, results.txt row labels and save_dir sub-folders are the registry names: every id this engine
    implements carries the reference's string and data_range (tests/golden/registry.json, parsed from test_demo.py)."""
    from ntire2022_esr_amd import registry
    ref = json.load(open(os.path.join(GOLD, "registry.json")))
    ents = registry._entries()
    assert sorted(ents) == registry.supported_ids()
    for mid, (disp, stem, dr, tile, _) in ents.items():
        assert f"{mid:02}_{disp}" == ref[str(mid)]["name"], mid
        assert dr == ref[str(mid)]["data_range"] and tile is None
        assert os.path.exists(os.path.join(REPO, "weights", stem + ".safetensors"))


def test_results_table_layout():
    r = {"-1_IMDN_baseline": dict(valid_ave_psnr=29.1312, valid_ave_runtime=1.2345, num_parameters=0.893936,
                                  flops=58.5315, activations=154.1407, valid_memory=471.76, num_conv=43,
                                  test_ave_psnr=28.78, test_ave_runtime=2.0)}
    t = H.results_table(r, include_test=False).split("\n")
    assert t[0].split("\t")[0].strip() == "Model" and len(t[0].split("\t")) == 8
    cells = [c.strip() for c in t[1].split("\t")]
    assert cells == ["-1_IMDN_baseline", "29.13", "1.23", "0.894", "58.53", "154.14", "471.76", "43"]
    t2 = H.results_table(r, include_test=True).split("\n")
    assert len(t2[0].split("\t")) == 11 and [c.strip() for c in t2[1].split("\t")][5] == "1.62"


def test_select_dataset_paths():
    v = H.select_dataset("/d", "valid")
    assert len(v) == 100 and v[0] == ("/d/DIV2K_valid_LR/0801x4.png", "/d/DIV2K_valid_HR/0801.png")
    t = H.select_dataset("/d", "test")
    assert t[-1] == ("/d/DIV2K_test_LR/1000.png", "/d/DIV2K_test_HR/1000.png")


def test_tiled_forward_equals_whole_image_for_a_pointwise_model():
    up = torch.nn.Upsample(scale_factor=4, mode="nearest")
    x = torch.rand(1, 3, 70, 90)
    assert torch.allclose(H.forward(x, up, tile=48, tile_overlap=32), up(x))
    assert torch.equal(H.forward(x, up, tile=None), up(x))


def test_shard_and_gather_single_process():
    assert D.shard(10, 1, 4) == [1, 5, 9] and D.shard(3, 3, 4) == []
    rows = [(i, 1.0 + i, 30.0 + i, float("nan")) for i in range(5)]
    a = D.gather_rows(rows, 5, 0, 1, torch.device("cpu"))
    assert a.shape == (5, 4) and list(a[:, 0]) == [0, 1, 2, 3, 4]
    with pytest.raises(RuntimeError):
        D.gather_rows(rows[:4], 5, 0, 1, torch.device("cpu"))
    assert D.ordered_mean([0.1, 0.2, 0.3]) == sum([0.1, 0.2, 0.3]) / 3


class _OracleModel:
    """CPU stand-in for the GPU module in host-logic tests: the pinned oracle port."""
    def __init__(self):
        from oracle import torch_port as TP
        self.sd, self.f = load_sd_torch("imdn_baseline"), TP.imdn

    def __call__(self, x):
        with torch.no_grad():
            return self.f(self.sd, x)


def _run_rank(rank, world, port, save_dir, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    torch.set_num_threads(2)
    r, w, _ = D.init_from_env(use_cuda=False)
    args = types.SimpleNamespace(data_dir=os.path.join(GOLD, "mini_div2k"), save_dir=save_dir, rank=r, world=w)
    pairs = H.select_dataset(args.data_dir, "valid")[:3]
    logger = logging.getLogger(f"t{rank}")
    res = H.run(_OracleModel(), "imdn", 1.0, None, logger, torch.device("cpu"), args, mode="valid", pairs=pairs)
    if r == 0:
        json.dump(res, open(out, "w"))
    if w > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def test_run_world2_gloo_equals_world1_and_reference(tmp_path):
    import torch.multiprocessing as mp
    ref = json.load(open(os.path.join(GOLD, "mini_div2k", "reference_psnr.json")))["imdn_baseline"]
    out1, out2 = str(tmp_path / "w1.json"), str(tmp_path / "w2.json")
    _run_rank(0, 1, 0, str(tmp_path / "s1"), out1)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    mp.spawn(_run_rank, args=(2, 29533, str(tmp_path / "s2"), out2), nprocs=2, join=True)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    r1, r2 = json.load(open(out1)), json.load(open(out2))
    assert r1["valid_psnr"] == r2["valid_psnr"]                      # bit-identical for W in {1,2}
    assert r1["valid_ave_psnr"] == r2["valid_ave_psnr"]
    assert len(r2["valid_runtime"]) == 3 and set(r2) == {"valid_runtime", "valid_psnr", "valid_memory",
                                                          "valid_ave_runtime", "valid_ave_psnr"}
    for a, b in zip(r1["valid_psnr"], ref["valid_psnr"]):
        assert abs(a - b) < 0.002                                    # per-image |dPSNR| budget
    assert abs(r1["valid_ave_psnr"] - ref["valid_ave_psnr"]) < 0.002
    # every rank wrote its own SR PNGs under save_dir/<model>/valid/
    assert sorted(os.listdir(str(tmp_path / "s2" / "imdn" / "valid"))) == ["0801.png", "0802.png", "0803.png"]


def test_ssim_formula_and_properties():
    """SSIM is parity-unpinned (the reference needs cv2): check against a dense 11x11-window evaluation of the
    published formula, plus identity / symmetry / the 3-channel convention."""
    rng = np.random.RandomState(5)
    a = rng.randint(0, 256, (40, 52, 3)).astype(np.uint8)
    b = np.clip(a.astype(int) + rng.randint(-20, 21, a.shape), 0, 255).astype(np.uint8)
    assert util.calculate_ssim(a, a, border=4) == pytest.approx(1.0, abs=1e-12)
    s_ab = util.calculate_ssim(a, b, border=4)
    assert s_ab == pytest.approx(util.calculate_ssim(b, a, border=4), abs=1e-12) and 0 < s_ab < 1
    # dense reference: full 2-D window, explicit loops over the valid region of the border-cropped arrays
    k = util._gaussian_kernel()
    win = np.outer(k, k)
    assert abs(k.sum() - 1) < 1e-15 and k[5] == k.max() and np.allclose(k, k[::-1])
    x, y = a[4:-4, 4:-4].astype(np.float64), b[4:-4, 4:-4].astype(np.float64)
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    vals = []
    for c in range(3):
        for i in range(5, x.shape[0] - 5):
            for j in range(5, x.shape[1] - 5):
                px, py = x[i - 5:i + 6, j - 5:j + 6, c], y[i - 5:i + 6, j - 5:j + 6, c]
                m1, m2 = (win * px).sum(), (win * py).sum()
                v1, v2, v12 = (win * px * px).sum() - m1 * m1, (win * py * py).sum() - m2 * m2, (win * px * py).sum() - m1 * m2
                vals.append(((2 * m1 * m2 + C1) * (2 * v12 + C2)) / ((m1 * m1 + m2 * m2 + C1) * (v1 + v2 + C2)))
    assert s_ab == pytest.approx(np.mean(vals), abs=1e-10)
    g = a[..., 0]
    assert util.calculate_ssim(g, g) == pytest.approx(1.0, abs=1e-12)
    with pytest.raises(ValueError):
        util.calculate_ssim(a, a[:-1])


def test_free_rider_registry_surface():
    """ids 6 / 22 / 26 reuse the RFDN / IMDN graphs with their own checkpoints (test_demo.py:66-72,175-181,203-209)."""
    from ntire2022_esr_amd import IMDN, RFDN
    from ntire2022_esr_amd.registry import load_checkpoint, supported_ids
    assert supported_ids() == [-1, 0, 4, 6, 8, 18, 22, 26, 40]
    for stem, m in (("team06_v1", RFDN(nf=50)), ("team22_rep_rfdn", RFDN(nf=40)), ("team26_imdn_nb7", IMDN(nb=7)),
                    ("team40_rfdn_pruned", RFDN(nf=40, block_residual=False, esa_f=12)),          # near riders
                    ("team08_sfdn", RFDN(block_residual=False, esa_conv_f=False))):
        missing, unexpected = m.load_state_dict(load_checkpoint(stem), strict=True)
        assert not missing and not unexpected


def test_bench_self_spawns_n_ranks():
    """`python bench.py --gpus 2` without WORLD_SIZE re-executes itself under torch.distributed.run (the driver's N > 1
    invocation): launcher, rendezvous on 127.0.0.1, MAX-reduction and rank gather, exercised on CPU with gloo.  No
    forward runs in this mode and it is labelled as not a measurement."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--launcher-selftest"],
                         capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["launcher_selftest"] is True and j["n_gpus"] == 2 and j["ranks_seen"] == [0, 1]
    # one record per rank, gathered over the process group: own elapsed time and a device identity, two distinct devices
    assert [r["rank"] for r in j["ranks"]] == [0, 1] and len({r["device"] for r in j["ranks"]}) == 2
    assert all(r["elapsed_s"] > 0 for r in j["ranks"])
    # mismatching WORLD_SIZE is refused instead of silently benchmarking the wrong job
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--launcher-selftest"],
                         capture_output=True, text=True, timeout=120, env=env2)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)


def test_harness_under_torch_distributed_run_two_ranks(tmp_path):
    """The harness end to end as a multi-GPU job is started: `python -m torch.distributed.run --nproc-per-node 2` (gloo on CPU; the
    launcher provides RANK / WORLD_SIZE / MASTER_*), against the same script run alone: the sharded, gathered results are
    bit-identical to the single-process ones (test_demo.py:467-475 averages in index order) and every rank wrote its images."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = os.path.join(REPO, "tests", "_harness_rank.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    o1, o2 = str(tmp_path / "w1.json"), str(tmp_path / "w2.json")
    a = subprocess.run([sys.executable, script, o1, str(tmp_path / "s1")], capture_output=True, text=True, timeout=600, env=env)
    assert a.returncode == 0, a.stderr[-2000:]
    b = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), script, o2, str(tmp_path / "s2")], capture_output=True, text=True, timeout=900, env=env)
    assert b.returncode == 0, b.stderr[-2000:]
    r1, r2 = json.load(open(o1)), json.load(open(o2))
    assert r1.pop("_world") == 1 and r2.pop("_world") == 2
    assert r1["valid_psnr"] == r2["valid_psnr"] and r1["valid_ave_psnr"] == r2["valid_ave_psnr"] and len(r2["valid_runtime"]) == 3
    assert sorted(os.listdir(str(tmp_path / "s2" / "imdn" / "valid"))) == ["0801.png", "0802.png", "0803.png"]
