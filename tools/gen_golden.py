#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference (authoring container only).

Runs only where /root/reference exists.  It never ships to the GPU box: what
travels is the DATA it writes (weights/ and tests/golden/), never reference
source.  Re-run with:   python tools/gen_golden.py

What it writes
  weights/<ckpt>.safetensors      the 4 hot-path checkpoints, key-for-key, fp32
  weights/manifest.json           key -> shape, sha256 per file
  tests/golden/e2e_<model>.npz    seeded inputs + reference fp32 outputs
  tests/golden/img_<model>.npz    utils/test.bmp 256x256 -> strided sample of the fp32 SR
                                  + a 64x64 bicubic LR proxy -> uint8 SR + PSNR
  tests/golden/blocks.npz         block-level in/out with the real block-1 weights
  tests/golden/metrics.npz/.json  uint2tensor4 / tensor2uint / modcrop / calculate_psnr pins
  tests/golden/summary.json       params / FLOPs / acts / #conv (reference model_summary)
  tests/golden/mini_div2k/        3 LR/HR PNG pairs + per-model reference PSNRs
  tests/golden/test.bmp           the reference's only image (data)
"""
import contextlib
import hashlib
import io
import json
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
WDIR = os.path.join(REPO, "weights")

import numpy as np
import torch
from PIL import Image
from safetensors.torch import save_file

torch.set_num_threads(8)


def _stub_cv2_torchvision():
    """utils.utils_image imports cv2 + torchvision (absent here); the 4 functions we
    pin (uint2tensor4, tensor2uint, modcrop, calculate_psnr) never touch them."""
    import matplotlib
    matplotlib.use("Agg")
    cv2 = types.ModuleType("cv2")
    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tvu.make_grid = lambda *a, **k: None
    tv.utils = tvu
    sys.modules.setdefault("cv2", cv2)
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.utils", tvu)


def load_reference_models():
    sys.path.insert(0, REF)
    os.chdir(REF)
    from models.imdn_baseline import IMDN
    from models.rfdn_baseline.RFDN import RFDN
    from models.team04_rlfn import RLFN_cut
    from models.team18_bsrn import BSRN

    def ld(name):
        return torch.load(os.path.join(REF, "model_zoo", name), map_location="cpu", weights_only=False)

    out = {}
    m = IMDN(in_nc=3, out_nc=3, nc=64, nb=8, upscale=4)
    sd = ld("imdn_baseline.pth")
    m.load_state_dict(sd, strict=True)
    out["imdn_baseline"] = (m.eval(), sd, 1.0)

    m = RFDN()
    sd = ld("rfdn_baseline.pth")
    m.load_state_dict(sd, strict=True)
    out["rfdn_baseline"] = (m.eval(), sd, 255.0)

    m = RLFN_cut(in_nc=3, out_nc=3)
    sd = ld("team04_rlfn.pth")
    m.load_state_dict(sd, strict=True)
    out["team04_rlfn"] = (m.eval(), sd, 255.0)

    with contextlib.redirect_stdout(io.StringIO()):
        m = BSRN(num_in_ch=3, num_feat=48, num_block=5, num_out_ch=3, upscale=4,
                 conv="BSConvU", upsampler="pixelshuffledirect")
    sd = ld("team18_bsrn.pth")["params"]
    m.load_state_dict(sd, strict=True)
    out["team18_bsrn"] = (m.eval(), sd, 1.0)
    for m, _, _ in out.values():
        for p in m.parameters():
            p.requires_grad = False
    return out


def load_free_riders():
    """Registry entries that reuse the in-scope graphs with their own checkpoints:
    id 6 `v1` and id 22 `RFDN40` are the rfdn_baseline graph (nf=50 / nf=40), id 26 is IMDN with nb=7; the near
    riders id 40 (pruned RFDN: nf=40, no in-block residual, ESA width fixed at 12) and id 8 (SFDN: residual folded
    into the checkpoint's weights, ESA without conv_f) are RFDN graphs with two switches."""
    from models.team06_v1 import v1
    from models.team22_rep_rfdn import RFDN40
    from models.imdn_baseline import IMDN
    from models.team40_rfdn_pruned import RFDN as RFDNPrune
    with contextlib.redirect_stdout(io.StringIO()):
        from models.team08_sfdn import RFDN as SFDN

    def ld(name):
        return torch.load(os.path.join(REF, "model_zoo", name), map_location="cpu", weights_only=False)

    out = {}
    for key, ctor, ck, dr in (("team06_v1", lambda: v1(in_nc=3, nf=50, num_modules=4, out_nc=3, upscale=4), "team06_v1.pth", 1.0),
                              ("team22_rep_rfdn", RFDN40, "team22_rep_rfdn.pth", 1.0),
                              ("team26_imdn_nb7", lambda: IMDN(in_nc=3, out_nc=3, nc=64, nb=7, upscale=4, act_mode='L',
                                                               upsample_mode='pixelshuffle'), "team26_imdn_nb7.pth", 1.0),
                              ("team40_rfdn_pruned", lambda: RFDNPrune(in_nc=3, nf=40, num_modules=4, out_nc=3, upscale=4),
                               "team40_rfdn_pruned.pth", 255.0),                                   # test_demo.py:302-308
                              ("team08_sfdn", SFDN, "team08_sfdn.pt", 1.0)):                         # test_demo.py:76-82
        with contextlib.redirect_stdout(io.StringIO()):
            m = ctor()
        sd = ld(ck)
        m.load_state_dict(sd, strict=True)
        out[key] = (m.eval(), sd, dr)
    return out


def write_riders(manifest):
    """weights + one small seeded vector for each free rider"""
    riders = load_free_riders()
    with torch.no_grad():
        for name, (m, sd, dr) in riders.items():
            tensors = {k: v.detach().float().contiguous().clone() for k, v in sd.items()}
            path = os.path.join(WDIR, name + ".safetensors")
            save_file(tensors, path)
            manifest[name] = {"file": name + ".safetensors", "sha256": sha256(path), "data_range": dr,
                              "num_tensors": len(tensors),
                              "num_elements": int(sum(v.numel() for v in tensors.values())),
                              "keys": {k: list(v.shape) for k, v in tensors.items()}}
            g = torch.Generator().manual_seed(2)
            xb = torch.rand(2, 3, 20, 36, generator=g) * dr
            np.savez(os.path.join(GOLD, f"e2e_{name}.npz"), xb=xb.numpy(), yb=m(xb).numpy(), data_range=np.float32(dr))


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def main():
    os.makedirs(GOLD, exist_ok=True)
    os.makedirs(WDIR, exist_ok=True)
    os.makedirs(os.path.join(GOLD, "mini_div2k"), exist_ok=True)
    _stub_cv2_torchvision()
    models = load_reference_models()
    import utils.utils_image as util
    from utils.model_summary import get_model_activation, get_model_flops

    # ---- weights (data) -------------------------------------------------
    manifest = {}
    for name, (m, sd, dr) in models.items():
        tensors = {k: v.detach().float().contiguous().clone() for k, v in sd.items()}
        path = os.path.join(WDIR, name + ".safetensors")
        save_file(tensors, path)
        manifest[name] = {
            "file": name + ".safetensors",
            "sha256": sha256(path),
            "data_range": dr,
            "num_tensors": len(tensors),
            "num_elements": int(sum(v.numel() for v in tensors.values())),
            "keys": {k: list(v.shape) for k, v in tensors.items()},
        }
    with open(os.path.join(WDIR, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)

    # ---- free riders: weights + one small seeded vector each -------------------
    write_riders(manifest)
    with open(os.path.join(WDIR, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)

    # ---- end-to-end seeded vectors ---------------------------------------
    with torch.no_grad():
        for name, (m, sd, dr) in models.items():
            g = torch.Generator().manual_seed(0)
            xa = torch.rand(1, 3, 48, 64, generator=g) * dr
            xb = torch.rand(2, 3, 20, 36, generator=g) * dr
            xc = torch.rand(1, 3, 17, 15, generator=g) * dr   # minimum-size / ragged edge case
            np.savez(os.path.join(GOLD, f"e2e_{name}.npz"),
                     xa=xa.numpy(), ya=m(xa).numpy(), xb=xb.numpy(), yb=m(xb).numpy(),
                     xc=xc.numpy(), yc=m(xc).numpy(), data_range=np.float32(dr))

    # ---- natural image (the reference's utils/test.bmp) -------------------
    bmp = os.path.join(REF, "utils", "test.bmp")
    with open(bmp, "rb") as f, open(os.path.join(GOLD, "test.bmp"), "wb") as g:
        g.write(f.read())
    img = np.array(Image.open(bmp).convert("RGB"))          # 256x256x3 uint8
    lr64 = np.array(Image.fromarray(img).resize((64, 64), Image.BICUBIC))
    with torch.no_grad():
        for name, (m, sd, dr) in models.items():
            x = util.uint2tensor4(img, dr)
            y = m(x)
            ys = y[0, :, ::5, ::5].numpy().copy()
            y_mean, y_l2 = y.double().mean().item(), y.double().pow(2).sum().sqrt().item()
            # NB: the reference's tensor2uint clamps IN PLACE (utils_image.py:205: .data.squeeze()
            # .float() are views, then .clamp_), so it must only ever see clones here.
            y8 = util.tensor2uint(y.clone(), dr)
            xl = util.uint2tensor4(lr64, dr)
            yl = m(xl)
            yl8 = util.tensor2uint(yl.clone(), dr)
            psnr = util.calculate_psnr(yl8, util.modcrop(img, 4), border=4)
            np.savez(os.path.join(GOLD, f"img_{name}.npz"),
                     sr_sample=ys, sr_mean=np.float64(y_mean), sr_l2=np.float64(y_l2),
                     sr_u8_sum=np.int64(y8.astype(np.int64).sum()),
                     sr_u8_crop=y8[400:528, 300:428].copy(),
                     lr64=lr64, lr64_sr_u8=yl8, lr64_sr_f32=yl.numpy(), lr64_psnr=np.float64(psnr))

    # ---- block-level vectors (real block-1 weights, 24x20 spatial) -------
    blk = {}
    with torch.no_grad():
        g = torch.Generator().manual_seed(1)
        m = models["imdn_baseline"][0]
        x = torch.randn(1, 64, 24, 20, generator=g) * 0.5
        blk["imdb_x"], blk["imdb_y"] = x.numpy(), m.model[1].sub[0](x).numpy()
        m = models["rfdn_baseline"][0]
        x = torch.randn(1, 50, 24, 20, generator=g) * 20
        blk["rfdb_x"], blk["rfdb_y"] = x.numpy(), m.B1(x).numpy()
        blk["rfdn_esa_x"], blk["rfdn_esa_y"] = x.numpy(), m.B1.esa(x).numpy()
        m = models["team04_rlfn"][0]
        x = torch.randn(1, 46, 24, 20, generator=g) * 20
        blk["rlfb_x"], blk["rlfb_y"] = x.numpy(), m.B1(x).numpy()
        blk["rlfn_esa_x"], blk["rlfn_esa_y"] = x.numpy(), m.B1.esa(x).numpy()
        m = models["team18_bsrn"][0]
        x = torch.randn(1, 48, 24, 20, generator=g) * 0.5
        blk["bsrn_rfdb_x"], blk["bsrn_rfdb_y"] = x.numpy(), m.B1(x).numpy()
        blk["bsrn_esa_x"], blk["bsrn_esa_y"] = x.numpy(), m.B1.esa(x).numpy()
        blk["bsconv_x"], blk["bsconv_y"] = x.numpy(), m.B1.c1_r(x).numpy()
    np.savez(os.path.join(GOLD, "blocks.npz"), **blk)

    # ---- metric / conversion pins (reference utils_image) ----------------
    rng = np.random.RandomState(0)
    a = rng.randint(0, 256, (37, 41, 3)).astype(np.uint8)
    b = np.clip(a.astype(np.int32) + rng.randint(-3, 4, a.shape), 0, 255).astype(np.uint8)
    met = {"a": a, "b": b}
    mj = {}
    mj["psnr_ab_border4"] = util.calculate_psnr(a, b, border=4)
    mj["psnr_ab_border0"] = util.calculate_psnr(a, b, border=0)
    mj["psnr_identical"] = util.calculate_psnr(a, a, border=4)
    rs = np.random.RandomState(0)
    p1 = rs.randint(0, 256, (64, 64, 3)).astype(np.uint8)
    p2 = np.clip(p1.astype(np.int32) + rs.randint(-2, 3, p1.shape), 0, 255).astype(np.uint8)
    met["p1"], met["p2"] = p1, p2
    mj["psnr_p1p2_border4"] = util.calculate_psnr(p1, p2, border=4)
    for dr in (1.0, 255.0):
        t = util.uint2tensor4(a, dr)
        met[f"u2t_{int(dr)}"] = t.numpy()
        # exact .5 ties (numpy round = half-to-even), out-of-range, negatives
        v = torch.tensor([[-3.0, 0.0, 0.5, 1.5, 2.5, 3.5, 127.5, 128.5, 254.5, 255.0, 255.49, 300.0]]) * (dr / 255.0)
        v = v.reshape(1, 1, 3, 4).repeat(1, 3, 1, 1)
        met[f"t2u_in_{int(dr)}"] = v.numpy().copy()
        met[f"t2u_out_{int(dr)}"] = util.tensor2uint(v.clone(), dr)
    mj["modcrop_1357x2041x3"] = list(util.modcrop(np.zeros((1357, 2041, 3), np.uint8), 4).shape)
    mj["modcrop_30x30"] = list(util.modcrop(np.zeros((30, 30), np.uint8), 4).shape)
    np.savez(os.path.join(GOLD, "metrics.npz"), **met)
    with open(os.path.join(GOLD, "metrics.json"), "w") as f:
        json.dump(mj, f, indent=1)

    # ---- complexity counters (reference model_summary) ---------------------
    summ = {}
    for name, (m, sd, dr) in models.items():
        with contextlib.redirect_stdout(io.StringIO()):
            acts, nconv = get_model_activation(m, (3, 256, 256))
            flops = get_model_flops(m, (3, 256, 256), False)
        summ[name] = {"activations": float(acts), "num_conv": int(nconv), "flops": float(flops),
                      "num_parameters": int(sum(p.numel() for p in m.parameters()))}
    with open(os.path.join(GOLD, "summary.json"), "w") as f:
        json.dump(summ, f, indent=1)

    # ---- mini DIV2K-shaped dataset (3 pairs cut from test.bmp) -----------
    # directory layout follows select_dataset (test_demo.py:344-361)
    md = os.path.join(GOLD, "mini_div2k")
    os.makedirs(os.path.join(md, "DIV2K_valid_LR"), exist_ok=True)
    os.makedirs(os.path.join(md, "DIV2K_valid_HR"), exist_ok=True)
    crops = [(0, 0, 96, 128), (100, 60, 128, 96), (150, 120, 102, 130)]  # y, x, h, w (last: not /4)
    res = {k: [] for k in models}
    for i, (y0, x0, h, w) in enumerate(crops):
        hr = img[y0:y0 + h, x0:x0 + w]
        lh, lw = h // 4, w // 4
        lr = np.array(Image.fromarray(hr[:lh * 4, :lw * 4]).resize((lw, lh), Image.BICUBIC))
        Image.fromarray(hr).save(os.path.join(md, "DIV2K_valid_HR", f"{801 + i:04}.png"))
        Image.fromarray(lr).save(os.path.join(md, "DIV2K_valid_LR", f"{801 + i:04}x4.png"))
        with torch.no_grad():
            for name, (m, sd, dr) in models.items():
                sr = util.tensor2uint(m(util.uint2tensor4(lr, dr)), dr)
                res[name].append(util.calculate_psnr(sr, util.modcrop(hr, 4), border=4))
    with open(os.path.join(md, "reference_psnr.json"), "w") as f:
        json.dump({k: {"valid_psnr": v, "valid_ave_psnr": sum(v) / len(v)} for k, v in res.items()}, f, indent=1)
    print("golden vectors written to", GOLD)


if __name__ == "__main__":
    if "--riders-only" in sys.argv:           # refresh the free riders without touching the other fixtures
        _stub_cv2_torchvision()
        sys.path.insert(0, REF)
        os.chdir(REF)
        man = json.load(open(os.path.join(WDIR, "manifest.json")))
        write_riders(man)
        json.dump(man, open(os.path.join(WDIR, "manifest.json"), "w"), indent=1)
    else:
        main()
