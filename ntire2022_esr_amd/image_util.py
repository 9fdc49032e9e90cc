"""Host-side image conversion and metric helpers around the forward path.

Restates the `utils/utils_image.py` functions `test_demo.run()` calls (mkdir :73-83)
with PIL instead of cv2 (PNG/BMP decoding is lossless, so the arrays are identical):
  imread_uint :122-134   imsave :137-141   uint2tensor4 :190-193
  tensor2uint :204-208   modcrop :442-455  calculate_psnr :490-503   calculate_ssim / ssim :509-554
Pinned against the reference's own outputs in tests/test_harness.py (tests/golden/metrics.*) -- except
SSIM, which needs cv2 to run in the reference and is therefore PARITY-UNPINNED: it is checked against an
independent dense evaluation of the published formula instead.
"""
import math
import os

import numpy as np
import torch
from PIL import Image


def mkdir(path):
    """utils_image.mkdir (utils/utils_image.py:73-75; run() calls it on the save path, test_demo.py:411)."""
    os.makedirs(path, exist_ok=True)


def mkdirs(paths):
    """utils_image.mkdirs (utils/utils_image.py:78-83): one path or an iterable of paths."""
    if isinstance(paths, str):
        mkdir(paths)
    else:
        for path in paths:
            mkdir(path)


def imread_uint(path, n_channels=3):
    """utils_image.imread_uint (utils/utils_image.py:122-134) with PIL in place of cv2.

    n_channels = 3: `cv2.imread(path, IMREAD_UNCHANGED)`, then GRAY2RGB for a 2-D result, BGR2RGB otherwise -- HxWx3 RGB; a gray image
    becomes GGG; a palette is expanded; an alpha channel (RGBA, gray + alpha, palette + tRNS) is DROPPED, not composited (cvtColor's
    4 -> 3 channel form); a 1-bit image is 0 / 255; a 16-bit gray image stays uint16 (IMREAD_UNCHANGED keeps the depth, the reference then
    divides by 255 like any other input -- restated, not fixed).  16-bit RGB(A) PNGs would come back as uint16 from cv2; PIL can only
    decode them to 8 bits, so they raise instead of silently differing.
    n_channels = 1: `cv2.imread(path, 0)` -- HxWx1 gray (colour images: PIL's ITU-R 601 luma; libpng's conversion inside cv2 may differ
    by one level on some pixels: PARITY-UNPINNED, the path never takes this branch, test_demo.py:419)."""
    img = Image.open(path)
    raw = img.tile[0][3] if getattr(img, "tile", None) and isinstance(img.tile[0][3], str) else ""
    if n_channels == 1:
        return np.expand_dims(np.array(img.convert("L")), axis=2)
    if n_channels != 3:
        raise ValueError("n_channels must be 1 or 3")
    if img.mode in ("I;16", "I;16B", "I;16L", "I"):
        g = np.array(img)
        if g.min() < 0 or g.max() > 65535:
            raise NotImplementedError(f"{path}: 32-bit integer image")
        g = g.astype(np.uint16)
        return np.stack([g, g, g], axis=2)
    if "16" in raw and img.mode in ("RGB", "RGBA"):
        raise NotImplementedError(f"{path}: 16-bit RGB PNG (cv2.IMREAD_UNCHANGED returns uint16; PIL decodes 8 bits only)")
    if img.mode in ("L", "1"):
        g = np.array(img.convert("L"))
        return np.stack([g, g, g], axis=2)
    return np.array(img.convert("RGB"))


def imsave(img, img_path):
    img = np.squeeze(img)
    os.makedirs(os.path.dirname(os.path.abspath(img_path)), exist_ok=True)
    if str(img_path).lower().endswith(".png"):
        # cv2.imwrite's default PNG setting is compression level 1 (IMWRITE_PNG_COMPRESSION, "best speed"); PIL's default is 6
        # and 3x slower on a 2040x1356 image.  Lossless either way: the decoded array is identical.
        Image.fromarray(img).save(img_path, compress_level=1)
    else:
        Image.fromarray(img).save(img_path)


def uint2tensor4(img, data_range):
    """HWC uint8 -> 1xCxHxW fp32 scaled to [0, data_range] (division by 255/data_range in fp32)."""
    if img.ndim == 2:
        img = np.expand_dims(img, axis=2)
    return torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float().div(255. / data_range).unsqueeze(0)


def tensor2uint(img, data_range):
    """clamp(0, data_range) -> *255/data_range in fp32 -> round half to even -> uint8, HWC.
    Unlike the reference (which clamps its argument IN PLACE through views, utils_image.py:205) the
    input tensor is left untouched; the returned array is identical."""
    t = img.detach().squeeze().float().clamp(0, 1 * data_range).cpu().numpy()
    if t.ndim == 3:
        t = np.transpose(t, (1, 2, 0))
    return np.uint8((t * 255.0 / data_range).round())


def modcrop(img_in, scale):
    img = np.copy(img_in)
    if img.ndim == 2:
        h, w = img.shape
        return img[:h - h % scale, :w - w % scale]
    if img.ndim == 3:
        h, w, _ = img.shape
        return img[:h - h % scale, :w - w % scale, :]
    raise ValueError('Wrong img ndim: [{:d}].'.format(img.ndim))


def calculate_psnr(img1, img2, border=0):
    """RGB PSNR on uint8 arrays in [0,255]: crop `border`, fp64 MSE over all samples, inf when identical."""
    if not img1.shape == img2.shape:
        raise ValueError('Input images must have the same dimensions.')
    h, w = img1.shape[:2]
    a = img1[border:h - border, border:w - border].astype(np.float64)
    b = img2[border:h - border, border:w - border].astype(np.float64)
    mse = np.mean((a - b) ** 2)
    if mse == 0:
        return float('inf')
    return 20 * math.log10(255.0 / math.sqrt(mse))


def _gaussian_kernel(ksize=11, sigma=1.5):
    """cv2.getGaussianKernel(11, 1.5): exp(-(i-(k-1)/2)^2 / (2 sigma^2)), normalised to sum 1."""
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    g = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return g / g.sum()


def _ssim(img1, img2):
    """ssim() utils_image.py:536-554: 11x11 Gaussian window = outer(k, k), 'valid' region only (the
    reference crops filter2D's output by 5 px per side, so its border mode never matters), float64."""
    from scipy.ndimage import correlate1d
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a, b = img1.astype(np.float64), img2.astype(np.float64)
    k = _gaussian_kernel()

    def blur(x):
        y = correlate1d(correlate1d(x, k, axis=0, mode="nearest"), k, axis=1, mode="nearest")
        return y[5:-5, 5:-5]

    mu1, mu2 = blur(a), blur(b)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    s1 = blur(a ** 2) - mu1_sq
    s2 = blur(b ** 2) - mu2_sq
    s12 = blur(a * b) - mu1_mu2
    m = ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean()


def calculate_ssim(img1, img2, border=0):
    """calculate_ssim utils_image.py:509-533.  For a 3-channel image the reference evaluates ssim() on the
    WHOLE HxWx3 array three times and averages (its loop index is unused), i.e. one evaluation."""
    if not img1.shape == img2.shape:
        raise ValueError('Input images must have the same dimensions.')
    h, w = img1.shape[:2]
    img1 = img1[border:h - border, border:w - border]
    img2 = img2[border:h - border, border:w - border]
    if img1.ndim == 2:
        return _ssim(img1, img2)
    if img1.ndim == 3:
        if img1.shape[2] == 3:
            return np.array([_ssim(img1, img2) for _ in range(1)] * 3).mean()
        if img1.shape[2] == 1:
            return _ssim(np.squeeze(img1), np.squeeze(img2))
    raise ValueError('Wrong input image dimensions.')
