"""CPU oracle of the training-time image transform -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product path
(multimodal-baby_amd/csrc/augment.hip behind cvcl_augment_frames) never does.

The reference composes (multimodal_data_module.py:244-256)

    RandomResizedCrop((224, 224), scale=(0.2, 1.))  ->  RandomApply([GaussianBlur([.1, 2.])], p=0.5)  (utils.py:94-103)
    ->  RandomHorizontalFlip()  ->  ToTensor()  ->  Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])  (:57)

on PIL images.  The pixel algorithms live in third-party dependencies that are unpinned in the reference's requirements.txt
(`Pillow`, `torchvision`, no versions) and absent from /root/reference:

* Pillow (12.2.0 in this image) -- `Image.crop`, `Image.resize(..., BILINEAR)` (libImaging/Resample.c: separable triangle
  filter widened by the down-scale factor, coefficients normalised in double and rounded to 22 fractional bits, horizontal
  pass then vertical pass, each rounded to uint8) and `ImageFilter.GaussianBlur(radius)` (libImaging/BoxBlur.c: three box
  blurs of a fractional radius per axis in 24-bit fixed point, all horizontal passes first).  Restated below from the
  published algorithm and PINNED against Pillow itself: tests/golden/augment_*.npz are outputs of Pillow run in this
  container (oracle/gen_golden_augment.py), and tests/test_augment_oracle.py also compares against the live library when
  it is importable.
* torchvision (not installed here) -- `RandomResizedCrop.get_params` (ten tries of area ~ U(scale) x log-uniform aspect
  ratio, centre-crop fallback), `RandomApply` / `RandomHorizontalFlip` (one `torch.rand(1)` each), `ToTensor` (uint8 / 255)
  and `Normalize` ((x - mean) / std in fp32).  Parameter SAMPLING is restated from torchvision's published algorithm with
  "parity unpinned" (no torchvision to run; the draws are random numbers, not pixel arithmetic); the pixel arithmetic of
  ToTensor / Normalize is pinned by construction (two IEEE fp32 operations).
"""
import math

import numpy as np

IMAGE_H = IMAGE_W = 224
MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)
PRECISION_BITS = 32 - 8 - 2


def synthetic_frame(seed, height, width, smooth=True):
    """Deterministic uint8 test frame [H][W][3] (numpy only, so fixtures need not store their inputs): low-frequency waves plus
    sparse +-1 noise, or uniform noise."""
    rng = np.random.default_rng(seed)
    if not smooth:
        return rng.integers(0, 256, (height, width, 3), dtype=np.uint8)
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    planes = []
    for c in range(3):
        fy, fx, ph = rng.uniform(0.01, 0.06), rng.uniform(0.01, 0.06), rng.uniform(0, 6.28)
        planes.append(127.5 + 80 * np.sin(fy * y + ph) * np.cos(fx * x + c) + 40 * np.sin(0.11 * (x + y) + ph * c))
    img = np.stack(planes, axis=2)
    img += rng.integers(-1, 2, img.shape) * (rng.random(img.shape) < 0.2)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


# ---- Pillow Resample.c: precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter -------------
def resample_coeffs(in_size, out_size):
    """Returns (bounds [out][2] = (first tap, tap count), coefficients [out][ksize] as int32 with 22 fractional bits)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        k = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            w = 1.0 - abs(t) if abs(t) < 1.0 else 0.0
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        for x in range(ksize):
            kk[xx, x] = int(k[x] * (1 << PRECISION_BITS) + (0.5 if k[x] >= 0 else -0.5))     # C cast truncates toward zero
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis0(img, out_size):
    """One resampling pass along axis 0 of a [n][...] uint8 array (rounded and clipped to uint8 as ImagingResample*_8bpc)."""
    bounds, kk = resample_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, xmax = bounds[xx]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(xmax):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resized_crop_u8(img, top, left, h, w, out_h=IMAGE_H, out_w=IMAGE_W):
    """torchvision F.resized_crop on a PIL image = Image.crop then Image.resize(BILINEAR): horizontal pass, then vertical pass
    (a pass whose size does not change is skipped by Pillow and is the identity anyway).  img: uint8 [H][W][3]."""
    crop = img[top:top + h, left:left + w]
    if w != out_w:
        crop = _resample_axis0(crop.transpose(1, 0, 2), out_w).transpose(1, 0, 2)
    if h != out_h:
        crop = _resample_axis0(crop, out_h)
    return np.ascontiguousarray(crop)


# ---- Pillow BoxBlur.c: ImagingGaussianBlur = 3 box blurs of a fractional radius per axis --------------------------------------
def gaussian_box_radius(radius, passes=3):
    """_gaussian_blur_radius: the box radius whose `passes`-fold application has the variance radius^2 (C `float` arithmetic except
    the square root and the floor, whose arguments carry double constants)."""
    f = np.float32
    radius = f(radius)
    sigma2 = radius * radius / f(passes)                                   # float
    L = f(math.sqrt(12.0 * float(sigma2) + 1.0))                           # double expression, stored to float
    l = f(math.floor((float(L) - 1.0) / 2.0))
    a = (f(2) * l + f(1)) * (l * (l + f(1)) - f(3) * sigma2)               # float
    a = a / (f(6) * (sigma2 - (l + f(1)) * (l + f(1))))
    return l + a


def box_params(float_radius):
    """ImagingHorizontalBoxBlur's fixed-point weights: ww per full pixel, fw per fractional edge pixel (24 fractional bits)."""
    float_radius = np.float32(float_radius)
    radius = int(float_radius)
    ww = int(np.float32(1 << 24) / (float_radius * np.float32(2) + np.float32(1)))
    fw = ((1 << 24) - (radius * 2 + 1) * ww) // 2
    return radius, ww, fw


def _box_blur_axis1(img, float_radius):
    """One horizontal box-blur pass over [rows][n][C] uint8: out[x] = (ww * sum_{|d| <= r} in[x + d] + fw * (in[x-r-1] + in[x+r+1])
    + 2^23) >> 24 with indices clamped to the line (edge replication), all in unsigned 32-bit integer arithmetic."""
    radius, ww, fw = box_params(float_radius)
    n = img.shape[1]
    src = img.astype(np.int64)
    idx = np.arange(n)
    acc = np.zeros_like(src)
    for d in range(-radius, radius + 1):
        acc += src[:, np.clip(idx + d, 0, n - 1)]
    far = src[:, np.clip(idx - radius - 1, 0, n - 1)] + src[:, np.clip(idx + radius + 1, 0, n - 1)]
    bulk = (acc * ww + far * fw) & 0xFFFFFFFF
    return (((bulk + (1 << 23)) & 0xFFFFFFFF) >> 24).astype(np.uint8)


def gaussian_blur_u8(img, sigma):
    """ImageFilter.GaussianBlur(radius=sigma) on an RGB uint8 image [H][W][3] (utils.py:100-103)."""
    r = gaussian_box_radius(sigma)
    out = img
    for _ in range(3):
        out = _box_blur_axis1(out, r)
    out = out.transpose(1, 0, 2)
    for _ in range(3):
        out = _box_blur_axis1(out, r)
    return np.ascontiguousarray(out.transpose(1, 0, 2))


def to_tensor_normalize(img_u8):
    """ToTensor + Normalize: uint8 [H][W][3] -> fp32 [3][H][W], (x / 255 - mean) / std with every operation in fp32."""
    x = img_u8.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    return ((x - MEAN[:, None, None]) / STD[:, None, None]).astype(np.float32)


def augment_frame(img, top, left, h, w, sigma, flip, out_h=IMAGE_H, out_w=IMAGE_W):
    """The composed transform for one frame with its random draws already made.  sigma <= 0: the blur was not applied."""
    x = resized_crop_u8(img, top, left, h, w, out_h, out_w)
    if sigma > 0:
        x = gaussian_blur_u8(x, sigma)
    if flip:
        x = x[:, ::-1]
    return to_tensor_normalize(np.ascontiguousarray(x))


# ---- torchvision RandomResizedCrop.get_params (published algorithm; parity unpinned: torchvision is not installed) -------------
def random_resized_crop_params(height, width, rng, scale=(0.2, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """rng: object with .uniform(a, b) and .randint(a, b_inclusive).  Returns (top, left, h, w)."""
    area = height * width
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = area * rng.uniform(scale[0], scale[1])
        aspect = math.exp(rng.uniform(log_ratio[0], log_ratio[1]))
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            top = rng.randint(0, height - h)
            left = rng.randint(0, width - w)
            return top, left, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w
