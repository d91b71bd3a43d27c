"""Training-time frame transform on the device.

Host-side mirror of the transform the reference composes in ``MultiModalDataModule.__init__``
(multimodal_data_module.py:244-256)::

    transforms.Compose([
        transforms.RandomResizedCrop((IMAGE_H, IMAGE_W), scale=(0.2, 1.)),
        transforms.RandomApply([GaussianBlur([.1, 2.])], p=0.5),          # utils.py:94-103
        transforms.RandomHorizontalFlip(),
        transforms.ToTensor(),
        normalizer,                                                       # :57
    ])

The reference applies it per frame to PIL images inside the DataLoader workers.  Here only the random DRAWS stay on the
host (a few numbers per frame, same distributions: ``sample_params`` draws a whole batch at once, ``sample_params_sequential``
uses the reference's generators -- torch's for the crop / apply / flip decisions, ``random`` for the blur sigma -- in its
per-frame order); every pixel operation runs in one launch of
``cvcl_augment_frames`` (csrc/augment.hip) on uint8 frames resident in HBM, bit-identical to Pillow / torchvision's
arithmetic.  There is no CPU pixel path: without the HIP library the call fails.
"""
import ctypes
import math
import random

import torch

from . import _hip as H

IMAGE_H = IMAGE_W = 224
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class FrameParams:
    """The per-frame random draws of one batch: crop boxes (top, left, h, w), blur sigmas (<= 0: not applied), flips."""

    def __init__(self, crop, sigma, flip):
        self.crop = torch.as_tensor(crop, dtype=torch.int32).reshape(-1, 4).contiguous()
        self.sigma = torch.as_tensor(sigma, dtype=torch.float32).reshape(-1).contiguous()
        self.flip = torch.as_tensor(flip, dtype=torch.int32).reshape(-1).contiguous()
        if not (self.crop.shape[0] == self.sigma.shape[0] == self.flip.shape[0]):
            raise ValueError("crop / sigma / flip must describe the same number of frames")

    def __len__(self):
        return self.crop.shape[0]

    @staticmethod
    def identity(n, height, width):
        """No crop, no blur, no flip: the reference's base transform (ToTensor + Normalize, :272-275) when H x W is the output size."""
        crop = torch.tensor([[0, 0, height, width]], dtype=torch.int32).repeat(n, 1)
        return FrameParams(crop, torch.zeros(n), torch.zeros(n, dtype=torch.int32))


def random_resized_crop_box(height, width, scale=(0.2, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), generator=None):
    """torchvision ``RandomResizedCrop.get_params``: up to ten (area, aspect) draws from torch's generator, centre crop otherwise.
    Returns (top, left, h, w)."""
    area = height * width
    log_ratio = torch.log(torch.tensor(ratio))
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1], generator=generator).item()
        aspect = torch.exp(torch.empty(1).uniform_(log_ratio[0].item(), log_ratio[1].item(), generator=generator)).item()
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            top = int(torch.randint(0, height - h + 1, size=(1,), generator=generator).item())
            left = int(torch.randint(0, width - w + 1, size=(1,), generator=generator).item())
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


class DeviceFrameAugment:
    """``augment(frames_u8)``: uint8 [B, H, W, 3] device frames -> fp32 [B, 3, 224, 224] normalised tensors.

    ``augment_frames=True`` draws the reference's random transform per frame; ``False`` is the base transform (ToTensor +
    Normalize, plus the resize to the output size if the frames are larger -- the reference's frames are stored at 224 x 224)."""

    def __init__(self, augment_frames=True, size=(IMAGE_H, IMAGE_W), scale=(0.2, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0),
                 blur_sigma=(0.1, 2.0), blur_p=0.5, flip_p=0.5, mean=IMAGENET_MEAN, std=IMAGENET_STD, generator=None):
        self.augment_frames = bool(augment_frames)
        self.size = (int(size[0]), int(size[1]))
        self.scale, self.ratio = tuple(scale), tuple(ratio)
        self.blur_sigma, self.blur_p, self.flip_p = tuple(blur_sigma), float(blur_p), float(flip_p)
        self.mean = (ctypes.c_float * 3)(*mean)
        self.std = (ctypes.c_float * 3)(*std)
        self.generator = generator

    def sample_params(self, n, height, width):
        """The draws for a batch of n frames, vectorised (one torch call per kind of draw instead of ~6 per frame, so the host
        keeps ahead of the device at batch 256): same distributions as the reference's per-frame transform -- ten (area,
        aspect) tries with the first fitting one kept and the centre-crop fallback, uniform box position, blur applied when
        rand <= p with sigma ~ U(blur_sigma), flip when rand < p -- but not the same random stream as
        ``sample_params_sequential`` (which the reference's 8 DataLoader workers do not share either)."""
        if not self.augment_frames:
            return FrameParams.identity(n, height, width)
        g = self.generator
        area = height * width * torch.empty(n, 10, dtype=torch.float64).uniform_(self.scale[0], self.scale[1], generator=g)
        log_ratio = torch.log(torch.tensor(self.ratio))
        aspect = torch.exp(torch.empty(n, 10, dtype=torch.float64).uniform_(log_ratio[0].item(), log_ratio[1].item(), generator=g))
        w = torch.round(torch.sqrt(area * aspect)).long()
        h = torch.round(torch.sqrt(area / aspect)).long()
        ok = (w > 0) & (w <= width) & (h > 0) & (h <= height)
        first = torch.where(ok.any(dim=1), ok.float().argmax(dim=1), torch.full((n,), -1, dtype=torch.long))
        rows = torch.arange(n)
        wsel, hsel = w[rows, first.clamp(min=0)], h[rows, first.clamp(min=0)]
        # fallback: the whole frame, cut to the nearest allowed aspect ratio, centred
        in_ratio = float(width) / float(height)
        if in_ratio < min(self.ratio):
            fw, fh = width, int(round(width / min(self.ratio)))
        elif in_ratio > max(self.ratio):
            fh, fw = height, int(round(height * max(self.ratio)))
        else:
            fw, fh = width, height
        fb = first < 0
        wsel = torch.where(fb, torch.full_like(wsel, fw), wsel)
        hsel = torch.where(fb, torch.full_like(hsel, fh), hsel)
        u = torch.rand(n, 2, dtype=torch.float64, generator=g)
        top = torch.minimum((u[:, 0] * (height - hsel + 1)).long(), height - hsel)          # uniform over 0 .. H - h
        left = torch.minimum((u[:, 1] * (width - wsel + 1)).long(), width - wsel)
        top = torch.where(fb, (height - hsel) // 2, top)
        left = torch.where(fb, (width - wsel) // 2, left)
        r = torch.rand(n, 3, generator=g)
        apply = ~(self.blur_p < r[:, 0])
        sigma = torch.where(apply, self.blur_sigma[0] + (self.blur_sigma[1] - self.blur_sigma[0]) * r[:, 1], torch.zeros(n))
        flip = (r[:, 2] < self.flip_p).int()
        return FrameParams(torch.stack([top, left, hsel, wsel], dim=1).int(), sigma, flip)

    def sample_params_sequential(self, n, height, width):
        """The reference's draws for n frames from the reference's generators in its per-frame order: crop box
        (RandomResizedCrop.get_params), apply? (RandomApply: skip when p < rand), sigma (random.uniform, only when applied),
        flip (rand < p)."""
        if not self.augment_frames:
            return FrameParams.identity(n, height, width)
        g = self.generator
        crop, sigma, flip = [], [], []
        for _ in range(n):
            crop.append(random_resized_crop_box(height, width, self.scale, self.ratio, g))
            skip = self.blur_p < torch.rand(1, generator=g).item()
            sigma.append(0.0 if skip else random.uniform(self.blur_sigma[0], self.blur_sigma[1]))
            flip.append(1 if torch.rand(1, generator=g).item() < self.flip_p else 0)
        return FrameParams(crop, sigma, flip)

    def __call__(self, frames, params=None, return_uint8=False):
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
            raise H.CvclError(f"expected uint8 [B, H, W, 3] frames, got {tuple(frames.shape)} {frames.dtype}")
        B, Hh, Ww, _ = frames.shape
        if params is None:
            params = self.sample_params(B, Hh, Ww)
        if len(params) != B:
            raise ValueError(f"{len(params)} parameter rows for {B} frames")
        c = params.crop
        if bool(((c[:, 2] < 1) | (c[:, 3] < 1) | (c[:, 0] < 0) | (c[:, 1] < 0) | (c[:, 0] + c[:, 2] > Hh) | (c[:, 1] + c[:, 3] > Ww)).any()):
            raise ValueError("crop box outside the frame")
        max_h = int(c[:, 2].max())
        frames = frames.contiguous()
        dev = frames.device
        crop_d, sigma_d, flip_d = (t.to(dev, non_blocking=True) for t in (params.crop, params.sigma, params.flip))
        oh, ow = self.size
        out = torch.empty(B, 3, oh, ow, dtype=torch.float32, device=dev)
        out8 = torch.empty(B, oh, ow, 3, dtype=torch.uint8, device=dev) if return_uint8 else None
        H.check(H.lib().cvcl_augment_frames(H.ptr(frames), B, Hh, Ww, H.ptr(crop_d), H.ptr(sigma_d), H.ptr(flip_d),
                                            ctypes.cast(self.mean, ctypes.c_void_p), ctypes.cast(self.std, ctypes.c_void_p), H.ptr(out),
                                            oh, ow, H.ptr(out8), max_h, H.stream_ptr()), "cvcl_augment_frames")
        return (out, out8) if return_uint8 else out
