"""GPU: cvcl_augment_frames (csrc/augment.hip through the C ABI) against the Pillow-pinned oracle -- bit-exact on the uint8
image and on the fp32 normalised tensor -- the Pillow golden fixture, and size-independent properties at the BASELINE batch."""
import zlib

import numpy as np
import pytest
import torch

import augment_oracle as A
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _run(frames_np, params, dev, size=(224, 224)):
    from multimodal.augment import DeviceFrameAugment, FrameParams
    aug = DeviceFrameAugment(size=size)
    crop = [p[:4] for p in params]
    out, out8 = aug(torch.from_numpy(frames_np).to(dev), FrameParams(crop, [p[4] for p in params], [p[5] for p in params]), return_uint8=True)
    return out.cpu().numpy(), out8.cpu().numpy()


def test_pillow_golden_bit_exact(dev):
    g = np.load(GOLDEN + "/augment_pil.npz")
    for i in range(int(g["n_cases"])):
        H, W, smooth = (int(v) for v in g[f"shape{i}"])
        frame = A.synthetic_frame(1000 + i, H, W, bool(smooth))
        top, left, h, w, flip = (int(v) for v in g[f"params{i}"])
        sigma = float(g[f"sigma{i}"])
        out, out8 = _run(frame[None], [(top, left, h, w, sigma, flip)], dev)
        assert zlib.crc32(out8[0].tobytes()) == int(g[f"final_crc{i}"]), f"case {i}"
        if f"final{i}" in g.files:
            assert np.array_equal(out8[0], g[f"final{i}"])
        if i == 1:
            assert np.array_equal(out[0][:, :16], g["tensor1_rows0_16"])
        # no blur, no flip: the resize alone
        r_out, r8 = _run(frame[None], [(top, left, h, w, 0.0, 0)], dev)
        assert zlib.crc32(r8[0].tobytes()) == int(g[f"resized_crc{i}"]), f"case {i}: resize"


@pytest.mark.parametrize("H,W,B", [(224, 224, 12), (240, 320, 6), (100, 75, 6), (460, 80, 3)])
def test_random_boxes_vs_oracle_bit_exact(dev, H, W, B):
    rng = np.random.default_rng(H + W)
    frames = np.stack([A.synthetic_frame(50 + i, H, W, smooth=bool(i % 2)) for i in range(B)])
    params = []
    for i in range(B):
        h, w = int(rng.integers(1, H + 1)), int(rng.integers(1, W + 1))
        if i == 0:
            h, w = H, W
        top, left = int(rng.integers(0, H - h + 1)), int(rng.integers(0, W - w + 1))
        params.append((top, left, h, w, float(rng.uniform(0.1, 2.0)) if i % 3 else 0.0, int(rng.integers(0, 2))))
    out, out8 = _run(frames, params, dev)
    for i, (top, left, h, w, sigma, flip) in enumerate(params):
        want = A.augment_frame(frames[i], top, left, h, w, sigma, flip)
        u8 = A.resized_crop_u8(frames[i], top, left, h, w)
        u8 = A.gaussian_blur_u8(u8, sigma) if sigma > 0 else u8
        u8 = u8[:, ::-1] if flip else u8
        assert np.array_equal(out8[i], u8), (i, params[i])
        assert np.array_equal(out[i], want), (i, params[i])                    # fp32, bit for bit


def test_other_output_size_vs_oracle(dev):
    frame = A.synthetic_frame(9, 150, 130)
    out, out8 = _run(frame[None], [(3, 4, 140, 120, 1.1, 1)], dev, size=(96, 128))
    want = A.augment_frame(frame, 3, 4, 140, 120, 1.1, 1, out_h=96, out_w=128)
    assert np.array_equal(out[0], want)


def test_full_batch_properties(dev):
    """BASELINE batch (256 frames of 224 x 224): identity draws == ToTensor + Normalize of the frames (torch's own fp32 CPU ops), flipping commutes with the transform, runs are deterministic, constants survive the blur."""
    import random
    from multimodal.augment import DeviceFrameAugment, FrameParams, IMAGENET_MEAN, IMAGENET_STD
    g = torch.Generator().manual_seed(0)
    frames = torch.randint(0, 256, (256, 224, 224, 3), dtype=torch.uint8, generator=g).to(dev)
    base = DeviceFrameAugment(augment_frames=False)
    got = base(frames)
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    want = (frames.cpu().permute(0, 3, 1, 2).float().div(255) - mean) / std          # torch's CPU (IEEE) ops, as the DataLoader workers run them
    assert torch.equal(got.cpu(), want)
    aug = DeviceFrameAugment(generator=torch.Generator().manual_seed(2))
    random.seed(2)
    p = aug.sample_params(256, 224, 224)
    a = aug(frames, p)
    assert torch.equal(a, aug(frames, p)) and torch.isfinite(a).all()
    flipped = aug(frames, FrameParams(p.crop, p.sigma, 1 - p.flip))
    assert torch.equal(flipped, a.flip(-1))
    # three frames spot-checked against the oracle inside the full batch (block-index arithmetic at B = 256)
    fcpu = frames.cpu().numpy()
    for i in (0, 131, 255):
        top, left, h, w = (int(v) for v in p.crop[i])
        assert np.array_equal(a[i].cpu().numpy(), A.augment_frame(fcpu[i], top, left, h, w, float(p.sigma[i]), int(p.flip[i])))
    const = torch.full((4, 224, 224, 3), 93, dtype=torch.uint8, device=dev)
    c = aug(const, FrameParams([[5, 9, 120, 100]] * 4, [0.1, 0.7, 1.3, 2.0], [0, 1, 0, 1]), return_uint8=True)[1]
    assert bool((c == 93).all())


def test_error_behaviour(dev):
    from multimodal import _hip as H
    from multimodal.augment import DeviceFrameAugment, FrameParams
    aug = DeviceFrameAugment()
    frames = torch.zeros(2, 224, 224, 3, dtype=torch.uint8, device=dev)
    with pytest.raises(ValueError):
        aug(frames, FrameParams([[0, 0, 225, 10], [0, 0, 5, 5]], [0, 0], [0, 0]))       # box outside the frame
    with pytest.raises(ValueError):
        aug(frames, FrameParams([[0, 0, 5, 5]], [0], [0]))                              # one row for two frames
    tall = torch.zeros(1, 700, 64, 3, dtype=torch.uint8, device=dev)
    with pytest.raises(H.CvclError):
        aug(tall, FrameParams([[0, 0, 700, 64]], [0], [0]))                             # does not fit the single-pass LDS plan
