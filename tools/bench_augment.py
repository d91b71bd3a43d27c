"""f3 probe: the device frame transform at the BASELINE batch (256 frames of 224 x 224) vs Pillow on the host cores.
python tools/bench_augment.py [--cpu-frames 64]"""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-baby_amd"))
from multimodal.augment import DeviceFrameAugment  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--cpu-frames", type=int, default=64)
a = ap.parse_args()
dev = torch.device("cuda:0")
frames = torch.randint(0, 256, (a.batch, 224, 224, 3), dtype=torch.uint8).to(dev)
aug = DeviceFrameAugment(generator=torch.Generator().manual_seed(0))
random.seed(0)
p = aug.sample_params(a.batch, 224, 224)
t0 = time.perf_counter()
for _ in range(10):
    p = aug.sample_params(a.batch, 224, 224)
t_draw = (time.perf_counter() - t0) / 10
for _ in range(3):
    aug(frames, p)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    aug(frames, p)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
byt = a.batch * (224 * 224 * 3 + 3 * 224 * 224 * 4)
print(f"device transform: {ms * 1e3:.0f} us per batch of {a.batch} ({a.batch / ms * 1e3:.0f} frames/s, {byt / ms / 1e6:.0f} GB/s of frame read + tensor "
      f"write); host draws {t_draw * 1e3:.1f} ms per batch")
try:
    from PIL import Image, ImageFilter
    torch.set_num_threads(1)                       # a DataLoader worker: one core per frame stream
    fr = frames[:a.cpu_frames].cpu().numpy()
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    t0 = time.perf_counter()
    for i in range(a.cpu_frames):
        top, left, h, w = (int(v) for v in p.crop[i])
        im = Image.fromarray(fr[i]).crop((left, top, left + w, top + h)).resize((224, 224), Image.BILINEAR)
        if float(p.sigma[i]) > 0:
            im = im.filter(ImageFilter.GaussianBlur(radius=float(p.sigma[i])))
        if int(p.flip[i]):
            im = im.transpose(Image.FLIP_LEFT_RIGHT)
        t = torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float().div(255).sub_(mean).div_(std)
    dt = time.perf_counter() - t0
    print(f"Pillow + torch on one host core: {dt / a.cpu_frames * 1e3:.2f} ms per frame ({a.cpu_frames / dt:.0f} frames/s)")
except ImportError:
    print("Pillow not importable: no host comparison")
