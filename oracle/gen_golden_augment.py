"""Generates tests/golden/augment_pil.npz: outputs of Pillow itself (the third-party library the reference's train transform
runs on, multimodal_data_module.py:244-256 / utils.py:94-103) for fixed frames and fixed random draws.  TEST INFRASTRUCTURE.

Run here (Pillow is importable in this container; version recorded in the fixture):  python oracle/gen_golden_augment.py
Each case applies  Image.crop -> Image.resize((224, 224), BILINEAR) -> [ImageFilter.GaussianBlur(sigma)] -> [FLIP_LEFT_RIGHT]
(what torchvision's RandomResizedCrop / RandomApply(GaussianBlur) / RandomHorizontalFlip do to a PIL image once their draws
are made); the inputs are oracle.synthetic_frame(seed) (numpy only, checksummed here), the outputs are stored in full for three
cases and as CRC-32 for all, and one case carries ToTensor + Normalize computed with torch's own fp32 operations."""
import os
import sys
import zlib

import numpy as np
import PIL
import torch
from PIL import Image, ImageFilter

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import augment_oracle as A  # noqa: E402  (only for synthetic_frame: the test inputs)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "augment_pil.npz")
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def main():
    # (H, W, smooth, (top, left, h, w), sigma (0 = no blur), flip)
    cases = [
        (224, 224, True, (0, 0, 224, 224), 0.0, 0),            # base transform: identity resize
        (224, 224, True, (31, 17, 150, 190), 1.37, 1),         # up-scaling crop, blur, flip
        (210, 224, False, (100, 120, 101, 100), 0.1, 0),       # area 0.2 crop of noise, smallest sigma
        (224, 200, True, (0, 3, 224, 167), 2.0, 1),            # full-height crop (vertical pass skipped), largest sigma
        (240, 300, True, (5, 20, 231, 275), 0.63, 0),          # down-scaling: widened filter support
        (120, 160, False, (7, 11, 1, 1), 0.0, 1),              # 1 x 1 crop
        (460, 80, True, (2, 5, 456, 70), 1.9, 0),              # 2x+ vertical down-scaling (7 taps), horizontal up-scaling
    ]
    out = {"pillow_version": np.array(PIL.__version__), "n_cases": np.array(len(cases))}
    for i, (H, W, smooth, box, sigma, flip) in enumerate(cases):
        img = A.synthetic_frame(1000 + i, H, W, smooth)
        top, left, h, w = box
        pil = Image.fromarray(img).crop((left, top, left + w, top + h)).resize((224, 224), Image.BILINEAR)
        resized = np.asarray(pil).copy()
        if sigma > 0:
            pil = pil.filter(ImageFilter.GaussianBlur(radius=sigma))
        if flip:
            pil = pil.transpose(Image.FLIP_LEFT_RIGHT)
        final = np.asarray(pil).copy()
        out[f"shape{i}"] = np.array([H, W, int(smooth)], dtype=np.int32)      # the frame is A.synthetic_frame(1000 + i, H, W, smooth)
        out[f"frame_crc{i}"] = np.array(zlib.crc32(img.tobytes()), dtype=np.uint32)
        out[f"params{i}"] = np.array([top, left, h, w, flip], dtype=np.int32)
        out[f"sigma{i}"] = np.array(sigma, dtype=np.float32)
        out[f"resized_crc{i}"] = np.array(zlib.crc32(resized.tobytes()), dtype=np.uint32)
        out[f"final_crc{i}"] = np.array(zlib.crc32(final.tobytes()), dtype=np.uint32)
        if i in (1, 4, 6):                                                    # full images for three cases, checksums for all
            out[f"final{i}"] = final
        if i == 1:      # ToTensor (uint8 -> float / 255, CHW) + Normalize (sub mean, div std) with torch's fp32 arithmetic
            t = torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
            t = t.sub_(torch.tensor(MEAN).view(3, 1, 1)).div_(torch.tensor(STD).view(3, 1, 1))
            out["tensor1_rows0_16"] = t[:, :16].contiguous().numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
