"""Kernel times of ONE eval-mode encode_image call at B = $B (default 1), C2 bf16: run under rocprofv3 --kernel-trace --stats."""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-baby_amd"))
import bench
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 1))
lit, ve, _ = bench.build_model(os.environ.get("CFG", "c2"), dev, precision="bf16")
lit.eval()
img, tok, ln = bench.synthetic_batch_on_device(B, 0, dev)
with torch.no_grad():
    for _ in range(50):
        lit.model.encode_image(img)
torch.cuda.synchronize()
