"""Experiment: consecutive frozen-trunk passes (independent: frozen weights, per-batch BN statistics) issued alternately on two
HIP streams with separate workspaces, so that the tail round / dependent-launch gaps / MFMA-bound phases of one pass are filled by
the other.  Timing only (the BN running-stat updates of the two streams are not ordered here)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multimodal-baby_amd"))
from multimodal import _hip as H
from multimodal.resnext import resnext50_32x4d, BN_MOMENTUM, BN_EPS
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 256)); steps = 40
m = resnext50_32x4d().to(dev); m.compute_dtype = torch.bfloat16; m.train()
x = torch.randn(B, 3, 224, 224, device=dev)
lib = H.lib(); dt = H.BF16
arr, keep = m._packed_layers(dt, dev)
nb = lib.cvcl_resnext50_workspace_bytes(dt, B, 224, 224)
def run(nstreams):
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    ws = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(nstreams)]
    fmap = [torch.empty(B, 7, 7, 2048, dtype=torch.bfloat16, device=dev) for _ in range(nstreams)]
    pooled = [torch.empty(B, 2048, device=dev) for _ in range(nstreams)]
    torch.cuda.synchronize()
    def step(i):
        s = i % nstreams
        with torch.cuda.stream(streams[s]):
            H.check(lib.cvcl_resnext50_fwd(dt, B, 224, 224, 1, H.ptr(x), arr, len(arr), H.ptr(ws[s]), nb, H.ptr(fmap[s]), H.ptr(pooled[s]),
                                           BN_MOMENTUM, BN_EPS, None, H.stream_ptr()), "fwd")
    for i in range(6): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for n in (1, 2, 1, 2, 3):
    print(f"streams={n}: {run(n):.3f} ms per trunk pass (B={B})")
