"""cvcl_gconv3x3_wgrad on the trunk's shapes at B = 256 (bf16): time per call; CVCL_GCONV_WGRAD_BAND=0 runs the tap-at-a-time form."""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "multimodal-baby_amd"))
from multimodal import _hip as H
dev = torch.device("cuda:0"); lib = H.lib(); B = 256
for C, S, stride in ((128, 56, 1), (256, 56, 2), (256, 28, 1), (512, 28, 2), (512, 14, 1), (1024, 14, 2), (1024, 7, 1)):
    So = (S - 1) // stride + 1
    x = torch.randn(B, S, S, C, device=dev).bfloat16(); dy = torch.randn(B, So, So, C, device=dev).bfloat16()
    dw = torch.empty(C, C // 32, 3, 3, device=dev)
    nb = lib.cvcl_gconv3x3_wgrad_workspace_bytes(B, S, S, C, stride)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    def run():
        H.check(lib.cvcl_gconv3x3_wgrad(H.ptr(x), H.ptr(dy), H.ptr(dw), B, S, S, C, 32, stride, H.ptr(ws), nb, H.stream_ptr()), "wgrad")
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    mb = (x.numel() + dy.numel()) * 2 / 1e6
    print(f"C {C:5d} {S:3d}x{S:<3d} stride {stride}: {e0.elapsed_time(e1) * 100:7.1f} us  (x + dy = {mb:6.1f} MB, one pass at 5.5 TB/s = {mb / 5.5:5.1f} us)")
