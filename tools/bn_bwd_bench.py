"""cvcl_bn_bwd (reduce + finalize + apply) on the trunk's shapes at B = 256, stand-alone: time and bytes per second.
bytes: mode 1 reads x, dy twice and writes dx (5 tensor passes); mode 2 reads x, out, dy twice and writes dx, g (8 passes)."""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "multimodal-baby_amd"))
from multimodal import _hip as H
dev = torch.device("cuda:0")
lib = H.lib()
for rows, C, mode in ((802816, 128, 1), (802816, 256, 2), (200704, 256, 1), (200704, 512, 2), (50176, 512, 1), (50176, 1024, 2), (12544, 2048, 2)):
    x = torch.randn(rows, C, device=dev).bfloat16(); out = torch.randn(rows, C, device=dev).bfloat16(); dy = torch.randn(rows, C, device=dev).bfloat16()
    scale = torch.rand(C, device=dev) + 0.5; shift = torch.randn(C, device=dev); mean = torch.randn(C, device=dev); rstd = torch.rand(C, device=dev) + 0.5
    gamma = torch.rand(C, device=dev) + 0.5
    prow = lib.cvcl_bn_bwd_partial_rows(H.BF16, rows, C)
    scratch = torch.empty(prow * 2 + 5, C, device=dev)
    dx = torch.empty_like(x); g = torch.empty_like(x) if mode == 2 else None
    def run():
        H.check(lib.cvcl_bn_bwd(H.BF16, mode, H.ptr(x), H.ptr(out), H.ptr(dy), H.ptr(scale), H.ptr(shift), H.ptr(mean), H.ptr(rstd), H.ptr(gamma),
                                H.ptr(scratch[prow * 2 + 3]), H.ptr(scratch[prow * 2 + 4]), H.ptr(dx), H.ptr(g), rows, C, H.ptr(scratch[:prow * 2]), prow,
                                H.ptr(scratch[prow * 2:prow * 2 + 3]), H.stream_ptr()), "bn_bwd")
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    passes = 5 if mode == 1 else 8
    mb = rows * C * 2 / 1e6
    print(f"[{rows:7d},{C:5d}] mode {mode}: {us:7.1f} us  {passes} x {mb:6.1f} MB -> {passes * mb / us / 1e3:5.2f} TB/s")
