"""Micro-benchmark of cvcl_gemm (bf16) on the MFMA-bound shapes; run with CVCL_GEMM256=0 / 1 to compare the two kernels."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multimodal-baby_amd"))
from multimodal import _hip as H
dev = torch.device("cuda:0")
shapes = [("l3.conv1", 50176, 512, 1024, 0), ("l3.conv3", 50176, 1024, 512, 0), ("l4.conv1", 12544, 1024, 2048, 0),
          ("l4.conv3", 12544, 2048, 1024, 0), ("l3.0.conv1", 200704, 512, 512, 0), ("l4.0.conv1", 50176, 1024, 1024, 0),
          ("vit.qkv", 50432, 2304, 768, 1), ("vit.proj", 50432, 768, 768, 1), ("vit.fc1", 50432, 3072, 768, 2), ("vit.fc2", 50432, 768, 3072, 1),
          ("qkv.plain", 50432, 2304, 768, 0), ("fc1.plain", 50432, 3072, 768, 0), ("fc1.bias", 50432, 3072, 768, 3), ("proj.plain", 50432, 768, 768, 0),
          ("4096^3", 4096, 4096, 4096, 0), ("8192^3", 8192, 8192, 8192, 0)]
for name, M, N, K, kind in shapes:
    a = (torch.randn(M, K, device=dev)).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    bias = torch.randn(N, device=dev) if kind else None
    res = torch.randn(M, N, device=dev).bfloat16() if kind == 1 else None
    act = H.ACT_GELU if kind == 2 else H.ACT_NONE
    if kind == 3: res = None
    f = lambda: H.gemm(a, w, out=out, bias=bias, act=act, residual=res)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:12s} M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us  {2*M*N*K/us/1e6:7.0f} TFLOP/s")

print("fp8 (cvcl_gemm_fp8, bias only):")
for name, M, N, K in [("vit.qkv", 50432, 2304, 768), ("vit.proj", 50432, 768, 768), ("vit.fc1", 50432, 3072, 768), ("vit.fc2", 50432, 768, 3072),
                      ("4096^3", 4096, 4096, 4096), ("8192^3", 8192, 8192, 8192)]:
    a8 = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev); w8 = torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev)
    sa = torch.ones(M, device=dev); sw = torch.ones(N, device=dev); bias = torch.zeros(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    f = lambda: H.check(H.lib().cvcl_gemm_fp8(H.ptr(a8), H.ptr(sa), K, H.ptr(w8), H.ptr(sw), K, H.ptr(out), N, H.ptr(bias), 0, None, 0, M, N, K,
                                              H.stream_ptr()), "fp8")
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"{name:12s} M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us  {2*M*N*K/us/1e6:7.0f} TFLOP/s")

print("fp8 MX variants (vit.fc1 GELU -> e4m3 + e8m0 blocks; vit.fc2 / vit.proj with block-scaled A + residual):")
def _t(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
for name, M, N, K, act, res in [("vit.fc1", 50432, 3072, 768, 2, False), ("vit.fc2", 50432, 768, 3072, 0, True), ("vit.proj", 50432, 768, 768, 0, True)]:
    a8 = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev); w8 = torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev)
    sa = torch.ones(M, device=dev); sw = torch.full((N,), 1e-3, device=dev); bias = torch.zeros(N, device=dev)
    ab = torch.full((M, K // 32), 127, dtype=torch.uint8, device=dev)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev); c8 = torch.empty(M, N, dtype=torch.uint8, device=dev)
    cb = torch.empty(M, N // 32, dtype=torch.uint8, device=dev)
    R = H.ptr(out) if res else None
    L = H.lib()
    base = _t(lambda: H.check(L.cvcl_gemm_fp8(H.ptr(a8), H.ptr(sa), K, H.ptr(w8), H.ptr(sw), K, H.ptr(out), N, H.ptr(bias), act, R, N, M, N, K, H.stream_ptr()), "fp8"))
    mxa = _t(lambda: H.check(L.cvcl_gemm_fp8_mx(H.ptr(a8), None, H.ptr(ab), K, H.ptr(w8), H.ptr(sw), K, H.ptr(out), N, None, None, 0, H.ptr(bias), 0, R, N, M, N, K, H.stream_ptr()), "mxa")) if act == 0 else float("nan")
    mxo = _t(lambda: H.check(L.cvcl_gemm_fp8_mx(H.ptr(a8), H.ptr(sa), None, K, H.ptr(w8), H.ptr(sw), K, None, 0, H.ptr(c8), H.ptr(cb), N, H.ptr(bias), act, None, 0, M, N, K, H.stream_ptr()), "mxo")) if not res else float("nan")
    print(f"{name:12s} per-row in / bf16 out {base:7.1f} us | MX in {mxa:7.1f} us | MX out {mxo:7.1f} us")
