"""How far the MFMA-bound shapes of cvcl_gemm are from the vendor library: torch.matmul (hipBLASLt / rocBLAS) on the same shapes.
Measurement aid only -- the product never calls the library."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multimodal-baby_amd"))
from multimodal import _hip as H
dev = torch.device("cuda:0")
shapes = [("l3.conv1", 50176, 512, 1024), ("l3.conv3", 50176, 1024, 512), ("l4.conv1", 12544, 1024, 2048), ("l4.conv3", 12544, 2048, 1024),
          ("l3.0.conv1", 200704, 512, 512), ("l4.0.conv1", 50176, 1024, 1024), ("even.3x512", 49152, 512, 1024), ("even.6x512", 49152, 1024, 512),
          ("vit.qkv", 50432, 2304, 768), ("vit.fc1", 50432, 3072, 768), ("vit.fc2", 50432, 768, 3072), ("4096^3", 4096, 4096, 4096), ("8192^3", 8192, 8192, 8192)]
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t_ours = timeit(lambda: H.gemm(a, w, out=out))
    wt = w.t()
    t_lib = timeit(lambda: torch.matmul(a, wt, out=out))
    fl = 2.0 * M * N * K
    print(f"{name:12s} M={M:6d} N={N:5d} K={K:5d}: cvcl {t_ours:8.1f} us {fl/t_ours/1e6:6.0f} TF | torch.matmul {t_lib:8.1f} us {fl/t_lib/1e6:6.0f} TF")
