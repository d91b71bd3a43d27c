"""One bf16 GEMM shape, repeated: python tools/gemm_one.py M N K [iters] (for PMC passes: tools/pmc_gemm.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multimodal-baby_amd"))
from multimodal import _hip as H
M, N, K = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for _ in range(iters): H.gemm(a, w, out=out)
torch.cuda.synchronize()
