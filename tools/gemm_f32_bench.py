import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"), "multimodal-baby_amd"))
from multimodal import _hip as H
dev = torch.device("cuda:0")
for M,N,K in [(256,512,2048),(256,256,512),(512,2048,256),(1280,1536,512),(1280,512,512),(1280,2048,512),(1280,512,2048),(512,512,1280),(2048,512,1280),(1536,512,1280),(2048,2048,512)]:
    a=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev); out=torch.empty(M,N,device=dev)
    f=lambda: H.gemm(a,w,out=out)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"M={M} N={N} K={K}: {e0.elapsed_time(e1)/20*1e3:.1f} us")
