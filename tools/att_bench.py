"""Timing of cvcl_attention alone at the C4 shapes (B = 256, 12 heads x 64; 197 and 257 tokens), for phase-ablation builds of the library
(-DCVCL_ATT_ABLATE=<bits>, csrc/vit.hip) selected with CVCL_HIP_LIB."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multimodal-baby_amd"))
from multimodal import _hip as H
dev = torch.device("cuda:0")
for T in (197, 257):
    B, heads, D = 256, 12, 768
    qkv = torch.randn(B * T, 3 * D, device=dev).bfloat16()
    out = torch.empty(B * T, D, dtype=torch.bfloat16, device=dev)
    def call():
        H.check(H.lib().cvcl_attention(H.BF16, H.ptr(qkv), None, H.ptr(out), B, T, heads, 64, 0.125, H.stream_ptr()), "cvcl_attention")
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    print(os.path.basename(os.environ.get("CVCL_HIP_LIB", "(product)")), f"T={T}: {e0.elapsed_time(e1) / 20 * 1000:.1f} us")
