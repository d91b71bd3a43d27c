"""Timing of the grouped-conv kernel on the trunk's shapes.  Phase ablation is a build option of the library
(-DCVCL_GCONV_ABLATE=<bits>, see csrc/resnext.hip): build a variant with
    CVCL_EXTRA_FLAGS=-DCVCL_GCONV_ABLATE=2 CVCL_LIB_SUFFIX=_abl2 python multimodal-baby_amd/build.py
and run this script with CVCL_HIP_LIB=multimodal-baby_amd/lib/libcvcl_hip_abl2.so."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multimodal-baby_amd"))
from multimodal import _hip as H
dev = torch.device("cuda:0")
def run(B, S, C, stride):
    cg = C // 32
    x = torch.randn(B, S, S, C, device=dev).to(torch.bfloat16)
    w = torch.randn(C, cg, 3, 3, device=dev)
    nb = H.lib().cvcl_packed_weight_bytes(H.BF16, H.PACK_GCONV3, C, cg, 3)
    wp = torch.empty(nb, dtype=torch.uint8, device=dev)
    H.check(H.lib().cvcl_pack_conv_weight(H.BF16, H.PACK_GCONV3, H.ptr(w), H.ptr(wp), C, cg, 3, H.stream_ptr()), "pack")
    sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    So = (S - 1) // stride + 1
    y = torch.empty(B, So, So, C, dtype=torch.bfloat16, device=dev)
    rows = H.lib().cvcl_gconv3x3_stats_rows(H.BF16, B, S, S, C, stride)
    st = torch.empty(rows, 2, C, device=dev)
    def call():
        H.check(H.lib().cvcl_gconv3x3(H.BF16, H.ptr(x), H.ptr(sc), H.ptr(sh), H.ptr(wp), H.ptr(y), H.ptr(st), rows, None, B, S, S, C, 32, stride, H.stream_ptr()), "gconv")
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1000
print("lib%s  layer1(56,128,s1): %.0f us   layer2.0(56,256,s2): %.0f us   layer2(28,256,s1): %.0f us   layer3(14,512): %.0f us" % (
    os.path.basename(os.environ.get("CVCL_HIP_LIB", "")), run(256, 56, 128, 1), run(256, 56, 256, 2), run(256, 28, 256, 1), run(256, 14, 512, 1)))
