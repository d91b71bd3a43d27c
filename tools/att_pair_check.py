"""Round 5: outputs of the attention entry points (bf16, MX, train + lse) on seeded inputs, as one digest line per case -- run twice with the
lab library (CVCL_ATT_PAIR=0 / 1) and diff: the paired-query-tile kernel must reproduce the one-tile-per-wave kernel bit for bit.
Also times cvcl_attention at the C4 shape."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "multimodal-baby_amd"))
from multimodal import _hip as H
dev = torch.device("cuda:0")


def digest(t):
    return hashlib.sha256(t.contiguous().cpu().view(torch.uint8).numpy().tobytes()).hexdigest()[:16]


for (B, T, heads) in ((3, 197, 12), (2, 65, 4), (2, 224, 2), (1, 100, 12), (2, 193, 6), (2, 33, 2)):
    D = heads * 64
    g = torch.Generator().manual_seed(B * 1000 + T)
    qkv = (torch.randn(B * T, 3 * D, generator=g) * 1.5).bfloat16().to(dev)
    out = torch.zeros(B * T, D, dtype=torch.bfloat16, device=dev)
    H.check(H.lib().cvcl_attention(H.BF16, H.ptr(qkv), None, H.ptr(out), B, T, heads, 64, 0.125, H.stream_ptr()), "cvcl_attention")
    out2 = torch.zeros_like(out)
    lse = torch.zeros(B * heads * T, device=dev)
    H.check(H.lib().cvcl_attention_train(H.ptr(qkv), H.ptr(out2), H.ptr(lse), B, T, heads, 64, 0.125, H.stream_ptr()), "cvcl_attention_train")
    o8 = torch.zeros(B * T, D, dtype=torch.uint8, device=dev)
    bs = torch.zeros((D // 128) * B * T * 4 if D % 128 == 0 else 4, dtype=torch.uint8, device=dev)
    dm = "-"
    if heads % 2 == 0:
        H.check(H.lib().cvcl_attention_mx(H.ptr(qkv), H.ptr(o8), H.ptr(bs), B, T, heads, 64, 0.125, H.stream_ptr()), "cvcl_attention_mx")
        dm = digest(o8) + digest(bs)
    torch.cuda.synchronize()
    # a float64 reference on a sample, so that "identical" cannot mean "identically wrong"
    q, k, v = [x.reshape(B, T, heads, 64).permute(0, 2, 1, 3).double() for x in qkv.cpu().float().split(D, dim=1)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).permute(0, 2, 1, 3).reshape(B * T, D)
    err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    print(f"B={B} T={T} heads={heads}: out {digest(out)} train {digest(out2)} lse {digest(lse)} mx {dm} err_vs_f64 {err:.2e}")
    assert err < 2e-2

if os.environ.get("ATT_TIME", "1") == "1":
    B, T, heads, D = 256, 197, 12, 768
    qkv = torch.randn(B * T, 3 * D, device=dev).bfloat16()
    out = torch.empty(B * T, D, dtype=torch.bfloat16, device=dev)
    o8 = torch.zeros(B * T, D, dtype=torch.uint8, device=dev); bs = torch.zeros((D // 128) * B * T * 4, dtype=torch.uint8, device=dev)
    for name, call in (("bf16", lambda: H.lib().cvcl_attention(H.BF16, H.ptr(qkv), None, H.ptr(out), B, T, heads, 64, 0.125, H.stream_ptr())),
                       ("mx", lambda: H.lib().cvcl_attention_mx(H.ptr(qkv), H.ptr(o8), H.ptr(bs), B, T, heads, 64, 0.125, H.stream_ptr()))):
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        print(f"# time {name} T=197 B=256: {e0.elapsed_time(e1) / 20 * 1000:.1f} us  (CVCL_ATT_PAIR={os.environ.get('CVCL_ATT_PAIR', 'default')})")
