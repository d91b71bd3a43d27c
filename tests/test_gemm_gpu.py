"""GPU parity: the fused MFMA GEMM (cvcl_gemm) vs fp64 matmul of the oracle's storage-point model.
fp32 mode: exact-fp32 MFMA, tolerance 2e-5 rel.  bf16 mode: operands/outputs rounded to bf16 with
fp32 accumulation -> compare against the same rounding applied in fp64 math, tolerance 1 bf16 ulp."""
import pytest
import torch

import cvcl_oracle as O
from conftest import maxrel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from multimodal import _hip
    return _hip


def _ref(A, W, *, bias=None, act=0, residual=None, a_scale=None, a_shift=None, a_relu=False, scale=1.0, bf16=False):
    q = O.bf16_round if bf16 else (lambda t: t)
    a = A.double()
    if a_scale is not None:
        a = torch.addcmul(a_shift.double(), a, a_scale.double()) if False else a * a_scale.double() + a_shift.double()
        if a_relu:
            a = torch.relu(a)
        a = q(a.float()).double()
    y = a @ W.double().t() * scale
    if bias is not None:
        y = y + bias.double()
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = O.gelu_erf(y)
    y = q(y.float()).double()
    if residual is not None:
        y = q((y + residual.double()).float()).double()
    return y


SHAPES = [(128, 128, 64), (256, 256, 256), (300, 130, 72), (1, 5, 8), (1000, 512, 2048), (6272, 128, 64), (77, 2304, 768)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_gemm_plain(H, dev, M, N, K, dt):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5          # asymmetric operands: catches transposes
    bf = dt == "bf16"
    tdt = torch.bfloat16 if bf else torch.float32
    Aq, Wq = A.to(tdt), W.to(tdt)
    y = H.gemm(Aq.to(dev), Wq.to(dev))
    ref = _ref(Aq.float(), Wq.float(), bf16=bf)
    tol = 8e-3 if bf else 2e-5                              # bf16: 1 ulp = 2^-8 relative to the element
    assert y.shape == (M, N) and y.dtype == tdt
    err = (y.double().cpu() - ref).abs()
    assert float((err / (ref.abs() + ref.abs().max() * 5e-2)).max()) < tol


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_gemm_fused_everything(H, dev, dt):
    """BN+ReLU operand prologue, bias, GELU/ReLU, residual and the BatchNorm statistics epilogue."""
    M, N, K = 3136, 256, 128
    g = torch.Generator().manual_seed(11)
    bf = dt == "bf16"
    tdt = torch.bfloat16 if bf else torch.float32
    A = torch.randn(M, K, generator=g).to(tdt)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(tdt)
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
    bias = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g).to(tdt)
    for act in (0, 1, 2):
        rows = H.gemm_stats_rows(H.BF16 if bf else H.F32, M, N, K, prologue=True, a_relu=True, bias=True, residual=True, act=act)
        stats = torch.full((rows, 2, N), float("nan"), device=dev)
        y = H.gemm(A.to(dev), W.to(dev), bias=bias.to(dev), act=act, residual=R.to(dev), a_scale=sc.to(dev),
                   a_shift=sh.to(dev), a_relu=True, stats=stats)
        ref = _ref(A.float(), W.float(), bias=bias, act=act, residual=R.float(), a_scale=sc, a_shift=sh, a_relu=True, bf16=bf)
        err = (y.double().cpu() - ref).abs()
        assert float((err / (ref.abs() + ref.abs().max() * 5e-2)).max()) < (1.6e-2 if bf else 3e-5), act
        # statistics are those of the tensor as stored
        s = stats.double().sum(dim=0).cpu()
        ys = y.double().cpu()
        assert maxrel(s[0], ys.sum(dim=0)) < 1e-5 and maxrel(s[1], (ys * ys).sum(dim=0)) < 1e-5


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_gemm_strided_gather(H, dev, dt):
    """1x1 stride-2 convolution (Bottleneck downsample) = GEMM over gathered rows of an NHWC tensor."""
    B, Hi, Wi, Cin, Cout = 3, 14, 14, 64, 128
    g = torch.Generator().manual_seed(5)
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32
    x = torch.randn(B, Hi, Wi, Cin, generator=g).to(tdt)
    W = (torch.randn(Cout, Cin, generator=g) / 8).to(tdt)
    Ho, Wo = Hi // 2, Wi // 2
    y = H.gemm(x.to(dev), W.to(dev), M=B * Ho * Wo, gather=(Ho, Wo, Hi, Wi, 2))
    ref = _ref(x[:, ::2, ::2].reshape(-1, Cin).float(), W.float(), bf16=dt == "bf16")
    err = (y.double().cpu() - ref).abs()
    assert float((err / (ref.abs() + ref.abs().max() * 5e-2)).max()) < (8e-3 if dt == "bf16" else 2e-5)


def test_gemm_linearity_full_size(H, dev):
    """Size-independent property at a BASELINE-size shape (layer1 conv3: M = 256*56*56): G(a+b) = G(a)+G(b) in fp32."""
    M, N, K = 256 * 56 * 56 // 8, 256, 128
    g = torch.Generator(device="cpu").manual_seed(1)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    a = torch.randn(M, K, device=dev)
    b = torch.randn(M, K, device=dev)
    lhs = H.gemm(a + b, W)
    rhs = H.gemm(a, W) + H.gemm(b, W)
    assert maxrel(lhs, rhs) < 1e-5


def _gemm8w(H, epi, A, W, C=None, stats=None, bias=None, act=0, R=None):
    import ctypes as Cc
    a = H.GemmArgs()
    M, K = A.shape
    N = W.shape[0]
    a.A, a.W, a.C = H.ptr(A), H.ptr(W), H.ptr(C)
    a.M, a.N, a.K, a.lda, a.ldw, a.ldc = M, N, K, K, K, N
    if stats is not None:
        a.stats, a.stats_rows = H.ptr(stats), stats.shape[0]
    if bias is not None:
        a.bias = H.ptr(bias)
    a.act = act
    if R is not None:
        a.R, a.ldr = H.ptr(R), N
    H.check(H.lib().cvcl_gemm8w(epi, Cc.byref(a), H.stream_ptr()), "cvcl_gemm8w")


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1000, 512, 512), (12544, 2048, 1024), (50176, 512, 1024), (777, 768, 3072),
                                   (4099, 256, 2048), (200704, 256, 512), (50400, 1024, 256)])
def test_gemm8w_exact_on_small_integers_and_race_free(dev, M, N, K):
    """8-wave 256 (224) x 256 kernel with the 4-stage global_load_lds ring, conv epilogue: with small-integer operands every
    partial sum is exact, so C must equal the float64 product bit for bit -- any stage read before its loads landed or
    overwritten before every wave had read it, a wrong swizzle or a fragment mix-up shows up as a wrong integer.  Repeated
    launches (different timing) must agree; the BN statistics rows must sum to the column sums of what was stored; the
    statistics-only form (C = NULL) must give the same rows."""
    from multimodal import _hip as H
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randint(-2, 3, (M, K), generator=g).float()
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    ref = (a.double() @ w.double().t())
    ad, wd = a.bfloat16().to(dev), w.bfloat16().to(dev)
    rows = H.lib().cvcl_gemm8w_stats_rows(M, N)
    outs = []
    for _ in range(4):
        C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        st = torch.full((rows, 2, N), float("nan"), device=dev)
        _gemm8w(H, 0, ad, wd, C, st)
        outs.append((C, st))
    st_only = torch.full((rows, 2, N), float("nan"), device=dev)
    _gemm8w(H, 0, ad, wd, None, st_only)
    torch.cuda.synchronize()
    assert torch.equal(st_only, outs[0][1])
    C0, st0 = outs[0]
    assert all(torch.equal(C0, c) and torch.equal(st0, s) for c, s in outs[1:])
    want = ref.float().bfloat16()                                            # the kernel rounds the exact sum to bf16
    assert torch.equal(C0.cpu(), want)
    col = want.double().sum(0)
    assert torch.equal(st0[:, 0].double().sum(0).cpu(), col)
    assert torch.allclose(st0[:, 1].double().sum(0).cpu(), (want.double() ** 2).sum(0), rtol=1e-6)


@pytest.mark.parametrize("B,Hi,Cin,Cout", [(256, 28, 512, 1024), (200, 14, 1024, 2048), (37, 56, 256, 512)])
def test_gemm8w_strided_gather_exact(dev, B, Hi, Cin, Cout):
    """Downsample branch of Bottleneck x.0 (1x1, stride 2) on the 8-wave kernel: rows gathered from the NHWC input by (b, 2 oy,
    2 ox), recomputed per tile; small integers -> bit-exact product and statistics (ragged last tile included)."""
    from multimodal import _hip as H
    g = torch.Generator().manual_seed(B + Hi)
    x = torch.randint(-2, 3, (B, Hi, Hi, Cin), generator=g).float()
    w = torch.randint(-2, 3, (Cout, Cin), generator=g).float()
    Ho = Hi // 2
    M = B * Ho * Ho
    gather = (Ho, Ho, Hi, Hi, 2)
    rows = H.gemm_stats_rows(H.BF16, M, Cout, Cin, gather=gather)
    assert rows == H.lib().cvcl_gemm8w_stats_rows(M, Cout), "shape not routed to the 8-wave kernel"
    st = torch.full((rows, 2, Cout), float("nan"), device=dev)
    H.prof_enable(True)
    y = H.gemm(x.bfloat16().to(dev), w.bfloat16().to(dev), M=M, gather=gather, stats=st)
    torch.cuda.synchronize()
    prof = H.prof_collect()
    H.prof_enable(False)
    assert prof.get("gemm8w", (0, 0))[1] == 1, prof
    want = (x[:, ::2, ::2].reshape(M, Cin).double() @ w.double().t()).float().bfloat16()
    assert torch.equal(y.cpu(), want)
    assert torch.equal(st[:, 0].double().sum(0).cpu(), want.double().sum(0))


@pytest.mark.parametrize("M,N,K,act,res", [(1024, 768, 768, 0, True), (3000, 3072, 768, 2, False), (5000, 768, 3072, 0, True),
                                           (600, 2304, 768, 0, False), (50432, 768, 768, 0, True), (9000, 2304, 768, 1, False)])
def test_gemm8w_linear_epilogue(dev, M, N, K, act, res):
    """EPI 1 (bias + activation + residual, the ViT linears; flat tile order over all column tiles, bias vector in LDS): vs
    float64 of the same bf16 operands within 1 bf16 ulp, deterministic, and -- with small-integer operands -- exact."""
    from multimodal import _hip as H
    g = torch.Generator().manual_seed(N + K)
    a = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g).bfloat16() if res else None
    y = a.double() @ w.double().t() + bias.double()
    if act == 1:
        y = torch.relu(y)
    if act == 2:
        y = 0.5 * y * (1 + torch.erf(y / 2 ** 0.5))
    y = y.float().bfloat16().double()
    if res:
        y = y + r.double()
    ad, wd, bd = a.to(dev), w.to(dev), bias.to(dev)
    rd = r.to(dev) if res else None
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    _gemm8w(H, 1, ad, wd, C, None, bd, act, rd)
    C2 = torch.empty_like(C)
    _gemm8w(H, 1, ad, wd, C2, None, bd, act, rd)
    assert torch.equal(C, C2)
    assert maxrel(C.float(), y.float()) < 8e-3
    if act != 2:                                    # exact on small integers (GELU is not an integer map)
        ai = torch.randint(-2, 3, (M, K), generator=g).float()
        wi = torch.randint(-2, 3, (N, K), generator=g).float()
        bi = torch.randint(-3, 4, (N,), generator=g).float()
        ri = torch.randint(-3, 4, (M, N), generator=g).float() if res else None
        yi = ai.double() @ wi.double().t() + bi.double()
        if act == 1:
            yi = torch.relu(yi)
        yi = yi.float().bfloat16().double()
        if res:
            yi = (yi + ri.double()).float().bfloat16().double()
        Ci = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        _gemm8w(H, 1, ai.bfloat16().to(dev), wi.bfloat16().to(dev), Ci, None, bi.to(dev), act, ri.bfloat16().to(dev) if res else None)
        assert torch.equal(Ci.double().cpu(), yi)


def _quant_rows_ref(y):
    """oracle of cvcl_quant_rows_fp8: per-row scale amax/448, OCP e4m3 round-to-nearest-even (torch.float8_e4m3fn)."""
    amax = y.abs().amax(dim=1)
    s = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    q = (y / s[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn)
    return q, s


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (1000, 256, 768), (777, 768, 3072), (4096, 2304, 768)])
def test_gemm_fp8_exact_on_small_integers(dev, M, N, K):
    """e4m3 x e4m3 -> fp32 on the block-scaled MFMA with unit block scales: small integers are exact in e4m3 and their sums
    in fp32, so C must equal the (bf16-rounded) float64 product exactly -- pins the fragment / swizzle / scale plumbing."""
    from multimodal import _hip as H
    g = torch.Generator().manual_seed(M + K)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float()
    sa = 2.0 ** torch.randint(-3, 2, (M,), generator=g).float()
    sw = 2.0 ** torch.randint(-3, 2, (N,), generator=g).float()
    ref = ((a.double() @ w.double().t()) * sa.double()[:, None] * sw.double()[None, :]).float().bfloat16()
    a8, w8 = a.to(torch.float8_e4m3fn).to(dev), w.to(torch.float8_e4m3fn).to(dev)
    sad, swd = sa.to(dev), sw.to(dev)
    C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    for _ in range(2):
        H.check(H.lib().cvcl_gemm_fp8(a8.data_ptr(), H.ptr(sad), K, w8.data_ptr(), H.ptr(swd), K, H.ptr(C), N, None, 0, None, 0,
                                      M, N, K, H.stream_ptr()), "cvcl_gemm_fp8")
    assert torch.equal(C.cpu(), ref)


@pytest.mark.parametrize("M,N,K,act,res", [(1000, 768, 768, 0, True), (600, 3072, 768, 2, False), (900, 768, 3072, 0, True)])
def test_fp8_linear_pipeline_vs_emulation(dev, M, N, K, act, res):
    """quantise activations (per row) and weights (per output channel) with cvcl_quant_rows_fp8 -- bit-identical to the
    torch.float8_e4m3fn emulation -- then cvcl_gemm_fp8 with bias / GELU / residual vs float64 maths on the dequantised
    operands, and the end-to-end quantisation error vs the unquantised linear stays at the fp8 level."""
    from multimodal import _hip as H
    g = torch.Generator().manual_seed(N + K)
    x = (torch.randn(M, K, generator=g) * 1.7).bfloat16()
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g) * 0.1
    r = torch.randn(M, N, generator=g).bfloat16() if res else None
    xq_ref, xs_ref = _quant_rows_ref(x.float())
    wq_ref, ws_ref = _quant_rows_ref(w)
    xd, wd = x.to(dev), w.to(dev)
    xq, xs = torch.empty(M, K, dtype=torch.uint8, device=dev), torch.empty(M, device=dev)
    wq, ws = torch.empty(N, K, dtype=torch.uint8, device=dev), torch.empty(N, device=dev)
    H.check(H.lib().cvcl_quant_rows_fp8(H.BF16, H.ptr(xd), K, None, None, 0.0, H.ptr(xq), H.ptr(xs), M, K, H.stream_ptr()), "quant")
    H.check(H.lib().cvcl_quant_rows_fp8(H.F32, H.ptr(wd), K, None, None, 0.0, H.ptr(wq), H.ptr(ws), N, K, H.stream_ptr()), "quant")
    assert torch.equal(xs.cpu(), xs_ref) and torch.equal(xq.cpu(), xq_ref.view(torch.uint8))
    assert torch.equal(ws.cpu(), ws_ref) and torch.equal(wq.cpu(), wq_ref.view(torch.uint8))
    y = (xq_ref.double() @ wq_ref.double().t()) * xs_ref.double()[:, None] * ws_ref.double()[None, :] + bias.double()
    if act == 2:
        y = 0.5 * y * (1 + torch.erf(y / 2 ** 0.5))
    y = y.float().bfloat16().double()
    if res:
        y = y + r.double()
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    bd = bias.to(dev)
    rd = r.to(dev) if res else None
    H.check(H.lib().cvcl_gemm_fp8(H.ptr(xq), H.ptr(xs), K, H.ptr(wq), H.ptr(ws), K, H.ptr(C), N, H.ptr(bd), act, H.ptr(rd), N, M, N, K,
                                  H.stream_ptr()), "cvcl_gemm_fp8")
    assert maxrel(C.float(), y.float()) < 8e-3
    full = x.double() @ w.double().t() + bias.double()
    if act == 2:
        full = 0.5 * full * (1 + torch.erf(full / 2 ** 0.5))
    if res:
        full = full + r.double()
    rel = float((C.double().cpu() - full).norm() / full.norm())
    assert rel < 0.05, rel


def _mx_quant_ref(y):
    """oracle of the MX block quantisation: per 32 elements, scale 2^ceil(log2(amax/448)) (e8m0 byte = exponent + 127), e4m3 RNE."""
    M, K = y.shape
    blk = y.reshape(M, K // 32, 32)
    amax = blk.abs().amax(dim=2)
    x = (amax / 448.0).float()
    bits = x.view(torch.int32)
    e = (bits >> 23) & 0xff
    e = e + ((bits & 0x7fffff) != 0).int()
    e = e.clamp(1, 254)
    scale = torch.pow(2.0, (e - 127).double()).float()
    q = (blk / scale[:, :, None]).clamp(-448, 448).to(torch.float8_e4m3fn)
    return q.reshape(M, K), e.to(torch.uint8), scale


def _mx_tile_scales(e):
    """[M][K/32] block-scale bytes -> the C-ABI's [K/128][M][4] tiling."""
    M, nb = e.shape
    return e.reshape(M, nb // 4, 4).permute(1, 0, 2).contiguous()


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (1000, 768, 768), (700, 768, 3072)])
def test_gemm_fp8_mx_input_exact(dev, M, N, K):
    """A operand with per-32-element e8m0 block scales applied by the scaled MFMA: integers x powers of two stay exact."""
    from multimodal import _hip as H
    g = torch.Generator().manual_seed(M + K)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float()
    eb = torch.randint(124, 130, (M, K // 32), generator=g).to(torch.uint8)                 # block scales 2^-3 .. 2^2
    sw = 2.0 ** torch.randint(-2, 2, (N,), generator=g).float()
    a_eff = a.reshape(M, K // 32, 32) * torch.pow(2.0, (eb.double() - 127))[:, :, None]
    ref = ((a_eff.reshape(M, K).double() @ w.double().t()) * sw.double()[None, :]).float().bfloat16()
    a8, w8, ebd, swd = a.to(torch.float8_e4m3fn).to(dev), w.to(torch.float8_e4m3fn).to(dev), _mx_tile_scales(eb).to(dev), sw.to(dev)
    C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    H.check(H.lib().cvcl_gemm_fp8_mx(a8.data_ptr(), None, H.ptr(ebd), K, w8.data_ptr(), H.ptr(swd), K, H.ptr(C), N, None, None, 0, None, 0,
                                     None, 0, M, N, K, H.stream_ptr()), "cvcl_gemm_fp8_mx")
    assert torch.equal(C.cpu(), ref)


@pytest.mark.parametrize("act", [0, 2])
def test_gemm_fp8_mx_output_matches_block_quantiser(dev, act):
    """fc1-style epilogue: bias (+GELU) -> bf16 rounding -> per-32-column e8m0 scale + e4m3 bytes, vs the oracle quantiser
    applied to the bf16 output of the plain fp8 GEMM on the same operands (bit-identical bytes and scales)."""
    from multimodal import _hip as H
    M, N, K = 900, 3072, 768
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(M, K, generator=g) * 1.3).bfloat16()
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g) * 0.1
    xd, wd, bd = x.to(dev), w.to(dev), bias.to(dev)
    xq, xs = torch.empty(M, K, dtype=torch.uint8, device=dev), torch.empty(M, device=dev)
    wq, ws = torch.empty(N, K, dtype=torch.uint8, device=dev), torch.empty(N, device=dev)
    H.check(H.lib().cvcl_quant_rows_fp8(H.BF16, H.ptr(xd), K, None, None, 0.0, H.ptr(xq), H.ptr(xs), M, K, H.stream_ptr()), "quant")
    H.check(H.lib().cvcl_quant_rows_fp8(H.F32, H.ptr(wd), K, None, None, 0.0, H.ptr(wq), H.ptr(ws), N, K, H.stream_ptr()), "quant")
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    H.check(H.lib().cvcl_gemm_fp8(H.ptr(xq), H.ptr(xs), K, H.ptr(wq), H.ptr(ws), K, H.ptr(C), N, H.ptr(bd), act, None, 0, M, N, K,
                                  H.stream_ptr()), "cvcl_gemm_fp8")
    c8 = torch.empty(M, N, dtype=torch.uint8, device=dev)
    cb = torch.empty(N // 128, M, 4, dtype=torch.uint8, device=dev)
    H.check(H.lib().cvcl_gemm_fp8_mx(H.ptr(xq), H.ptr(xs), None, K, H.ptr(wq), H.ptr(ws), K, None, 0, H.ptr(c8), H.ptr(cb), N, H.ptr(bd), act,
                                     None, 0, M, N, K, H.stream_ptr()), "cvcl_gemm_fp8_mx")
    q_ref, e_ref, _ = _mx_quant_ref(C.float().cpu())
    assert torch.equal(cb.cpu(), _mx_tile_scales(e_ref)) and torch.equal(c8.cpu(), q_ref.view(torch.uint8))


def test_gemm_training_gelu_epilogues_match_separate_passes(dev):
    """The ViT-MLP training epilogues of cvcl_gemm: (a) act = GELU with C_pre also stores the pre-activation u and C = gelu(u);
    (b) G multiplies the product by gelu'(G) (data gradient through the GELU).  Both equal, bit for bit, the plain GEMM followed
    by the standalone cvcl_gelu_bf16 pass, and gelu / gelu' agree with torch's erf GELU in float64."""
    from multimodal import _hip as H
    M, N, K = 1000, 512, 256
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).bfloat16().to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().to(dev)
    bias = (torch.randn(N, generator=g) * 0.2).to(dev)
    u_ref = H.gemm(x, w, bias=bias)
    g_ref = torch.empty_like(u_ref)
    H.check(H.lib().cvcl_gelu_bf16(H.ptr(u_ref), None, H.ptr(g_ref), u_ref.numel(), H.stream_ptr()), "gelu")
    u = torch.empty_like(u_ref)
    gg = H.gemm(x, w, bias=bias, act=H.ACT_GELU, pre_out=u)
    assert torch.equal(u, u_ref) and torch.equal(gg, g_ref)
    exact = torch.nn.functional.gelu(u_ref.double().cpu())
    assert float((g_ref.double().cpu() - exact).abs().max()) < 2e-2 * float(exact.abs().max())
    # (b): dY [M, N2] @ W2 [N2 -> N] times gelu'(u)
    N2 = 128
    dy = torch.randn(M, N2, generator=g).bfloat16().to(dev)
    w2t = (torch.randn(N, N2, generator=g) / N2 ** 0.5).bfloat16().to(dev)          # the "W^T copy": [N][N2]
    dg = H.gemm(dy, w2t)
    du_ref = torch.empty_like(dg)
    H.check(H.lib().cvcl_gelu_bf16(H.ptr(u_ref), H.ptr(dg), H.ptr(du_ref), dg.numel(), H.stream_ptr()), "gelu bwd")
    du = H.gemm(dy, w2t, gelu_grad_of=u_ref)
    assert torch.equal(du, du_ref)
    uu = u_ref.double().cpu().requires_grad_(True)
    torch.nn.functional.gelu(uu).backward(dg.double().cpu())
    err = float((du_ref.double().cpu() - uu.grad).abs().max() / uu.grad.abs().max())
    assert err < 1e-2, err


def _gemm_args(H, A, W, C=None, stats=None, a_scale=None, a_shift=None, c_scale=None, c_shift=None, R=None, r_scale=None, r_shift=None):
    a = H.GemmArgs()
    M, K = A.shape
    N = W.shape[0]
    a.A, a.W, a.C = H.ptr(A), H.ptr(W), H.ptr(C)
    a.M, a.N, a.K, a.lda, a.ldw, a.ldc = M, N, K, K, K, N
    if a_scale is not None:
        a.a_scale, a.a_shift, a.a_relu = H.ptr(a_scale), H.ptr(a_shift), 1
    if stats is not None:
        a.stats, a.stats_rows = H.ptr(stats), stats.shape[0]
    if c_scale is not None:
        a.c_scale, a.c_shift, a.act = H.ptr(c_scale), H.ptr(c_shift), H.ACT_RELU
        a.R, a.ldr = H.ptr(R), N
        if r_scale is not None:
            a.r_scale, a.r_shift = H.ptr(r_scale), H.ptr(r_shift)
    return a


@pytest.mark.parametrize("M,N,K", [(200704, 256, 256), (131073, 512, 128), (802816, 256, 256)])
def test_gemm_pro_plain_operand_streams_conv1_of_layer2(dev, M, N, K):
    """Round 6: a plain bf16 product with K = 128 | 256, N % 256 == 0 and M >= 2^17 rows (conv1 of ResNeXt layer2.0 at B = 256 is the
    last shape: reference torchvision Bottleneck.conv1 reached from multimodal/multimodal.py:101) is routed to the W-in-registers
    streaming kernel of gemm_pro.hip with the operand multiplied as stored.  Integer operands: output and column statistics exact,
    partial rows and accumulators alike (below 2^17 rows the dispatcher keeps the tiled kernels)."""
    import ctypes as Cc
    from multimodal import _hip as H
    g = torch.Generator().manual_seed(M % 977 + N + K)
    A = torch.randint(-2, 3, (M, K), generator=g).to(torch.bfloat16).to(dev)
    W = torch.randint(-1, 2, (N, K), generator=g).to(torch.bfloat16).to(dev)
    probe = _gemm_args(H, A, W, torch.empty(1, device=dev), torch.empty(1, 2, N, device=dev))
    assert H.lib().cvcl_gemm_pro_supported(Cc.byref(probe)) == 1
    rows = H.gemm_stats_rows(H.BF16, M, N, K)
    assert rows == H.lib().cvcl_gemm_pro_stats_rows(M, N), "shape not routed to the streaming kernel"
    st = torch.full((rows, 2, N), float("nan"), device=dev)
    C = H.gemm(A, W, stats=st)
    acc = torch.zeros(8, 2, N, dtype=torch.int64, device=dev)
    C2 = H.gemm(A, W, stats_acc=acc)
    torch.cuda.synchronize()
    step = 65536
    for i0 in range(0, M, step):                              # (float64 reference in slices: 800 k x 256 doubles at a time)
        ref = (A[i0:i0 + step].double() @ W.double().t()).float().to(torch.bfloat16)
        assert torch.equal(C[i0:i0 + step], ref) and torch.equal(C2[i0:i0 + step], ref)
    cd = C.double()
    want = torch.stack((cd.sum(0), (cd * cd).sum(0)))
    assert torch.equal(st.double().sum(0), want)
    assert torch.equal(acc.sum(0).double() / 2.0 ** 24, want)


@pytest.mark.parametrize("M,N,K", [(12544, 256, 128), (5000, 512, 256), (63, 256, 128), (100352, 256, 128), (7777, 768, 256)])
def test_gemm_pro_bn_relu_operand_three_epilogues(dev, M, N, K):
    """cvcl_gemm with the producer's BatchNorm + ReLU on the operand and K = 128 | 256 (conv3 of ResNeXt layers 1-2 on the raw
    grouped-convolution output) runs the bandwidth-bound kernel of gemm_pro.hip: statistics only / store + statistics /
    Bottleneck tail, against float64 of the storage-point model (operand rounded to bf16 after the affine + ReLU, output rounded,
    statistics of the rounded output), ragged M included; deterministic."""
    import ctypes as Cc
    from multimodal import _hip as H
    g = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 2 + 0.5).bfloat16()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.5
    act = torch.relu(a.float() * sc + sh).bfloat16()                          # what conv3 consumes (storage-point model)
    y = (act.double() @ w.double().t()).float().bfloat16()
    ad, wd, scd, shd = a.to(dev), w.to(dev), sc.to(dev), sh.to(dev)
    rows = H.lib().cvcl_gemm_pro_stats_rows(M, N)
    probe = torch.empty(rows, 2, N, device=dev)
    args = _gemm_args(H, ad, wd, None, probe, scd, shd)
    assert H.lib().cvcl_gemm_pro_supported(Cc.byref(args)) == 1
    assert rows == H.lib().cvcl_gemm_stats_rows(H.BF16, Cc.byref(args))
    # store + statistics
    C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    st = torch.full((rows, 2, N), float("nan"), device=dev)
    H.check(H.lib().cvcl_gemm(H.BF16, Cc.byref(_gemm_args(H, ad, wd, C, st, scd, shd)), H.stream_ptr()), "cvcl_gemm")
    err = (C.double().cpu() - y.double()).abs() / (y.double().abs() + float(y.abs().max()) * 5e-2)
    assert float(err.max()) < 8e-3
    s = st.double().sum(0).cpu()
    cd = C.double().cpu()
    assert maxrel(s[0], cd.sum(0)) < 1e-5 and maxrel(s[1], (cd * cd).sum(0)) < 1e-5
    # statistics only: same rows as sums of the same rounded outputs (other per-workgroup split is allowed: compare totals)
    st2 = torch.full((rows, 2, N), float("nan"), device=dev)
    H.check(H.lib().cvcl_gemm(H.BF16, Cc.byref(_gemm_args(H, ad, wd, None, st2, scd, shd)), H.stream_ptr()), "cvcl_gemm")
    s2 = st2.double().sum(0).cpu()
    assert maxrel(s2[0], cd.sum(0)) < 1e-5 and maxrel(s2[1], (cd * cd).sum(0)) < 1e-5
    st3 = torch.empty_like(st2)
    H.check(H.lib().cvcl_gemm(H.BF16, Cc.byref(_gemm_args(H, ad, wd, None, st3, scd, shd)), H.stream_ptr()), "cvcl_gemm")
    assert torch.equal(st2, st3)
    # Bottleneck tail: relu(bn3(round(acc)) + identity) and + normalised downsample branch
    cs, cb = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    rs, rb = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g).bfloat16()
    for with_ds in (False, True):
        idn = r.double() * rs.double() + rb.double() if with_ds else r.double()
        ref = torch.relu(cd * cs.double() + cb.double() + idn).float().bfloat16()
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        args = _gemm_args(H, ad, wd, out, None, scd, shd, cs.to(dev), cb.to(dev), r.to(dev),
                          rs.to(dev) if with_ds else None, rb.to(dev) if with_ds else None)
        keep = (args, )
        H.check(H.lib().cvcl_gemm(H.BF16, Cc.byref(args), H.stream_ptr()), "cvcl_gemm")
        e = (out.double().cpu() - ref.double()).abs() / (ref.double().abs() + float(ref.abs().max()) * 5e-2)
        assert float(e.max()) < 8e-3, with_ds


def test_gemm_pro_exact_on_small_integers(dev):
    """Identity affine + ReLU on integer operands: every product and partial sum is exact -> bit-exact output (pins the W-in-
    registers fragment layout, the A-tile swizzle and the double buffering across many tiles per workgroup)."""
    import ctypes as Cc
    from multimodal import _hip as H
    M, N, K = 40000, 512, 256
    g = torch.Generator().manual_seed(3)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    ref = (torch.relu(a).double() @ w.double().t()).float().bfloat16()
    one, zero = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    ad, wd = a.bfloat16().to(dev), w.bfloat16().to(dev)
    rows = H.lib().cvcl_gemm_pro_stats_rows(M, N)
    outs = []
    for _ in range(3):
        C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        st = torch.full((rows, 2, N), float("nan"), device=dev)
        H.check(H.lib().cvcl_gemm(H.BF16, Cc.byref(_gemm_args(H, ad, wd, C, st, one, zero)), H.stream_ptr()), "cvcl_gemm")
        outs.append((C, st))
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0][0], c) and torch.equal(outs[0][1], s) for c, s in outs[1:])
    assert torch.equal(outs[0][0].cpu(), ref)
    assert torch.equal(outs[0][1][:, 0].double().sum(0).cpu(), ref.double().sum(0))


@pytest.mark.parametrize("kernel,M,N,K", [("gemm_glds", 6272, 128, 64), ("gemm8w", 50176, 512, 512), ("gemm8w", 12544, 2048, 512),
                                          ("gemm_pro", 12544, 256, 128), ("gemm_pro", 5000, 512, 256), ("generic_f32", 300, 130, 72)])
def test_gemm_centred_storage(dev, kernel, M, N, K):
    """Centred storage of a raw convolution output (include/cvcl_hip.h "Centred storage"): with args.centre the convolution
    epilogues store round(A'W^T - centre[n]), take the statistics of that, and the Bottleneck tail applies its affine to that --
    in every kernel the dispatcher can pick for a conv (128 x 128 direct-to-LDS, 8-wave, BN-prologue in its three modes, and
    the generic fp32 kernel).  Integer operands and integer centres: every product and partial sum is exact -> bit-exact."""
    import ctypes as Cc
    from multimodal import _hip as H
    g = torch.Generator().manual_seed(M + N)
    f32 = kernel == "generic_f32"
    tdt = torch.float32 if f32 else torch.bfloat16
    dt = H.F32 if f32 else H.BF16
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    cen = torch.randint(-40, 41, (N,), generator=g).float()
    pro = kernel == "gemm_pro"
    act = torch.relu(a) if pro else a
    y = (act.double() @ w.double().t() - cen.double()).float()
    ref = y.to(tdt)
    assert torch.equal(ref.float(), y) or not f32
    ad, wd, cd_ = a.to(tdt).to(dev), w.to(tdt).to(dev), cen.to(dev)
    one, zero = torch.ones(K, device=dev), torch.zeros(K, device=dev)

    def args(C, stats, tail=None):
        t = _gemm_args(H, ad, wd, C, stats, one if pro else None, zero if pro else None, *(tail or ()))
        t.centre = H.ptr(cd_)
        return t
    probe = args(torch.empty(1, dtype=tdt, device=dev), torch.empty(1, device=dev))
    rows = H.lib().cvcl_gemm_stats_rows(dt, Cc.byref(probe))
    if kernel == "gemm_pro":
        assert H.lib().cvcl_gemm_pro_supported(Cc.byref(probe)) == 1 and rows == H.lib().cvcl_gemm_pro_stats_rows(M, N)
    elif kernel == "gemm8w":
        assert rows == H.lib().cvcl_gemm8w_stats_rows(M, N)
    C = torch.full((M, N), float("nan"), dtype=tdt, device=dev)
    st = torch.full((rows, 2, N), float("nan"), device=dev)
    H.check(H.lib().cvcl_gemm(dt, Cc.byref(args(C, st)), H.stream_ptr()), "cvcl_gemm")
    assert torch.equal(C.cpu(), ref)
    s = st.double().sum(0).cpu()
    assert torch.equal(s[0], ref.double().sum(0)) and maxrel(s[1], (ref.double() ** 2).sum(0)) < 1e-6
    if f32:
        return
    if kernel in ("gemm_pro", "gemm8w"):                                       # statistics only (nothing stored)
        st2 = torch.full((rows, 2, N), float("nan"), device=dev)
        H.check(H.lib().cvcl_gemm(dt, Cc.byref(args(None, st2)), H.stream_ptr()), "cvcl_gemm")
        assert torch.equal(st2.double().sum(0).cpu()[0], ref.double().sum(0))
    if kernel in ("gemm_pro", "gemm_glds"):                                    # Bottleneck tail on the centred product
        cs, cb = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
        r = torch.randn(M, N, generator=g).bfloat16()
        want = torch.relu(torch.addcmul(cb, ref.float(), cs) + r.float()).bfloat16()      # fmaf(round(acc - c), cs, cb) + idn
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        t = args(out, None, (cs.to(dev), cb.to(dev), r.to(dev)))
        keep = (t,)
        H.check(H.lib().cvcl_gemm(dt, Cc.byref(t), H.stream_ptr()), "cvcl_gemm")
        e = (out.double().cpu() - want.double()).abs() / (want.double().abs() + float(want.abs().max()) * 5e-2)
        assert float(e.max()) < 8e-3
    with pytest.raises(H.CvclError):                                           # centre belongs to the convolution epilogues
        H.gemm(ad, wd, bias=torch.zeros(N, device=dev), centre=cd_)


@pytest.mark.parametrize("M", [12544, 5000, 63])
def test_gemm_pro_tail_with_recomputed_downsample(dev, M):
    """Bottleneck tail of a stage's first block with the downsample branch recomputed in the kernel (cvcl_gemm_args.A2 / W2,
    gemm_pro.hip PRO_TAIL_DS; torchvision Bottleneck.forward: out = relu(bn3(conv3(.)) + downsample(x)) of layer1.0):
    out = relu(round(relu(bn2(A)) W^T - c) cs + cb + round(X W2^T - c2) rs + rb) against (a) the SAME call with the branch stored
    by a separate cvcl_gemm and read through R -- bit for bit on integer operands -- and (b) float64 of the storage-point model
    on random operands."""
    import ctypes as Cc
    from multimodal import _hip as H
    N, K, K2 = 256, 128, 64
    g = torch.Generator().manual_seed(M)
    for exact in (True, False):
        if exact:
            a = torch.randint(-3, 4, (M, K), generator=g).float()
            w = torch.randint(-2, 3, (N, K), generator=g).float()
            x = torch.randint(-3, 4, (M, K2), generator=g).float()
            w2 = torch.randint(-2, 3, (N, K2), generator=g).float()
            sc, sh = torch.ones(K), torch.zeros(K)
            cen, cen2 = torch.randint(-8, 9, (N,), generator=g).float(), torch.randint(-8, 9, (N,), generator=g).float()
        else:
            a = torch.randn(M, K, generator=g) * 2 + 0.5
            w = torch.randn(N, K, generator=g) / K ** 0.5
            x = torch.randn(M, K2, generator=g)
            w2 = torch.randn(N, K2, generator=g) / K2 ** 0.5
            sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.5
            cen, cen2 = torch.randn(N, generator=g), torch.randn(N, generator=g)
        cs, cb = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
        rs, rb = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
        ab, wb, xb, w2b = a.bfloat16(), w.bfloat16(), x.bfloat16(), w2.bfloat16()
        act = torch.relu(ab.float() * sc + sh).bfloat16()
        y3 = (act.double() @ wb.double().t() - cen.double()).float().bfloat16()
        yd = (xb.double() @ w2b.double().t() - cen2.double()).float().bfloat16()
        want = torch.relu(torch.addcmul(cb, y3.float(), cs) + torch.addcmul(rb, yd.float(), rs)).bfloat16()
        d = {k: v.to(dev) for k, v in dict(a=ab, w=wb, x=xb, w2=w2b, sc=sc, sh=sh, cen=cen, cen2=cen2, cs=cs, cb=cb, rs=rs, rb=rb).items()}
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        t = _gemm_args(H, d["a"], d["w"], out, None, d["sc"], d["sh"], d["cs"], d["cb"], None, d["rs"], d["rb"])
        t.R = None
        t.centre = H.ptr(d["cen"])
        t.A2, t.W2, t.K2, t.lda2, t.ldw2, t.centre2 = H.ptr(d["x"]), H.ptr(d["w2"]), K2, K2, K2, H.ptr(d["cen2"])
        assert H.lib().cvcl_gemm_pro_supported(Cc.byref(t)) == 1
        H.check(H.lib().cvcl_gemm(H.BF16, Cc.byref(t), H.stream_ptr()), "cvcl_gemm")
        # the stored form: downsample through a separate cvcl_gemm, then the tail reading it through R
        rd = H.gemm(d["x"], d["w2"], centre=d["cen2"])
        out2 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        t2 = _gemm_args(H, d["a"], d["w"], out2, None, d["sc"], d["sh"], d["cs"], d["cb"], rd, d["rs"], d["rb"])
        t2.centre = H.ptr(d["cen"])
        H.check(H.lib().cvcl_gemm(H.BF16, Cc.byref(t2), H.stream_ptr()), "cvcl_gemm")
        torch.cuda.synchronize()
        if exact:
            assert torch.equal(rd.cpu(), yd) and torch.equal(out.cpu(), out2.cpu())
        # random operands: a product whose bf16 rounding falls the other way (fp32 vs float64 accumulation) moves a term by one
        # ulp of ITS magnitude, which the sum of two large terms of opposite sign does not bound -- so: all but a few elements
        # within one bf16 ulp of the result, and every element within one ulp of the larger term
        mag = (y3.float().abs() * cs + yd.float().abs() * rs + want.float().abs()).double()       # the magnitudes that were rounded
        for got, ref in ((out.cpu(), want), (out.cpu(), out2.cpu())):
            err = (got.double() - ref.double()).abs()
            assert float((err / (ref.double().abs() + float(ref.abs().max()) * 5e-2) > 8e-3).double().mean()) < 2e-3, exact
            assert float((err / (mag * 2.0 ** -7 + 1e-3)).max()) < 1.5, exact


# K-major operands + fused row sums (round 5): the gradient GEMMs of the trainable tail read dY / X / W as they lie.  Shapes cover
# both kernels behind the flag (split-K VALU kernel for the small products, the 128 x 128 fp32 MFMA kernel for the large ones and
# whenever row sums are asked for), ragged edges, and extents that are not multiples of 4 (scalar operand path).
@pytest.mark.parametrize("M,N,K", [(256, 512, 256), (512, 512, 1280), (1536, 512, 1280), (130, 72, 300), (7, 5, 12), (513, 130, 77),
                                   (2048, 512, 4096)])
@pytest.mark.parametrize("ta,tw", [(True, True), (False, True), (True, False)])
def test_gemm_f32_kmajor_operands_and_rowsum(H, dev, M, N, K, ta, tw):
    g = torch.Generator().manual_seed(M * 3 + N * 5 + K * 7 + ta * 2 + tw)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    Ad = (A.t().contiguous() if ta else A).to(dev)              # [K, M] in memory when ta
    Wd = (W.t().contiguous() if tw else W).to(dev)              # [K, N] in memory when tw
    rs = torch.full((M,), float("nan"), device=dev) if ta else None
    y = H.gemm(Ad, Wd, a_trans=ta, w_trans=tw, a_rowsum=rs, bias=bias.to(dev))
    ref = A.double() @ W.double().t() + bias.double()
    assert y.shape == (M, N)
    assert maxrel(y.cpu(), ref) < 2e-5
    if ta:
        ref_rs = A.double().sum(1)
        assert float((rs.double().cpu() - ref_rs).abs().max()) < 2e-5 * float(A.abs().sum(1).max())
        y2 = H.gemm(Ad, Wd, a_trans=ta, w_trans=tw, bias=bias.to(dev))          # without the row sums: the same product
        assert maxrel(y2.cpu(), ref) < 2e-5


def test_linear_backward_reads_operands_in_place(dev):
    """nn.Linear backward through ops.linear_backward (dW = dY^T X with the bias gradient fused, dX = dY W) at the text
    transformer's shapes (reference multimodal/multimodal.py:553-573: d_model 512, FFN 2048, B L = 1280 rows) vs float64."""
    from multimodal import ops
    g = torch.Generator().manual_seed(11)
    for M, K, N in [(1280, 512, 1536), (1280, 2048, 512), (256, 2048, 512), (100, 24, 40)]:
        x, w, dy = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(M, N, generator=g)
        dx, dw, db = ops.linear_backward(x.to(dev), w.to(dev), dy.to(dev), (True, True, True))
        assert maxrel(dx.cpu(), dy.double() @ w.double()) < 2e-5
        assert maxrel(dw.cpu(), dy.double().t() @ x.double()) < 2e-5
        assert maxrel(db.cpu(), dy.double().sum(0)) < 2e-5
        _, dw2, db2 = ops.linear_backward(x.to(dev), w.to(dev), dy.to(dev), (False, True, False))
        assert db2 is None and maxrel(dw2.cpu(), dw.cpu()) < 2e-5     # (without the row sums the small-product kernel may run: another summation order)
        _, _, db3 = ops.linear_backward(x.to(dev), w.to(dev), dy.to(dev), (False, False, True))
        assert maxrel(db3.cpu(), dy.double().sum(0)) < 2e-5


@pytest.mark.parametrize("M,N,K", [(1280, 1536, 512), (1280, 512, 2048), (512, 2048, 1280), (333, 130, 77)])
@pytest.mark.parametrize("ta,tw", [(False, False), (True, True), (False, True)])
def test_gemm_f32_split_arithmetic(H, dev, M, N, K, ta, tw):
    """cvcl_gemm_args.f32_split: fp32 operands on the bf16 MFMA as hi + lo bf16 parts (hi.hi + hi.lo + lo.hi, fp32 accumulation) --
    the text transformer's linears and their gradients in the bf16 configurations.  ~2^-16 per product: two orders of magnitude
    closer to float64 than a bf16 GEMM of the same operands, and within 3e-5 of the result's scale."""
    g = torch.Generator().manual_seed(M + 3 * N + 5 * K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    Ad = (A.t().contiguous() if ta else A).to(dev)
    Wd = (W.t().contiguous() if tw else W).to(dev)
    rs = torch.empty(M, device=dev) if ta else None
    y = H.gemm(Ad, Wd, a_trans=ta, w_trans=tw, a_rowsum=rs, bias=bias.to(dev), split=True)
    ref = A.double() @ W.double().t() + bias.double()
    err = maxrel(y.cpu(), ref)
    bf = maxrel((A.bfloat16().double() @ W.bfloat16().double().t() + bias.double()), ref)
    exact = maxrel(H.gemm(Ad, Wd, a_trans=ta, w_trans=tw, bias=bias.to(dev)).cpu(), ref)
    print(f"split {err:.2e}  exact fp32 {exact:.2e}  bf16 operands {bf:.2e}")
    assert err < 3e-5 and err < bf / 50
    if ta:
        assert float((rs.double().cpu() - A.double().sum(1)).abs().max()) < 2e-5 * float(A.abs().sum(1).max())


@pytest.mark.parametrize("M,N,K", [(1280, 512, 512), (64, 64, 256)])
def test_gemm_f32_split_operands_beyond_the_bf16_range(H, dev, M, N, K):
    """A finite value that rounds to bf16 infinity (3.39e38 < |x| <= fp32 max: a sliver of the fp32 range) gets the largest finite
    bf16 as its hi part (hi = inf made lo = x - inf = -inf, and inf times the other operand's lo part NaN): the row comes
    out as the exact fp32 GEMM gives it -- finite where 3.4e38 w fits, +inf where it overflows -- never NaN; an inf / NaN operand
    stays non-finite; finite rows are untouched."""
    g = torch.Generator().manual_seed(K)
    A = torch.randn(M, K, generator=g)
    W = torch.rand(N, K, generator=g) + 0.5                          # positive weights: row 3 of the product is +inf, not inf - inf
    A[3].abs_()
    A[3, 5] = 3.4e38
    A[7, 1] = float("inf")
    A[9, 2] = float("nan")
    y = H.gemm(A.to(dev), W.to(dev), split=True).cpu()
    exact = H.gemm(A.to(dev), W.to(dev)).cpu()
    assert not bool(torch.isnan(y[3]).any()) and bool((y[3] > 1e38).all()) and not bool(torch.isnan(exact[3]).any())
    fin = torch.isfinite(exact[3]) & (exact[3] < 3.2e38)             # (away from the overflow edge the two arithmetics agree)
    assert int(fin.sum()) > 0 and maxrel(y[3][fin], exact[3][fin]) < 1e-4
    assert not bool(torch.isfinite(y[7]).any()) and not bool(torch.isfinite(y[9]).any())
    keep = torch.ones(M, dtype=torch.bool)
    keep[[3, 7, 9]] = False
    assert bool(torch.isfinite(y[keep]).all()) and maxrel(y[keep], exact[keep]) < 3e-5
