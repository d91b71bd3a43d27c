"""GPU parity: BatchNorm statistics of a 1x1 convolution's output from the Gram matrix of its operand (csrc/bn_gram.hip:
cvcl_conv1x1_gram + cvcl_bn_from_gram) -- torchvision Bottleneck.forward bn3(conv3(relu(bn2(.)))) / the downsample branch, reached from
multimodal/multimodal.py:101.  The Gram data are exact on small integers (incl. a ragged last tile and the BN + ReLU prologue); the
resulting (scale, shift) and running statistics agree with float64 BatchNorm of the explicitly formed product and with the
statistics-only GEMM pass + cvcl_bn_finalize they replace."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from multimodal import _hip
    return _hip


def _gram(H, dev, a, K, a_scale=None, a_shift=None, relu=False):
    M = a.shape[0]
    nb = H.lib().cvcl_conv1x1_gram_workspace_bytes(K)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    out = C.c_void_p()
    H.check(H.lib().cvcl_conv1x1_gram(H.ptr(a), a.shape[1], M, K, H.ptr(a_scale), H.ptr(a_shift), int(relu), H.ptr(ws), nb, C.byref(out),
                                      H.stream_ptr()), "cvcl_conv1x1_gram")
    torch.cuda.synchronize()
    off = out.value - ws.data_ptr()
    flat = ws[off:off + (K * K + K) * 8].view(torch.float64).cpu()
    G, s = flat[:K * K].view(K, K).clone(), flat[K * K:].clone()
    assert torch.equal(G, G.t())
    return G, s, ws, out


# (M = 64 / 130: most workgroups own no tile; 33000 / 50000 / 70001: whole rounds with a ragged last one -- workgroups with fewer tiles
# than rounds * NPF steps still meet every barrier of the producer / consumer protocol, for each K)
@pytest.mark.parametrize("M,K", [(64, 64), (64, 128), (130, 256), (1000, 128), (4099, 256), (50000, 128), (70001, 256), (33000, 64),
                                 (33000, 256), (70001, 64)])
@pytest.mark.parametrize("prologue", [False, True])
def test_gram_exact_on_small_integers(H, dev, M, K, prologue):
    g = torch.Generator().manual_seed(M + K)
    a = torch.randint(-3, 4, (M, K), generator=g).float().bfloat16()
    ad = a.to(dev)
    if prologue:                                   # a' = relu(2 a - 1): still small integers
        sc, sh = torch.full((K,), 2.0, device=dev), torch.full((K,), -1.0, device=dev)
        G, s, _, _ = _gram(H, dev, ad, K, sc, sh, relu=True)
        ap = torch.relu(a.double() * 2 - 1)
    else:
        G, s, _, _ = _gram(H, dev, ad, K)
        ap = a.double()
    assert torch.equal(s, ap.sum(0))
    assert torch.equal(G, ap.t() @ ap)
    G2, s2, _, _ = _gram(H, dev, ad, K, *((sc, sh, True) if prologue else ()))
    assert torch.equal(G, G2) and torch.equal(s, s2)          # deterministic


@pytest.mark.parametrize("M,K,N,centred,deferred", [(12544, 128, 256, True, False), (20000, 256, 512, False, True), (9000, 64, 256, True, True)])
def test_bn_from_gram_vs_float64_and_vs_the_statistics_pass(H, dev, M, K, N, centred, deferred):
    """y = relu(bn2(a)) W^T: BatchNorm(y) from the Gram data against float64 on the explicitly formed product, and against the route
    it replaces (statistics-only cvcl_gemm pass over the same operand + cvcl_bn_finalize)."""
    g = torch.Generator().manual_seed(K + N)
    a = (torch.randn(M, K, generator=g) * 1.5 + 0.3).bfloat16()
    a_scale, a_shift = 0.5 + torch.rand(K, generator=g), 0.2 * torch.randn(K, generator=g)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    gamma, beta = 0.75 + 0.5 * torch.rand(N, generator=g), 0.1 * torch.randn(N, generator=g)
    plain = K == 64                                 # (the downsample branch: no prologue)
    ad, wd = a.to(dev), w.to(dev)
    scd, shd = (None, None) if plain else (a_scale.to(dev), a_shift.to(dev))
    gd, bd = gamma.to(dev), beta.to(dev)
    # float64 reference on the bf16 operand the convolution multiplies
    ap = a.double() if plain else torch.relu(a.double() * a_scale.double() + a_shift.double()).float().bfloat16().double()
    y = ap @ w.double().t()
    mean, var = y.mean(0), y.var(0, unbiased=False)
    centre = (mean + 0.05 * torch.randn(N, generator=g).double()).float() if centred else None
    cd = centre.to(dev) if centred else None
    ref_scale = gamma.double() / torch.sqrt(var.float().double() + 1e-5)
    ref_shift = beta.double() - (mean - (centre.double() if centred else 0.0)) * ref_scale
    G, s, ws, out = _gram(H, dev, ad, K, scd, shd, relu=not plain)
    scale, shift = torch.empty(N, device=dev), torch.empty(N, device=dev)
    rm, rv, nbt = torch.zeros(N, device=dev), torch.ones(N, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    mom = torch.zeros(2, 2048, device=dev) if deferred else None
    H.check(H.lib().cvcl_bn_from_gram(out, K, M, H.ptr(wd), K, N, H.ptr(gd), H.ptr(bd), H.ptr(rm), H.ptr(rv), H.ptr(nbt), 0.1, 1e-5,
                                      H.ptr(scale), H.ptr(shift), H.ptr(mom), 2048, H.ptr(cd), H.stream_ptr()), "cvcl_bn_from_gram")
    torch.cuda.synchronize()
    assert float(((scale.double().cpu() - ref_scale).abs() / ref_scale.abs()).max()) < 2e-5
    assert float((shift.double().cpu() - ref_shift).abs().max()) < 2e-5 * float(ref_shift.abs().max() + 1)
    unb = var * M / (M - 1)
    if deferred:
        assert float((mom[0, :N].double().cpu() - mean).abs().max()) < 1e-5 * float(mean.abs().max() + 1)
        assert float(((mom[1, :N].double().cpu() - unb).abs() / unb).max()) < 2e-5
        assert int(nbt) == 0 and float(rm.abs().max()) == 0.0
    else:
        assert float((rm.double().cpu() - 0.1 * mean).abs().max()) < 1e-5 * float(mean.abs().max() + 1)
        assert float(((rv.double().cpu() - (0.9 + 0.1 * unb)).abs() / (0.9 + 0.1 * unb)).max()) < 2e-5
        assert int(nbt) == 1
    if not plain:
        # the route it replaces: statistics-only GEMM pass (sums of the ROUNDED outputs) + bn_finalize
        rows = H.gemm_stats_rows(H.BF16, M, N, K, prologue=True, a_relu=True)
        st = torch.empty(rows, 2, N, device=dev)
        a_ = H.GemmArgs()
        a_.A, a_.W, a_.C = H.ptr(ad), H.ptr(wd), None
        a_.M, a_.N, a_.K, a_.lda, a_.ldw, a_.ldc = M, N, K, K, K, N
        a_.a_scale, a_.a_shift, a_.a_relu = H.ptr(scd), H.ptr(shd), 1
        a_.stats, a_.stats_rows = H.ptr(st), rows
        a_.centre = H.ptr(cd)
        H.check(H.lib().cvcl_gemm(H.BF16, C.byref(a_), H.stream_ptr()), "cvcl_gemm statistics pass")
        s2, h2 = torch.empty(N, device=dev), torch.empty(N, device=dev)
        rm2, rv2, nbt2 = torch.zeros(N, device=dev), torch.ones(N, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
        H.check(H.lib().cvcl_bn_finalize(H.ptr(st), rows, M, H.ptr(gd), H.ptr(bd), H.ptr(rm2), H.ptr(rv2), H.ptr(nbt2), 0.1, 1e-5, H.ptr(s2),
                                         H.ptr(h2), H.ptr(cd), N, H.stream_ptr()), "cvcl_bn_finalize")
        torch.cuda.synchronize()
        # (the pass takes the statistics of the bf16-ROUNDED product: the two routes differ by the mean of the rounding errors)
        assert float(((scale - s2).abs() / s2.abs()).max()) < 2e-4
        assert float((shift - h2).abs().max()) < 2e-4 * float(h2.abs().max() + 1)


@pytest.mark.parametrize("ratio", [10.0, 30.0, 100.0])
@pytest.mark.parametrize("M,K,N", [(200704, 128, 512), (50176, 256, 1024)])
def test_bn_from_gram_where_the_variance_can_cancel(H, dev, M, K, N, ratio):
    """var = w^T G w / M - (w.s / M)^2 subtracts two numbers of size mean_y^2 to leave sigma_y^2: operand columns whose mean is
    `ratio` x their spread, and weights of one sign so that the OUTPUT mean dominates too (|mean_y| / sigma_y ~ ratio sqrt(K): what a
    trained BN2 -> conv3 with a large beta / gamma can present).  The Gram partials are fp32 MFMA accumulations over M / 256 rows;
    the affine BatchNorm applies -- scale = gamma / sqrt(var + eps) -- must still agree with float64 on the explicit product."""
    g = torch.Generator().manual_seed(int(ratio) + K)
    sigma = 0.25
    a = (torch.randn(M, K, generator=g) * sigma + ratio * sigma).bfloat16()         # plain operand (no prologue): columns mean / std = ratio
    w = ((0.5 + torch.rand(N, K, generator=g)) / K).bfloat16()                      # all positive: y's mean = sum_k w_k mu_k dominates
    gamma, beta = 0.75 + 0.5 * torch.rand(N, generator=g), 0.1 * torch.randn(N, generator=g)
    ad, wd, gd, bd = a.to(dev), w.to(dev), gamma.to(dev), beta.to(dev)
    # float64 reference in two chunks (memory)
    wdbl = w.double().to(dev)
    s1 = torch.zeros(N, dtype=torch.float64, device=dev)
    s2 = torch.zeros(N, dtype=torch.float64, device=dev)
    mean0 = (ad[:4096].double() @ wdbl.t()).mean(0)                                  # shifted sums: no cancellation in the reference
    for i in range(0, M, 16384):
        y = ad[i:i + 16384].double() @ wdbl.t() - mean0
        s1 += y.sum(0)
        s2 += (y * y).sum(0)
    mean = (s1 / M + mean0).cpu()
    var = (s2 / M - (s1 / M) ** 2).cpu()
    assert float((mean.abs() / var.sqrt()).min()) > ratio                           # the regime the test is about
    G, s, ws, out = _gram(H, dev, ad, K)
    scale, shift = torch.empty(N, device=dev), torch.empty(N, device=dev)
    rm, rv, nbt = torch.zeros(N, device=dev), torch.ones(N, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    cen = mean.float().to(dev)                                                       # centred storage near the batch mean, as in the trunk
    H.check(H.lib().cvcl_bn_from_gram(out, K, M, H.ptr(wd), K, N, H.ptr(gd), H.ptr(bd), H.ptr(rm), H.ptr(rv), H.ptr(nbt), 0.1, 1e-5,
                                      H.ptr(scale), H.ptr(shift), None, 0, H.ptr(cen), H.stream_ptr()), "cvcl_bn_from_gram")
    torch.cuda.synchronize()
    ref_scale = gamma.double() / torch.sqrt(var + 1e-5)
    ref_shift = beta.double() - (mean - cen.double().cpu()) * ref_scale
    rel_scale = float(((scale.double().cpu() - ref_scale).abs() / ref_scale).max())
    err_shift = float((shift.double().cpu() - ref_shift).abs().max())
    # the Gram entries themselves against float64 (what bounds everything downstream)
    Gref = (ad.double().t() @ ad.double()).cpu()
    rel_G = float(((G - Gref).abs() / Gref.abs()).max())
    print(f"ratio {ratio} K {K}: |mean_y|/sigma_y >= {float((mean.abs() / var.sqrt()).min()):.0f}, G rel {rel_G:.2e}, "
          f"scale rel {rel_scale:.2e}, shift abs {err_shift:.2e}")
    assert rel_scale < 1e-3, (ratio, rel_scale, rel_G)
    assert err_shift < 1e-3 * float(ref_shift.abs().max() + 1), (ratio, err_shift)
