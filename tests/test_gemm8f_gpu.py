"""GPU parity: the 8-wave e4m3 GEMM (csrc/gemm8f_kernel.h) that cvcl_gemm_fp8 / cvcl_gemm_fp8_mx select for the large ViT shapes
(>= 96 tiles of 256 x 256) -- exact on small integers for every operand path (per-row scales, MX input + residual, MX output),
ragged last tiles and several tiles per workgroup included, and bit-identical MX output against the block quantiser applied to the
bf16 output of the same product."""
import pytest
import torch

from test_gemm_gpu import _mx_quant_ref, _mx_tile_scales

pytestmark = pytest.mark.gpu


def _ints(M, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float()
    sw = 2.0 ** torch.randint(-3, 2, (N,), generator=g).float()
    bias = torch.randint(-8, 9, (N,), generator=g).float()
    return g, a, w, sw, bias


@pytest.mark.parametrize("M,N,K,with_bias", [(4100, 2304, 768, True), (8192, 768, 3072, False), (16500, 2304, 768, True), (3000, 3072, 768, True), (4099, 2304, 768, True)])
def test_gemm8f_per_row_scales_exact_on_small_integers(dev, M, N, K, with_bias):
    from multimodal import _hip as H
    g, a, w, sw, bias = _ints(M, N, K, M + N)
    sa = 2.0 ** torch.randint(-3, 2, (M,), generator=g).float()
    ref = (a.double() @ w.double().t()) * sa.double()[:, None] * sw.double()[None, :]
    if with_bias:
        ref = ref + bias.double()
    ref = ref.float().bfloat16()
    a8, w8 = a.to(torch.float8_e4m3fn).to(dev), w.to(torch.float8_e4m3fn).to(dev)
    sad, swd, bd = sa.to(dev), sw.to(dev), bias.to(dev)
    C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    for _ in range(2):
        H.check(H.lib().cvcl_gemm_fp8(a8.data_ptr(), H.ptr(sad), K, w8.data_ptr(), H.ptr(swd), K, H.ptr(C), N, H.ptr(bd) if with_bias else None,
                                      0, None, 0, M, N, K, H.stream_ptr()), "cvcl_gemm_fp8")
    bad = (C.cpu() != ref)
    assert int(bad.sum()) == 0, (int(bad.sum()), bad.nonzero()[:5])


@pytest.mark.parametrize("M,N,K", [(8200, 768, 768), (9000, 768, 3072), (33000, 768, 768), (21500, 768, 768), (21501, 768, 3072), (16000, 1536, 768)])
def test_gemm8f_mx_input_with_residual_exact(dev, M, N, K):
    """proj / fc2 form: A with e8m0 block scales (applied by the scaled MFMA) + bias + residual.  (The dispatcher gives N = 768 to the
    8-wave kernel only when the last round of tiles is >= 85 % full -- M = 21 500: 252 tiles in one round; M = 16 000 x N = 1536: 378
    tiles -- and to the 128 x 128 kernel otherwise: both are covered.)"""
    from multimodal import _hip as H
    g, a, w, sw, bias = _ints(M, N, K, M + K)
    eb = torch.randint(124, 130, (M, K // 32), generator=g).to(torch.uint8)                 # block scales 2^-3 .. 2^2
    r = torch.randint(-20, 21, (M, N), generator=g).float().bfloat16()
    a_eff = a.reshape(M, K // 32, 32) * torch.pow(2.0, (eb.double() - 127))[:, :, None]
    y = ((a_eff.reshape(M, K).double() @ w.double().t()) * sw.double()[None, :] + bias.double()).float().bfloat16()
    ref = (y.float() + r.float()).bfloat16()
    a8, w8 = a.to(torch.float8_e4m3fn).to(dev), w.to(torch.float8_e4m3fn).to(dev)
    ebd, swd, bd, rd = _mx_tile_scales(eb).to(dev), sw.to(dev), bias.to(dev), r.to(dev)
    C = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    H.check(H.lib().cvcl_gemm_fp8_mx(a8.data_ptr(), None, H.ptr(ebd), K, w8.data_ptr(), H.ptr(swd), K, H.ptr(C), N, None, None, 0, H.ptr(bd), 0,
                                     H.ptr(rd), N, M, N, K, H.stream_ptr()), "cvcl_gemm_fp8_mx")
    bad = (C.cpu() != ref)
    assert int(bad.sum()) == 0, (int(bad.sum()), bad.nonzero()[:5])
    # in place (C == R), as the ViT's residual stream runs it
    H.check(H.lib().cvcl_gemm_fp8_mx(a8.data_ptr(), None, H.ptr(ebd), K, w8.data_ptr(), H.ptr(swd), K, H.ptr(rd), N, None, None, 0, H.ptr(bd), 0,
                                     H.ptr(rd), N, M, N, K, H.stream_ptr()), "cvcl_gemm_fp8_mx")
    assert torch.equal(rd.cpu(), ref)


@pytest.mark.parametrize("M,act", [(4100, 2), (4100, 0), (16500, 2), (4103, 2)])
def test_gemm8f_mx_output_matches_block_quantiser(dev, M, act):
    """fc1 form: bias (+ GELU) -> bf16 rounding -> per-32-column e8m0 scale + e4m3 bytes, bit-identical to the oracle quantiser applied
    to the bf16 output of the same product and epilogue."""
    from multimodal import _hip as H
    N, K = 3072, 768
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, K, generator=g) * 1.3).bfloat16()
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias = (torch.randn(N, generator=g) * 0.1).to(dev)
    xd, wd = x.to(dev), w.to(dev)
    xq, xs = torch.empty(M, K, dtype=torch.uint8, device=dev), torch.empty(M, device=dev)
    wq, ws = torch.empty(N, K, dtype=torch.uint8, device=dev), torch.empty(N, device=dev)
    H.check(H.lib().cvcl_quant_rows_fp8(H.BF16, H.ptr(xd), K, None, None, 0.0, H.ptr(xq), H.ptr(xs), M, K, H.stream_ptr()), "quant")
    H.check(H.lib().cvcl_quant_rows_fp8(H.F32, H.ptr(wd), K, None, None, 0.0, H.ptr(wq), H.ptr(ws), N, K, H.stream_ptr()), "quant")
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    H.check(H.lib().cvcl_gemm_fp8(H.ptr(xq), H.ptr(xs), K, H.ptr(wq), H.ptr(ws), K, H.ptr(C), N, H.ptr(bias), act, None, 0, M, N, K,
                                  H.stream_ptr()), "cvcl_gemm_fp8")
    c8 = torch.empty(M, N, dtype=torch.uint8, device=dev)
    cb = torch.empty(N // 128, M, 4, dtype=torch.uint8, device=dev)
    H.check(H.lib().cvcl_gemm_fp8_mx(H.ptr(xq), H.ptr(xs), None, K, H.ptr(wq), H.ptr(ws), K, None, 0, H.ptr(c8), H.ptr(cb), N, H.ptr(bias), act,
                                     None, 0, M, N, K, H.stream_ptr()), "cvcl_gemm_fp8_mx")
    q_ref, e_ref, _ = _mx_quant_ref(C.float().cpu())
    assert torch.equal(cb.cpu(), _mx_tile_scales(e_ref)) and torch.equal(c8.cpu(), q_ref.view(torch.uint8))
    # and the bf16 output itself against float64 maths on the dequantised operands
    xf = xq.view(torch.float8_e4m3fn).float().cpu().double() * xs.cpu().double()[:, None]
    wf = wq.view(torch.float8_e4m3fn).float().cpu().double() * ws.cpu().double()[:, None]
    y = xf[:512] @ wf.t() + bias.cpu().double()
    if act == 2:
        y = 0.5 * y * (1 + torch.erf(y / 2 ** 0.5))
    err = float((C[:512].double().cpu() - y).abs().max() / y.abs().max())
    assert err < 8e-3, err
