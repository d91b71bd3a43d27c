"""Finalize-on-load (round 6; csrc/resnext.hip "finalize-on-load", include/cvcl_hip.h "BatchNorm accumulators"): in the bf16
train-mode trunk the convolutions whose consumer normalises a fixed set of channels ACCUMULATE their per-channel batch statistics
(int64 fixed point, one row per XCD) and that consumer -- the grouped 3x3's prologue (BN1 of every Bottleneck), the Gram launch ahead of
conv3 (BN2 of layers 1-2) -- forms its channels' (scale, shift) itself instead of waiting for a ``cvcl_bn_finalize`` launch
(reference: the train-mode nn.BatchNorm2d of every torchvision Bottleneck, reached at multimodal/multimodal.py:101).

* the accumulate mode of ``cvcl_gemm`` against its partial-row mode on integer operands (exact in both);
* everything a trunk pass leaves behind with the switch on against the launch sequence of rounds 1-5 (switch off) -- features, the
  layer-4 map, all 53 layers' running statistics and num_batches_tracked, at the benchmark's B = 256 and at B = 32, in place and in
  the deferred-statistics form of the two trunk streams: the two forms differ only in how a sum of ~10^5 terms is rounded, and the
  pass is bit-reproducible run to run in either."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT, maxrel

pytestmark = pytest.mark.gpu
WORKER = os.path.join(ROOT, "tests", "fol_worker.py")


@pytest.fixture(scope="module")
def H():
    from multimodal import _hip
    return _hip


@pytest.mark.parametrize("M,N,K", [(50176, 512, 1024), (12544, 2048, 1024), (200704, 256, 512), (802816, 128, 256), (3136, 256, 64)])
def test_gemm_accumulated_statistics_equal_the_partial_rows(H, dev, M, N, K):
    """Integer-valued operands: every product, every partial sum and every fixed-point addend is exact, so the accumulators
    (8 XCD rows, 24 fractional bits) must hold exactly what the partial rows sum to -- for the 8-wave kernel's launches of layers
    2-4, the 128 x 128 kernel of layer 1 and a small ragged shape; twice into the same accumulator = twice the sums."""
    g = torch.Generator().manual_seed(M % 1000 + N)
    A = torch.randint(-2, 3, (M, K), generator=g).to(torch.bfloat16).to(dev)
    W = torch.randint(-1, 2, (N, K), generator=g).to(torch.bfloat16).to(dev)
    rows = H.gemm_stats_rows(H.BF16, M, N, K)
    st = torch.zeros(rows, 2, N, dtype=torch.float32, device=dev)
    out_rows = H.gemm(A, W, stats=st)
    acc = torch.zeros(8, 2, N, dtype=torch.int64, device=dev)
    out_acc = H.gemm(A, W, stats_acc=acc)
    torch.cuda.synchronize()
    assert torch.equal(out_rows, out_acc)
    want = st.double().sum(dim=0)
    got = acc.sum(dim=0).double() / 2.0 ** 24
    assert torch.equal(got, want), float((got - want).abs().max())
    assert float(want[1].max()) > 0
    if M <= 50176:
        assert int((acc != 0).any(dim=2).any(dim=1).sum()) == 8                  # every XCD's workgroups added to their own row
    H.gemm(A, W, stats_acc=acc)
    torch.cuda.synchronize()
    assert torch.equal(acc.sum(dim=0).double() / 2.0 ** 24, 2 * want)


def _run(setting, path):
    env = dict(os.environ)
    env["CVCL_FINALIZE_ON_LOAD"] = setting
    r = subprocess.run([sys.executable, WORKER, str(path)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return torch.load(path, weights_only=False)


def test_trunk_pass_with_accumulated_statistics_vs_the_finalize_launches(tmp_path):
    on, on2, off = _run("1", tmp_path / "on.pt"), _run("1", tmp_path / "on2.pt"), _run("0", tmp_path / "off.pt")
    assert on["env"] == "1" and off["env"] == "0"
    worst = {}
    for case in ("B32_s0", "B256_s0", "B256_s2"):
        a, a2, b = on[case], on2[case], off[case]
        assert set(a) == set(b) and len(a) > 3 * 53
        for k in a:
            assert torch.equal(a[k], a2[k]), (case, k)                                # bit-reproducible run to run (no float atomics)
            if k.endswith("num_batches_tracked"):
                assert int(a[k]) == int(b[k]) == 3, (case, k)
                continue
            kind = "stats" if "running_" in k else "features"
            worst[kind] = max(worst.get(kind, 0.0), maxrel(a[k], b[k]))
        cos = torch.nn.functional.cosine_similarity(a["feats2"].double(), b["feats2"].double(), dim=1)
        assert float(cos.min()) > 0.9999, (case, float(cos.min()))
        # in place (one stream) and deferred (two streams) statistics are the same numbers, as before
        if case == "B256_s2":
            for k in a:
                assert torch.equal(a[k], on["B256_s0"][k]), k
    print("accumulated vs finalize launches: max-rel", worst)
    # Both forms add up the SAME fp32 partial sums, one in float64, the other in 2^-24 fixed point: a partial of magnitude >= 1 converts
    # exactly (its ulp is >= 2^-24), so the totals -- and everything downstream -- are normally identical (observed: max-rel 0.0 in all
    # three cases); a partial below 1 loses what lies under 2^-24, which can move an affine by an ulp and flip a bf16 rounding downstream
    assert worst["stats"] < 2e-3 and worst["features"] < 5e-2, worst


def test_non_finite_statistics_stay_loud(H, dev):
    """A partial sum that is not finite (or beyond what the fixed point holds) must not turn into plausible statistics: the channel's
    accumulator is poisoned (>= 2^62), which ``bn_slice_affine`` turns into a NaN affine -- what cvcl_bn_finalize makes of a
    non-finite row."""
    M, N, K = 12544, 256, 512
    g = torch.Generator().manual_seed(2)
    A = torch.randint(-2, 3, (M, K), generator=g).to(torch.bfloat16)
    W = torch.randint(-1, 2, (N, K), generator=g).to(torch.bfloat16)
    A[100, 0] = float("inf")                                      # output row 100 is inf / NaN (inf x 0) in every column
    acc = torch.zeros(8, 2, N, dtype=torch.int64, device=dev)
    H.gemm(A.to(dev), W.to(dev), stats_acc=acc)
    torch.cuda.synchronize()
    poisoned = acc[:, 1, :] >= 2 ** 62
    assert bool((poisoned.sum(dim=0) == 1).all())                 # every channel, in the row of the XCD whose workgroup met the inf, only there
    assert int((acc[:, 1, :][~poisoned] < 0).sum()) == 0
