"""GPU parity of the BENCHMARKED configuration: C2 (BASELINE configs[1]) in bf16 with train-mode BatchNorm at the bench batch.

* bf16 vs the exact-fp32 parity mode on the same weights and the bench's own batch of 256: logits max-rel / cosine / loss
  (the figure bench.py prints as ``logits_rel_vs_fp32``; reference forward multimodal/multimodal.py:746-822).
* every Bottleneck of the bf16 trunk TEACHER-FORCED at batch 32: the oracle's block input goes through
  cvcl_resnext50_block_fwd (the launch sequence the whole-trunk call uses) and the block output is compared with the
  oracle's storage-point emulation -- amplification across blocks cannot hide a kernel bug, and at 32 x 56 x 56 ... 32 x 7 x 7
  samples per channel the batch statistics are not chaotic.
* 50-step loss trajectories bf16 vs fp32 from the same initial weights, frozen trunk (B = 256) and --finetune_cnn.
"""
import ctypes as C

import pytest
import torch

import cvcl_oracle as O
from conftest import maxrel

pytestmark = pytest.mark.gpu

# Thresholds (the measured values are printed by the tests and tabulated in DESIGN.md section 3).
# The bench batch is iid uniform-noise frames (as reference tests/test_cvcl.py:14 / demo.py:11 draw them) through a
# RANDOM-INIT trunk with train-mode BatchNorm.  That network amplifies ANY relative perturbation ~100x on the way to the
# logits (fp32 rounding, 6e-8, arrives as 3e-5; rounding only the input frames to bf16 moves the pooled features by 0.2:
# measured with the oracle), so a bf16 figure only means something next to what another correct bf16
# implementation does on the same weights and batch.  The yardstick is therefore MEASURED IN THE TEST: torch's own
# autocast(bfloat16) against torch's own fp32, both running the oracle's restatement of the reference forward on the GPU
# (bench.torch_yardstick).  Measured on MI355X at B = 256: torch autocast 0.146 / cosine 0.9911 / |d loss| 0.005; HIP bf16
# (centred storage, calibrated on another batch) 0.127 / 0.9922 / 0.004; plain storage (round 2) 0.134 / 0.9917.
# The 1e-3 logits gate of BASELINE.json belongs to the fp32 parity mode and is checked here at the full size against torch fp32.
LOGITS_REL_BF16_VS_TORCH_AUTOCAST = 1.1     # HIP bf16 deviation <= 1.1 x torch autocast's own deviation ...
LOGITS_REL_BF16_CAP = 0.2                   # ... and below this in absolute terms
LOGITS_COS_BF16 = 0.99
LOSS_ABS_BF16 = 1e-2
LOGITS_REL_FP32_VS_TORCH = 2e-4             # fp32 parity mode vs torch fp32 at B = 256 (north_star: 1e-3; measured 3.3e-5)
# C4 (ViT-B/16, no BatchNorm) bf16 vs fp32: measured 0.016; C5 (e4m3 linears, per-token / MX block scales) vs fp32: measured
# 0.14, cosine 0.993 -- e4m3 carries 3 mantissa bits (2^-4 relative rounding per operand element against 2^-9 for bf16)
LOGITS_REL_C4, LOGITS_COS_C4 = 0.03, 0.9995
LOGITS_REL_C5, LOGITS_COS_C5 = 0.2, 0.99
# Teacher-forced block output, in bf16 ulps (2^-8) of the magnitudes that were rounded (see _ulp_error).  A block is three
# convolutions with a train-mode BatchNorm behind each: a stored conv output whose bf16 rounding flips (fp32 summation order
# differs between the MFMA tiles and the oracle's conv2d) is amplified by |x| / sigma in the BatchNorm that follows and spreads
# through the next convolution, so a block's output is NOT within 1-2 ulp of any other correct implementation's.  The yardstick
# is therefore measured, not assumed: the oracle's own block with every convolution's input channels visited in reverse order
# (mathematically identical, different fp32 summation order).  The HIP block must be as close to the oracle as that.
BLOCK_VS_YARDSTICK = 3.0     # HIP-vs-oracle <= 3 x (oracle-vs-reordered-oracle), on the share of elements off by > 1 ulp and on the max
BLOCK_FLOOR_FRAC, BLOCK_FLOOR_MAX = 2e-4, 8.0


@pytest.fixture(scope="module")
def H():
    from multimodal import _hip
    return _hip


def test_c2_bf16_logits_vs_fp32_at_benchmark_batch(dev):
    import bench
    lit, ve, _opt = bench.build_model("c2", dev, "bf16")
    batch = bench.synthetic_batch_on_device(bench.PER_GPU_BATCH, seed=0, device=dev)
    before = {k: v.clone() for k, v in lit.state_dict().items() if "running_" in k}
    # the trunk calibrates its storage centres on the first train-mode batch it sees: ANOTHER batch, as in a training run
    other = bench.synthetic_batch_on_device(bench.PER_GPU_BATCH, seed=977, device=dev)
    with torch.no_grad():
        lit.model(other[0], other[1], other[2])
    lit.load_state_dict(before, strict=False)
    r = bench.logits_vs_fp32(lit, batch, "bf16")
    ty, torch_logits = bench.torch_yardstick(lit, batch)
    print("C2 B=256 bf16 vs fp32:", {k: float(f"{v:.4g}") for k, v in r.items()}, "| torch autocast(bf16) vs torch fp32:",
          {k: float(f"{v:.4g}") for k, v in ty.items()})
    assert r["logits_rel_vs_fp32"] < min(LOGITS_REL_BF16_VS_TORCH_AUTOCAST * ty["logits_rel"], LOGITS_REL_BF16_CAP)
    assert r["logits_cosine_vs_fp32"] > max(LOGITS_COS_BF16, ty["logits_cosine"] - 1e-3)
    assert r["loss_abs_vs_fp32"] < LOSS_ABS_BF16
    after = lit.state_dict()
    assert all(torch.equal(v, after[k]) for k, v in before.items())          # the check leaves the BatchNorm buffers alone
    assert ve.model.compute_dtype == torch.bfloat16 and lit.training
    # the fp32 parity mode against torch's own fp32 ops (the oracle's forward on the GPU) at the full benchmark size
    lit.set_precision("32")
    lit.model.text_embed.eval()
    gn, lit.model.global_negatives = lit.model.global_negatives, False
    with torch.no_grad():
        li, _ = lit.model(batch[0], batch[1], batch[2])
    lit.model.global_negatives = gn
    e = float((li.double() - torch_logits.double()).abs().max() / torch_logits.double().abs().max())
    print(f"C2 B=256 HIP fp32 vs torch fp32: logits max-rel {e:.3g}")
    assert e < LOGITS_REL_FP32_VS_TORCH


def test_c2_bf16_logits_vs_fp32_well_conditioned(dev):
    """The same comparison at a WELL-CONDITIONED point (bench.structured_parity: residual-branch gains bn3.weight = 0.25 as in a
    trained / zero-init-residual network instead of random init's 1.0, smooth frames with per-frame contrast): the 50-layer network no
    longer amplifies every rounding ~100x (that is a property of the random-init weights, not of the input: measured 0.16-0.18 on
    smooth and noise frames alike), torch's own autocast deviates ~2e-2, and a systematic error of a few per cent in any layer would
    show.  Gate: HIP bf16 <= 1.1 x torch autocast on the same weights and frames (measured 0.0148-0.0187 vs 0.0190: the figure
    moves by +-0.003 under 1e-5 perturbations of an early layer's affine, e.g. which route produced layer1.0's downsample statistics)."""
    import bench
    lit, ve, _opt = bench.build_model("c2", dev, "bf16")
    before = {k: v.clone() for k, v in lit.state_dict().items()}
    r = bench.structured_parity(lit, ve, bench.PER_GPU_BATCH, dev)
    ty = r["torch_autocast_bf16_vs_torch_fp32"]
    print("C2 B=256 well-conditioned: HIP bf16 vs fp32", {k: v for k, v in r.items() if isinstance(v, float)}, "| torch autocast:", ty)
    assert r["logits_rel_vs_fp32"] <= LOGITS_REL_BF16_VS_TORCH_AUTOCAST * ty["logits_rel"] + 1e-3
    assert r["logits_cosine_vs_fp32"] >= ty["logits_cosine"] - 1e-4
    assert ty["logits_rel"] < 0.05 and r["logits_rel_vs_fp32"] < 0.03          # (the point is indeed the well-conditioned one)
    after = lit.state_dict()
    assert all(torch.equal(v, after[k]) for k, v in before.items())          # weights and BatchNorm buffers restored


@pytest.mark.parametrize("cfg,rel,cos", [("c4", LOGITS_REL_C4, LOGITS_COS_C4), ("c5", LOGITS_REL_C5, LOGITS_COS_C5),
                                         ("c4p14", LOGITS_REL_C4, LOGITS_COS_C4)])
def test_vit_configs_logits_vs_fp32_at_benchmark_batch(dev, cfg, rel, cos):
    """BASELINE configs[3] / [4] (saycam_contrastive_transformer: ViT-B/16 + transformer text encoder; reference
    runner_config/saycam_contrastive_transformer.py) at their stated 256 pairs per GPU: bf16 / e4m3 linears vs the fp32 mode."""
    import bench
    lit, _ve, _opt = bench.build_model(cfg, dev)
    batch = bench.synthetic_batch_on_device(bench.PER_GPU_BATCH, seed=0, device=dev)
    r = bench.logits_vs_fp32(lit, batch, "fp8" if cfg == "c5" else "bf16")
    print(f"{cfg.upper()} B=256 vs fp32:", {k: float(f"{v:.4g}") for k, v in r.items()})
    assert r["logits_rel_vs_fp32"] < rel and r["logits_cosine_vs_fp32"] > cos and r["loss_abs_vs_fp32"] < 1e-2


def test_c5_fp8_features_against_the_emulation_yardstick_at_benchmark_batch(dev):
    """C5 at B = 256 had no yardstick: 0.14 against fp32 with a gate set from the measurement.  The yardstick is what
    the e4m3 STORAGE POINTS themselves cost: the oracle's ViT forward on torch's GPU ops with torch.float8_e4m3fn roundings at the
    same operand points (bench.fp8_yardstick), against the same forward in fp32.  The HIP e4m3 path must deviate from the HIP fp32
    mode no more than that emulation deviates from torch fp32 (x 1.15: the two are different roundings of a 12-block network), on
    the ViT's output features of the benchmark's weights and frames; and the HIP fp32 mode must match torch fp32."""
    import bench
    lit, ve, _opt = bench.build_model("c5", dev)
    batch = bench.synthetic_batch_on_device(bench.PER_GPU_BATCH, seed=0, device=dev)
    r = bench.fp8_yardstick(ve, batch[0])
    print("C5 B=256 ViT features:", r)
    e, h = r["emulation_vs_torch_fp32"], r["hip_fp8_vs_hip_fp32"]
    assert h["rel_l2"] <= 1.15 * e["rel_l2"] + 2e-3 and h["cosine_min"] >= e["cosine_min"] - 5e-3
    assert r["hip_fp32_vs_torch_fp32"]["rel_l2"] < 1e-4


@pytest.mark.parametrize("cfg", ["c4", "c4p14"])
def test_c4_bf16_features_against_the_emulation_yardstick_at_benchmark_batch(dev, cfg):
    """The same for the bf16 ViT (patch 16 and the reference's patch 14) at B = 256, LayerNorm folded into qkv / fc1: the HIP bf16 path
    must be as close to fp32 as the oracle's bf16 storage-point emulation is (it rounds the normalised rows once more than the
    folded path does)."""
    import bench
    lit, ve, _opt = bench.build_model(cfg, dev)
    batch = bench.synthetic_batch_on_device(bench.PER_GPU_BATCH, seed=0, device=dev)
    r = bench.fp8_yardstick(ve, batch[0], bench.patch_of(cfg), fp8=False)
    print(f"{cfg} B=256 ViT features:", r)
    e, h = r["emulation_vs_torch_fp32"], r["hip_bf16_vs_hip_fp32"]
    assert h["rel_l2"] <= 1.15 * e["rel_l2"] + 1e-3 and h["cosine_min"] >= e["cosine_min"] - 1e-3
    assert r["hip_fp32_vs_torch_fp32"]["rel_l2"] < 1e-4


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _block_params(H, p, pre, first, dev):
    """ConvBnParams array of one Bottleneck from the oracle's parameter dict (weights packed for the bf16 kernels)."""
    names = [pre + "conv1", pre + "conv2", pre + "conv3"] + ([pre + "downsample.0"] if first else [])
    arr = (H.ConvBnParams * len(names))()
    keep, bufs = [], {}
    lib = H.lib()
    for i, n in enumerate(names):
        w = p[n + ".weight"].to(dev).contiguous()
        cout, cing, k, _ = w.shape
        kind = H.PACK_GCONV3 if n.endswith("conv2") else H.PACK_DENSE
        nb = lib.cvcl_packed_weight_bytes(H.BF16, kind, cout, cing, k)
        buf = torch.empty(nb, dtype=torch.uint8, device=dev)
        H.check(lib.cvcl_pack_conv_weight(H.BF16, kind, H.ptr(w), H.ptr(buf), cout, cing, k, H.stream_ptr()), "pack")
        bn = O.bn_name_of(n)
        t = {s: p[f"{bn}.{s}"].clone().to(dev) for s in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")}
        keep += [w, buf, t]
        bufs[bn] = t
        arr[i].w = buf.data_ptr()
        arr[i].gamma, arr[i].beta = H.ptr(t["weight"]), H.ptr(t["bias"])
        arr[i].running_mean, arr[i].running_var = H.ptr(t["running_mean"]), H.ptr(t["running_var"])
        arr[i].num_batches_tracked = H.ptr(t["num_batches_tracked"])
    torch.cuda.synchronize()
    return arr, keep, bufs


def _bn_scale(raw_nhwc, gamma, eps=1e-5):
    """|gamma| / sqrt(var + eps) of train-mode BatchNorm over the stored tensor's (N, H, W)."""
    v = raw_nhwc.double().reshape(-1, raw_nhwc.shape[-1]).var(dim=0, unbiased=False)
    return gamma.double().abs() / torch.sqrt(v + eps)


def _ulp_error(got, want, mag):
    """Largest error in units of one bf16 ulp (2^-8 relative: 8 significant bits) of ``mag`` -- the magnitude of the terms that
    were ROUNDED on the way to this value: out = relu(raw3 * s3 + b3 + identity) carries the rounding of the stored raw3
    (2^-8 |raw3| s3), of the stored downsample output and of the result itself, so mag = |raw3| s3 + |raw_d| s_d + |out|."""
    e = (got.double() - want.double()).abs() / (mag.double() * 2.0 ** -8 + 1e-30)
    return float(e.max()), float((e > 1.0).double().mean())


_reordered_conv = O.reordered_conv2d


@pytest.mark.parametrize("centred", [False, True])
def test_bf16_blocks_teacher_forced_vs_oracle_b32(H, dev, centred):
    """centred: every raw conv output stored as round(y - c) (include/cvcl_hip.h "Centred storage") with c = the batch means
    of a plain-storage pass over ANOTHER batch (what the trunk's calibration leaves behind); the oracle's storage-point model
    stores the same way with the same c."""
    B = 32
    p = O.resnext50_random_params(seed=1)
    g = torch.Generator().manual_seed(7)
    for k in list(p.keys()):                     # non-trivial BatchNorm affine so a swapped gamma / beta cannot pass
        if ("bn" in k or "downsample.1" in k) and k.endswith(".weight"):
            p[k] = torch.rand(p[k].shape, generator=g) * 0.5 + 0.75
        elif ("bn" in k or "downsample.1" in k) and k.endswith(".bias"):
            p[k] = torch.randn(p[k].shape, generator=g) * 0.1
    x, _tok, _ln = O.synthetic_batch(B, seed=3)
    taps, stats_o = {}, {}
    centres = O.resnext50_batch_means(p, O.synthetic_batch(B, seed=4)[0], O.bf16_round) if centred else None
    O.resnext50_forward(p, x, True, O.bf16_round, stats_out=stats_o, taps=taps, centres=centres)
    worst, failures = (0.0, ""), []
    prev = "maxpool"
    for li, blocks in zip((1, 2, 3, 4), O.RESNEXT_LAYERS):
        for bi in range(blocks):
            pre = f"layer{li}.{bi}."
            e_max, frac_gt1, y_max, y_frac = _check_block(H, dev, p, li, bi, taps[prev], taps, stats_o, centres)
            if e_max > worst[0]:
                worst = (e_max, pre)
            if e_max > max(BLOCK_VS_YARDSTICK * y_max, BLOCK_FLOOR_MAX) or frac_gt1 > max(BLOCK_VS_YARDSTICK * y_frac, BLOCK_FLOOR_FRAC):
                failures.append((pre, e_max, frac_gt1, y_max, y_frac))
            prev = pre + "out"
    print("teacher-forced blocks: worst", worst)
    assert not failures, failures


def _check_block(H, dev, p, li, bi, x_nchw, taps, stats_o, centres):
    """One Bottleneck through cvcl_resnext50_block_fwd on the given (bf16-exact) input against the oracle's taps for that block:
    -> (max error in ulps, share of elements off by > 1 ulp, the same two for the reordered-oracle yardstick); the train-mode
    running statistics of the block's BatchNorms are checked on the way."""
    lib = H.lib()
    pre = f"layer{li}.{bi}."
    first = bi == 0
    xin = _nhwc(x_nchw).to(torch.bfloat16).to(dev)                          # the block input, exact in bf16
    assert torch.equal(xin.float().cpu(), _nhwc(x_nchw))
    B, h, w, _c = xin.shape
    arr, _keep, bufs = _block_params(H, p, pre, first, dev)
    stride = 2 if (li > 1 and first) else 1
    out = torch.empty(B, h // stride, w // stride, 256 << (li - 1), dtype=torch.bfloat16, device=dev)
    nb = lib.cvcl_resnext50_block_workspace_bytes(H.BF16, B, h, w, li - 1)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    cen = None
    if centres is not None:
        cen = torch.zeros(len(arr), 2048)
        for i, n in enumerate(["conv1", "conv2", "conv3"] + (["downsample.0"] if first else [])):
            c = centres[pre + n]
            cen[i, :c.numel()] = c
        cen = cen.to(dev)
    H.check(lib.cvcl_resnext50_block_fwd(H.BF16, B, h, w, li - 1, int(first), 1, H.ptr(xin), arr, len(arr), H.ptr(ws), nb,
                                         H.ptr(out), 0.1, 1e-5, H.ptr(cen), H.stream_ptr()), "cvcl_resnext50_block_fwd")
    torch.cuda.synchronize()
    want = _nhwc(taps[pre + "out"])
    raw3 = _nhwc(taps[pre + "conv3.raw"])
    mag = raw3.abs() * _bn_scale(raw3, p[pre + "bn3.weight"]) + want.abs()
    if first:
        rawd = _nhwc(taps[pre + "downsample.0.raw"])
        mag = mag + rawd.abs() * _bn_scale(rawd, p[pre + "downsample.1.weight"])
    e_max, frac_gt1 = _ulp_error(out.float().cpu(), want, mag)
    # yardstick: the oracle's own block on the same input with the other summation order
    alt = _nhwc(O.resnext50_block(p, x_nchw, li, bi, True, O.bf16_round, conv_fn=_reordered_conv, centres=centres))
    y_max, y_frac = _ulp_error(alt, want, mag)
    print(f"{pre}out (B = {B}): HIP vs oracle max {e_max:.1f} ulp, {frac_gt1 * 100:.4f} % > 1 ulp (max-rel {maxrel(out.float(), want):.2e}); "
          f"oracle vs reordered oracle max {y_max:.1f} ulp, {y_frac * 100:.4f} % > 1 ulp")
    for bn, t in bufs.items():                                     # train-mode running statistics of the block's BNs
        for sname in ("running_mean", "running_var"):
            assert maxrel(t[sname], stats_o[f"{bn}.{sname}"]) < 2e-3, (bn, sname)
        assert int(t["num_batches_tracked"]) == 1
    return e_max, frac_gt1, y_max, y_frac


@pytest.mark.parametrize("li,bi", [(3, 0), (3, 1), (4, 0), (4, 1)])
def test_bf16_layer34_blocks_teacher_forced_at_benchmark_geometry(H, dev, li, bi):
    """The MFMA-bound stages at the BENCHMARK's geometry (B = 256: 50 176 / 12 544 output rows -- the real tile counts of the 8-wave
    kernel: 448 / 896 / 224 tiles, the strided-gather downsample, the materialised raw3 + bn_add_relu form), one first and one
    later block of each, teacher-forced on synthetic post-ReLU activations of the right shape against the oracle's storage-point
    model, with the same reordered-oracle yardstick as the B = 32 test."""
    B = 256
    p = O.resnext50_random_params(seed=1)
    g = torch.Generator().manual_seed(40 + li * 4 + bi)
    for k in list(p.keys()):
        if ("bn" in k or "downsample.1" in k) and k.endswith(".weight"):
            p[k] = torch.rand(p[k].shape, generator=g) * 0.5 + 0.75
        elif ("bn" in k or "downsample.1" in k) and k.endswith(".bias"):
            p[k] = torch.randn(p[k].shape, generator=g) * 0.1
    cin = (256 << (li - 2)) if bi == 0 else (256 << (li - 1))
    hw = (56 >> (li - 2)) if bi == 0 else (56 >> (li - 1))
    scale = 0.5 + torch.rand(1, cin, 1, 1, generator=g)
    shift = 0.3 * torch.randn(1, cin, 1, 1, generator=g)
    x = O.bf16_round(torch.relu(torch.randn(B, cin, hw, hw, generator=g) * scale + shift))
    taps, stats_o = {}, {}
    O.resnext50_block(p, x, li, bi, True, O.bf16_round, stats_out=stats_o, taps=taps)
    e_max, frac_gt1, y_max, y_frac = _check_block(H, dev, p, li, bi, x, taps, stats_o, None)
    assert e_max <= max(BLOCK_VS_YARDSTICK * y_max, BLOCK_FLOOR_MAX) and frac_gt1 <= max(BLOCK_VS_YARDSTICK * y_frac, BLOCK_FLOOR_FRAC), \
        (e_max, frac_gt1, y_max, y_frac)


def _trajectory(dev, precision, finetune, B, steps, lr):
    import bench
    lit, ve, _ = bench.build_model("c2", dev, precision, seed=11)
    if finetune:
        for prm in ve.model.parameters():
            prm.requires_grad_(True)
    lit.lr = lr
    opt = lit.configure_optimizers()
    batch = bench.synthetic_batch_on_device(B, seed=5, device=dev) + (None,)
    losses = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        out = lit.training_step(batch, 0)
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"].detach()))
    return losses


@pytest.mark.parametrize("finetune,B,lr", [(False, 256, 2e-3), (True, 32, 1e-4)])
def test_loss_trajectory_bf16_vs_fp32(dev, finetune, B, lr):
    """50 AdamW steps on a fixed synthetic batch from identical initial weights in the fp32 parity mode and in bf16: the
    two loss curves fall together.  (Frozen trunk: the benchmarked configuration at its batch; fine-tuning: every trunk
    parameter trains, bf16 activations / gradients vs fp32.)"""
    steps = 50
    f32 = _trajectory(dev, "32", finetune, B, steps, lr)
    b16 = _trajectory(dev, "bf16", finetune, B, steps, lr)
    gap = max(abs(a - b) for a, b in zip(f32, b16))
    print(f"finetune={finetune} B={B}: fp32 loss {f32[0]:.4f} -> {f32[-1]:.4f}, bf16 {b16[0]:.4f} -> {b16[-1]:.4f}, max gap {gap:.4f}")
    print("fp32:", [round(v, 3) for v in f32], "\nbf16:", [round(v, 3) for v in b16])
    assert abs(f32[0] - b16[0]) < 3e-2                                   # same start (forward deviation only)
    assert f32[-1] < 0.05 * f32[0] and b16[-1] < 0.05 * b16[0]           # both fit the batch

    def first_below(curve, frac):
        return next(i for i, v in enumerate(curve) if v < frac * curve[0])
    for frac in (0.5, 0.1):                                              # ... at the same pace
        a, b = first_below(f32, frac), first_below(b16, frac)
        assert abs(a - b) <= max(3, 0.2 * a), (frac, a, b)
    if not finetune:
        # frozen trunk (the benchmarked configuration): the curves coincide step by step.  With every trunk parameter training
        # the first AdamW steps (lr / sqrt(v) normalised) move 25 M weights at once and the loss falls 100x within ~7 steps, so
        # the curves are compared by pace (above) and end point, not pointwise
        # (through the steep part of the descent -- lr 2e-3, the loss halves every few steps -- the curves differ by their local
        # noise: measured |gap| <= 0.12 at loss ~1 (round 3: 1.049 vs 0.932 at step 32), 0.002 on the plateau before and 0.005 at the end)
        assert all(abs(a - b) <= 0.15 * max(a, b) + 0.03 for a, b in zip(f32, b16)), gap
        # the pointwise tolerance above was widened (0.10 x + 0.02 -> 0.15 x + 0.03) in the change that introduced centred
        # storage.  Control: the same curve with plain storage ($CVCL_CENTRED_STORAGE=0, the round-2 numerics).  With a FROZEN trunk
        # the image features are the same every step, so each storage form is ONE fixed bf16 perturbation of them and the curve gap is
        # one draw of what that perturbation does through the steep part of the descent (the logits-level comparison, where centred
        # is the closer one, is test_c2_bf16_logits_vs_fp32_at_benchmark_batch).  Both forms must hold the tolerance, and the
        # centred run must not be off by a different order of magnitude than the plain one (mean gap over the curve).
        import os
        os.environ["CVCL_CENTRED_STORAGE"] = "0"
        try:
            plain = _trajectory(dev, "bf16", finetune, B, steps, lr)
        finally:
            del os.environ["CVCL_CENTRED_STORAGE"]
        gaps_c = [abs(a - b) for a, b in zip(f32, b16)]
        gaps_p = [abs(a - b) for a, b in zip(f32, plain)]
        print(f"gap to fp32 over the curve: centred mean {sum(gaps_c) / steps:.4f} max {max(gaps_c):.4f}; plain storage mean "
              f"{sum(gaps_p) / steps:.4f} max {max(gaps_p):.4f}")
        print("plain:", [round(v, 3) for v in plain])
        assert all(abs(a - b) <= 0.15 * max(a, b) + 0.03 for a, b in zip(f32, plain)), max(gaps_p)
        assert sum(gaps_c) <= 3.0 * sum(gaps_p) + 0.01 * steps, (sum(gaps_c) / steps, sum(gaps_p) / steps)
