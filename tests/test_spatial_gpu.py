"""GPU parity of the embedding_type='spatial' branch (reference multimodal/multimodal.py:96-99, 181-185, 579-580, 757-787):
per-location image features x per-word text features with 'max' / 'mean' similarity, vs golden vectors produced by the
reference's own MultiModalModel and vs the oracle end to end."""
import argparse
import contextlib
import io
import sys

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import cvcl_oracle as O
from conftest import ROOT, load_golden, maxrel

pytestmark = pytest.mark.gpu
sys.path.insert(0, ROOT)


class _ImgEnc(nn.Module):
    def __init__(self, f):
        super().__init__()
        self.f = nn.Parameter(f.clone())

    def forward(self, x):
        return self.f, None


class _TxtEnc(nn.Module):
    def __init__(self, f):
        super().__init__()
        self.f = nn.Parameter(f.clone())

    def forward(self, x, x_len=None):
        return self.f, self.f, None


@pytest.mark.parametrize("sim", ["max", "mean"])
def test_spatial_head_golden(dev, sim):
    """normalise + spatial similarity + InfoNCE through multimodal.MultiModalModel on given features == reference."""
    from multimodal.multimodal import MultiModalModel
    g = load_golden("spatial_" + sim)
    args = argparse.Namespace(sim=sim, embedding_type="spatial", normalize_features=True, temperature=0.07, fix_temperature=False)
    # the HIP path keeps per-location features as NHWC rows: hand the model an NCHW *view* of NHWC storage, like the encoder does
    fi = g["image_raw"].permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    model = MultiModalModel(_ImgEnc(fi), _TxtEnc(g["text_raw"]), args).to(dev)
    lens = g["lens"].to(dev)
    out = model.calculate_contrastive_loss(torch.zeros(6, 1, device=dev), torch.zeros(6, 1, device=dev), lens)
    assert maxrel(out[5], g["logits_per_image"]) < 2e-5 and torch.equal(out[6], out[5].t())
    for i, k in enumerate("infonce image_accuracy text_accuracy image_entropy text_entropy".split()):
        assert abs(float(out[i].detach()) - float(g[k])) < 2e-5, k
    out[0].backward()
    assert maxrel(model.image_embed.f.grad, g["d_image_raw"]) < 5e-5
    # pad positions hold zero vectors: F.normalize's backward divides by eps there and torch.amax splits the gradient
    # among the tied (all-zero) locations, the kernel gives it to the first one -- those rows feed padding_idx, which
    # receives no gradient in the model, so only real words are compared
    real = (torch.arange(g["text_raw"].shape[1])[None, :] < g["lens"][:, None])
    assert maxrel(model.text_embed.f.grad.cpu()[real], g["d_text_raw"][real]) < 5e-5
    assert abs(float(model.logit_neg_log_temperature.grad) - float(g["d_neg_log_temp"])) < 1e-4 * max(1.0, abs(float(g["d_neg_log_temp"])))


@pytest.mark.parametrize("sim", ["max", "mean"])
@pytest.mark.parametrize("Bi,HW,Bt,L,E", [(256, 49, 256, 5, 512), (9, 4, 7, 25, 40), (12, 4, 700, 13, 24)])
def test_spatial_logits_oracle_sizes(dev, sim, Bi, HW, Bt, L, E):
    """C2-sized batch (256 images x 49 locations x 256 utterances), a ragged small case and one with more (utterance, word)
    columns than the max kernel keeps in LDS at a time (700 x 13 > 8192: the chunk walk a data-parallel global batch padded to
    25 words takes) vs the oracle, incl. gradients."""
    from multimodal.multimodal import MultiModalModel
    g = torch.Generator().manual_seed(Bi + L)
    side = int(HW ** 0.5)
    fi = torch.randn(Bi, E, side, side, generator=g)
    ft = torch.randn(Bt, L, E, generator=g)
    lens = torch.randint(1, L + 1, (Bt,), generator=g)
    ft = ft * (torch.arange(L)[None, :, None] < lens[:, None, None])
    nlt = torch.tensor(2.0)
    fo, to, no = fi.clone().requires_grad_(), ft.clone().requires_grad_(), nlt.clone().requires_grad_()
    lpi, _ = O.spatial_similarity_logits(F.normalize(fo, p=2, dim=1), F.normalize(to, p=2, dim=-1), lens, no, sim)
    d = torch.randn(Bi, Bt, generator=g)
    (lpi * d).sum().backward()
    args = argparse.Namespace(sim=sim, embedding_type="spatial", normalize_features=True, temperature=float(torch.exp(-nlt)),
                              fix_temperature=False)
    model = MultiModalModel(_ImgEnc(fi.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)), _TxtEnc(ft), args).to(dev)
    out = model(torch.zeros(1, device=dev), torch.zeros(1, device=dev), lens.to(dev))
    assert maxrel(out[0], lpi.detach()) < 5e-5
    (out[0] * d.to(dev)).sum().backward()
    real = (torch.arange(L)[None, :] < lens[:, None])
    assert maxrel(model.image_embed.f.grad, fo.grad) < 2e-4 and maxrel(model.text_embed.f.grad.cpu()[real], to.grad[real]) < 2e-4
    assert abs(float(model.logit_neg_log_temperature.grad) - float(no.grad)) < 2e-4 * max(1.0, abs(float(no.grad)))


@pytest.mark.parametrize("sim", ["max", "mean"])
def test_spatial_end_to_end_vs_oracle(dev, sim):
    """train.py objects with --embedding_type spatial (fp32 parity mode): state_dict layout of the reference's
    nn.Sequential vision model, loss vs the oracle (ResNeXt trunk -> 1x1 projection -> normalise -> spatial similarity)."""
    import train
    argv = (f"--dataset synthetic --batch_size 4 --gpus 1 --text_encoder embedding --embedding_dim 32 --embedding_type spatial "
            f"--sim {sim} --normalize_features --lambda_lm 0 --optimize_unused --fast_dev_run --checkpoint_callback False "
            f"--logger False").split()
    with contextlib.redirect_stdout(io.StringIO()):
        trainer, lit = train.main(argv)
    sd = lit.state_dict()
    for k in ("vision_encoder.model.0.weight", "vision_encoder.model.1.running_mean", "vision_encoder.model.4.0.conv2.weight",
              "vision_encoder.model.7.2.bn3.bias", "vision_encoder.model.8.weight", "vision_encoder.model.8.bias"):
        assert k in sd, k
    assert sd["vision_encoder.model.8.weight"].shape == (32, 2048, 1, 1)
    trainable = sorted(n for n, p in lit.named_parameters() if p.requires_grad and n.startswith("vision_encoder"))
    assert trainable == ["vision_encoder.model.8.bias", "vision_encoder.model.8.weight"]
    # one more step by hand against the oracle
    from multimodal.multimodal_data_module import SyntheticDataModule
    dm = SyntheticDataModule(train._setup_parser().parse_args(argv))
    dm.setup()
    x, y, y_len, _ = next(iter(dm.train_dataloader()))
    lit.train()
    p = {k[len("vision_encoder.model."):]: v.detach().cpu() for k, v in sd.items() if k.startswith("vision_encoder.model.")}
    names = ["conv1", "bn1", None, None, "layer1", "layer2", "layer3", "layer4"]
    po = {}
    for k, v in p.items():
        idx, rest = k.split(".", 1)
        if int(idx) < 8:
            po[f"{names[int(idx)]}.{rest}"] = v.clone()
    _pooled, fmap = O.resnext50_forward(po, x, True, None, stats_out={})
    feat = F.conv2d(fmap, p["8.weight"], p["8.bias"])
    table = sd["text_encoder.embedding.weight"].detach().cpu()
    txt = F.embedding(y, table, padding_idx=0)
    nlt = lit.model.logit_neg_log_temperature.detach().cpu()
    lpi, lpt = O.spatial_similarity_logits(F.normalize(feat, p=2, dim=1), F.normalize(txt, p=2, dim=-1), y_len, nlt, sim)
    ref = O.contrastive_loss(lpi, lpt)[0]
    out = lit.training_step((x.to(dev), y.to(dev), y_len.to(dev), None), 0)
    out["loss"].backward()
    assert abs(float(out["loss"].detach()) - float(ref)) < 2e-3 * max(1.0, abs(float(ref)))
    gw = lit.vision_encoder.model[8].weight.grad
    assert gw is not None and torch.isfinite(gw).all() and float(gw.abs().sum()) > 0
    assert lit.text_encoder.embedding.weight.grad is not None
    # eval-mode HIP-graph replay of this Sequential-wrapped trunk (the caches live on the wrapped ResNet): identical to eager, and a
    # pass at another batch shape in between does not disturb the captured workspaces
    if sim == "max":
        lit.eval()
        ve = lit.vision_encoder
        xd = x.to(dev)
        with torch.no_grad():
            f0, m0 = ve(xd)
            ve.enable_hip_graphs(True)
            f1, m1 = ve(xd)
            ve(xd[:2])
            f2, m2 = ve(xd)
        assert torch.equal(f0, f1) and torch.equal(m0, m1) and torch.equal(f0, f2) and torch.equal(m0, m2)
        ve.enable_hip_graphs(False)


@pytest.mark.parametrize("precision", [32, 16])
def test_finetune_cnn_with_spatial_embeddings_vs_oracle(dev, precision):
    """--finetune_cnn together with --embedding_type spatial (reference multimodal.py:175-185: autograd through
    nn.Sequential(trunk, Conv2d(2048, E, 1))): the loss gradient reaches the 1x1 projection AND, through the layer-4 map, every
    trunk parameter.  fp32 parity mode: projection / trunk gradients vs oracle autograd (direction and norm, as
    test_trunk_parameter_grads_vs_oracle_fp32 explains); bf16 mode: the same chain through the f32 <-> bf16 casts, checked against
    the oracle only loosely (see below)."""
    import train
    argv = (f"--dataset synthetic --batch_size 4 --gpus 1 --text_encoder embedding --embedding_dim 32 --embedding_type spatial "
            f"--sim max --normalize_features --lambda_lm 0 --optimize_unused --finetune_cnn --fast_dev_run --checkpoint_callback False "
            f"--logger False --precision {precision} --seed 11").split()
    with contextlib.redirect_stdout(io.StringIO()):
        trainer, lit = train.main(argv)
    ve = lit.vision_encoder.model
    assert ve[4][0].conv2.weight.requires_grad and ve[8].weight.requires_grad
    from multimodal.multimodal_data_module import SyntheticDataModule
    dm = SyntheticDataModule(train._setup_parser().parse_args(argv))
    dm.setup()
    x, y, y_len, _ = next(iter(dm.train_dataloader()))
    lit.train()
    sd = {k: v.detach().cpu().clone() for k, v in lit.state_dict().items()}
    p = {k[len("vision_encoder.model."):]: v for k, v in sd.items() if k.startswith("vision_encoder.model.")}
    names = ["conv1", "bn1", None, None, "layer1", "layer2", "layer3", "layer4"]
    po = {}
    for k, v in p.items():
        idx, rest = k.split(".", 1)
        if int(idx) < 8:
            fl = v.dtype.is_floating_point and "running" not in rest
            po[f"{names[int(idx)]}.{rest}"] = v.clone().requires_grad_() if fl else v.clone()
    w8, b8 = p["8.weight"].clone().requires_grad_(), p["8.bias"].clone().requires_grad_()
    _pooled, fmap = O.resnext50_forward(po, x, True, None, stats_out={})
    feat = F.conv2d(fmap, w8, b8)
    txt = F.embedding(y, sd["text_encoder.embedding.weight"], padding_idx=0)
    nlt = lit.model.logit_neg_log_temperature.detach().cpu()
    lpi, lpt = O.spatial_similarity_logits(F.normalize(feat, p=2, dim=1), F.normalize(txt, p=2, dim=-1), y_len, nlt, "max")
    ref = O.contrastive_loss(lpi, lpt)[0]
    ref.backward()
    lit.zero_grad(set_to_none=True)
    out = lit.training_step((x.to(dev), y.to(dev), y_len.to(dev), None), 0)
    out["loss"].backward()
    torch.cuda.synchronize()
    # (bf16: 4 images of 64 x 64 leave 16 values per channel for layer 4's train-mode BatchNorm -- storage rounding moves the loss
    # itself by tens of per cent at this size; the bf16 case checks that the chain is connected and sane, the fp32 case the values)
    tol = 2e-3 if precision == 32 else 0.5
    assert abs(float(out["loss"].detach()) - float(ref.detach())) < tol * max(1.0, abs(float(ref.detach())))

    def cos(a, b):
        return float(F.cosine_similarity(a.detach().cpu().flatten().double(), b.flatten().double(), dim=0))
    got = dict(ve.named_parameters())
    assert cos(got["8.weight"].grad, w8.grad) > (0.9999 if precision == 32 else 0.5)
    checked = 0
    for k, v in po.items():
        if not v.requires_grad:
            continue
        top, rest = k.split(".", 1)
        g = got[f"{names.index(top)}.{rest}"].grad
        assert g is not None and g.dtype == torch.float32 and torch.isfinite(g).all(), k
        if precision == 32:
            c = cos(g, v.grad)
            l2 = float((g.cpu().double() - v.grad.double()).norm() / v.grad.double().norm())
            assert c > 0.998 and l2 < 0.06, (k, c, l2)
        else:
            assert float(g.abs().sum()) > 0, k
        checked += 1
    assert checked == 159


def test_global_spatial_match_map_guard(dev):
    """Data-parallel global negatives with sim='max' replicate a [N_g HW, N_g L] fp32 match map and its gradient on every rank
    (2 x 20 GB at 8 ranks x 256 pairs x 7x7 x 25 words): a size beyond the device is refused with a message naming the ways out,
    instead of an out-of-memory kill inside the step; the sizes the configurations use pass."""
    from multimodal import parallel
    parallel.check_spatial_global_bytes(2048 * 49, 2048 * 25, "max", dev)                      # 8 ranks x 256 pairs: 41 GB, fits
    parallel.check_spatial_global_bytes(10 ** 7, 10 ** 7, "mean", dev)                         # sim='mean' forms no match map
    with pytest.raises(RuntimeError, match="local_negatives"):
        parallel.check_spatial_global_bytes(16384 * 49, 16384 * 25, "max", dev)                # 64 ranks x 256: 2.6 TB
