"""A second, non-torchvision anchor for the oracle's ResNeXt restatement.

torchvision -- where the reference's ``resnext50_32x4d`` lives (multimodal/multimodal.py:155-158, utils.py:207-209,
pyproject.toml:7) -- is neither under /root/reference nor installable here, so the oracle's Bottleneck arithmetic is "parity
unpinned".  The ``transformers`` wheel in this image ships an INDEPENDENT implementation of the same published architecture
family: ``ResNetModel`` with ``layer_type="bottleneck"``, ``downsample_in_bottleneck=False`` is ResNet-50 v1.5 -- stride on the 3x3,
projection shortcut on every stage's first block, 7x7/2 stem + 3x3/2 max-pool (pad 1), BatchNorm eps 1e-5 / momentum 0.1 with the
unbiased running variance, ReLU after the add.  With groups = 1 and 64 channels per "group" the oracle's ``resnext50_*``
functions ARE that network, so running both on the same weights pins everything in the restatement except the
``groups = 32`` argument of the 3x3 convolution (one keyword of ``F.conv2d``) and the 32x4d widths (pinned by the 25 028 904
parameter count, tests/test_oracle_golden.py): stem, stride placement, downsample wiring, block counts, BatchNorm train / eval
semantics and the running-statistics update, the pooled output.

Build-container / CPU test (transformers is a test-time dependency only; nothing of it travels or ships)."""
import pytest
import torch

import cvcl_oracle as O

transformers = pytest.importorskip("transformers")


def _hf_to_oracle(sd):
    """transformers ResNetModel state_dict -> the oracle's torchvision-style names."""
    out = {}

    def put(conv, bn, src):
        out[conv + ".weight"] = sd[src + ".convolution.weight"].clone()
        for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            out[f"{bn}.{k}"] = sd[f"{src}.normalization.{k}"].clone()

    put("conv1", "bn1", "embedder.embedder")
    for li, blocks in enumerate(O.RESNEXT_LAYERS, start=1):
        for bi in range(blocks):
            src = f"encoder.stages.{li - 1}.layers.{bi}"
            for j in (1, 2, 3):
                put(f"layer{li}.{bi}.conv{j}", f"layer{li}.{bi}.bn{j}", f"{src}.layer.{j - 1}")
            if bi == 0:
                put(f"layer{li}.{bi}.downsample.0", f"layer{li}.{bi}.downsample.1", f"{src}.shortcut")
    return out


@pytest.fixture()
def plain_resnet50(monkeypatch):
    monkeypatch.setattr(O, "RESNEXT_GROUPS", 1)
    monkeypatch.setattr(O, "RESNEXT_WIDTH_PER_GROUP", 64)
    torch.manual_seed(11)
    cfg = transformers.ResNetConfig()                       # the defaults are ResNet-50 v1.5
    assert cfg.layer_type == "bottleneck" and list(cfg.depths) == list(O.RESNEXT_LAYERS) and not cfg.downsample_in_bottleneck
    m = transformers.ResNetModel(cfg)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():                                   # non-trivial BatchNorm state: a wiring error must not hide behind identities
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.copy_(0.5 + torch.rand(mod.weight.shape, generator=g))
                mod.bias.copy_(0.2 * torch.randn(mod.bias.shape, generator=g))
                mod.running_mean.copy_(0.1 * torch.randn(mod.running_mean.shape, generator=g))
                mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=g))
                assert mod.eps == O.BN_EPS and mod.momentum == O.BN_MOMENTUM
    return m


def test_conv_specs_at_groups_1_are_resnet50(plain_resnet50):
    sd = plain_resnet50.state_dict()
    p = _hf_to_oracle(sd)
    specs = O.resnext50_conv_specs()
    assert len(specs) == 53 and len(p) == 53 * 6 == len(sd)                  # every tensor of the independent model is consumed
    for name, cin, cout, k, stride, pad, groups in specs:
        assert tuple(p[name + ".weight"].shape) == (cout, cin // groups, k, k), name


@pytest.mark.parametrize("training", [False, True])
def test_oracle_bottleneck_wiring_matches_an_independent_resnet50(plain_resnet50, training):
    m = plain_resnet50.double()                             # float64 on both sides: a wiring difference is O(1), rounding is 1e-13
    p = _hf_to_oracle(m.state_dict())
    x = torch.randn(4, 3, 96, 96, generator=torch.Generator().manual_seed(3)).double()
    stats = {}
    pooled, fmap = O.resnext50_forward(p, x, training, stats_out=stats)
    m.train(training)
    with torch.no_grad():
        out = m(x)
    ref_map, ref_pool = out.last_hidden_state, out.pooler_output.flatten(1)
    assert fmap.shape == ref_map.shape == (4, 2048, 3, 3)
    assert float((fmap - ref_map).abs().max()) < 1e-9 * float(ref_map.abs().max())
    assert float((pooled - ref_pool).abs().max()) < 1e-9 * float(ref_pool.abs().max())
    if training:                                            # running-statistics EMA with the unbiased variance, every layer
        after = _hf_to_oracle(m.state_dict())
        assert len(stats) == 53 * 3
        for k, v in stats.items():
            if k.endswith("num_batches_tracked"):
                assert int(v) == int(after[k]) == 1, k
            else:
                assert float((v - after[k]).abs().max()) < 1e-10 * max(1.0, float(after[k].abs().max())), k
