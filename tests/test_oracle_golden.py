"""CPU: the oracle reproduces every golden vector generated from the reference (oracle/gen_golden.py)."""
import math

import pytest
import torch

import cvcl_oracle as O
from conftest import load_golden, maxrel

TOL = 5e-6


def _weights(g):
    return {k[2:]: v for k, v in g.items() if k.startswith("w.")}


def test_text_embedding_fwd_bwd():
    g = load_golden("text_embedding")
    w = _weights(g)
    ret, out = O.embedding_meanpool(w["embedding.weight"], g["x"], g["x_len"])
    assert maxrel(ret, g["ret"]) < TOL and maxrel(out, g["output"]) < TOL
    d = O.embedding_meanpool_grad(g["d_ret"], g["x"], g["x_len"], w["embedding.weight"].shape[0])
    assert maxrel(d, g["d_table"]) < TOL
    assert float(d[0].abs().max()) == 0.0            # padding_idx row: no gradient


def test_text_lstm():
    g = load_golden("text_lstm")
    ret, out = O.lstm_text(_weights(g), g["x"], g["x_len"])
    assert maxrel(ret, g["ret"]) < TOL and maxrel(out, g["output"]) < TOL


@pytest.mark.parametrize("pos", ["learned", "sinusoidal", "no_pos_embed"])
def test_text_transformer(pos):
    g = load_golden(f"text_transformer_{pos}")
    w = _weights(load_golden("text_transformer_weights"))
    w.update(_weights(g))
    ret, out = O.transformer_text(w, g["x"], g["x_len"], pos)
    assert maxrel(ret, g["ret"]) < TOL and maxrel(out, g["output"]) < TOL


@pytest.mark.parametrize("name,norm", [("sq16", True), ("sq16_learned", True), ("sq37_nonorm", False), ("sq130", True)])
def test_head_square(name, norm):
    g = load_golden("head_" + name)
    fi = g["image_raw"].clone().requires_grad_(True)
    ft = g["text_raw"].clone().requires_grad_(True)
    nlt = g["neg_log_temp"].reshape(()).clone().requires_grad_(True)
    a = O.l2_normalize(fi) if norm else fi
    b = O.l2_normalize(ft) if norm else ft
    lpi, lpt = O.similarity_logits(a, b, nlt)
    out = O.contrastive_loss(lpi, lpt)
    assert maxrel(lpi, g["logits_per_image"]) < TOL
    for i, k in enumerate("infonce image_accuracy text_accuracy image_entropy text_entropy".split()):
        assert abs(float(out[i]) - float(g[k])) < 2e-5, k
    out[0].backward()
    assert maxrel(fi.grad, g["d_image_raw"]) < 2e-5 and maxrel(ft.grad, g["d_text_raw"]) < 2e-5
    if "d_neg_log_temp" in g:
        assert abs(float(nlt.grad) - float(g["d_neg_log_temp"])) < 1e-5 * max(1.0, abs(float(g["d_neg_log_temp"])))
    # closed-form dlogits (what the HIP backward implements) == autograd
    lp = lpi.detach().clone().requires_grad_(True)
    O.contrastive_loss(lp, lp.t())[0].backward()
    assert maxrel(O.infonce_dlogits(lpi.detach()), lp.grad) < 1e-5


def test_text_bilstm_and_cbow():
    g = load_golden("text_bilstm")
    ret, out = O.bilstm_text(_weights(g), g["x"], g["x_len"])
    assert maxrel(ret, g["ret"]) < TOL and maxrel(out, g["output"]) < TOL
    for c in (1, 2):
        g = load_golden(f"text_cbow{c}")
        assert maxrel(O.cbow_text(_weights(g), g["x"], c), g["output"]) < TOL


@pytest.mark.parametrize("kind", ["lstm", "embedding"])
def test_lm_ce_loss(kind):
    """language-model branch: token-wise loss, labels and the three masked means vs the reference (golden)."""
    g = load_golden("lm_" + kind)
    w = _weights(g)
    if kind == "lstm":
        _r, out = O.lstm_text(w, g["x"], g["x_len"])
    else:
        _r, out = O.embedding_meanpool(w["embedding.weight"], g["x"], g["x_len"])
    loss, labels = O.lm_ce_loss(out, w["embedding.weight"], g["out_bias"], g["x"], kind == "lstm")
    assert torch.equal(labels, g["labels"]) and float((loss - g["loss"]).abs().max()) < 1e-4
    means, counts = O.lm_loss_summaries(loss, labels)
    assert float((torch.stack(means) - g["means"]).abs().max()) < 1e-4 and torch.equal(torch.stack(counts), g["counts"])


@pytest.mark.parametrize("sim", ["max", "mean"])
def test_spatial_similarity(sim):
    """embedding_type='spatial' (reference multimodal.py:757-780): logits, loss scalars and feature / temperature gradients."""
    import torch.nn.functional as F
    g = load_golden("spatial_" + sim)
    fi = g["image_raw"].clone().requires_grad_(True)
    ft = g["text_raw"].clone().requires_grad_(True)
    nlt = g["neg_log_temp"].reshape(()).clone().requires_grad_(True)
    lpi, lpt = O.spatial_similarity_logits(F.normalize(fi, p=2, dim=1), F.normalize(ft, p=2, dim=-1), g["lens"], nlt, sim)
    out = O.contrastive_loss(lpi, lpt)
    assert maxrel(lpi, g["logits_per_image"]) < TOL
    for i, k in enumerate("infonce image_accuracy text_accuracy image_entropy text_entropy".split()):
        assert abs(float(out[i]) - float(g[k])) < 2e-5, k
    out[0].backward()
    assert maxrel(fi.grad, g["d_image_raw"]) < 2e-5 and maxrel(ft.grad, g["d_text_raw"]) < 2e-5
    assert abs(float(nlt.grad) - float(g["d_neg_log_temp"])) < 1e-5 * max(1.0, abs(float(g["d_neg_log_temp"])))


@pytest.mark.parametrize("name", ["eval_4x1", "eval_1x4"])
def test_head_nonsquare(name):
    g = load_golden("head_" + name)
    lpi, lpt = O.similarity_logits(O.l2_normalize(g["image_raw"]), O.l2_normalize(g["text_raw"]),
                                   g["neg_log_temp"].reshape(()))
    assert maxrel(lpi, g["logits_per_image"]) < TOL and maxrel(lpt, g["logits_per_text"]) < TOL
    assert lpi.shape == (g["image_raw"].shape[0], g["text_raw"].shape[0])


def test_vit_tiny():
    g = load_golden("vit_tiny")
    y = O.vit_forward(_weights(g), g["x"], 8, 2)
    assert maxrel(y, g["cls"]) < TOL


def test_vit_tiny_non_native_resolution():
    """interpolate_pos_encoding (vit:210-230): the oracle's restatement, and the host-side mirror's method (a parameter transform
    in plain torch), against the reference's own output on 6 x 5 and 3 x 7 patch grids (native: 4 x 4)."""
    from functools import partial

    import torch.nn as nn
    from multimodal import vision_transformer_dino_mugs as vits
    g, gi = load_golden("vit_tiny"), load_golden("vit_tiny_interp")
    w = _weights(g)
    m = vits.VisionTransformer(img_size=[32], patch_size=8, embed_dim=32, depth=2, num_heads=2, mlp_ratio=4, qkv_bias=True,
                               norm_layer=partial(nn.LayerNorm, eps=1e-6))
    m.load_state_dict(w)
    for tag in ("a", "b"):
        x = gi["x_" + tag]
        assert maxrel(O.vit_forward(w, x, 8, 2), gi["cls_" + tag]) < TOL
        gh, gw = x.shape[2] // 8, x.shape[3] // 8
        with torch.no_grad():
            got = m.interpolate_pos_encoding(torch.empty(1, gh * gw + 1, 32), x.shape[2], x.shape[3])
        assert torch.equal(got, O.vit_interpolate_pos_encoding(w["pos_embed"], gh, gw))
    with torch.no_grad():
        assert m.interpolate_pos_encoding(torch.empty(1, 17, 32), 32, 32) is m.pos_embed


def test_cvcl_step_c1():
    """Reference training_step (VisionEncoder wrapper + TextEncoder + MultiModalLitModel) on the C1 shape."""
    g = load_golden("cvcl_step_c1")
    assert list(g["step_keys"]) == ["batch_size", "image_accuracy", "image_entropy", "infonce_loss", "loss",
                                    "text_accuracy", "text_entropy"]
    img, tok, ln = O.synthetic_batch(4, seed=int(g["img_seed"][0]))
    assert torch.equal(tok, g["tokens"]) and torch.equal(ln, g["lengths"])
    p = {"image_embed.model." + k: v for k, v in O.resnext50_random_params(int(g["resnext_seed"][0])).items()}
    p["image_embed.model.fc.weight"], p["image_embed.model.fc.bias"] = g["fc_weight"], g["fc_bias"]
    E = g["fc_weight"].shape[0]
    emb = torch.zeros(2350, E)
    emb[g["emb_rows"].long()] = g["emb_values"]
    p["text_embed.embedding.weight"] = emb
    p["logit_neg_log_temperature"] = torch.tensor(-math.log(0.07))
    out = O.cvcl_contrastive_loss(p, img, tok, ln, normalize_features=False, training=True)
    assert abs(float(out[0]) - float(g["infonce_loss"])) < 1e-5
    assert abs(float(out[3]) - float(g["image_entropy"])) < 1e-5
    assert list(g["feature_map_shape"]) == [4, 2048, 7, 7]


def test_resnext_anchors():
    """ResNeXt is parity-unpinned (torchvision absent): anchor on parameter count, key layout and
    self-consistency (eval-mode batched == per-sample; grouped conv == per-group dense conv)."""
    p = O.resnext50_random_params(0)
    n = sum(v.numel() for k, v in p.items() if k.endswith(".weight") or k.endswith(".bias"))
    assert n + 2048 * 1000 + 1000 == 25_028_904
    assert p["layer1.0.conv2.weight"].shape == (128, 4, 3, 3) and p["layer4.0.downsample.0.weight"].shape == (2048, 1024, 1, 1)
    assert len(O.resnext50_conv_specs()) == 53
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    pooled, fmap = O.resnext50_forward(p, x, training=False)
    p0, _ = O.resnext50_forward(p, x[:1], training=False)
    assert maxrel(p0, pooled[:1]) < 1e-5
    w = p["layer1.0.conv2.weight"]
    h = torch.randn(1, 128, 8, 8, generator=torch.Generator().manual_seed(1))
    full = torch.nn.functional.conv2d(h, w, None, 1, 1, 1, 32)
    g3 = torch.nn.functional.conv2d(h[:, 12:16], w[12:16], None, 1, 1)
    assert maxrel(g3, full[:, 12:16]) < 1e-5


def test_batch_norm_forms_agree():
    """The explicit BatchNorm of the oracle and its F.batch_norm form (what nn.BatchNorm2d executes; used by the CPU
    baseline) give the same outputs and the same updated running statistics, in train and eval mode."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 16, 9, 7, generator=g) * 1.7 + 0.3
    p = {"bn.weight": torch.rand(16, generator=g) + 0.5, "bn.bias": torch.randn(16, generator=g),
         "bn.running_mean": torch.randn(16, generator=g), "bn.running_var": torch.rand(16, generator=g) + 0.5,
         "bn.num_batches_tracked": torch.tensor(7)}
    for training in (True, False):
        s1, s2 = {}, {}
        y1 = O.batch_norm(x, p, "bn", training, s1)
        y2 = O.batch_norm(x, p, "bn", training, s2, impl="torch")
        assert float((y1 - y2).abs().max()) < 1e-5
        assert set(s1) == set(s2)
        for k in s1:
            assert float((s1[k].double() - s2[k].double()).abs().max()) < 1e-5, k
    assert int(s1.get("bn.num_batches_tracked", torch.tensor(8))) == 8 or not s1
    # and through the whole train step of the ResNeXt + embedding configuration (tiny frames)
    pr = O.cvcl_random_params(32, seed=2)
    img, tok, ln = O.synthetic_batch(3, seed=4)
    img = img[:, :, :64, :64].contiguous()
    a = O.cvcl_contrastive_loss(pr, img, tok, ln, normalize_features=True, training=True)
    b = O.cvcl_contrastive_loss(pr, img, tok, ln, normalize_features=True, training=True, bn_impl="torch")
    assert abs(float(a[0]) - float(b[0])) < 1e-4


def test_eval_trials_fixture_is_self_consistent_with_the_oracle():
    """tests/golden/eval_trials.npz (reference validation_step idx 1 + eval.py record): accuracy = [argmax == 0], entropy =
    get_entropy(logits row) (oracle restatement of multimodal/utils.py:106-108), eval.py's softmax list / pred of the same row."""
    g = load_golden("eval_trials")
    rows = g["logits_per_text_row"]
    for i in range(rows.shape[0]):
        assert int(g["accuracy"][i]) == int(int(torch.argmax(rows[i])) == 0)
        assert abs(float(O.get_entropy(rows[i])) - float(g["entropy"][i])) < 1e-5
        assert float((torch.softmax(rows[i].double(), -1) - g["image_softmax"][i]).abs().max()) < 1e-6
        assert int(g["image_pred"][i]) == int(torch.argmax(rows[i]))
        c = str(g["categories"][i])
        assert [str(k) for k in g["logged_keys"][i]] == sorted(["val_accuracy", "val_entropy", f"val_accuracy_{c}"])
