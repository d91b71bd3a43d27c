"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

TEST INFRASTRUCTURE.  Imports ``/root/reference/multimodal`` under ``sys.modules`` stubs for
the third-party packages this image lacks, drives the reference's own
classes on small seeded inputs and writes inputs + weights + expected outputs as ``.npz``
fixtures.  The reference Python never travels to the GPU box; only these vectors do.

Also asserts, case by case, that ``oracle/cvcl_oracle.py`` reproduces the reference (this is
what "oracle pinned" means) and prints the max deviation.

    python oracle/gen_golden.py            # rewrites tests/golden/
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import types
from functools import partial

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)
import cvcl_oracle as O  # noqa: E402


# ----------------------------------------------------------------------------- stubs
def install_stubs():
    def stub(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
        return m

    class _T:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    tv = stub("torchvision")
    tv.models = stub("torchvision.models")
    tv.transforms = stub(
        "torchvision.transforms",
        **{n: _T for n in "Normalize Compose Resize ToTensor CenterCrop RandomResizedCrop RandomApply "
                          "RandomHorizontalFlip Lambda".split()},
        InterpolationMode=types.SimpleNamespace(BICUBIC=3))

    class LM(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            self.__dict__.setdefault("_logged", {})[a[0]] = a[1]

    stub("pytorch_lightning", LightningModule=LM, LightningDataModule=object, seed_everything=lambda s: None)
    stub("clip")
    stub("spacy", load=lambda n: (lambda text: [types.SimpleNamespace(text=t) for t in text.split()]))
    stub("pycocoevalcap")
    for m in "bleu meteor rouge cider spice".split():
        stub(f"pycocoevalcap.{m}")
        stub(f"pycocoevalcap.{m}.{m}", **{m.capitalize(): object})
    sys.path.insert(0, REF)


def T(a):
    return a.detach().cpu().numpy()


def maxrel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **{k: (T(v) if torch.is_tensor(v) else np.asarray(v))
                                                            for k, v in arrs.items()})
    sz = os.path.getsize(os.path.join(OUT, name + ".npz"))
    print(f"  wrote {name}.npz ({sz/1024:.1f} KiB)")


def sd_np(mod, prefix=""):
    return {prefix + k: v for k, v in mod.state_dict().items()}


# ----------------------------------------------------------------------------- cases
def small_vocab(n=50):
    v = {"<pad>": 0, "<unk>": 1, "<sos>": 2, "<eos>": 3}
    for i in range(4, n):
        v[f"w{i}"] = i
    return v


def ragged_tokens(B, L, vocab, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(3, L + 1, (B,), generator=g)
    lens[0] = L                                  # at least one full-length row
    lens[-1] = 3                                 # and a minimal one (<sos> w <eos>)
    x = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n = int(lens[b])
        x[b, 0] = 2
        x[b, 1:n - 1] = torch.randint(4, vocab, (n - 2,), generator=g)
        x[b, n - 1] = 3
    return x, lens


def text_args(kind, E, pos="no_pos_embed"):
    return argparse.Namespace(text_encoder=kind, embedding_type="flat", embedding_dim=E, crange=1,
                              dropout_i=0.5 if kind == "lstm" else 0.0, dropout_o=0.0, pos_embed_type=pos)


def case_text_encoders(mm):
    vocab = small_vocab(50)
    E = 32
    shared_saved = False
    x, xl = ragged_tokens(6, 7, 50, seed=1)
    for kind, pos in (("embedding", "no_pos_embed"), ("lstm", "no_pos_embed"),
                      ("transformer", "learned"), ("transformer", "sinusoidal"), ("transformer", "no_pos_embed")):
        torch.manual_seed(10)
        te = mm.TextEncoder(vocab, 2048, text_args(kind, E, pos)).eval()
        if kind == "transformer" and pos == "learned":
            with torch.no_grad():
                te.pos_embed.normal_(0, 0.5)
        with torch.no_grad():
            ret, out, attns = te(x, xl)
        sd = sd_np(te)
        if kind == "embedding":
            o_ret, o_out = O.embedding_meanpool(sd["embedding.weight"], x, xl)
        elif kind == "lstm":
            o_ret, o_out = O.lstm_text(sd, x, xl)
        else:
            o_ret, o_out = O.transformer_text(sd, x, xl, pos)
        e1, e2 = maxrel(o_ret, ret), maxrel(o_out, out)
        print(f"text/{kind}/{pos}: oracle-vs-reference rel err ret {e1:.2e} out {e2:.2e}")
        assert e1 < 2e-6 and e2 < 2e-6
        # gradient of the table through the mean-pool, for the embedding encoder
        extra = {}
        if kind == "embedding":
            te.zero_grad()
            r2, _, _ = te(x, xl)
            g = torch.linspace(-1, 1, r2.numel()).reshape(r2.shape)
            (r2 * g).sum().backward()
            d_ref = te.embedding.weight.grad
            d_or = O.embedding_meanpool_grad(g, x, xl, 50)
            assert maxrel(d_or, d_ref) < 2e-6
            extra = {"d_ret": g, "d_table": d_ref}
        tag = kind if kind != "transformer" else f"transformer_{pos}"
        w = {"w." + k: v for k, v in sd.items() if not k.startswith("encoder_layer.")}
        if kind == "transformer":
            # same seed -> identical layer/embedding weights in the three pos-embed cases: store once
            if not shared_saved:
                save("text_transformer_weights", **{k: v for k, v in w.items() if k != "w.pos_embed"})
                shared_saved = True
            w = {k: v for k, v in w.items() if k == "w.pos_embed"}
        save(f"text_{tag}", x=x, x_len=xl, ret=ret, output=out, **w, **extra)


def case_text_extra(mm):
    """the remaining --text_encoder choices: bilstm (flat + spatial outputs) and cbow (spatial only)."""
    vocab = small_vocab(50)
    E = 32
    x, xl = ragged_tokens(6, 7, 50, seed=2)
    torch.manual_seed(12)
    a = text_args("bilstm", E)
    te = mm.TextEncoder(vocab, 2048, a).eval()
    with torch.no_grad():
        ret, out, _ = te(x, xl)
    sd = sd_np(te)
    o_ret, o_out = O.bilstm_text(sd, x, xl)
    print(f"text/bilstm: oracle-vs-reference rel err ret {maxrel(o_ret, ret):.2e} out {maxrel(o_out, out):.2e}")
    assert maxrel(o_ret, ret) < 2e-6 and maxrel(o_out, out) < 2e-6
    save("text_bilstm", x=x, x_len=xl, ret=ret, output=out, **{"w." + k: v for k, v in sd.items()})
    for crange in (1, 2):
        torch.manual_seed(13)
        a = text_args("cbow", E)
        a.embedding_type, a.crange = "spatial", crange
        te = mm.TextEncoder(vocab, 2048, a).eval()
        with torch.no_grad():
            ret, out, _ = te(x, xl)
        sd = sd_np(te)
        o_out = O.cbow_text(sd, x, crange)
        print(f"text/cbow{crange}: oracle-vs-reference rel err {maxrel(o_out, out):.2e}")
        assert maxrel(o_out, out) < 2e-6 and torch.equal(ret, out)
        save(f"text_cbow{crange}", x=x, x_len=xl, output=out, **{"w." + k: v for k, v in sd.items()})


def case_head(mm):
    """normalise + similarity + InfoNCE 10-tuple on given features, through the reference's own
    MultiModalModel.forward / calculate_contrastive_loss with trivial encoders."""
    class FeatEnc(nn.Module):              # stands in for VisionEncoder / TextEncoder outputs
        def __init__(self, feats, text):
            super().__init__()
            self.f = nn.Parameter(feats.clone())
            self.text = text

        def forward(self, x, x_len=None):
            if self.text:
                return self.f, self.f.unsqueeze(1), None
            return self.f, None

    for n, ni, nt, norm, fix in (("sq16", 16, 16, True, True), ("sq16_learned", 16, 16, True, False),
                                 ("sq37_nonorm", 37, 37, False, False), ("eval_4x1", 4, 1, True, True),
                                 ("eval_1x4", 1, 4, True, True), ("sq130", 130, 130, True, True)):
        g = torch.Generator().manual_seed(100 + ni)
        E = 48
        fi = torch.randn(ni, E, generator=g)
        ft = torch.randn(nt, E, generator=g)
        if ni == nt:                                  # make the diagonal informative but not trivial
            ft = ft + 0.8 * fi
        args = argparse.Namespace(sim="max", embedding_type="flat", normalize_features=norm,
                                  temperature=0.07, fix_temperature=fix)
        model = mm.MultiModalModel(FeatEnc(fi, False), FeatEnc(ft, True), args)
        dummy = torch.zeros(ni, 1)
        arrs = dict(image_raw=fi, text_raw=ft, neg_log_temp=model.logit_neg_log_temperature.detach().reshape(1))
        if ni == nt:
            out = model.calculate_contrastive_loss(dummy, dummy, dummy)
            loss = out[0]
            loss.backward()
            names = "infonce image_accuracy text_accuracy image_entropy text_entropy".split()
            arrs.update({k: out[i].detach().reshape(1) for i, k in enumerate(names)})
            assert torch.equal(out[6], out[5].t())          # logits_per_text is exactly the transpose
            arrs.update(logits_per_image=out[5], image_features=out[7],
                        d_image_raw=model.image_embed.f.grad, d_text_raw=model.text_embed.f.grad)
            if not fix:
                arrs["d_neg_log_temp"] = model.logit_neg_log_temperature.grad.reshape(1)
            # oracle check
            a = O.l2_normalize(fi) if norm else fi
            b = O.l2_normalize(ft) if norm else ft
            lpi, lpt = O.similarity_logits(a, b, model.logit_neg_log_temperature.detach())
            o = O.contrastive_loss(lpi, lpt)
            errs = [maxrel(lpi, out[5])] + [abs(float(o[i]) - float(out[i])) for i in range(5)]
            print(f"head/{n}: logits rel {errs[0]:.2e}  scalars abs {max(errs[1:]):.2e}")
            assert errs[0] < 2e-6 and max(errs[1:]) < 2e-5
            dl = O.infonce_dlogits(lpi)
            dl_ref = torch.autograd.grad(O.contrastive_loss(lpi.requires_grad_(), lpi.t())[0], lpi)[0]
            assert maxrel(dl, dl_ref) < 1e-5
        else:
            with torch.no_grad():
                lpi, lpt = model(dummy, dummy, dummy)
            arrs.update(logits_per_image=lpi, logits_per_text=lpt)
            a, b = O.l2_normalize(fi), O.l2_normalize(ft)
            o1, o2 = O.similarity_logits(a, b, model.logit_neg_log_temperature)
            assert maxrel(o1, lpi) < 2e-6 and maxrel(o2, lpt) < 2e-6
            print(f"head/{n}: ok")
        save(f"head_{n}", **arrs)


def formula_fill_(t: torch.Tensor, tag: int, scale: float):
    """Deterministic weight fill re-creatable anywhere without shipping weights."""
    n = t.numel()
    idx = torch.arange(n, dtype=torch.float64)
    v = torch.sin(idx * 0.7310585786 + tag * 1.6180339887) * scale
    with torch.no_grad():
        t.copy_(v.reshape(t.shape).to(t.dtype))
    return t


def vit_formula_state(model_sd):
    """Formula-filled ViT state dict (same function is used by tests through this module)."""
    out = {}
    for i, (k, v) in enumerate(sorted(model_sd.items())):
        t = torch.empty_like(v)
        if k.endswith("norm1.weight") or k.endswith("norm2.weight") or k == "norm.weight":
            formula_fill_(t, i, 0.1)
            t += 1.0
        elif k.endswith(".bias"):
            formula_fill_(t, i, 0.05)
        elif k in ("cls_token", "pos_embed"):
            formula_fill_(t, i, 0.1)
        else:
            fan_in = v[0].numel()
            formula_fill_(t, i, 1.7 / np.sqrt(fan_in))
        out[k] = t
    return out


def case_vit(vits):
    # tiny ViT through the reference class, weights stored
    torch.manual_seed(3)
    m = vits.VisionTransformer(img_size=[32], patch_size=8, embed_dim=32, depth=2, num_heads=2, mlp_ratio=4,
                               qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6)).eval()
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if n_.endswith("bias"):
                p_.normal_(0, 0.1)
            elif "norm" in n_:
                p_.normal_(1, 0.1)
            else:
                p_.normal_(0, 0.15)
    x = torch.randn(3, 3, 32, 32, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        y = m(x)
    sd = sd_np(m)
    yo = O.vit_forward(sd, x, 8, 2)
    e = maxrel(yo, y)
    print(f"vit/tiny: oracle-vs-reference rel err {e:.2e}")
    assert e < 5e-6
    save("vit_tiny", x=x, cls=y, **{"w." + k: v for k, v in sd.items()})
    # the same model at two non-native resolutions (interpolate_pos_encoding, vit:210-230): 6 x 5 and 3 x 7 patches
    extra = {}
    for tag, (hh, ww) in (("a", (48, 40)), ("b", (24, 56))):
        xi = torch.randn(2, 3, hh, ww, generator=torch.Generator().manual_seed(40 + hh))
        with torch.no_grad():
            yi = m(xi)
        e = maxrel(O.vit_forward(sd, xi, 8, 2), yi)
        print(f"vit/tiny {hh}x{ww}: oracle-vs-reference rel err {e:.2e}")
        assert e < 5e-6
        extra["x_" + tag] = xi
        extra["cls_" + tag] = yi
    save("vit_tiny_interp", **extra)

    # full ViT-B/16 and ViT-B/14 (formula weights, B=1): output only
    for patch in (16, 14):
        m = vits.vit_base(patch_size=patch, num_classes=0).eval()
        sd = vit_formula_state(m.state_dict())
        m.load_state_dict(sd)
        x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(5))
        with torch.no_grad():
            y = m(x)
        yo = O.vit_forward(sd, x, patch, 12)
        e = maxrel(yo, y)
        print(f"vit/b{patch}: tokens {m.pos_embed.shape[1]}  oracle-vs-reference rel err {e:.2e}")
        assert e < 2e-5
        save(f"vit_b{patch}", x_seed=np.array([5]), cls=y, keys=np.array(sorted(sd.keys())),
             shapes=np.array([list(sd[k].shape) + [0] * (4 - sd[k].dim()) for k in sorted(sd.keys())]))


class _StubStage(nn.Module):
    def __init__(self, owner, li):
        super().__init__()
        self.owner, self.li = [owner], li

    def forward(self, h):
        o = self.owner[0]
        return O.resnext50_stage(o.params(), h, self.li, o.training)


class StubResNeXt(nn.Module):
    """torchvision-shaped container (conv1/bn1/layer1-4/fc names) whose arithmetic is the oracle's;
    only used to drive the REFERENCE VisionEncoder's wrapper logic (Hook on layer4, fc swap, freeze)."""

    def __init__(self, pretrained=False, **kw):
        super().__init__()
        p = O.resnext50_random_params(7)
        self._names = []
        for k, v in p.items():
            safe = k.replace(".", "__")
            if v.dtype.is_floating_point and not k.endswith("running_mean") and not k.endswith("running_var"):
                self.register_parameter(safe, nn.Parameter(v))
            else:
                self.register_buffer(safe, v)
            self._names.append((k, safe))
        for li in (1, 2, 3, 4):
            setattr(self, f"layer{li}", _StubStage(self, li))
        self.fc = nn.Linear(2048, 1000)

    def params(self):
        return {k: getattr(self, s) for k, s in self._names}

    def forward(self, x):
        h = O.resnext50_stem(self.params(), x, self.training)
        for li in (1, 2, 3, 4):
            h = getattr(self, f"layer{li}")(h)
        return self.fc(h.mean(dim=(2, 3)))


def case_cvcl_step(mm, lit_mod):
    """BASELINE config 1 shape through the reference's own VisionEncoder / TextEncoder /
    MultiModalLitModel.training_step (B=4 here to keep the fixture small), ResNeXt arithmetic
    supplied by the oracle through a torchvision-shaped stub.  Pins: wrapper logic (Hook ->
    layer4 map, fc swap, freezing), the 10-tuple, the training_step dict and its key set,
    trainable-parameter set, gradients of fc / embedding."""
    sys.modules["torchvision"].models.resnext50_32x4d = StubResNeXt
    sys.modules["torchvision.models"].resnext50_32x4d = StubResNeXt
    with open(os.path.join(REF, "multimodal", "vocab.json")) as f:
        vocab = json.load(f)
    args = argparse.Namespace(
        embedding_type="flat", embedding_dim=32, pretrained_cnn=False, cnn_model="resnext50_32x4d",
        cnn_dino=False, vit_dino=False, finetune_cnn=False, text_encoder="embedding", captioning=False,
        attention=False, attention_gate=False, crange=1, dropout_i=0.0, dropout_o=0.0,
        pos_embed_type="no_pos_embed", normalize_features=False, sim="max", temperature=0.07,
        fix_temperature=False, tie=True, bias=True, lr=1e-4, weight_decay=0.1, lambda_mm=1.0, lambda_lm=0.0,
        lambda_ar=0.0, optimize_unused=True, lr_scheduler=False, optimizer=torch.optim.AdamW)
    torch.manual_seed(0)
    ve = mm.VisionEncoder(args)
    te = mm.TextEncoder(vocab, ve.last_cnn_out_dim, args)
    lit = lit_mod.MultiModalLitModel(ve, te, args)
    lit.train()
    img, tok, ln = O.synthetic_batch(4, seed=0)
    out = lit.training_step((img, tok, ln, [["a b c"]] * 4), 0)
    out["loss"].backward()
    keys = sorted(out.keys())
    trainable = sorted(n for n, p_ in lit.named_parameters() if p_.requires_grad)
    with_grad = sorted(n for n, p_ in lit.named_parameters() if p_.grad is not None)
    sd = lit.state_dict()
    print("training_step keys:", keys)
    print("trainable:", trainable)
    print("with grad:", with_grad)
    # oracle on the same weights (pre-step running stats were the init ones)
    p0 = {"image_embed.model." + k: v for k, v in O.resnext50_random_params(7).items()}
    p0["image_embed.model.fc.weight"] = sd["model.image_embed.model.fc.weight"]
    p0["image_embed.model.fc.bias"] = sd["model.image_embed.model.fc.bias"]
    p0["text_embed.embedding.weight"] = sd["model.text_embed.embedding.weight"]
    p0["logit_neg_log_temperature"] = sd["model.logit_neg_log_temperature"]
    stats = {}
    o = O.cvcl_contrastive_loss(p0, img, tok, ln, normalize_features=False, training=True, stats_out=stats)
    e_loss = abs(float(o[0]) - float(out["infonce_loss"]))
    print(f"cvcl_step: oracle-vs-reference |loss diff| {e_loss:.2e}")
    assert e_loss < 1e-5
    # feature map from the reference Hook
    lit.zero_grad()
    with torch.no_grad():
        tup = lit.model.calculate_contrastive_loss(img, tok, ln)
    save("cvcl_step_c1",
         step_keys=np.array(keys), trainable=np.array(trainable), with_grad=np.array(with_grad),
         infonce_loss=out["infonce_loss"].reshape(1), loss=out["loss"].detach().reshape(1),
         image_accuracy=out["image_accuracy"].reshape(1), text_accuracy=out["text_accuracy"].reshape(1),
         image_entropy=out["image_entropy"].reshape(1), text_entropy=out["text_entropy"].reshape(1),
         logged=np.array(sorted(lit.__dict__.get("_logged", {}).keys())),
         fc_weight=sd["model.image_embed.model.fc.weight"], fc_bias=sd["model.image_embed.model.fc.bias"],
         tokens=tok, lengths=ln, img_seed=np.array([0]), resnext_seed=np.array([7]),
         emb_rows=torch.unique(tok), emb_values=sd["model.text_embed.embedding.weight"][torch.unique(tok)],
         emb_seed_note=np.array(["embedding rows used by the batch are stored; others are irrelevant"]),
         logits_per_image=tup[5], feature_map_mean=tup[8].mean(dim=(2, 3)),
         feature_map_shape=np.array(tup[8].shape),
         d_fc_bias=lit.model.image_embed.model.fc.bias.grad if lit.model.image_embed.model.fc.bias.grad is not None
         else torch.zeros(1))
    # state-dict layout of the lit model (checkpoint compat contract): names minus the
    # stub's flattened resnext names
    names = [k for k in sd.keys() if "__" not in k]
    save("lit_state_dict_keys", keys=np.array(names),
         shapes=np.array([list(sd[k].shape) + [0] * (4 - sd[k].dim()) for k in names]))



def periodic_fill_(t: torch.Tensor, tag: int, std: float, period: int = 1021):
    """Weights that survive gzip: a seeded normal table of `period` values repeated over the flattened tensor (exactly
    periodic data deflates ~1000:1, so a 31 MB checkpoint becomes a fixture of a few hundred KiB)."""
    g = torch.Generator().manual_seed(1000 + tag)
    table = torch.randn(period, generator=g) * std
    idx = torch.arange(t.numel()) % period
    with torch.no_grad():
        t.copy_(table[idx].reshape(t.shape))


def case_reference_checkpoint(mm, lit_mod, vits):
    """f1: a Lightning-layout checkpoint WRITTEN BY THE REFERENCE'S OWN CLASSES (multimodal_lit.py:74 save_hyperparameters ->
    hyper_parameters = the constructor arguments, i.e. the pickled VisionEncoder / TextEncoder modules and the args
    namespace; :134-149 load_from_checkpoint reads it back) + the logits the reference computes from it.  pytorch_lightning
    is absent, so the dict Lightning 1.6 would write is assembled here key by key; every object inside it is an instance of
    the reference's classes (multimodal.multimodal.VisionEncoder / TextEncoder, the vendored VisionTransformer), pickled by
    their dotted names.  The ViT is one ViT-B-wide block on 32 x 32 frames (VisionEncoder hard-codes the 768-wide head,
    multimodal.py:118-122), weights periodic so the file compresses."""
    import gzip
    import shutil
    import tempfile
    with open(os.path.join(REF, "multimodal", "vocab.json")) as f:
        vocab = json.load(f)
    args = argparse.Namespace(
        embedding_type="flat", embedding_dim=32, pretrained_cnn=False, cnn_model="models/TC-S-resnext.tar",
        cnn_dino=False, vit_dino=True, finetune_cnn=False, text_encoder="embedding", captioning=False,
        attention=False, attention_gate=False, crange=1, dropout_i=0.0, dropout_o=0.0,
        pos_embed_type="no_pos_embed", normalize_features=True, sim="max", temperature=0.07,
        fix_temperature=False, tie=True, bias=True, lr=1e-4, weight_decay=0.1, lambda_mm=1.0, lambda_lm=0.0,
        lambda_ar=0.0, optimize_unused=True, lr_scheduler=False, optimizer=torch.optim.AdamW, seed=0)
    orig = mm.load_model
    mm.load_model = lambda name, pretrained: vits.VisionTransformer(
        img_size=[32], patch_size=16, embed_dim=768, depth=1, num_heads=12, mlp_ratio=4, qkv_bias=True,
        norm_layer=partial(nn.LayerNorm, eps=1e-6))
    try:
        torch.manual_seed(5)
        ve = mm.VisionEncoder(args)
        te = mm.TextEncoder(vocab, ve.last_cnn_out_dim, args)
        lit = lit_mod.MultiModalLitModel(ve, te, args)
    finally:
        mm.load_model = orig
    for i, (n_, p_) in enumerate(lit.named_parameters()):
        if p_.numel() >= 4096:
            periodic_fill_(p_.data, i, 0.03 if "embedding" not in n_ else 0.5)
        elif p_.dim() >= 1:
            with torch.no_grad():
                p_.copy_(torch.randn(p_.shape, generator=torch.Generator().manual_seed(2000 + i)) * 0.1 + (1.0 if "norm" in n_ and n_.endswith("weight") else 0.0))
    with torch.no_grad():
        lit.model.text_embed.embedding.weight[0].zero_()          # padding row
    lit.eval()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 3, 32, 32, generator=g)
    tok, ln = lit.tokenize(["ball", "look at the ball", "car"])
    with torch.no_grad():
        lpi, lpt = lit(x, tok, ln)
        fi, ft = lit.encode_image(x), lit.encode_text(tok, ln)
    ckpt = {"epoch": 3, "global_step": 120, "pytorch-lightning_version": "1.6.0",
            "state_dict": lit.state_dict(),
            "hyper_parameters": {"vision_encoder": ve, "text_encoder": te, "args": args},
            "optimizer_states": [], "lr_schedulers": [], "callbacks": {}}
    with tempfile.TemporaryDirectory() as d:
        raw = os.path.join(d, "ref.ckpt")
        torch.save(ckpt, raw)
        dst = os.path.join(OUT, "ref_lit_vit.ckpt.gz")
        with open(raw, "rb") as fi_, gzip.GzipFile(dst, "wb", compresslevel=9, mtime=0) as fo_:
            shutil.copyfileobj(fi_, fo_)
        print(f"  wrote ref_lit_vit.ckpt.gz ({os.path.getsize(raw)/2**20:.1f} MiB raw -> {os.path.getsize(dst)/1024:.1f} KiB)")
    sd = lit.state_dict()
    save("ref_lit_vit_io", x=x, tokens=tok, lengths=ln, logits_per_image=lpi, logits_per_text=lpt, image_features=fi,
         text_features=ft, keys=np.array(sorted(sd.keys())), temperature=sd["model.logit_neg_log_temperature"].reshape(1),
         n_params=np.array([sum(p_.numel() for p_ in lit.parameters())]))
    case_eval_trials(lit)


def case_eval_trials(lit):
    """f2: the evaluation callers, driven on the REFERENCE with the checkpoint model above (eval mode).
    * ``MultiModalLitModel.validation_step(batch, i, dataloader_idx=1)`` (multimodal_lit.py:456-511): one 4-image trial per
      batch, x [1, 4, C, H, W] reshaped to the batch dim, ``logits_per_text[0]``, accuracy / entropy / per-category accuracy
      as the reference logs them (the stub LightningModule.log records every call);
    * ``validation_step(batch, i, dataloader_idx=0)``: the val pairs through calculate_joint_loss (:266-309) in eval mode;
    * the per-trial record of eval.py:196-232 for --eval_type image (one label, four frames: softmax(logits_per_text)[0],
      argmax) and --eval_type text (one frame, four labels: softmax(logits_per_image)[0], argmax) -- the reference model is
      called exactly as eval.py calls it (``model(img, label, label_len)``), the three post-processing expressions are
      eval.py's own."""
    cats = ["ball", "car", "kitty", "dog", "chair", "bottle"]
    n_tr = len(cats)
    g = torch.Generator().manual_seed(21)
    x_tr = torch.randn(n_tr, 4, 3, 32, 32, generator=g)
    calls = []
    lit.log = lambda name, value, *a, **k: calls.append((name, float(value)))
    acc, ent, logit_rows, keys = [], [], [], []
    toks, lens = [], []
    for i, c in enumerate(cats):
        y, y_len = lit.tokenize([c])                                  # <sos> c <eos>, padded to 25 (multimodal_lit.py:163-183)
        toks.append(y[0]); lens.append(y_len[0])
        calls.clear()
        with torch.no_grad():
            ret = lit.validation_step((x_tr[i:i + 1], y, y_len, [[c]]), i, dataloader_idx=1)
            _lpi, lpt = lit.model(x_tr[i], y, y_len)
        d = dict(calls)
        assert set(d) == {"val_accuracy", "val_entropy", f"val_accuracy_{c}"}, d
        assert d["val_accuracy"] == ret["accuracy"] == d[f"val_accuracy_{c}"]
        acc.append(ret["accuracy"]); ent.append(d["val_entropy"]); logit_rows.append(lpt[0]); keys.append(sorted(d))
    # eval.py:196-232, --eval_type image
    img_soft = [torch.softmax(r[None], dim=-1).detach().cpu().numpy().tolist()[0] for r in logit_rows]
    img_pred = [int(torch.argmax(r[None], dim=-1).item()) for r in logit_rows]
    # --eval_type text: frame 0 of every trial against four labels (target first)
    txt_soft, txt_pred, txt_tok, txt_len = [], [], [], []
    for i in range(n_tr):
        labels = [cats[i]] + [cats[(i + k) % n_tr] for k in (1, 2, 3)]
        y, y_len = lit.tokenize(labels)
        with torch.no_grad():
            lpi, _ = lit.model(x_tr[i, :1], y, y_len)
        txt_soft.append(torch.softmax(lpi, dim=-1).detach().cpu().numpy().tolist()[0])
        txt_pred.append(int(torch.argmax(lpi, dim=-1).item()))
        txt_tok.append(y); txt_len.append(y_len)
    # dataloader_idx 0: the val pairs (batch of 5 image / utterance pairs)
    xv = torch.randn(5, 3, 32, 32, generator=g)
    yv, lv = lit.tokenize(["look at the ball", "car", "a kitty here", "the dog", "bottle on the chair"])
    calls.clear()
    with torch.no_grad():
        rv = lit.validation_step((xv, yv, lv, [["x"]] * 5), 0, dataloader_idx=0)
    assert not calls                                                  # validation logs at epoch end only (empty_log, :461-463)
    val = {k: float(v) for k, v in rv.items() if torch.is_tensor(v) and v.numel() == 1 or isinstance(v, (int, float))}
    print("eval trials: accuracy", acc, "val step keys", sorted(rv.keys()))
    save("eval_trials", x_trials=x_tr, tokens=torch.stack(toks), lengths=torch.stack(lens), categories=np.array(cats),
         logits_per_text_row=torch.stack(logit_rows), accuracy=np.array(acc), entropy=np.array(ent),
         logged_keys=np.array(keys), image_softmax=np.array(img_soft), image_pred=np.array(img_pred),
         text_tokens=torch.stack(txt_tok), text_lengths=torch.stack(txt_len), text_softmax=np.array(txt_soft),
         text_pred=np.array(txt_pred), x_val=xv, val_tokens=yv, val_lengths=lv,
         val_keys=np.array(sorted(val.keys())), val_values=np.array([val[k] for k in sorted(val.keys())]),
         val_all_keys=np.array(sorted(rv.keys())))


def case_tokenizer(lit_mod):
    with open(os.path.join(REF, "multimodal", "vocab.json")) as f:
        vocab = json.load(f)
    self = types.SimpleNamespace(vocab=vocab, nlp=sys.modules["spacy"].load("x"))
    texts = ["ball", "puzzle", "car", "look at the ball", "zzzunknownzzz here"]
    tok, ln = lit_mod.MultiModalLitModel.tokenize(self, texts)
    print("tokenizer:", tok[:3, :3].tolist(), ln.tolist())
    save("tokenizer", texts=np.array(texts), tokens=tok, lengths=ln, vocab_size=np.array([len(vocab)]))


def case_lm(mm):
    """language-model cross entropy (lambda_lm > 0 configs): reference LanguageModel.calculate_ce_loss (tokenwise) on the
    LSTM (regressional) and embedding (non-regressional) text encoders, eval mode, + the masked means of
    multimodal_lit.py:284-300 and the gradients of the mean loss."""
    vocab = small_vocab(50)
    E = 32
    x, xl = ragged_tokens(6, 7, 50, seed=3)
    x[:, 0] = 2                                            # <sos> ... <eos> like real utterances
    for b in range(x.shape[0]):
        x[b, int(xl[b]) - 1] = 3
    for kind in ("lstm", "embedding"):
        torch.manual_seed(11)
        te = mm.TextEncoder(vocab, 2048, text_args(kind, E, "no_pos_embed")).eval()
        lm = mm.LanguageModel(te, argparse.Namespace(tie=True, bias=True)).eval()
        with torch.no_grad():
            lm.output_layer.bias.normal_(0, 0.3)
        loss, outputs, logits, attns, labels = lm.calculate_ce_loss(x, xl, tokenwise=True)
        mask = labels != 0
        mean = loss.sum() / mask.sum()
        lm.zero_grad()
        mean.backward()
        sd = sd_np(te)
        if kind == "lstm":
            _r, o_out = O.lstm_text(sd, x, xl)
        else:
            _r, o_out = O.embedding_meanpool(sd["embedding.weight"], x, xl)
        o_loss, o_labels = O.lm_ce_loss(o_out, sd["embedding.weight"], lm.output_layer.bias.detach(), x, kind == "lstm")
        (s0, s1, s2), (n0, n1, n2) = O.lm_loss_summaries(o_loss, o_labels)
        print(f"lm/{kind}: oracle tokenwise loss rel {maxrel(o_loss, loss.detach()):.2e}, mean {abs(float(s0) - float(mean)):.2e}")
        assert maxrel(o_loss, loss.detach()) < 5e-6 and torch.equal(o_labels, labels)
        w = {"w." + k: v for k, v in sd.items()}
        grads = {"g." + k: v.grad for k, v in te.named_parameters() if v.grad is not None}
        save(f"lm_{kind}", x=x, x_len=xl, out_bias=lm.output_layer.bias.detach(), loss=loss.detach(), labels=labels,
             means=torch.stack([s0, s1, s2]), counts=torch.stack([n0, n1, n2]), d_out_bias=lm.output_layer.bias.grad, **w, **grads)


def case_spatial(mm):
    """embedding_type='spatial' (reference multimodal.py:757-780): per-location image features x per-word text features,
    'max' and 'mean' similarity, through the reference's own MultiModalModel.forward / calculate_contrastive_loss."""
    class ImgEnc(nn.Module):
        def __init__(self, f):
            super().__init__()
            self.f = nn.Parameter(f.clone())

        def forward(self, x):
            return self.f, None

    class TxtEnc(nn.Module):
        def __init__(self, f):
            super().__init__()
            self.f = nn.Parameter(f.clone())

        def forward(self, x, x_len=None):
            return self.f, self.f, None

    for sim in ("max", "mean"):
        g = torch.Generator().manual_seed(7 if sim == "max" else 8)
        B, E, Hh, Ww, L = 6, 24, 3, 3, 5
        lens = torch.tensor([5, 3, 4, 2, 5, 1])
        fi = torch.randn(B, E, Hh, Ww, generator=g)
        ft = torch.randn(B, L, E, generator=g)
        ft = ft + 0.5 * fi.mean(dim=(2, 3))[:, None, :]
        ft = ft * (torch.arange(L)[None, :, None] < lens[:, None, None])            # pad positions embed to zero rows
        args = argparse.Namespace(sim=sim, embedding_type="spatial", normalize_features=True, temperature=0.07,
                                  fix_temperature=False)
        model = mm.MultiModalModel(ImgEnc(fi), TxtEnc(ft), args)
        dummy = torch.zeros(B, 1)
        out = model.calculate_contrastive_loss(dummy, dummy, lens)
        out[0].backward()
        names = "infonce image_accuracy text_accuracy image_entropy text_entropy".split()
        arrs = dict(image_raw=fi, text_raw=ft, lens=lens, neg_log_temp=model.logit_neg_log_temperature.detach().reshape(1),
                    logits_per_image=out[5], d_image_raw=model.image_embed.f.grad, d_text_raw=model.text_embed.f.grad,
                    d_neg_log_temp=model.logit_neg_log_temperature.grad.reshape(1))
        arrs.update({k: out[i].detach().reshape(1) for i, k in enumerate(names)})
        a = torch.nn.functional.normalize(fi, p=2, dim=1)
        b = torch.nn.functional.normalize(ft, p=2, dim=-1)
        lpi, lpt = O.spatial_similarity_logits(a, b, lens, model.logit_neg_log_temperature.detach(), sim)
        print(f"spatial/{sim}: oracle logits rel {maxrel(lpi, out[5]):.2e}")
        assert maxrel(lpi, out[5]) < 5e-6 and torch.equal(out[6], out[5].t())
        save(f"spatial_{sim}", **arrs)


def main():
    install_stubs()
    from multimodal import multimodal as mm
    from multimodal import multimodal_lit as lit_mod
    from multimodal import vision_transformer_dino_mugs as vits
    torch.set_num_threads(8)
    case_text_encoders(mm)
    case_text_extra(mm)
    case_head(mm)
    case_spatial(mm)
    case_lm(mm)
    case_vit(vits)
    case_tokenizer(lit_mod)
    case_cvcl_step(mm, lit_mod)
    case_reference_checkpoint(mm, lit_mod, vits)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
