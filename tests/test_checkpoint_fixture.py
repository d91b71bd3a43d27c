"""f1 -- a checkpoint WRITTEN BY THE REFERENCE'S OWN CLASSES loads onto the product's classes and serves the reference's logits.

tests/golden/ref_lit_vit.ckpt.gz was produced by oracle/gen_golden.py::case_reference_checkpoint from the reference itself
(Lightning 1.6 layout: state_dict + hyper_parameters holding the pickled reference VisionEncoder / TextEncoder instances and the
args namespace, multimodal_lit.py:74,134-149); ref_lit_vit_io.npz holds inputs and the logits / features the reference computed.
CPU: the file unpickles (dotted names resolve to the product's modules, attributes the kernels need are recovered) and the
state_dict layout matches.  GPU: the loaded model reproduces the stored logits in the fp32 parity mode."""
import gzip
import shutil

import pytest
import torch

from conftest import GOLDEN, load_golden, maxrel


@pytest.fixture(scope="module")
def ckpt_path(tmp_path_factory):
    import os
    dst = tmp_path_factory.mktemp("refckpt") / "ref_lit_vit.ckpt"
    with gzip.open(os.path.join(GOLDEN, "ref_lit_vit.ckpt.gz"), "rb") as fi, open(dst, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    return str(dst)


def test_reference_written_checkpoint_loads_on_product_classes(ckpt_path):
    from multimodal import vision_transformer_dino_mugs as vits
    from multimodal.multimodal import TextEncoder, VisionEncoder
    from multimodal.multimodal_lit import MultiModalLitModel
    g = load_golden("ref_lit_vit_io")
    lit = MultiModalLitModel.load_from_checkpoint(ckpt_path)
    assert type(lit.vision_encoder) is VisionEncoder and type(lit.text_encoder) is TextEncoder
    assert type(lit.vision_encoder.model) is vits.VisionTransformer
    assert sorted(lit.state_dict().keys()) == [str(k) for k in g["keys"]]
    assert sum(p.numel() for p in lit.parameters()) == int(g["n_params"][0])
    assert lit.args["embedding_dim"] == 32 and lit.args["vit_dino"] and lit.model.normalize_features
    assert isinstance(lit.model.logit_neg_log_temperature, torch.nn.Parameter)
    assert abs(float(lit.model.logit_neg_log_temperature.detach()) - float(g["temperature"][0])) < 1e-7
    vit = lit.vision_encoder.model
    assert vit.patch_size == 16 and vit.embed_dim == 768 and vit.blocks[0].attn.num_heads == 12    # what vit_hip reads
    assert not any(p.requires_grad for n, p in vit.named_parameters() if not n.startswith("head."))   # frozen trunk, as saved
    tok, ln = lit.tokenize(["ball", "look at the ball", "car"])
    assert torch.equal(tok, g["tokens"]) and torch.equal(ln, g["lengths"])
    with pytest.raises(FileNotFoundError):
        MultiModalLitModel.load_model("cvcl", checkpoint_path=ckpt_path + ".missing")


@pytest.mark.gpu
def test_reference_written_checkpoint_reproduces_reference_logits(ckpt_path, dev):
    from multimodal.multimodal_lit import MultiModalLitModel
    g = load_golden("ref_lit_vit_io")
    lit, _pre = MultiModalLitModel.load_model("cvcl", checkpoint_path=ckpt_path)
    lit.to(dev).eval()
    lit.set_precision("32")
    x, tok, ln = g["x"].to(dev), g["tokens"].to(dev), g["lengths"].to(dev)
    with torch.no_grad():
        lpi, lpt = lit(x, tok, ln)
        fi, ft = lit.encode_image(x), lit.encode_text(tok, ln)
    assert maxrel(lpi, g["logits_per_image"]) < 1e-4 and maxrel(lpt, g["logits_per_text"]) < 1e-4      # BASELINE gate: 1e-3
    assert maxrel(fi, g["image_features"]) < 1e-4 and maxrel(ft, g["text_features"]) < 1e-5
    lit.set_precision("bf16")
    with torch.no_grad():
        lpi16, _ = lit(x, tok, ln)
    assert maxrel(lpi16, g["logits_per_image"]) < 3e-2
