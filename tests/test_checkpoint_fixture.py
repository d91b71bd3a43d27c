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


@pytest.mark.gpu
def test_eval_callers_reproduce_the_reference(ckpt_path, dev):
    """f2 -- the evaluation callers against the REFERENCE's own outputs (tests/golden/eval_trials.npz, written by
    oracle/gen_golden.py::case_eval_trials with the reference's MultiModalLitModel.validation_step, multimodal_lit.py:456-511,
    and the per-trial record of the reference's eval.py:196-232), on the checkpoint the reference wrote:
    validation_step idx 1 (one 4-image trial per batch: accuracy, entropy, logged keys incl. the per-category accuracy),
    validation_step idx 0 (the val pairs through calculate_joint_loss in eval mode), and eval.py's batched trial evaluation
    for --eval_type image / text (softmax list, pred)."""
    import eval as eval_entry
    from multimodal.multimodal_lit import MultiModalLitModel
    g = load_golden("eval_trials")
    lit, _pre = MultiModalLitModel.load_model("cvcl", checkpoint_path=ckpt_path)
    lit.to(dev).eval()
    lit.set_precision("32")
    calls = []
    lit.log = lambda name, value, *a, **k: calls.append((name, float(value)))
    cats = [str(c) for c in g["categories"]]
    x_tr = g["x_trials"]
    for i, c in enumerate(cats):
        y, y_len = g["tokens"][i:i + 1].to(dev), g["lengths"][i:i + 1].to(dev)
        calls.clear()
        with torch.no_grad():
            ret = lit.validation_step((x_tr[i:i + 1].to(dev), y, y_len, [[c]]), i, dataloader_idx=1)
        d = dict(calls)
        assert sorted(d) == [str(k) for k in g["logged_keys"][i]], (c, sorted(d))
        row = g["logits_per_text_row"][i]
        clear = float(torch.sort(row, descending=True).values[:2].diff().abs()) > 1e-3 * float(row.abs().max())
        if clear:                                                          # (a near-tie may fall either way within 1e-4 rel)
            assert ret["accuracy"] == int(g["accuracy"][i]) == d["val_accuracy"] == d[f"val_accuracy_{c}"], c
        assert abs(d["val_entropy"] - float(g["entropy"][i])) < 1e-4, (c, d["val_entropy"], float(g["entropy"][i]))
    # eval.py: --eval_type image (all six trials in ONE device pass: trial t = the t-th diagonal block) and the batch-1 loop
    trials = [(x_tr[i:i + 1], g["tokens"][i:i + 1], g["lengths"][i:i + 1], [[cats[i]]]) for i in range(len(cats))]
    for group in (trials, None):
        outs = (eval_entry.evaluate_trials(lit, trials, "image", dev) if group is not None
                else [eval_entry.evaluate_trials(lit, [t], "image", dev)[0] for t in trials])
        for i, (soft, pred) in enumerate(outs):
            want = g["image_softmax"][i]
            assert float((torch.tensor(soft) - want).abs().max()) < 2e-5, i
            if float(torch.sort(want, descending=True).values[:2].diff().abs()) > 1e-4:
                assert pred == int(g["image_pred"][i]), i
    # --eval_type text: one frame, four labels
    ttr = [(x_tr[i:i + 1, :1], g["text_tokens"][i:i + 1], g["text_lengths"][i:i + 1], [[cats[i]]]) for i in range(len(cats))]
    for i, (soft, pred) in enumerate(eval_entry.evaluate_trials(lit, ttr, "text", dev)):
        want = g["text_softmax"][i]
        assert float((torch.tensor(soft) - want).abs().max()) < 2e-5, i
        if float(torch.sort(want, descending=True).values[:2].diff().abs()) > 1e-4:
            assert pred == int(g["text_pred"][i]), i
    # validation_step idx 0: the val pairs
    calls.clear()
    with torch.no_grad():
        rv = lit.validation_step((g["x_val"].to(dev), g["val_tokens"].to(dev), g["val_lengths"].to(dev), [["x"]] * 5), 0, dataloader_idx=0)
    assert not calls and sorted(rv.keys()) == [str(k) for k in g["val_all_keys"]]
    for k, v in zip(g["val_keys"], g["val_values"]):
        assert abs(float(rv[str(k)]) - float(v)) < 1e-4 * max(1.0, abs(float(v))), (k, float(rv[str(k)]), float(v))
