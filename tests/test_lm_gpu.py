"""GPU parity of the language-model loss branch (lambda_lm > 0; reference multimodal.py:861-890, multimodal_lit.py:266-309)
vs golden vectors produced by the reference's own LanguageModel.calculate_ce_loss."""
import argparse
import contextlib
import io
import sys

import pytest
import torch

import cvcl_oracle as O
from conftest import ROOT, load_golden, maxrel

pytestmark = pytest.mark.gpu
sys.path.insert(0, ROOT)


def _build(kind, dev, g):
    from multimodal.multimodal import LanguageModel, TextEncoder
    vocab = {f"w{i}": i for i in range(50)}
    args = argparse.Namespace(text_encoder=kind, embedding_type="flat", embedding_dim=32, crange=1, dropout_i=0.0, dropout_o=0.0,
                              pos_embed_type="no_pos_embed", captioning=False, attention=False, attention_gate=False, tie=True, bias=True)
    with contextlib.redirect_stdout(io.StringIO()):
        te = TextEncoder(vocab, 2048, args)
        lm = LanguageModel(te, args)
    sd = te.state_dict()
    for k, v in g.items():
        if k.startswith("w."):
            sd[k[2:]].copy_(v)
    with torch.no_grad():
        lm.output_layer.bias.copy_(g["out_bias"])
    return te.to(dev).eval(), lm.to(dev).eval()


@pytest.mark.parametrize("kind", ["lstm", "embedding"])
def test_lm_ce_loss_golden(dev, kind):
    from multimodal import ops
    g = load_golden("lm_" + kind)
    te, lm = _build(kind, dev, g)
    y, y_len = g["x"].to(dev), g["x_len"].to(dev)
    loss, outputs, logits, attns, labels = lm.calculate_ce_loss(y, y_len, tokenwise=True)
    Lp = g["loss"].shape[1]
    # loss = logsumexp - logit[label]: compare on the scale of the logits (the tied-embedding predictor of the embedding
    # encoder has losses of 1e-5 next to logits of 30: fp32 cancellation noise, same in the reference)
    scale = max(1.0, float(logits.detach().abs().max()))
    assert torch.equal(labels.cpu()[:, :Lp], g["labels"])
    assert float((loss.detach().cpu()[:, :Lp] - g["loss"]).abs().max()) < 2e-6 * scale
    assert float(loss.detach().cpu()[:, Lp:].abs().sum()) == 0.0
    means, counts = ops.lm_loss_summaries(loss.reshape(-1), labels.reshape(-1))
    assert float((means.detach().cpu() - g["means"]).abs().max()) < 2e-6 * scale and torch.equal(counts.cpu().long(), g["counts"].long())
    mean_direct = lm.calculate_ce_loss(y, y_len, tokenwise=False)[0]
    assert abs(float(mean_direct.detach()) - float(g["means"][0])) < 2e-6 * scale
    means[0].backward()
    got = dict(te.named_parameters())
    for k, v in g.items():
        if k.startswith("g."):
            assert float((got[k[2:]].grad.cpu() - v).abs().max()) < 1e-4 * float(v.abs().max()) + 5e-7, k      # (softmax - onehot) cancels to ~1e-5 for the tied embedding predictor
    assert float((lm.output_layer.bias.grad.cpu() - g["d_out_bias"]).abs().max()) < 1e-4 * float(g["d_out_bias"].abs().max()) + 5e-7
    assert lm.output_layer.weight is te.embedding.weight


def test_joint_loss_training_step(dev):
    """train.py objects with lambda_mm = lambda_lm = 1 on the LSTM text encoder: joint loss = InfoNCE + LM cross entropy (both
    vs the oracle), logged keys of the reference, gradients reach the LSTM and the tied table through both branches."""
    import train
    argv = ("--dataset synthetic --batch_size 4 --gpus 1 --text_encoder lstm --embedding_dim 32 --lambda_mm 1 --lambda_lm 1 "
            "--optimize_unused --fast_dev_run --checkpoint_callback False --logger False").split()
    with contextlib.redirect_stdout(io.StringIO()):
        trainer, lit = train.main(argv)
    for k in ("train_ce_loss", "train_ce_loss_wo_sos", "train_ce_loss_wo_sos_eos", "train_infonce_loss", "train_loss",
              "train_ce_loss_epoch", "train_perplexity_epoch"):
        assert k in trainer.logged_metrics, (k, sorted(trainer.logged_metrics))
    from multimodal.multimodal_data_module import SyntheticDataModule
    dm = SyntheticDataModule(train._setup_parser().parse_args(argv))
    dm.setup()
    x, y, y_len, _ = next(iter(dm.train_dataloader()))
    lit.eval()                                        # no dropout: comparable with the oracle's eval-mode LSTM
    for p in lit.parameters():
        p.grad = None
    with torch.enable_grad():
        out = lit.calculate_joint_loss((x.to(dev), y.to(dev), y_len.to(dev), None), "train", lambda *a, **k: None)
    sd = {k[len("text_encoder."):]: v.detach().cpu() for k, v in lit.state_dict().items() if k.startswith("text_encoder.")}
    _r, o_out = O.lstm_text(sd, y, y_len)
    o_loss, o_labels = O.lm_ce_loss(o_out, sd["embedding.weight"], lit.language_model.output_layer.bias.detach().cpu(), y, True)
    (m0, m1, m2), _n = O.lm_loss_summaries(o_loss, o_labels)
    assert abs(float(out["ce_loss"]) - float(m0)) < 1e-4 and abs(float(out["ce_loss_wo_sos_eos"]) - float(m2)) < 1e-4
    assert abs(float(out["loss"].detach()) - (float(out["infonce_loss"]) + float(m0))) < 2e-4
    out["loss"].backward()
    assert lit.text_encoder.lstm.weight_hh_l0.grad is not None and lit.text_encoder.embedding.weight.grad is not None
    assert lit.language_model.output_layer.bias.grad is not None
