"""GPU parity: backward of the trainable text encoders (transformer, LSTM) vs torch autograd through the oracle on
the CPU, dropout disabled (p = 0) so the comparison is exact; plus the statistical / consistency properties of the
hash-RNG dropout kernels.  fp32, tolerance 1e-4 rel on every parameter gradient."""
import argparse
import contextlib
import io

import pytest
import torch

import cvcl_oracle as O
from conftest import maxrel

pytestmark = pytest.mark.gpu


def _te(kind, pos, E, V, dropout_i=0.0):
    from multimodal.multimodal import TextEncoder
    vocab = {f"w{i}": i for i in range(V)}
    args = argparse.Namespace(text_encoder=kind, embedding_type="flat", embedding_dim=E, crange=1, dropout_i=dropout_i,
                              dropout_o=0.0, pos_embed_type=pos)
    with contextlib.redirect_stdout(io.StringIO()):
        return TextEncoder(vocab, 2048, args)


def _tokens(B, L, V, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(2, L + 1, (B,), generator=g)
    lens[0] = L
    tok = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        tok[b, :int(lens[b])] = torch.randint(1, V, (int(lens[b]),), generator=g)
    return tok, lens


@pytest.mark.parametrize("B,L,E,V", [(7, 6, 64, 40), (64, 5, 512, 300)])
def test_transformer_text_backward(dev, B, L, E, V):
    torch.manual_seed(B)
    te = _te("transformer", "learned", E, V)
    with torch.no_grad():
        te.pos_embed.normal_(0, 0.3)
    tok, lens = _tokens(B, L, V, 5)
    g_out = torch.randn(B, E, generator=torch.Generator().manual_seed(9))
    # oracle gradients (CPU autograd through the functional restatement)
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in te.state_dict().items()}
    r_o, _ = O.transformer_text(sd, tok, lens, "learned")
    (r_o * g_out).sum().backward()
    # HIP: eval mode => every dropout p = 0, gradients enabled
    te = te.to(dev).eval()
    ret, out, _ = te(tok.to(dev), lens.to(dev))
    assert maxrel(ret, r_o) < 2e-5
    (ret * g_out.to(dev)).sum().backward()
    checked = 0
    for name, prm in te.named_parameters():
        if name.startswith("encoder_layer."):
            assert prm.grad is None                      # the dead duplicate gets no gradient (Appendix C.2)
            continue
        ref = sd[name].grad
        assert prm.grad is not None, name
        e = maxrel(prm.grad, ref)
        assert e < 1e-4, (name, e)
        checked += 1
    assert checked >= 13
    assert float(te.embedding.weight.grad[0].abs().max()) == 0.0           # padding_idx row


@pytest.mark.parametrize("B,L,E,V", [(5, 7, 32, 30), (64, 5, 512, 300)])
def test_lstm_text_backward(dev, B, L, E, V):
    torch.manual_seed(L)
    te = _te("lstm", "no_pos_embed", E, V)
    tok, lens = _tokens(B, L, V, 6)
    g_out = torch.randn(B, E, generator=torch.Generator().manual_seed(3))
    sd = {k: v.clone().requires_grad_(True) for k, v in te.state_dict().items()}
    r_o, _ = O.lstm_text(sd, tok, lens)
    (r_o * g_out).sum().backward()
    te = te.to(dev).eval()
    ret, _, _ = te(tok.to(dev), lens.to(dev))
    assert maxrel(ret, r_o) < 2e-5
    (ret * g_out.to(dev)).sum().backward()
    for name, prm in te.named_parameters():
        e = maxrel(prm.grad, sd[name].grad)
        assert e < 1e-4, (name, e)


def test_dropout_kernel_properties(dev):
    from multimodal import text_train as T
    x = torch.ones(4096, 128, device=dev, requires_grad=True)
    y = T.DropoutAdd.apply(x, None, 0.1, 1234, 0, 1)
    kept = (y != 0).float().mean().item()
    assert abs(kept - 0.9) < 0.01 and abs(y.mean().item() - 1.0) < 0.02           # inverted dropout keeps the mean
    assert torch.allclose(y[y != 0], torch.full_like(y[y != 0], 1 / 0.9))
    y.sum().backward()
    assert torch.equal(x.grad != 0, y != 0)                                        # backward uses the same mask
    y2 = T.DropoutAdd.apply(x, None, 0.1, 1234, 0, 1)
    assert torch.equal(y, y2)                                                      # same seed -> same mask
    assert not torch.equal(y, T.DropoutAdd.apply(x, None, 0.1, 99, 0, 1))
    # locked dropout: mask [B,1,E] shared over the L positions of each sequence
    B, L, E = 32, 7, 64
    z = T.DropoutAdd.apply(torch.ones(B * L, E, device=dev), None, 0.5, 7, L, E).view(B, L, E)
    assert torch.equal(z[:, 0], z[:, 3]) and torch.equal(z[:, 0], z[:, L - 1])
    assert abs((z != 0).float().mean().item() - 0.5) < 0.05
    # residual: y = dropout(x) + r
    r = torch.randn(16, 8, device=dev)
    assert torch.equal(T.DropoutAdd.apply(torch.zeros(16, 8, device=dev), r, 0.3, 5, 0, 1), r)


def test_attention_small_backward_with_dropout(dev):
    """Directional-derivative check of the fused attention forward/backward with probability dropout active."""
    from multimodal import text_train as T
    B, L, E, nh = 3, 6, 32, 4
    g = torch.Generator().manual_seed(2)
    qkv = torch.randn(B * L, 3 * E, generator=g).to(dev)
    tok, _ = _tokens(B, L, 20, 8)
    tok = tok.to(dev)
    go = torch.randn(B * L, E, generator=g).to(dev)
    d = torch.randn(B * L, 3 * E, generator=g).to(dev)

    def f(z):
        return (T.AttentionSmall.apply(z, tok, nh, 0.25, 4242) * go).sum()

    z = qkv.clone().requires_grad_(True)
    f(z).backward()
    analytic = float((z.grad * d).sum())
    eps = 1e-2
    numeric = float((f(qkv + eps * d) - f(qkv - eps * d)) / (2 * eps))
    assert abs(analytic - numeric) < 2e-2 * max(1.0, abs(numeric)), (analytic, numeric)


def test_c4_train_step_runs_with_dropout(dev):
    """One training step of the transformer text encoder in train mode (dropout 0.1 active) changes every live
    parameter and produces finite gradients."""
    te = _te("transformer", "learned", 64, 50).to(dev).train()
    tok, lens = _tokens(16, 5, 50, 1)
    opt = torch.optim.AdamW(te.parameters(), lr=1e-2)
    before = {n: p.detach().clone() for n, p in te.named_parameters()}
    ret, _, _ = te(tok.to(dev), lens.to(dev))
    (ret ** 2).mean().backward()
    opt.step()
    for n, p in te.named_parameters():
        if n.startswith("encoder_layer."):
            continue
        assert torch.isfinite(p.grad).all(), n
        if n != "embedding.weight":
            assert not torch.equal(p.detach(), before[n]), n


def _extra_te(kind, embedding_type, E, V, crange=1):
    import argparse, contextlib, io
    from multimodal.multimodal import TextEncoder
    vocab = {f"w{i}": i for i in range(V)}
    args = argparse.Namespace(text_encoder=kind, embedding_type=embedding_type, embedding_dim=E, crange=crange, dropout_i=0.0,
                              dropout_o=0.0, pos_embed_type="no_pos_embed", captioning=False, attention=False, attention_gate=False)
    with contextlib.redirect_stdout(io.StringIO()):
        return TextEncoder(vocab, 2048, args)


def test_bilstm_golden_and_backward(dev):
    """text_encoder='bilstm' (reference :513-552): golden forward from the reference, gradients vs oracle autograd."""
    from conftest import load_golden
    g = load_golden("text_bilstm")
    te = _extra_te("bilstm", "flat", 32, 50)
    sd = te.state_dict()
    for k, v in g.items():
        if k.startswith("w."):
            sd[k[2:]].copy_(v)
    te = te.to(dev).eval()
    ret, out, _ = te(g["x"].to(dev), g["x_len"].to(dev))
    assert maxrel(ret, g["ret"]) < 2e-5 and maxrel(out, g["output"]) < 2e-5
    tok, lens = _tokens(9, 11, 50, 4)
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in te.state_dict().items()}
    r_o, o_o = O.bilstm_text(p, tok, lens)
    gr = torch.randn(r_o.shape, generator=torch.Generator().manual_seed(1))
    go = torch.randn(o_o.shape, generator=torch.Generator().manual_seed(2))
    ((r_o * gr).sum() + (o_o * go).sum()).backward()
    for prm in te.parameters():
        prm.grad = None
    ret, out, _ = te(tok.to(dev), lens.to(dev))
    assert maxrel(ret, r_o) < 2e-5 and maxrel(out, o_o) < 2e-5
    ((ret * gr.to(dev)).sum() + (out * go.to(dev)).sum()).backward()
    for name, prm in te.named_parameters():
        assert maxrel(prm.grad, p[name].grad) < 2e-4, name


@pytest.mark.parametrize("crange", [1, 2])
def test_cbow_golden_and_backward(dev, crange):
    from conftest import load_golden
    g = load_golden(f"text_cbow{crange}")
    te = _extra_te("cbow", "spatial", 32, 50, crange)
    te.state_dict()["embedding.weight"].copy_(g["w.embedding.weight"])
    te = te.to(dev).eval()
    ret, out, _ = te(g["x"].to(dev), g["x_len"].to(dev))
    assert maxrel(out, g["output"]) < 2e-5 and ret is out
    p = {"embedding.weight": g["w.embedding.weight"].clone().requires_grad_(True)}
    go = torch.randn(g["output"].shape, generator=torch.Generator().manual_seed(5))
    (O.cbow_text(p, g["x"], crange) * go).sum().backward()
    (out * go.to(dev)).sum().backward()
    assert maxrel(te.embedding.weight.grad, p["embedding.weight"].grad) < 2e-5
