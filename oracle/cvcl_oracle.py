"""CPU oracle for the CVCL contrastive hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain restatement (torch-CPU tensor arithmetic, fp32 or fp64) of the
algorithm the reference runs on the path SURVEY.md section 8 names.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the
product path (``multimodal-baby_amd/``) never does and fails loudly without its HIP
library.

Every function cites the reference file:line it follows (paths relative to
``/root/reference``).  It is a *functional* restatement: parameters arrive as a flat
``dict`` keyed with the reference's own ``state_dict`` names, so the reference's modules
can be run against it in this container (``oracle/gen_golden.py``) and the product's
modules on the GPU box.

Pinning status
--------------
* text encoders, L2 normalise, similarity, InfoNCE, ViT, ``training_step`` dict:
  PINNED against the reference itself, imported in the build container under
  ``sys.modules`` stubs by ``oracle/gen_golden.py`` (fixtures in ``tests/golden``).
* ``resnext50_32x4d``: PARITY UNPINNED.  The arithmetic lives in
  ``torchvision==0.19.0`` (``pyproject.toml:7``), which is neither vendored in the
  reference nor installed here.  The restatement below follows torchvision's published
  ``ResNet(Bottleneck, [3,4,6,3], groups=32, width_per_group=4)`` definition (v1.5:
  stride on the 3x3) and is anchored on the reference's call sites
  (``multimodal/multimodal.py:96-102,155-158,192``; ``multimodal/utils.py:207-209``),
  the 25 028 904-parameter / 4.23 GMAC counts and self-consistency tests.  Second anchor
  (round 6): with groups = 1 these functions are ResNet-50 v1.5 and agree with the
  independent ``transformers.ResNetModel`` on the same weights to 1e-9 in float64
  (``tests/test_oracle_resnet_anchor.py``: wiring, strides, BatchNorm train / eval
  semantics) -- the ``groups = 32`` keyword and the widths are what stays unpinned.

``quant`` argument: ``None`` keeps everything fp32 (the reference numerics).  Passing
``bf16_round`` emulates the storage points of the HIP bf16 path (operands and stored
activations rounded to bf16, fp32 accumulation and fp32 statistics), which lets the
bf16 kernels be checked far tighter than the raw bf16-vs-fp32 gap.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Quant = Optional[Callable[[Tensor], Tensor]]

PAD_TOKEN_ID, UNK_TOKEN_ID, SOS_TOKEN_ID, EOS_TOKEN_ID = 0, 1, 2, 3  # multimodal_data_module.py:47-50
MAX_LEN_UTTERANCE = 25                                               # multimodal_data_module.py:37


def bf16_round(t: Tensor) -> Tensor:
    """Round-to-nearest-even to bf16 and back (storage-point emulation)."""
    return t.to(torch.bfloat16).to(t.dtype)


def _q(quant: Quant, t: Tensor) -> Tensor:
    return t if quant is None else quant(t)


# --------------------------------------------------------------------------------------
# a4  embedding text encoder                                   multimodal.py:496-503
# --------------------------------------------------------------------------------------
def embedding_meanpool(table: Tensor, x: Tensor, x_len: Tensor) -> Tuple[Tensor, Tensor]:
    """``ret = sum_l table[x[b,l]] / x_len[b]``, ``output = table[x]``.

    Follows multimodal.py:496 (gather), :503 (sum over L divided by the true length);
    pad positions contribute row 0 of the table (zero at init, zero grad: padding_idx=0,
    multimodal.py:311-312).
    """
    emb = F.embedding(x, table, padding_idx=PAD_TOKEN_ID)   # (B, L, E); row 0 receives no gradient (:311-312)
    ret = emb.sum(dim=1) / x_len.unsqueeze(1)        # int64 length promotes like the reference
    return ret, emb


def embedding_meanpool_grad(d_ret: Tensor, x: Tensor, x_len: Tensor, vocab: int) -> Tensor:
    """Gradient of ``embedding_meanpool`` w.r.t. the table.

    ``dTable[x[b,l]] += d_ret[b] / len[b]`` for every position with ``x != 0``
    (padding_idx row gets no gradient), accumulated in (b, l) order.
    """
    B, L = x.shape
    d = torch.zeros(vocab, d_ret.shape[1], dtype=d_ret.dtype)
    contrib = d_ret / x_len.unsqueeze(1)
    for b in range(B):
        for l in range(L):
            t = int(x[b, l])
            if t != PAD_TOKEN_ID:
                d[t] += contrib[b]
    return d


# --------------------------------------------------------------------------------------
# a5  LSTM text encoder                                        multimodal.py:513-552
# --------------------------------------------------------------------------------------
def lstm_text(p: Dict[str, Tensor], x: Tensor, x_len: Tensor, prefix: str = "") -> Tuple[Tensor, Tensor]:
    """One-layer uni-directional LSTM over variable-length sequences (eval mode).

    multimodal.py:516 zeros h0/c0 (:671-688); :522-534 pack + ``nn.LSTM``; :538 pad;
    :552 ``ret = hidden.mean(dim=0)`` = the hidden state at each sequence's last valid
    step.  Gate order i,f,g,o with ``b_ih + b_hh`` (torch ``nn.LSTM``).
    Returns (ret [B,E], raw_output [B,Lmax,E] zero beyond each length).
    """
    table = p[prefix + "embedding.weight"]
    w_ih, w_hh = p[prefix + "lstm.weight_ih_l0"], p[prefix + "lstm.weight_hh_l0"]
    b = p[prefix + "lstm.bias_ih_l0"] + p[prefix + "lstm.bias_hh_l0"]
    B, L = x.shape
    H = w_hh.shape[1]
    emb = F.embedding(x, table, padding_idx=PAD_TOKEN_ID)
    h = torch.zeros(B, H, dtype=table.dtype)
    c = torch.zeros(B, H, dtype=table.dtype)
    Lmax = int(x_len.max())
    out = torch.zeros(B, Lmax, H, dtype=table.dtype)
    for t in range(Lmax):
        gates = emb[:, t] @ w_ih.t() + h @ w_hh.t() + b
        i, f, g, o = gates.split(H, dim=1)
        c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h_new = torch.sigmoid(o) * torch.tanh(c_new)
        live = (x_len > t).unsqueeze(1)
        c = torch.where(live, c_new, c)
        h = torch.where(live, h_new, h)
        out[:, t] = torch.where(live, h_new, torch.zeros_like(h_new))
    return h, out


def bilstm_text(p: Dict[str, Tensor], x: Tensor, x_len: Tensor, prefix: str = "") -> Tuple[Tensor, Tensor]:
    """Bidirectional one-layer LSTM (multimodal.py:513-552 with text_encoder == 'bilstm'): the backward direction runs over
    each sequence from its last valid token to its first; per-step outputs are the mean of both directions (:540-547) and
    the flat feature the mean of the two final hidden states (:552).  Returns (ret [B,E], raw_output [B,Lmax,E])."""
    table = p[prefix + "embedding.weight"]
    B, L = x.shape
    emb = F.embedding(x, table, padding_idx=PAD_TOKEN_ID)
    Lmax = int(x_len.max())

    def run(sfx, reverse):
        w_ih, w_hh = p[prefix + "lstm.weight_ih_l0" + sfx], p[prefix + "lstm.weight_hh_l0" + sfx]
        b = p[prefix + "lstm.bias_ih_l0" + sfx] + p[prefix + "lstm.bias_hh_l0" + sfx]
        Hd = w_hh.shape[1]
        h = torch.zeros(B, Hd, dtype=table.dtype)
        c = torch.zeros(B, Hd, dtype=table.dtype)
        outs = [None] * Lmax
        steps = range(Lmax - 1, -1, -1) if reverse else range(Lmax)
        for t in steps:
            gates = emb[:, t] @ w_ih.t() + h @ w_hh.t() + b
            i, f, g, o = gates.split(Hd, dim=1)
            c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h_new = torch.sigmoid(o) * torch.tanh(c_new)
            live = (x_len > t).unsqueeze(1)
            c = torch.where(live, c_new, c)
            h = torch.where(live, h_new, h)
            outs[t] = torch.where(live, h_new, torch.zeros_like(h_new))
        return h, torch.stack(outs, dim=1)

    h_f, out_f = run("", False)
    h_b, out_b = run("_reverse", True)
    return (h_f + h_b) / 2, (out_f + out_b) / 2


def cbow_text(p: Dict[str, Tensor], x: Tensor, crange: int, prefix: str = "") -> Tensor:
    """Continuous bag of words (multimodal.py:505-511, spatial embeddings only): every position gets the sum of its
    neighbours within +-crange (zeros outside the sequence tensor, the position itself excluded) / (2 crange)."""
    emb = F.embedding(x, p[prefix + "embedding.weight"], padding_idx=PAD_TOKEN_ID)
    B, L, E = emb.shape
    out = torch.zeros_like(emb)
    for j in range(L):
        lo, hi = max(j - crange, 0), min(j + crange, L - 1)
        out[:, j] = (emb[:, lo:hi + 1].sum(dim=1) - emb[:, j]) / (2 * crange)
    return out


# --------------------------------------------------------------------------------------
# a6  transformer text encoder                                 multimodal.py:553-573
# --------------------------------------------------------------------------------------
def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def transformer_text(p: Dict[str, Tensor], x: Tensor, x_len: Tensor, pos_embed_type: str,
                     prefix: str = "", nhead: int = 8) -> Tuple[Tensor, Tensor]:
    """One post-norm ``nn.TransformerEncoderLayer`` (d=E, 8 heads, ff 2048, ReLU, eps 1e-5), eval mode.

    multimodal.py:555 key padding mask ``x == 0``; :558 (L,B,E); :561-563 + pos_embed[:L];
    :565 encoder; :573 ``sum over ALL L positions / true length`` (pads included, SURVEY
    Appendix C.1).  Weights come from ``transformer_encoder.layers.0.*`` (the live copy;
    ``encoder_layer.*`` is the dead duplicate, Appendix C.2).
    """
    lp = prefix + "transformer_encoder.layers.0."
    table = p[prefix + "embedding.weight"]
    B, L = x.shape
    E = table.shape[1]
    hd = E // nhead
    h = F.embedding(x, table, padding_idx=PAD_TOKEN_ID)    # (B, L, E)
    if pos_embed_type in ("sinusoidal", "learned"):
        h = h + p[prefix + "pos_embed"][:L, 0].unsqueeze(0)
    pad = (x == PAD_TOKEN_ID)                               # (B, L) keys to ignore
    qkv = h @ p[lp + "self_attn.in_proj_weight"].t() + p[lp + "self_attn.in_proj_bias"]
    q, k, v = qkv.split(E, dim=-1)
    q = q.view(B, L, nhead, hd).transpose(1, 2)             # (B, h, L, hd)
    k = k.view(B, L, nhead, hd).transpose(1, 2)
    v = v.view(B, L, nhead, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    s = s.masked_fill(pad[:, None, None, :], float("-inf"))
    a = torch.softmax(s, dim=-1) @ v                        # (B, h, L, hd)
    a = a.transpose(1, 2).reshape(B, L, E)
    a = a @ p[lp + "self_attn.out_proj.weight"].t() + p[lp + "self_attn.out_proj.bias"]
    h = layer_norm(h + a, p[lp + "norm1.weight"], p[lp + "norm1.bias"], 1e-5)
    f = torch.relu(h @ p[lp + "linear1.weight"].t() + p[lp + "linear1.bias"])
    f = f @ p[lp + "linear2.weight"].t() + p[lp + "linear2.bias"]
    h = layer_norm(h + f, p[lp + "norm2.weight"], p[lp + "norm2.bias"], 1e-5)
    ret = h.sum(dim=1) / x_len.unsqueeze(1)
    return ret, h


# --------------------------------------------------------------------------------------
# a7  encode_* normalisation                                   multimodal.py:736,743
# --------------------------------------------------------------------------------------
def l2_normalize(x: Tensor, eps: float = 1e-12) -> Tensor:
    """``F.normalize(x, p=2, dim=-1)``: ``x / max(||x||_2, eps)``."""
    n = torch.linalg.vector_norm(x, 2, dim=-1, keepdim=True)   # zero-safe subgradient, as F.normalize
    return x / torch.clamp(n, min=eps)


# --------------------------------------------------------------------------------------
# a8  similarity logits                                        multimodal.py:755,783-787
# --------------------------------------------------------------------------------------
def similarity_logits(image_features: Tensor, text_features: Tensor,
                      logit_neg_log_temperature: Tensor) -> Tuple[Tensor, Tensor]:
    match = image_features @ text_features.t()               # :755
    logit_scale = logit_neg_log_temperature.exp()            # :784
    return match * logit_scale, match.t() * logit_scale      # :786-787


# f4  language-model cross entropy                              multimodal.py:861-890, multimodal_lit.py:266-300
def lm_ce_loss(outputs: Tensor, out_weight: Tensor, out_bias, y: Tensor, regressional: bool):
    """logits = output_layer(outputs) (:859); regressional (LSTM): predict token l+1 from position l (:879-883), else
    labels = y; token-wise F.cross_entropy with ignore_index = PAD (:884-889) -> (loss [B,L'], labels [B,L'])."""
    logits = outputs @ out_weight.t()
    if out_bias is not None:
        logits = logits + out_bias
    if regressional:
        logits = logits[:, :-1]
        labels = y[:, 1:1 + logits.size(1)]
    else:
        labels = y
    loss = F.cross_entropy(logits.transpose(-2, -1), labels, ignore_index=PAD_TOKEN_ID, reduction="none")
    return loss, labels


def lm_loss_summaries(ce_loss: Tensor, labels: Tensor):
    """the three masked means of multimodal_lit.py:284-300: all non-pad tokens, without <sos>, without <sos>/<eos>."""
    mask = labels != PAD_TOKEN_ID
    n0 = mask.sum()
    l0 = ce_loss.sum() / n0
    mask = mask & (labels != SOS_TOKEN_ID)
    n1 = mask.sum()
    l1 = (ce_loss * mask).sum() / n1
    mask = mask & (labels != EOS_TOKEN_ID)
    n2 = mask.sum()
    l2 = (ce_loss * mask).sum() / n2
    return (l0, l1, l2), (n0, n1, n2)


# f4  spatial similarity (embedding_type == "spatial")         multimodal.py:757-787
def spatial_similarity_logits(image_features: Tensor, text_features: Tensor, text_length: Tensor,
                              logit_neg_log_temperature: Tensor, sim: str = "max") -> Tuple[Tensor, Tensor]:
    """image_features [Bi,E,H,W], text_features [Bt,L,E] (per-word outputs), text_length [Bt].
    mean: sum of all location x word dot products / (H W len[t])                       (:761-769)
    max:  for every word the best location, summed over ALL L positions / len[t]       (:770-780)"""
    Bi, E, Hh, Ww = image_features.shape
    img = image_features.reshape(Bi, E, Hh * Ww)
    if sim == "mean":
        match = torch.einsum("iep,tle->it", img, text_features) / (Hh * Ww * text_length)
    elif sim == "max":
        mm = torch.einsum("iep,tle->itlp", img, text_features)
        match = mm.amax(dim=3).sum(dim=2) / text_length
    else:
        raise ValueError(sim)
    logit_scale = logit_neg_log_temperature.exp()
    return match * logit_scale, match.t() * logit_scale


# --------------------------------------------------------------------------------------
# a9  symmetric InfoNCE + accuracies + entropies               multimodal.py:801-818, utils.py:106-108
# --------------------------------------------------------------------------------------
def get_entropy(logits: Tensor, dim: int = -1) -> Tensor:
    log_p = torch.log_softmax(logits, dim=dim)               # utils.py:107
    return (torch.softmax(log_p, dim=dim) * -log_p).sum(dim=dim)   # utils.py:108


def cross_entropy_diag(logits: Tensor) -> Tensor:
    """``F.cross_entropy(logits, arange(N))`` with mean reduction (multimodal.py:803-810)."""
    lse = torch.logsumexp(logits, dim=-1)
    return (lse - logits.diagonal()).mean()


def contrastive_loss(logits_per_image: Tensor, logits_per_text: Tensor):
    """Returns (infonce, image_acc, text_acc, image_entropy, text_entropy) as 0-dim tensors."""
    n = logits_per_image.shape[0]
    gt = torch.arange(n)
    infonce = (cross_entropy_diag(logits_per_image) + cross_entropy_diag(logits_per_text)) / 2   # :808-810
    image_acc = (logits_per_image.argmax(dim=-1) == gt).sum() / n                                 # :813-815
    text_acc = (logits_per_text.argmax(dim=-1) == gt).sum() / n                                   # :814-816
    image_ent = get_entropy(logits_per_image).mean()                                              # :817
    text_ent = get_entropy(logits_per_text).mean()                                                # :818
    return infonce, image_acc, text_acc, image_ent, text_ent


def infonce_dlogits(logits_per_image: Tensor) -> Tensor:
    """d(infonce)/d(logits_per_image) with logits_per_text its transpose."""
    n = logits_per_image.shape[0]
    eye = torch.eye(n, dtype=logits_per_image.dtype)
    pr = torch.softmax(logits_per_image, dim=1)
    pc = torch.softmax(logits_per_image, dim=0)
    return ((pr - eye) + (pc - eye)) / (2 * n)


# --------------------------------------------------------------------------------------
# a2  ResNeXt-50 32x4d  (torchvision definition; PARITY UNPINNED, see header)
# --------------------------------------------------------------------------------------
RESNEXT_LAYERS = (3, 4, 6, 3)
RESNEXT_GROUPS = 32
RESNEXT_WIDTH_PER_GROUP = 4
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def resnext50_conv_specs():
    """[(name, cin, cout, k, stride, pad, groups)] in torchvision state_dict order,
    each followed by a BatchNorm named by ``bn_name_of``."""
    specs = [("conv1", 3, 64, 7, 2, 3, 1)]
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), RESNEXT_LAYERS), start=1):
        width = int(planes * (RESNEXT_WIDTH_PER_GROUP / 64.0)) * RESNEXT_GROUPS
        for bi in range(blocks):
            stride = 2 if (li > 1 and bi == 0) else 1
            pre = f"layer{li}.{bi}."
            specs.append((pre + "conv1", inplanes, width, 1, 1, 0, 1))
            specs.append((pre + "conv2", width, width, 3, stride, 1, RESNEXT_GROUPS))
            specs.append((pre + "conv3", width, planes * 4, 1, 1, 0, 1))
            if bi == 0:
                specs.append((pre + "downsample.0", inplanes, planes * 4, 1, stride, 0, 1))
            inplanes = planes * 4
    return specs


def bn_name_of(conv_name: str) -> str:
    if conv_name == "conv1":
        return "bn1"
    if conv_name.endswith("downsample.0"):
        return conv_name[:-1] + "1"
    return conv_name[:-5] + "bn" + conv_name[-1]


def batch_norm(x: Tensor, p: Dict[str, Tensor], name: str, training: bool,
               stats_out: Optional[Dict[str, Tensor]] = None, impl: str = "explicit", centre: Optional[Tensor] = None) -> Tensor:
    """``nn.BatchNorm2d`` (eps 1e-5, momentum 0.1): batch statistics + running-stat EMA with
    the unbiased variance in training mode, running statistics in eval mode.  ``stats_out``
    receives the *updated* running buffers (the reference mutates them in place, also when the
    CNN is frozen: SURVEY 0.4).  ``impl="torch"`` runs the same layer through ``F.batch_norm`` on
    copies of the running buffers -- exactly what ``nn.BatchNorm2d.forward`` executes on the host;
    the CPU baseline uses it so that it times what the reference would run (tests/test_oracle_golden.py
    checks the two forms against each other).
    ``centre`` (the HIP path's centred storage, include/cvcl_hip.h): ``x`` is the tensor stored as y - centre; BatchNorm is
    invariant under that shift, only the running mean (train) / the mean that is subtracted (eval) need the centre back."""
    w, b = p[name + ".weight"], p[name + ".bias"]
    rm, rv = p[name + ".running_mean"], p[name + ".running_var"]
    c = 0.0 if centre is None else centre
    if impl == "torch":
        assert centre is None
        rm2, rv2 = rm.detach().clone(), rv.detach().clone()
        y = F.batch_norm(x, rm2, rv2, w, b, training, BN_MOMENTUM, BN_EPS)
        if training and stats_out is not None:
            stats_out[name + ".running_mean"], stats_out[name + ".running_var"] = rm2, rv2
            stats_out[name + ".num_batches_tracked"] = p[name + ".num_batches_tracked"] + 1
        return y
    if training:
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=(0, 2, 3))
        var = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
        if stats_out is not None:
            stats_out[name + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * (mean.detach() + c)
            stats_out[name + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var.detach() * n / max(n - 1, 1)
            stats_out[name + ".num_batches_tracked"] = p[name + ".num_batches_tracked"] + 1
    else:
        mean, var = rm - c, rv
    scale = w / torch.sqrt(var + BN_EPS)
    shift = b - mean * scale
    return x * scale[None, :, None, None] + shift[None, :, None, None]


def _centre_of(centres, pp, name):
    """Storage centre of conv ``name``'s raw output: None (plain), the string "running_mean" (the HIP library's eval-mode
    default: every stored tensor is y - running_mean) or a dict conv name -> [C] tensor (calibrated / tracked centres)."""
    if centres is None:
        return None
    if isinstance(centres, str):
        assert centres == "running_mean"
        return pp[bn_name_of(name) + ".running_mean"]
    return centres.get(name)


def _conv_bn(pp, inp, name, stride, pad, groups, relu, training, quant, stats_out, taps, bn_impl="explicit", conv_fn=None,
             centres=None):
    y = (conv_fn or F.conv2d)(_q(quant, inp), _q(quant, pp[name + ".weight"]), None, stride, pad, 1, groups)
    c = _centre_of(centres, pp, name)
    if c is not None:
        y = y - c[None, :, None, None]
    y = _q(quant, y)                                        # raw conv output as stored (centred storage: y - c)
    if taps is not None:
        taps[name + ".raw"] = y
    y = batch_norm(y, pp, bn_name_of(name), training, stats_out, bn_impl, c)
    return torch.relu(y) if relu else y


def resnext50_stem(pp, x, training, quant: Quant = None, stats_out=None, taps=None, bn_impl="explicit", centres=None,
                  conv_fn=None) -> Tensor:
    """conv1 7x7/2 -> bn1 -> relu -> maxpool 3x3/2 pad 1."""
    h = _conv_bn(pp, x, "conv1", 2, 3, 1, True, training, quant, stats_out, taps, bn_impl, conv_fn, centres)
    h = _q(quant, F.max_pool2d(h, 3, 2, 1))
    if taps is not None:
        taps["maxpool"] = h
    return h


def resnext50_block(pp, h, li: int, bi: int, training, quant: Quant = None, stats_out=None, taps=None, bn_impl="explicit",
                   conv_fn=None, centres=None) -> Tensor:
    """One torchvision ``Bottleneck`` of ``layer{li}``: 1x1 -> grouped 3x3 (stride here, v1.5) -> 1x1, + identity /
    1x1-stride-s downsample on the first block, ReLU after the add.  ``conv_fn`` (tests only) replaces ``F.conv2d`` -- e.g.
    by the same convolution with its input channels visited in another order, to measure what fp32 summation order alone
    does to a block's output."""
    pre = f"layer{li}.{bi}."
    stride = 2 if (li > 1 and bi == 0) else 1
    a = (training, quant, stats_out, taps, bn_impl, conv_fn, centres)
    o = _q(quant, _conv_bn(pp, h, pre + "conv1", 1, 0, 1, True, *a))
    o = _q(quant, _conv_bn(pp, o, pre + "conv2", stride, 1, RESNEXT_GROUPS, True, *a))
    o = _conv_bn(pp, o, pre + "conv3", 1, 0, 1, False, *a)
    idn = _conv_bn(pp, h, pre + "downsample.0", stride, 0, 1, False, *a) if bi == 0 else h
    h = _q(quant, torch.relu(o + idn))
    if taps is not None:
        taps[pre + "out"] = h
    return h


def resnext50_stage(pp, h, li: int, training, quant: Quant = None, stats_out=None, taps=None, bn_impl="explicit",
                   centres=None, conv_fn=None) -> Tensor:
    """``layer{li}``: Bottleneck x RESNEXT_LAYERS[li-1]."""
    for bi in range(RESNEXT_LAYERS[li - 1]):
        h = resnext50_block(pp, h, li, bi, training, quant, stats_out, taps, bn_impl, conv_fn, centres)
    return h


def resnext50_forward(p: Dict[str, Tensor], x: Tensor, training: bool, quant: Quant = None,
                      prefix: str = "", stats_out: Optional[Dict[str, Tensor]] = None,
                      taps: Optional[Dict[str, Tensor]] = None, bn_impl: str = "explicit", centres=None, conv_fn=None):
    """torchvision ``ResNet.forward`` for resnext50_32x4d up to and including avgpool+flatten.

    Returns (pooled [B,2048] fp32, layer4 feature map [B,2048,7,7]).  The ``fc`` is applied
    by the caller (it is the trainable projection swapped in at multimodal.py:192).
    ``quant`` marks the HIP bf16 path's storage points: conv operands, raw conv outputs
    (BN statistics are taken from the stored tensor), normalised activations, block outputs.
    ``taps`` (optional) collects intermediate tensors by name for layer-wise kernel tests.
    ``stats_out`` keys carry no prefix.
    ``centres`` models the HIP path's centred storage of the raw conv outputs (``_centre_of``): stored = quant(y - c).
    ``conv_fn`` (tests only) replaces ``F.conv2d``, e.g. by ``reordered_conv2d``: same mathematics, other fp32 summation
    order -- the yardstick of what bf16 storage does to a chaotic network.
    """
    pp = {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)} if prefix else p
    h = resnext50_stem(pp, x, training, quant, stats_out, taps, bn_impl, centres, conv_fn)
    for li in (1, 2, 3, 4):
        h = resnext50_stage(pp, h, li, training, quant, stats_out, taps, bn_impl, centres, conv_fn)
    pooled = h.mean(dim=(2, 3))
    return pooled, h


def reordered_conv2d(x, w, bias, stride, pad, dil, groups):
    """``F.conv2d`` with the input channels of every group visited in reverse order: same mathematics, other fp32 summation order."""
    cg = w.shape[1]
    idx = torch.arange(x.shape[1]).view(groups, cg).flip(1).reshape(-1)
    return F.conv2d(x[:, idx].contiguous(), w.flip(1).contiguous(), bias, stride, pad, dil, groups)


def resnext50_batch_means(p: Dict[str, Tensor], x: Tensor, quant: Quant = None) -> Dict[str, Tensor]:
    """conv name -> per-channel batch mean of its raw output in a plain-storage train-mode pass: what the HIP trunk's
    calibration pass leaves behind as storage centres (multimodal/resnext.py ``_calibrated_centres``)."""
    taps: Dict[str, Tensor] = {}
    resnext50_forward(p, x, True, quant, taps=taps)
    return {k[:-4]: v.mean(dim=(0, 2, 3)) for k, v in taps.items() if k.endswith(".raw")}


def resnext50_random_params(seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    """torchvision's init: kaiming-normal(fan_out, relu) convs, BN weight 1 / bias 0,
    ``zero_init_residual=False``; running_mean 0, running_var 1.  (No fc: the caller owns it.)"""
    g = torch.Generator().manual_seed(seed)
    p: Dict[str, Tensor] = {}
    for name, cin, cout, k, stride, pad, groups in resnext50_conv_specs():
        fan_out = cout * k * k
        p[name + ".weight"] = (torch.randn(cout, cin // groups, k, k, generator=g) * math.sqrt(2.0 / fan_out)).to(dtype)
        bn = bn_name_of(name)
        p[bn + ".weight"] = torch.ones(cout, dtype=dtype)
        p[bn + ".bias"] = torch.zeros(cout, dtype=dtype)
        p[bn + ".running_mean"] = torch.zeros(cout, dtype=dtype)
        p[bn + ".running_var"] = torch.ones(cout, dtype=dtype)
        p[bn + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return p


def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    y = x @ w.t()
    return y if b is None else y + b


# --------------------------------------------------------------------------------------
# a3  DINO ViT                                                 vision_transformer_dino_mugs.py
# --------------------------------------------------------------------------------------
def gelu_erf(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))    # nn.GELU default, vit:88


def vit_interpolate_pos_encoding(pos_embed: Tensor, gh: int, gw: int) -> Tensor:
    """``VisionTransformer.interpolate_pos_encoding`` (vit:210-230): the [1, 1 + N, D] learned table resampled (bicubic,
    scale factors (gh + 0.1) / sqrt(N), (gw + 0.1) / sqrt(N) -- the reference's guard against a floor() one short) to a gh x gw
    patch grid; identity when the grid is the native square one."""
    N = pos_embed.shape[1] - 1
    if gh * gw == N and gh == gw:
        return pos_embed
    D = pos_embed.shape[-1]
    s = int(math.sqrt(N))
    grid = pos_embed[:, 1:].reshape(1, s, s, D).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=((gh + 0.1) / math.sqrt(N), (gw + 0.1) / math.sqrt(N)), mode="bicubic")
    assert grid.shape[-2] == gh and grid.shape[-1] == gw
    return torch.cat([pos_embed[:, :1], grid.permute(0, 2, 3, 1).reshape(1, gh * gw, D)], dim=1)


def fp8_rows(y: Tensor) -> Tensor:
    """e4m3 storage-point emulation with one scale per row (amax / 448; 1 for an all-zero row): the per-token activation scales and
    per-output-channel weight scales of the product's fp8 linears (BASELINE configs[4]; this repo's storage points, not the reference's)."""
    amax = y.abs().amax(dim=-1, keepdim=True)
    s = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    return (y / s).clamp(-448, 448).to(torch.float8_e4m3fn).to(y.dtype) * s


def fp8_mx(y: Tensor) -> Tensor:
    """e4m3 with one power-of-two (e8m0) scale per 32 consecutive elements of the last dimension: 2^ceil(log2(amax / 448))."""
    shp = y.shape
    blk = y.reshape(-1, shp[-1] // 32, 32)
    x = (blk.abs().amax(dim=2) / 448.0).float()
    bits = x.view(torch.int32)
    e = ((bits >> 23) & 0xff) + ((bits & 0x7fffff) != 0).int()
    e = e.clamp(1, 254)
    scale = torch.pow(2.0, (e - 127).double()).to(y.dtype)[:, :, None]
    return ((blk / scale).clamp(-448, 448).to(torch.float8_e4m3fn).to(y.dtype) * scale).reshape(shp)


def fp8_folded_linear(h: Tensor, W: Tensor, b: Tensor, gamma: Tensor, beta: Tensor, eps: float) -> Tensor:
    """LayerNorm(h) W^T + b with the e4m3 storage points of the product's FOLDED form (round 5; csrc/gemm8f_kernel.h KIND 3 / 4):
    the RAW rows h (already at their bf16 storage point) MX-quantised per 32 elements, W diag(gamma) quantised per output channel,
    and the normalisation applied to the product:  rstd (h8 . W8^T) - mean rstd s + (b + W beta),  s = the row sums of the
    dequantised weight; mean / rstd are those of the stored rows."""
    h8 = fp8_mx(h)
    W8 = fp8_rows(W * gamma[None, :])
    mean = h.mean(dim=-1, keepdim=True)
    rstd = torch.rsqrt(h.var(dim=-1, unbiased=False, keepdim=True) + eps)
    return rstd * (h8 @ W8.t()) - (rstd * mean) * W8.sum(dim=1) + (b + W @ beta)


def vit_forward(p: Dict[str, Tensor], x: Tensor, patch: int, num_heads: int, quant: Quant = None,
                prefix: str = "", eps: float = 1e-6, taps: Optional[Dict[str, Tensor]] = None, fp8: bool = False,
                fp8_fold: bool = False) -> Tensor:
    """``VisionTransformer.forward`` (vit:245-250): prepare_tokens (:232-243; pos-embed
    interpolation :210-230, the identity at the native resolution) -> depth x pre-LN
    ``Block`` (:133-149; attention :106-130 scale head_dim**-0.5, MLP :87-103 GELU-erf)
    -> LayerNorm -> cls token.  Returns [B, D] (the ``head`` is applied by the caller,
    multimodal.py:91-92).  ``fp8``: the e4m3 storage points of the product's configs[4] path on top of ``quant`` -- the operands of
    the four linears of every block: LayerNorm outputs per row, attention and GELU outputs per 32-element block, weights per row;
    ``fp8_fold``: LayerNorm folded into qkv / fc1 (fp8_folded_linear) as the product does at benchmark sizes."""
    pp = {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)} if prefix else p
    B, C, H, W = x.shape
    D = pp["cls_token"].shape[-1]
    gh, gw = H // patch, W // patch
    # PatchEmbed conv k=s=patch == unfold + GEMM (vit:162,166)
    cols = x.reshape(B, C, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * patch * patch)
    wpe = pp["patch_embed.proj.weight"].reshape(D, -1)
    tok = _q(quant, cols) @ _q(quant, wpe).t() + pp["patch_embed.proj.bias"]
    h = torch.cat([pp["cls_token"].expand(B, -1, -1), tok], dim=1) + vit_interpolate_pos_encoding(pp["pos_embed"], gh, gw)   # :237-241
    h = _q(quant, h)
    T = h.shape[1]
    hd = D // num_heads
    depth = 1 + max(int(k.split(".")[1]) for k in pp if k.startswith("blocks."))
    for i in range(depth):
        bp = f"blocks.{i}."
        qa = (lambda t: fp8_rows(t)) if fp8 else (lambda t: _q(quant, t))           # per-row e4m3 | the bf16 storage point
        qm = (lambda t: fp8_mx(_q(quant, t))) if fp8 else (lambda t: _q(quant, t))  # MX e4m3 of the bf16-rounded tensor
        qw = (lambda t: fp8_rows(t)) if fp8 else (lambda t: _q(quant, t))
        if fp8 and fp8_fold:                                                             # (the product's folded form, see above)
            qkv = _q(quant, fp8_folded_linear(h, pp[bp + "attn.qkv.weight"], pp[bp + "attn.qkv.bias"], pp[bp + "norm1.weight"],
                                              pp[bp + "norm1.bias"], eps))
        else:
            y = qa(layer_norm(h, pp[bp + "norm1.weight"], pp[bp + "norm1.bias"], eps))
            qkv = _q(quant, y @ qw(pp[bp + "attn.qkv.weight"]).t() + pp[bp + "attn.qkv.bias"])
        qkv = qkv.reshape(B, T, 3, num_heads, hd).permute(2, 0, 3, 1, 4)                # :119
        q, k, v = qkv[0], qkv[1], qkv[2]
        a = torch.softmax((q @ k.transpose(-2, -1)) * (hd ** -0.5), dim=-1)             # :123-124
        o = qm((_q(quant, a) @ v).transpose(1, 2).reshape(B, T, D))                     # :127
        o = o @ qw(pp[bp + "attn.proj.weight"]).t() + pp[bp + "attn.proj.bias"]
        h = _q(quant, h + _q(quant, o))                                                  # :146
        if fp8 and fp8_fold:
            f = qm(gelu_erf(fp8_folded_linear(h, pp[bp + "mlp.fc1.weight"], pp[bp + "mlp.fc1.bias"], pp[bp + "norm2.weight"],
                                              pp[bp + "norm2.bias"], eps)))
        else:
            y = qa(layer_norm(h, pp[bp + "norm2.weight"], pp[bp + "norm2.bias"], eps))
            f = qm(gelu_erf(y @ qw(pp[bp + "mlp.fc1.weight"]).t() + pp[bp + "mlp.fc1.bias"]))
        f = f @ qw(pp[bp + "mlp.fc2.weight"]).t() + pp[bp + "mlp.fc2.bias"]
        h = _q(quant, h + _q(quant, f))                                                  # :147
        if taps is not None:
            taps[bp + "out"] = h
    cls = layer_norm(h[:, 0], pp["norm.weight"], pp["norm.bias"], eps)                   # :249-250
    return cls


# --------------------------------------------------------------------------------------
# a1/a7/a8/a9 composed:  MultiModalModel.forward / calculate_contrastive_loss
# --------------------------------------------------------------------------------------
def cvcl_forward(p: Dict[str, Tensor], image: Tensor, text: Tensor, text_len: Tensor, *,
                 vision: str = "resnext", text_encoder: str = "embedding", normalize_features: bool,
                 training: bool, quant: Quant = None, vit_patch: int = 14, vit_heads: int = 12,
                 pos_embed_type: str = "no_pos_embed", stats_out=None, bn_impl: str = "explicit"):
    """``MultiModalModel.forward(..., return_image_features=True, return_text_outputs=True)``
    (multimodal.py:746-794), flat embedding branch.  ``p`` uses ``MultiModalModel.state_dict()``
    names: ``image_embed.model.*``, ``text_embed.*``, ``logit_neg_log_temperature``."""
    ip = "image_embed.model."
    if vision == "resnext":
        pooled, fmap = resnext50_forward(p, image, training, quant, ip, stats_out, bn_impl=bn_impl)
        img = linear(pooled, p[ip + "fc.weight"], p[ip + "fc.bias"])           # multimodal.py:101,192
    else:
        cls = vit_forward(p, image, vit_patch, vit_heads, quant, ip)
        img = linear(cls, p[ip + "head.weight"], p[ip + "head.bias"])          # multimodal.py:91-92,190
        fmap = None
    tp = "text_embed."
    if text_encoder == "embedding":
        txt, tout = embedding_meanpool(p[tp + "embedding.weight"], text, text_len)
    elif text_encoder == "lstm":
        txt, tout = lstm_text(p, text, text_len, tp)
    elif text_encoder == "transformer":
        txt, tout = transformer_text(p, text, text_len, pos_embed_type, tp)
    else:
        raise ValueError(text_encoder)
    if normalize_features:
        img, txt = l2_normalize(img), l2_normalize(txt)                        # multimodal.py:736,743
    lpi, lpt = similarity_logits(img, txt, p["logit_neg_log_temperature"])
    return lpi, lpt, img, fmap, tout


def cvcl_contrastive_loss(p, image, text, text_len, **kw):
    """``MultiModalModel.calculate_contrastive_loss`` 10-tuple (multimodal.py:796-822)."""
    lpi, lpt, img, fmap, tout = cvcl_forward(p, image, text, text_len, **kw)
    return contrastive_loss(lpi, lpt) + (lpi, lpt, img, fmap, tout)


# --------------------------------------------------------------------------------------
# synthetic inputs and the CPU baseline train step
# --------------------------------------------------------------------------------------
IMAGENET_MEAN = (0.485, 0.456, 0.406)    # multimodal_data_module.py:57
IMAGENET_STD = (0.229, 0.224, 0.225)


def synthetic_batch(batch: int, seed: int = 0, n_words: int = 3, vocab: int = 2350, pad_to: Optional[int] = None):
    """Images ``rand -> ImageNet normalise`` (tests/test_cvcl.py:14, data_module.py:57); utterances
    ``<sos> w1..wn <eos>`` with ``w ~ U{4..vocab-1}`` (saycam_dm.py:101-105)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(batch, 3, 224, 224, generator=g)
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    img = (img - mean) / std
    words = torch.randint(4, vocab, (batch, n_words), generator=g)
    tok = torch.cat([torch.full((batch, 1), SOS_TOKEN_ID), words, torch.full((batch, 1), EOS_TOKEN_ID)], dim=1)
    ln = torch.full((batch,), n_words + 2, dtype=torch.long)
    if pad_to is not None and pad_to > tok.shape[1]:
        tok = F.pad(tok, (0, pad_to - tok.shape[1]), value=PAD_TOKEN_ID)
    return img, tok.long(), ln


def cvcl_random_params(embedding_dim: int, seed: int = 0, vocab: int = 2350, temperature: float = 0.07):
    """Random-init parameter dict for the ResNeXt + embedding configuration (C1/C2)."""
    g = torch.Generator().manual_seed(seed + 1)
    p = {"image_embed.model." + k: v for k, v in resnext50_random_params(seed).items()}
    bound = 1.0 / math.sqrt(2048)
    p["image_embed.model.fc.weight"] = (torch.rand(embedding_dim, 2048, generator=g) * 2 - 1) * bound
    p["image_embed.model.fc.bias"] = (torch.rand(embedding_dim, generator=g) * 2 - 1) * bound
    emb = torch.randn(vocab, embedding_dim, generator=g)
    emb[PAD_TOKEN_ID] = 0                                     # nn.Embedding(padding_idx=0)
    p["text_embed.embedding.weight"] = emb
    p["logit_neg_log_temperature"] = torch.tensor(-math.log(temperature), dtype=torch.float32)
    return p


TRAINABLE_FROZEN_CNN = ("image_embed.model.fc.weight", "image_embed.model.fc.bias", "text_embed.embedding.weight")


class CpuTrainStep:
    """One contrastive train step of the frozen-CNN configuration on host cores: forward
    (BN in train mode: SURVEY 0.4), InfoNCE, backward of the trainable set, AdamW
    (multimodal_lit.py:112-114, 227-261, 445-447).  Used by bench.py's cpu_baseline leg."""

    def __init__(self, p, lr=1e-4, weight_decay=0.1, normalize_features=True, bn_impl="explicit", learn_temperature=False):
        self.p = dict(p)
        self.normalize = normalize_features
        self.bn_impl = bn_impl
        self.train = [k for k in TRAINABLE_FROZEN_CNN] + (["logit_neg_log_temperature"] if learn_temperature else [])
        for k in self.train:
            self.p[k] = self.p[k].clone().requires_grad_(True)
        self.opt = torch.optim.AdamW([self.p[k] for k in self.train], lr=lr, weight_decay=weight_decay)

    def step(self, image, text, text_len):
        stats = {}
        out = cvcl_contrastive_loss(self.p, image, text, text_len, normalize_features=self.normalize,
                                    training=True, stats_out=stats, bn_impl=self.bn_impl)
        loss = out[0]
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        self.opt.step()
        with torch.no_grad():
            for k, v in stats.items():
                self.p["image_embed.model." + k] = v
        return float(loss.detach())
