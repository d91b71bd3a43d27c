"""torch.autograd bridges onto the HIP C ABI for the contrastive head and the trainable projections.

PyTorch owns device memory, autograd bookkeeping and the optimizer; every arithmetic step on the
path is a libcvcl_hip kernel.  Functions raise if handed CPU tensors (no fallback by design).
"""
from __future__ import annotations

import torch

from . import _hip as H

_F = torch.float32


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


class EmbedMeanPool(torch.autograd.Function):
    """multimodal/multimodal.py:496-503 (reference): embedding gather, sum over L, divide by length."""

    @staticmethod
    def forward(ctx, table, tok, length, want_output: bool):
        B, L = tok.shape
        V, E = table.shape
        ret = torch.empty(B, E, dtype=_F, device=table.device)
        out = torch.empty(B, L, E, dtype=_F, device=table.device) if want_output else None
        H.check(H.lib().cvcl_embed_meanpool_fwd(H.ptr(table, _F), H.ptr(tok, torch.int64), H.ptr(length, torch.int64),
                                               H.ptr(ret), H.ptr(out), B, L, E, V, H.stream_ptr()),
                "cvcl_embed_meanpool_fwd")
        ctx.save_for_backward(tok, length)
        ctx.shape = (V, E)
        ctx.mark_non_differentiable(*([out] if out is not None else []))
        ctx.set_materialize_grads(False)       # (no zero-filled [B, L, E] gradient for the non-differentiable per-word output)
        return ret, out

    @staticmethod
    def backward(ctx, d_ret, _d_out):
        tok, length = ctx.saved_tensors
        if d_ret is None:
            return None, None, None, None
        V, E = ctx.shape
        B, L = tok.shape
        d_table = torch.empty(V, E, dtype=_F, device=d_ret.device)
        H.check(H.lib().cvcl_embed_meanpool_bwd(H.ptr(d_ret.contiguous(), _F), H.ptr(tok), H.ptr(length),
                                               H.ptr(d_table), B, L, E, V, H.stream_ptr()),
                "cvcl_embed_meanpool_bwd")
        return d_table, None, None, None


class L2Normalize(torch.autograd.Function):
    """F.normalize(x, p=2, dim=-1) (reference multimodal/multimodal.py:736,743)."""

    @staticmethod
    def forward(ctx, x, eps: float):
        x = x.contiguous()
        N, E = x.shape
        y = torch.empty_like(x)
        norm = torch.empty(N, dtype=_F, device=x.device)
        H.check(H.lib().cvcl_l2norm_fwd(H.ptr(x, _F), H.ptr(y), H.ptr(norm), N, E, eps, H.stream_ptr()), "cvcl_l2norm_fwd")
        ctx.save_for_backward(y, norm)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        y, norm = ctx.saved_tensors
        N, E = y.shape
        dx = torch.empty_like(y)
        H.check(H.lib().cvcl_l2norm_bwd(H.ptr(y), H.ptr(norm), H.ptr(dy.contiguous(), _F), H.ptr(dx), N, E, ctx.eps,
                                       H.stream_ptr()), "cvcl_l2norm_bwd")
        return dx, None


class SimLogits(torch.autograd.Function):
    """logits_per_image = (img @ txt.T) * exp(neg_log_temp) (reference multimodal/multimodal.py:755,783-787)."""

    @staticmethod
    def forward(ctx, img, txt, neg_log_temp):
        img, txt = img.contiguous(), txt.contiguous()
        Ni, E = img.shape
        Nt = txt.shape[0]
        nlt = neg_log_temp.reshape(1).contiguous()
        logits = torch.empty(Ni, Nt, dtype=_F, device=img.device)
        H.check(H.lib().cvcl_sim_logits_fwd(H.ptr(img, _F), H.ptr(txt, _F), H.ptr(nlt, _F), H.ptr(logits), Ni, Nt, E,
                                           H.stream_ptr()), "cvcl_sim_logits_fwd")
        ctx.save_for_backward(img, txt, nlt, logits)
        ctx.temp_shape = neg_log_temp.shape
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        img, txt, nlt, logits = ctx.saved_tensors
        Ni, E = img.shape
        Nt = txt.shape[0]
        need_i, need_t, need_s = ctx.needs_input_grad
        d_img = torch.empty_like(img) if need_i else None
        d_txt = torch.empty_like(txt) if need_t else None
        d_s = torch.empty(1, dtype=_F, device=img.device) if need_s else None
        nb = H.lib().cvcl_sim_logits_bwd_workspace_bytes(Ni, Nt, E)
        ws = _ws(nb, img.device)
        H.check(H.lib().cvcl_sim_logits_bwd(H.ptr(img), H.ptr(txt), H.ptr(nlt), H.ptr(logits),
                                           H.ptr(d_logits.contiguous(), _F), H.ptr(d_img), H.ptr(d_txt), H.ptr(d_s),
                                           Ni, Nt, E, H.ptr(ws), nb, H.stream_ptr()), "cvcl_sim_logits_bwd")
        return d_img, d_txt, (d_s.reshape(ctx.temp_shape) if need_s else None)


class SpatialMaxLogits(torch.autograd.Function):
    """embedding_type='spatial', sim='max' (reference multimodal/multimodal.py:770-787):
    logits[i][t] = exp(nlt) * sum_l max_p <img[i,p,:], txt[t,l,:]> / len[t].
    img_rows [Bi*HW, E] (per-location features, NHWC order), txt_rows [Bt*L, E] (per-word outputs)."""

    @staticmethod
    def forward(ctx, img_rows, txt_rows, length, neg_log_temp, Bi, HW, Bt, L):
        img_rows, txt_rows = img_rows.contiguous(), txt_rows.contiguous()
        dev = img_rows.device
        nlt = neg_log_temp.reshape(1).contiguous()
        mm = H.gemm(img_rows, txt_rows)                                  # [Bi*HW, Bt*L] match map, fp32
        logits = torch.empty(Bi, Bt, dtype=_F, device=dev)
        arg = torch.empty(Bi, Bt * L, dtype=torch.uint8, device=dev)
        H.check(H.lib().cvcl_spatial_max_fwd(H.ptr(mm), H.ptr(length, torch.int64), H.ptr(nlt, _F), H.ptr(logits), H.ptr(arg),
                                             Bi, HW, Bt, L, H.stream_ptr()), "cvcl_spatial_max_fwd")
        ctx.save_for_backward(img_rows, txt_rows, length, nlt, logits, arg)
        ctx.dims = (Bi, HW, Bt, L)
        ctx.temp_shape = neg_log_temp.shape
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        img_rows, txt_rows, length, nlt, logits, arg = ctx.saved_tensors
        Bi, HW, Bt, L = ctx.dims
        dev = img_rows.device
        E = img_rows.shape[1]
        need_i, need_t, _nl, need_s = ctx.needs_input_grad[:4]
        d_mm = torch.empty(Bi * HW, Bt * L, dtype=_F, device=dev)
        d_s = torch.empty(1, dtype=_F, device=dev) if need_s else None
        s = H.stream_ptr()
        H.check(H.lib().cvcl_spatial_max_bwd(H.ptr(d_logits.contiguous(), _F), H.ptr(arg), H.ptr(length), H.ptr(nlt), H.ptr(logits),
                                             H.ptr(d_mm), H.ptr(d_s), Bi, HW, Bt, L, s), "cvcl_spatial_max_bwd")

        # the operands of both gradient GEMMs are read K-major in place (H.gemm a_trans / w_trans): no transposed copies
        d_img = H.gemm(d_mm, txt_rows, w_trans=True) if need_i else None                    # [Bi*HW, E] = d_mm . txt_rows
        d_txt = H.gemm(d_mm, img_rows, a_trans=True, w_trans=True) if need_t else None      # [Bt*L, E] = d_mm^T . img_rows
        return d_img, d_txt, None, (d_s.reshape(ctx.temp_shape) if need_s else None), None, None, None, None


class TokenCrossEntropy(torch.autograd.Function):
    """F.cross_entropy(logits [R,V], labels [R], ignore_index, reduction='none') (reference multimodal.py:884-889)."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        logits = logits.contiguous()
        R, V = logits.shape
        loss = torch.empty(R, dtype=_F, device=logits.device)
        lse = torch.empty(R, dtype=_F, device=logits.device)
        H.check(H.lib().cvcl_token_ce_fwd(H.ptr(logits, _F), H.ptr(labels, torch.int64), H.ptr(loss), H.ptr(lse), R, V, ignore_index,
                                          H.stream_ptr()), "cvcl_token_ce_fwd")
        ctx.save_for_backward(logits, labels, lse)
        ctx.ignore = ignore_index
        return loss

    @staticmethod
    def backward(ctx, d_loss):
        logits, labels, lse = ctx.saved_tensors
        R, V = logits.shape
        d_logits = torch.empty_like(logits)
        H.check(H.lib().cvcl_token_ce_bwd(H.ptr(logits), H.ptr(labels), H.ptr(lse), H.ptr(d_loss.contiguous(), _F), H.ptr(d_logits),
                                          R, V, ctx.ignore, H.stream_ptr()), "cvcl_token_ce_bwd")
        return d_logits, None, None


class LmLossSummaries(torch.autograd.Function):
    """(means [3], counts [3]) of the token-wise LM loss: all non-pad tokens / without <sos> / without <sos>, <eos>
    (reference multimodal_lit.py:284-300)."""

    @staticmethod
    def forward(ctx, loss, labels, pad, sos, eos):
        loss = loss.contiguous()
        R = loss.numel()
        means = torch.empty(3, dtype=_F, device=loss.device)
        counts = torch.empty(3, dtype=_F, device=loss.device)
        H.check(H.lib().cvcl_lm_loss_summaries(H.ptr(loss, _F), H.ptr(labels, torch.int64), None, H.ptr(means), H.ptr(counts), None,
                                               R, pad, sos, eos, H.stream_ptr()), "cvcl_lm_loss_summaries")
        ctx.save_for_backward(labels, counts)
        ctx.meta = (R, pad, sos, eos)
        ctx.mark_non_differentiable(counts)
        return means, counts

    @staticmethod
    def backward(ctx, d_means, _d_counts):
        labels, counts = ctx.saved_tensors
        R, pad, sos, eos = ctx.meta
        d_loss = torch.empty(R, dtype=_F, device=d_means.device)
        H.check(H.lib().cvcl_lm_loss_summaries(None, H.ptr(labels), H.ptr(d_means.contiguous(), _F), None, H.ptr(counts), H.ptr(d_loss),
                                               R, pad, sos, eos, H.stream_ptr()), "cvcl_lm_loss_summaries")
        return d_loss, None, None, None, None


class InfoNCE(torch.autograd.Function):
    """Symmetric InfoNCE + accuracies + entropies (reference multimodal/multimodal.py:801-818)."""

    @staticmethod
    def forward(ctx, logits):
        logits = logits.contiguous()
        N = logits.shape[0]
        assert logits.shape[1] == N, "the contrastive loss needs square logits"
        dev = logits.device
        scalars = torch.empty(5, dtype=_F, device=dev)
        row_lse = torch.empty(N, dtype=_F, device=dev)
        col_lse = torch.empty(N, dtype=_F, device=dev)
        nb = H.lib().cvcl_infonce_workspace_bytes(N)
        ws = _ws(nb, dev)
        H.check(H.lib().cvcl_infonce_fwd(H.ptr(logits, _F), N, H.ptr(scalars), H.ptr(row_lse), H.ptr(col_lse),
                                        H.ptr(ws), nb, H.stream_ptr()), "cvcl_infonce_fwd")
        ctx.save_for_backward(logits, row_lse, col_lse)
        loss = scalars[0].clone()
        metrics = scalars[1:].clone()
        ctx.mark_non_differentiable(metrics)
        ctx.set_materialize_grads(False)
        return loss, metrics

    @staticmethod
    def backward(ctx, d_loss, _d_metrics):
        if d_loss is None:
            return None
        logits, row_lse, col_lse = ctx.saved_tensors
        N = logits.shape[0]
        d_logits = torch.empty_like(logits)
        g = d_loss.reshape(1).to(_F).contiguous()
        H.check(H.lib().cvcl_infonce_bwd(H.ptr(logits), H.ptr(row_lse), H.ptr(col_lse), H.ptr(g), H.ptr(d_logits), N,
                                        H.stream_ptr()), "cvcl_infonce_bwd")
        return d_logits


class LinearF32(torch.autograd.Function):
    """nn.Linear on fp32 operands (fc 2048->E, multimodal/multimodal.py:192; ViT head :190; the text transformer's linears
    :553-573).  ``split`` False: the exact-fp32 MFMA GEMM (the parity mode); True (the bf16 configurations): the same operands on the
    bf16 MFMA with every element split into two bf16 parts -- ~2^-16 relative, ~4x faster (cvcl_gemm_args.f32_split)."""

    @staticmethod
    def forward(ctx, x, weight, bias, split=False):
        x = x.contiguous()
        y = H.gemm(x, weight.contiguous(), bias=bias, split=split)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.split = bool(split)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        return linear_backward(x, weight, dy.contiguous(), (need_x, need_w, need_b and ctx.has_bias), split=ctx.split) + (None,)


def linear_backward(x, weight, dy, needs, split=False):
    """(dx, dw, db) of y = x W^T + b for dy [M, N]: dW[N,K] = dY^T X and dX[M,K] = dY W as GEMMs over operands read K-major in
    place; db = the column sums of dY comes out of the dW GEMM's own operand loads (a_rowsum).  needs = (dx, dw, db) wanted."""
    M, K = x.shape
    N = weight.shape[0]
    need_x, need_w, need_b = needs
    dx = dw = db = None
    if need_b:
        db = torch.empty(N, dtype=_F, device=x.device)
    if need_w:                                       # A' = dY^T: dY is [K' = M][N] in memory; W' = X^T: X is [K' = M][K]
        dw = H.gemm(dy, x, a_trans=True, w_trans=True, a_rowsum=db, split=split)
    elif need_b:
        H.check(H.lib().cvcl_colsum_f32(H.ptr(dy), H.ptr(db), M, N, H.stream_ptr()), "cvcl_colsum_f32")
    if need_x:                                       # W' = W^T: W is [K' = N][K] in memory
        dx = H.gemm(dy, weight.contiguous(), w_trans=True, split=split)
    return dx, dw, db


def token_cross_entropy(logits, labels, ignore_index=0):
    return TokenCrossEntropy.apply(logits, labels, ignore_index)


def lm_loss_summaries(loss, labels, pad=0, sos=2, eos=3):
    return LmLossSummaries.apply(loss, labels, pad, sos, eos)


def spatial_max_logits(img_rows, txt_rows, length, neg_log_temp, Bi, HW, Bt, L):
    return SpatialMaxLogits.apply(img_rows, txt_rows, length, neg_log_temp, Bi, HW, Bt, L)


def embed_meanpool(table, tok, length, want_output=True):
    return EmbedMeanPool.apply(table, tok, length, want_output)


def l2_normalize(x, eps: float = 1e-12):
    return L2Normalize.apply(x, eps)


def sim_logits(img, txt, neg_log_temp):
    return SimLogits.apply(img, txt, neg_log_temp)


def infonce(logits):
    """-> (loss, metrics[4] = image_accuracy, text_accuracy, image_entropy, text_entropy)"""
    return InfoNCE.apply(logits)


def linear_f32(x, weight, bias=None, split=False):
    return LinearF32.apply(x, weight, bias, split)


def _embed_gather(table, tok, pos=None):
    B, L = tok.shape
    V, E = table.shape
    x = torch.empty(B * L, E, dtype=_F, device=table.device)
    H.check(H.lib().cvcl_embed_gather_pos(H.ptr(table.detach(), _F), H.ptr(tok, torch.int64), H.ptr(pos), H.ptr(x), B, L, E, V,
                                         H.stream_ptr()), "cvcl_embed_gather_pos")
    return x


def lstm_text(table, lstm, tok, length):
    """Embedding + one-layer uni-directional nn.LSTM over variable-length sequences, eval mode
    (reference multimodal/multimodal.py:513-552).  -> (h at each sequence's last step [B,H], outputs [B,Lmax,H]).
    x W_ih^T for all steps is one GEMM; each step is one recurrent GEMM (gates of the input added through the
    residual epilogue) plus the cell kernel."""
    if lstm.bidirectional or lstm.num_layers != 1:
        raise NotImplementedError("only the one-layer uni-directional LSTM text encoder is on the contrastive path")
    B, L = tok.shape
    Hd = lstm.hidden_size
    dev = table.device
    with torch.no_grad():
        x = _embed_gather(table, tok)
        bias = (lstm.bias_ih_l0 + lstm.bias_hh_l0).detach().contiguous()
        gx = H.gemm(x, lstm.weight_ih_l0.detach().contiguous(), bias=bias)             # [B*L, 4H]
        w_hh = lstm.weight_hh_l0.detach().contiguous()
        h = torch.zeros(B, Hd, dtype=_F, device=dev)                                   # init_hidden zeros (:671-688)
        c = torch.zeros(B, Hd, dtype=_F, device=dev)
        out = torch.empty(B, L, Hd, dtype=_F, device=dev)
        gates = torch.empty(B, 4 * Hd, dtype=_F, device=dev)
        lib, s = H.lib(), H.stream_ptr()
        for t in range(L):
            a = H.GemmArgs()
            a.A, a.W, a.C = H.ptr(h), H.ptr(w_hh), H.ptr(gates)
            a.M, a.N, a.K, a.lda, a.ldw, a.ldc = B, 4 * Hd, Hd, Hd, Hd, 4 * Hd
            a.R, a.ldr = gx.data_ptr() + t * 4 * Hd * 4, L * 4 * Hd                   # row b of step t inside gx
            H.check(lib.cvcl_gemm(H.F32, a, s), "cvcl_gemm")
            H.check(lib.cvcl_lstm_cell(H.ptr(gates), H.ptr(length, torch.int64), t, H.ptr(h), H.ptr(c), H.ptr(out), B, L, Hd, s),
                    "cvcl_lstm_cell")
        lmax = int(length.max())              # pad_packed_sequence trims to the longest sequence (the reference syncs here too)
    return h, out[:, :lmax]


def transformer_text(table, layer, pos_embed, tok, length):
    """Embedding (+pos) + one post-norm nn.TransformerEncoderLayer with key-padding mask + sum/len, eval mode
    (reference multimodal/multimodal.py:553-573).  -> (ret [B,E], outputs [B,L,E])."""
    B, L = tok.shape
    E = table.shape[1]
    nh = layer.self_attn.num_heads
    lib, s = H.lib(), H.stream_ptr()
    with torch.no_grad():
        pos = None if pos_embed is None else pos_embed.detach()[:L, 0].contiguous().float()
        x = _embed_gather(table, tok, pos)                                              # [B*L, E]
        sa = layer.self_attn
        qkv = H.gemm(x, sa.in_proj_weight.detach().contiguous(), bias=sa.in_proj_bias.detach().contiguous())
        att = torch.empty(B * L, E, dtype=_F, device=x.device)
        H.check(lib.cvcl_attention(H.F32, H.ptr(qkv), H.ptr(tok, torch.int64), H.ptr(att), B, L, nh, E // nh,
                                   float((E // nh) ** -0.5), s), "cvcl_attention")
        y = H.gemm(att, sa.out_proj.weight.detach().contiguous(), bias=sa.out_proj.bias.detach().contiguous(), residual=x)
        h1 = torch.empty_like(y)
        H.check(lib.cvcl_layernorm(H.F32, H.ptr(y), E, H.ptr(layer.norm1.weight.detach()), H.ptr(layer.norm1.bias.detach()),
                                   layer.norm1.eps, H.ptr(h1), 1, B * L, E, s), "cvcl_layernorm")
        f = H.gemm(h1, layer.linear1.weight.detach().contiguous(), bias=layer.linear1.bias.detach().contiguous(), act=H.ACT_RELU)
        y2 = H.gemm(f, layer.linear2.weight.detach().contiguous(), bias=layer.linear2.bias.detach().contiguous(), residual=h1)
        h2 = torch.empty_like(y2)
        H.check(lib.cvcl_layernorm(H.F32, H.ptr(y2), E, H.ptr(layer.norm2.weight.detach()), H.ptr(layer.norm2.bias.detach()),
                                   layer.norm2.eps, H.ptr(h2), 1, B * L, E, s), "cvcl_layernorm")
        ret = torch.empty(B, E, dtype=_F, device=x.device)
        H.check(lib.cvcl_seq_sum_div(H.ptr(h2), H.ptr(length, torch.int64), H.ptr(ret), B, L, E, s), "cvcl_seq_sum_div")
    return ret, h2.view(B, L, E)
