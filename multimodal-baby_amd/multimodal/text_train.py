"""Trainable text encoders: torch.autograd bridges for the transformer and LSTM text encoders on libcvcl_hip.

The reference trains these under Lightning's ``.train()`` (multimodal/multimodal.py:513-573): the one-layer
``nn.TransformerEncoderLayer`` with dropout 0.1 (attention probabilities, after out_proj, inside the FFN, after
linear2) and the LSTM behind ``LockedDropout(dropout_i)``.  Every arithmetic step below is a HIP kernel; PyTorch
carries the autograd graph and the parameter tensors.  Dropout masks come from a counter-based hash of a per-call
seed drawn from torch's CPU generator (reproducible under ``seed_everything``; not bitwise torch's Philox stream).
"""
from __future__ import annotations

import torch

from . import _hip as H
from .ops import LinearF32, _F, linear_backward


def _seed() -> int:
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


class EmbedGatherPos(torch.autograd.Function):
    """x[b*L + l] = table[tok[b,l]] (+ pos[l]);  bwd: deterministic per-row scatter (padding row 0 gets none)."""

    @staticmethod
    def forward(ctx, table, pos, tok):
        B, L = tok.shape
        V, E = table.shape
        x = torch.empty(B * L, E, dtype=_F, device=table.device)
        pos2 = None if pos is None else pos[:L].reshape(L, E).contiguous()
        H.check(H.lib().cvcl_embed_gather_pos(H.ptr(table, _F), H.ptr(tok, torch.int64), H.ptr(pos2), H.ptr(x), B, L, E, V,
                                             H.stream_ptr()), "cvcl_embed_gather_pos")
        ctx.save_for_backward(tok)
        ctx.meta = (V, E, None if pos is None else tuple(pos.shape))
        return x

    @staticmethod
    def backward(ctx, dx):
        (tok,) = ctx.saved_tensors
        V, E, pos_shape = ctx.meta
        B, L = tok.shape
        dx = dx.contiguous()
        d_table = d_pos = None
        if ctx.needs_input_grad[0]:
            d_table = torch.empty(V, E, dtype=_F, device=dx.device)
            H.check(H.lib().cvcl_embed_rows_bwd(H.ptr(dx), H.ptr(tok), H.ptr(d_table), B * L, E, V, H.stream_ptr()),
                    "cvcl_embed_rows_bwd")
        if pos_shape is not None and ctx.needs_input_grad[1]:
            d_pos = torch.zeros(pos_shape, dtype=_F, device=dx.device)
            flat = torch.empty(L * E, dtype=_F, device=dx.device)
            H.check(H.lib().cvcl_colsum_f32(H.ptr(dx), H.ptr(flat), B, L * E, H.stream_ptr()), "cvcl_colsum_f32")
            d_pos.view(pos_shape[0], E)[:L] = flat.view(L, E)
        return d_table, d_pos, None


class DropoutAdd(torch.autograd.Function):
    """y = dropout(x, p) (+ residual); the mask is a pure function of (seed, index), so backward re-derives it."""

    @staticmethod
    def forward(ctx, x, residual, p, seed, period, inner):
        x = x.contiguous()
        y = torch.empty_like(x)
        r = None if residual is None else residual.contiguous()
        H.check(H.lib().cvcl_dropout(H.ptr(x, _F), H.ptr(r), H.ptr(y), x.numel(), p, seed, period, inner, H.stream_ptr()),
                "cvcl_dropout")
        ctx.cfg = (p, seed, period, inner, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed, period, inner, has_res = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        H.check(H.lib().cvcl_dropout(H.ptr(dy), None, H.ptr(dx), dy.numel(), p, seed, period, inner, H.stream_ptr()), "cvcl_dropout")
        return dx, (dy if has_res else None), None, None, None, None


class LinearReluF32(torch.autograd.Function):
    """relu(x W^T + b) with the ReLU fused into the GEMM epilogue; backward masks dy with the saved output."""

    @staticmethod
    def forward(ctx, x, weight, bias, split=False):
        x = x.contiguous()
        y = H.gemm(x, weight.contiguous(), bias=bias, act=H.ACT_RELU, split=split)
        ctx.save_for_backward(x, weight, y)
        ctx.split = bool(split)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dz = torch.empty_like(y)
        H.check(H.lib().cvcl_relu_bwd(H.ptr(y), H.ptr(dy.contiguous()), H.ptr(dz), y.numel(), H.stream_ptr()), "cvcl_relu_bwd")
        return _linear_bwd(x, weight, dz, ctx.needs_input_grad, True, ctx.split) + (None,)


def _linear_bwd(x, weight, dy, needs, has_bias, split=False):
    return linear_backward(x, weight, dy, (needs[0], needs[1], needs[2] and has_bias), split=split)


class LayerNormF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        rows, D = x.shape
        y = torch.empty_like(x)
        H.check(H.lib().cvcl_layernorm(H.F32, H.ptr(x, _F), D, H.ptr(gamma.contiguous()), H.ptr(beta.contiguous()), eps, H.ptr(y), 1,
                                       rows, D, H.stream_ptr()), "cvcl_layernorm")
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        rows, D = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dyxh = torch.empty_like(x)
        s = H.stream_ptr()
        H.check(H.lib().cvcl_layernorm_bwd(H.ptr(x), H.ptr(gamma.contiguous()), H.ptr(dy), ctx.eps, H.ptr(dx), H.ptr(dyxh), rows, D, s),
                "cvcl_layernorm_bwd")
        dg = torch.empty(D, dtype=_F, device=x.device)
        db = torch.empty(D, dtype=_F, device=x.device)
        H.check(H.lib().cvcl_colsum_f32(H.ptr(dyxh), H.ptr(dg), rows, D, s), "cvcl_colsum_f32")
        H.check(H.lib().cvcl_colsum_f32(H.ptr(dy), H.ptr(db), rows, D, s), "cvcl_colsum_f32")
        return dx, dg, db, None


class AttentionSmall(torch.autograd.Function):
    """nn.MultiheadAttention core (key padding mask where tok == 0, dropout on the probabilities), T <= 32."""

    @staticmethod
    def forward(ctx, qkv, tok, heads, p, seed):
        B, L = tok.shape
        E = qkv.shape[1] // 3
        qkv = qkv.contiguous()
        out = torch.empty(B * L, E, dtype=_F, device=qkv.device)
        scale = float((E // heads) ** -0.5)
        H.check(H.lib().cvcl_attention_small(H.ptr(qkv, _F), H.ptr(tok, torch.int64), None, H.ptr(out), None, B, L, heads, E // heads,
                                            scale, p, seed, H.stream_ptr()), "cvcl_attention_small")
        ctx.save_for_backward(qkv, tok)
        ctx.cfg = (heads, p, seed, scale)
        return out

    @staticmethod
    def backward(ctx, d_out):
        qkv, tok = ctx.saved_tensors
        heads, p, seed, scale = ctx.cfg
        B, L = tok.shape
        E = qkv.shape[1] // 3
        d_qkv = torch.empty_like(qkv)
        H.check(H.lib().cvcl_attention_small(H.ptr(qkv), H.ptr(tok), H.ptr(d_out.contiguous()), None, H.ptr(d_qkv), B, L, heads,
                                            E // heads, scale, p, seed, H.stream_ptr()), "cvcl_attention_small")
        return d_qkv, None, None, None, None


class SeqSumDiv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, length, B, L):
        E = x.shape[1]
        ret = torch.empty(B, E, dtype=_F, device=x.device)
        H.check(H.lib().cvcl_seq_sum_div(H.ptr(x.contiguous(), _F), H.ptr(length, torch.int64), H.ptr(ret), B, L, E, H.stream_ptr()),
                "cvcl_seq_sum_div")
        ctx.save_for_backward(length)
        ctx.dims = (B, L, E)
        return ret

    @staticmethod
    def backward(ctx, d_ret):
        (length,) = ctx.saved_tensors
        B, L, E = ctx.dims
        dx = torch.empty(B * L, E, dtype=_F, device=d_ret.device)
        H.check(H.lib().cvcl_seq_sum_div_bwd(H.ptr(d_ret.contiguous()), H.ptr(length), H.ptr(dx), B, L, E, H.stream_ptr()),
                "cvcl_seq_sum_div_bwd")
        return dx, None, None, None


def transformer_text_train(table, layer, pos_embed, tok, length, training: bool, split: bool = False):
    """Differentiable embedding(+pos) -> post-norm TransformerEncoderLayer -> sum/len (multimodal.py:553-573).  ``split``: the four
    linears (and their gradients) on the bf16 MFMA with hi / lo split fp32 operands (ops.LinearF32) -- the bf16 configurations."""
    B, L = tok.shape
    if L > 32:
        raise NotImplementedError("utterances are at most 25 tokens (MAX_LEN_UTTERANCE); got L > 32")
    sa = layer.self_attn
    p_attn = float(sa.dropout) if training else 0.0
    p1 = float(layer.dropout1.p) if training else 0.0
    pf = float(layer.dropout.p) if training else 0.0
    p2 = float(layer.dropout2.p) if training else 0.0
    x = EmbedGatherPos.apply(table, pos_embed, tok)
    qkv = LinearF32.apply(x, sa.in_proj_weight, sa.in_proj_bias, split)
    att = AttentionSmall.apply(qkv, tok, sa.num_heads, p_attn, _seed())
    o = LinearF32.apply(att, sa.out_proj.weight, sa.out_proj.bias, split)
    y = DropoutAdd.apply(o, x, p1, _seed(), 0, 1)                          # x + dropout1(self_attn(x))
    h1 = LayerNormF32.apply(y, layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)
    f = LinearReluF32.apply(h1, layer.linear1.weight, layer.linear1.bias, split)
    if pf > 0:
        f = DropoutAdd.apply(f, None, pf, _seed(), 0, 1)
    g = LinearF32.apply(f, layer.linear2.weight, layer.linear2.bias, split)
    y2 = DropoutAdd.apply(g, h1, p2, _seed(), 0, 1)                        # h1 + dropout2(ffn(h1))
    h2 = LayerNormF32.apply(y2, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
    ret = SeqSumDiv.apply(h2, length, B, L)
    return ret, h2.view(B, L, -1)


class LstmCore(torch.autograd.Function):
    """One-layer uni-directional nn.LSTM over [B*L, E] inputs with per-sequence lengths; returns the hidden state at
    each sequence's last step.  Forward saves gate activations / cell states / previous hidden states; backward is
    BPTT: per step one recurrent GEMM + the cell kernel, then three big GEMMs for dW_ih, dW_hh, dX."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, length, B, L):
        Hd = w_hh.shape[1]
        dev = x.device
        lib, s = H.lib(), H.stream_ptr()
        bias = (b_ih + b_hh).contiguous()
        gx = H.gemm(x.contiguous(), w_ih.contiguous(), bias=bias)                       # [B*L, 4H]
        w_hh_c = w_hh.contiguous()
        h = torch.zeros(B, Hd, dtype=_F, device=dev)
        c = torch.zeros(B, Hd, dtype=_F, device=dev)
        gates = torch.empty(B, 4 * Hd, dtype=_F, device=dev)
        gact = torch.zeros(B * L, 4 * Hd, dtype=_F, device=dev)
        csave = torch.empty(B * L, Hd, dtype=_F, device=dev)
        hprev = torch.empty(B * L, Hd, dtype=_F, device=dev)
        out = torch.empty(B, L, Hd, dtype=_F, device=dev)
        for t in range(L):
            a = H.GemmArgs()
            a.A, a.W, a.C = H.ptr(h), H.ptr(w_hh_c), H.ptr(gates)
            a.M, a.N, a.K, a.lda, a.ldw, a.ldc = B, 4 * Hd, Hd, Hd, Hd, 4 * Hd
            a.R, a.ldr = gx.data_ptr() + t * 4 * Hd * 4, L * 4 * Hd
            H.check(lib.cvcl_gemm(H.F32, a, s), "cvcl_gemm")
            H.check(lib.cvcl_lstm_cell_train(H.ptr(gates), H.ptr(length, torch.int64), t, H.ptr(h), H.ptr(c), H.ptr(out), H.ptr(gact),
                                             H.ptr(csave), H.ptr(hprev), B, L, Hd, s), "cvcl_lstm_cell_train")
        ctx.save_for_backward(x, w_ih, w_hh_c, length, gact, csave, hprev)
        ctx.dims = (B, L, Hd)
        ctx.set_materialize_grads(False)       # the unused one of (h, out) arrives as None instead of a zero tensor
        return h, out

    @staticmethod
    def backward(ctx, dh_final, d_out):
        x, w_ih, w_hh, length, gact, csave, hprev = ctx.saved_tensors
        B, L, Hd = ctx.dims
        dev = x.device
        lib, s = H.lib(), H.stream_ptr()
        dh = dh_final.contiguous().clone() if dh_final is not None else torch.zeros(B, Hd, dtype=_F, device=dev)
        dc = torch.zeros(B, Hd, dtype=_F, device=dev)
        dG = torch.empty(B * L, 4 * Hd, dtype=_F, device=dev)
        carry = torch.empty(B, Hd, dtype=_F, device=dev)
        dh_next = torch.empty_like(dh)
        if d_out is not None:
            d_out = d_out.contiguous()
        for t in range(L - 1, -1, -1):
            if d_out is not None:             # per-step outputs feed the language-model branch: out[b,t] = h_t while running
                H.check(lib.cvcl_lstm_add_dout(H.ptr(dh), H.ptr(d_out, _F), H.ptr(length), t, B, L, Hd, s), "cvcl_lstm_add_dout")
            H.check(lib.cvcl_lstm_cell_bwd(H.ptr(gact), H.ptr(csave), H.ptr(length), t, H.ptr(dh), H.ptr(dc), H.ptr(dG), H.ptr(carry),
                                           B, L, Hd, s), "cvcl_lstm_cell_bwd")
            a = H.GemmArgs()                                                       # dh_{t-1} = dG_t . W_hh + carry
            a.A, a.W, a.C = dG.data_ptr() + t * 4 * Hd * 4, H.ptr(w_hh), H.ptr(dh_next)      # W' = W_hh^T: W_hh read K-major in place
            a.M, a.N, a.K, a.lda, a.ldw, a.ldc = B, Hd, 4 * Hd, L * 4 * Hd, Hd, Hd
            a.w_trans = 1
            a.R, a.ldr = H.ptr(carry), Hd
            H.check(lib.cvcl_gemm(H.F32, a, s), "cvcl_gemm")
            dh, dh_next = dh_next, dh
        needs = ctx.needs_input_grad
        dx = dwi = dwh = db = None
        want_b = needs[3] or needs[4]
        if want_b:
            db = torch.empty(4 * Hd, dtype=_F, device=dev)
        b_done = False
        if needs[1]:                          # dW_ih = dG^T X (+ the bias gradient = row sums of dG^T from the same operand loads)
            dwi = H.gemm(dG, x, a_trans=True, w_trans=True, a_rowsum=db if want_b else None)
            b_done = want_b
        if needs[2]:
            dwh = H.gemm(dG, hprev, a_trans=True, w_trans=True, a_rowsum=db if (want_b and not b_done) else None)
            b_done = want_b
        if want_b and not b_done:
            H.check(lib.cvcl_colsum_f32(H.ptr(dG), H.ptr(db), B * L, 4 * Hd, s), "cvcl_colsum_f32")
        if needs[0]:
            dx = H.gemm(dG, w_ih.contiguous(), w_trans=True)
        return dx, dwi, dwh, db, (db.clone() if db is not None else None), None, None, None


class SeqReverse(torch.autograd.Function):
    """y[b,t] = x[b, len-1-t] inside each sequence, zeros beyond (x: [B*L, E] rows); self-adjoint."""

    @staticmethod
    def forward(ctx, x, length, B, L):
        y = torch.empty_like(x)
        H.check(H.lib().cvcl_seq_reverse(H.ptr(x.contiguous(), _F), H.ptr(length, torch.int64), H.ptr(y), B, L, x.shape[-1],
                                         H.stream_ptr()), "cvcl_seq_reverse")
        ctx.save_for_backward(length)
        ctx.dims = (B, L)
        return y

    @staticmethod
    def backward(ctx, dy):
        (length,) = ctx.saved_tensors
        B, L = ctx.dims
        dx = torch.empty_like(dy)
        H.check(H.lib().cvcl_seq_reverse(H.ptr(dy.contiguous(), _F), H.ptr(length), H.ptr(dx), B, L, dy.shape[-1], H.stream_ptr()),
                "cvcl_seq_reverse")
        return dx, None, None, None


class Mean2(torch.autograd.Function):
    """0.5 * (a + b): mean of the forward and backward LSTM directions (reference :540-547, :552)."""

    @staticmethod
    def forward(ctx, a, b):
        y = torch.empty_like(a)
        H.check(H.lib().cvcl_scale_add_f32(H.ptr(a.contiguous(), _F), H.ptr(b.contiguous(), _F), 0.5, H.ptr(y), a.numel(), H.stream_ptr()),
                "cvcl_scale_add_f32")
        return y

    @staticmethod
    def backward(ctx, dy):
        g = torch.empty_like(dy)
        H.check(H.lib().cvcl_scale_add_f32(H.ptr(dy.contiguous(), _F), None, 0.5, H.ptr(g), dy.numel(), H.stream_ptr()),
                "cvcl_scale_add_f32")
        return g, g


class Cbow(torch.autograd.Function):
    """continuous bag of words over [B, L, E] (reference :505-511); the window operator is symmetric."""

    @staticmethod
    def forward(ctx, x, B, L, crange):
        y = torch.empty_like(x)
        H.check(H.lib().cvcl_cbow(H.ptr(x.contiguous(), _F), H.ptr(y), B, L, x.shape[-1], crange, H.stream_ptr()), "cvcl_cbow")
        ctx.dims = (B, L, crange)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, L, crange = ctx.dims
        dx = torch.empty_like(dy)
        H.check(H.lib().cvcl_cbow(H.ptr(dy.contiguous(), _F), H.ptr(dx), B, L, dy.shape[-1], crange, H.stream_ptr()), "cvcl_cbow")
        return dx, None, None, None


def cbow_text_train(table, tok, crange: int):
    B, L = tok.shape
    x = EmbedGatherPos.apply(table, None, tok)
    return Cbow.apply(x, B, L, int(crange)).view(B, L, table.shape[1])


def bilstm_text_train(table, lstm, tok, length, dropout_i: float, training: bool):
    """Differentiable embedding -> LockedDropout -> bidirectional LSTM (multimodal.py:513-552 with 'bilstm'):
    (mean of the two final hidden states [B,E], mean of the two directions' outputs [B,Lmax,E])."""
    if not lstm.bidirectional or lstm.num_layers != 1:
        raise NotImplementedError("bilstm_text_train expects a one-layer bidirectional nn.LSTM")
    B, L = tok.shape
    E = table.shape[1]
    x = EmbedGatherPos.apply(table, None, tok)
    if training and dropout_i:
        x = DropoutAdd.apply(x, None, float(dropout_i), _seed(), L, E)
    h_f, out_f = LstmCore.apply(x, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0, length, B, L)
    xr = SeqReverse.apply(x, length, B, L)
    h_b, out_r = LstmCore.apply(xr, lstm.weight_ih_l0_reverse, lstm.weight_hh_l0_reverse, lstm.bias_ih_l0_reverse,
                                lstm.bias_hh_l0_reverse, length, B, L)
    Hd = out_r.shape[-1]
    out_b = SeqReverse.apply(out_r.reshape(B * L, Hd), length, B, L).view(B, L, Hd)
    ret = Mean2.apply(h_f, h_b)
    out = Mean2.apply(out_f.contiguous(), out_b)
    return ret, out[:, :int(length.max())]


def lstm_text_train(table, lstm, tok, length, dropout_i: float, training: bool):
    """Differentiable embedding -> LockedDropout(dropout_i) -> LSTM -> last hidden state (multimodal.py:513-552)."""
    if lstm.bidirectional or lstm.num_layers != 1:
        raise NotImplementedError("only the one-layer uni-directional LSTM text encoder is on the contrastive path")
    B, L = tok.shape
    E = table.shape[1]
    x = EmbedGatherPos.apply(table, None, tok)
    if training and dropout_i:
        x = DropoutAdd.apply(x, None, float(dropout_i), _seed(), L, E)        # mask [B,1,E] shared over time (:46-53)
    h, out = LstmCore.apply(x, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0, length, B, L)
    return h, out[:, :int(length.max())]       # pad_packed_sequence trims to the longest sequence (reference syncs too)
