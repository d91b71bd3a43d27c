"""Fine-tuning path of the DINO ViT (``--finetune_cnn`` with ``--vit_dino``): the whole trunk as ONE autograd node.

Reference: vision_transformer_dino_mugs.py:87-149 (Mlp / Attention / Block) and :232-250 (prepare_tokens, forward) under
torch.autograd.  Here the forward is the frozen path's launch sequence (vit_hip.py) that additionally keeps what the backward
needs -- the residual stream entering each norm, the normalised rows, qkv, the attention output with its log-sum-exp, the MLP
pre-activation and its GELU -- and the backward walks the blocks in reverse with explicit kernels:

    linear        dX = cvcl_gemm(dY, W^T copy)        dW, db = cvcl_gemm_tn_colsum(dY, X) (one pass over dY)
    attention     cvcl_attention_bwd (MFMA, probabilities rebuilt from the saved log-sum-exp)
    LayerNorm     cvcl_layernorm_bwd_rows (+ the residual gradient that bypasses the norm, in the same pass)
    GELU          in the GEMM epilogues: fc1 stores the pre-activation next to its GELU, the fc2 data-gradient GEMM multiplies
                  by gelu'(pre-activation) (cvcl_gelu_bf16 is the standalone form, kept for tests)
    tokens        cvcl_vit_tokens_bwd (patch rows -> patch-embedding weight gradient; batch sums -> pos_embed / cls_token)

bf16 storage, fp32 accumulation and fp32 parameter gradients; every kernel is deterministic.  Needs head_dim 64 and
32 < T <= 288 tokens (ViT-S/B/L at patch 16 or 14, 224 x 224).  There is no torch fallback: without the HIP library it fails."""
import torch

from . import _hip as H
from . import vit_hip

_F = torch.float32


def _linear_wgrad(dy2d: torch.Tensor, x2d: torch.Tensor, k_keep=None):
    """-> (dW [N, k_keep], db [N]) fp32 of y = x W^T + b from one pass over dY (cvcl_gemm_tn_colsum)."""
    M, N = dy2d.shape
    K = x2d.shape[1]
    k_keep = K if k_keep is None else k_keep
    lib = H.lib()
    nb = lib.cvcl_gemm_tn_colsum_workspace_bytes(M, N, K)
    ws = torch.empty(nb, dtype=torch.uint8, device=dy2d.device)
    dw = torch.empty(N, k_keep, dtype=_F, device=dy2d.device)
    db = torch.empty(N, dtype=_F, device=dy2d.device)
    H.check(lib.cvcl_gemm_tn_colsum(H.ptr(dy2d), N, H.ptr(x2d), K, M, N, K, H.ptr(dw), k_keep, H.ptr(db), H.ptr(ws), nb, H.stream_ptr()),
            "cvcl_gemm_tn_colsum")
    return dw, db


def _transpose_bf16(w: torch.Tensor) -> torch.Tensor:
    N, K = w.shape
    out = torch.empty(K, N, dtype=w.dtype, device=w.device)
    H.check(H.lib().cvcl_transpose(H.BF16, H.ptr(w), H.ptr(out), N, K, H.stream_ptr()), "cvcl_transpose")
    return out


def _ln_bwd(x, x_stride, gamma, dy, dy_f32, dy_stride, eps, add, dx, dx_stride, rows, D):
    """-> (dgamma, dbeta) fp32 [D]; dx written in place of the buffer given."""
    lib, s = H.lib(), H.stream_ptr()
    npart = lib.cvcl_layernorm_bwd_rows_partials(rows)
    part = torch.empty(npart, 2 * D, dtype=_F, device=x.device)
    H.check(lib.cvcl_layernorm_bwd_rows(H.ptr(x), x_stride, H.ptr(gamma), H.ptr(dy), int(dy_f32), dy_stride, eps, H.ptr(add), H.ptr(dx),
                                        dx_stride, H.ptr(part), rows, D, s), "cvcl_layernorm_bwd_rows")
    out = torch.empty(2 * D, dtype=_F, device=x.device)
    H.check(lib.cvcl_colsum_f32(H.ptr(part), H.ptr(out), npart, 2 * D, s), "cvcl_colsum_f32")
    return out[:D], out[D:]


def _gelu(u, d_y=None):
    y = torch.empty_like(u)
    H.check(H.lib().cvcl_gelu_bf16(H.ptr(u), H.ptr(d_y), H.ptr(y), u.numel(), H.stream_ptr()), "cvcl_gelu_bf16")
    return y


def trunk_params(model):
    """The trunk's parameters in the order VitTrunk.backward returns their gradients (the head is applied by VisionEncoder)."""
    ps = [model.patch_embed.proj.weight, model.patch_embed.proj.bias, model.cls_token, model.pos_embed]
    for blk in model.blocks:
        ps += [blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.qkv.bias, blk.attn.proj.weight, blk.attn.proj.bias,
               blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias]
    ps += [model.norm.weight, model.norm.bias]
    return ps


class VitTrunk(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, *params):
        if model.compute_dtype != torch.bfloat16:
            raise NotImplementedError("fine-tuning the ViT runs in bf16 (--precision bf16); the fp32 parity mode covers the frozen ViT only")
        if x.dtype != _F or x.dim() != 4 or x.shape[1] != 3:
            raise H.CvclError(f"expected NCHW fp32 images, got {tuple(x.shape)} {x.dtype}")
        x = x.contiguous()
        B, _, Hh, Ww = x.shape
        p, D = model.patch_size, model.embed_dim
        dt, cd = torch.bfloat16, H.BF16
        lib, s, dev = H.lib(), H.stream_ptr(), x.device
        n_p = (Hh // p) * (Ww // p)
        T = n_p + 1
        if T != model.pos_embed.shape[1]:
            raise NotImplementedError("positional-embedding interpolation (non-native resolution) is not on the hot path")
        w = vit_hip._packed(model, dt, dev)
        heads = w["blocks"][0]["heads"] if w["blocks"] else 1
        if w["blocks"] and (D // heads != 64 or not 32 < T <= 288):
            raise NotImplementedError(f"ViT fine-tuning needs head_dim 64 and 32 < tokens <= 288 (got head_dim {D // heads}, {T} tokens)")
        R = B * T
        cols = torch.empty(B * n_p, w["Kpad"], dtype=dt, device=dev)
        H.check(lib.cvcl_im2col_patches(cd, H.ptr(x), H.ptr(cols), B, Hh, Ww, p, w["Kpad"], s), "cvcl_im2col_patches")
        tok = H.gemm(cols, w["pe_w"], bias=w["pe_b"])
        h = torch.empty(R, D, dtype=dt, device=dev)
        H.check(lib.cvcl_vit_assemble_tokens(cd, H.ptr(tok), H.ptr(w["cls"]), H.ptr(w["pos"]), H.ptr(h), B, T, D, s), "cvcl_vit_assemble_tokens")
        saved = []
        for bw in w["blocks"]:
            h_in = h
            y1 = torch.empty(R, D, dtype=dt, device=dev)
            vit_hip._ln(cd, h_in, D, bw["n1w"], bw["n1b"], bw["eps"], y1, False, R, D)
            qkv = H.gemm(y1, bw["qkv_w"], bias=bw["qkv_b"])
            att = torch.empty(R, D, dtype=dt, device=dev)
            lse = torch.empty(B, bw["heads"], T, dtype=_F, device=dev)
            H.check(lib.cvcl_attention_train(H.ptr(qkv), H.ptr(att), H.ptr(lse), B, T, bw["heads"], 64, bw["scale"], s), "cvcl_attention_train")
            h_mid = H.gemm(att, bw["proj_w"], bias=bw["proj_b"], residual=h_in)
            y2 = torch.empty(R, D, dtype=dt, device=dev)
            vit_hip._ln(cd, h_mid, D, bw["n2w"], bw["n2b"], bw["eps"], y2, False, R, D)
            u = torch.empty(R, bw["fc1_w"].shape[0], dtype=dt, device=dev)
            g = H.gemm(y2, bw["fc1_w"], bias=bw["fc1_b"], act=H.ACT_GELU, pre_out=u)      # u kept for the backward, g = gelu(u)
            h = H.gemm(g, bw["fc2_w"], bias=bw["fc2_b"], residual=h_mid)
            saved.append((h_in, y1, qkv, att, lse, h_mid, y2, u, g))
        cls = torch.empty(B, D, dtype=_F, device=dev)
        vit_hip._ln(cd, h, T * D, w["nw"], w["nb"], w["neps"], cls, True, B, D)
        ctx.model, ctx.w, ctx.saved, ctx.h_last, ctx.cols = model, w, saved, h, cols
        ctx.dims = (B, T, D, n_p, 3 * p * p)
        ctx.needs = [prm is not None and prm.requires_grad for prm in params]
        return cls

    @staticmethod
    def backward(ctx, d_cls):
        w, saved = ctx.w, ctx.saved
        B, T, D, n_p, Kpe = ctx.dims
        R = B * T
        lib, s = H.lib(), H.stream_ptr()
        dev = d_cls.device
        dt = torch.bfloat16
        d_cls = d_cls.contiguous().to(_F)
        # final norm on the cls rows only: dh is zero on every other token
        dh = torch.zeros(R, D, dtype=dt, device=dev)
        g_nw, g_nb = _ln_bwd(ctx.h_last, T * D, w["nw"], d_cls, True, D, w["neps"], None, dh, T * D, B, D)
        grads_blocks = []
        for bw, (h_in, y1, qkv, att, lse, h_mid, y2, u, g) in zip(reversed(w["blocks"]), reversed(saved)):
            # h_out = h_mid + fc2(gelu(fc1(norm2(h_mid))))
            d_u = H.gemm(dh, _transpose_bf16(bw["fc2_w"]), gelu_grad_of=u)      # [R, Dm]: (dh W2) * gelu'(u) in the epilogue
            g_fc2w, g_fc2b = _linear_wgrad(dh, g)
            d_y2 = H.gemm(d_u, _transpose_bf16(bw["fc1_w"]))                    # [R, D]
            g_fc1w, g_fc1b = _linear_wgrad(d_u, y2)
            dh_mid = torch.empty(R, D, dtype=dt, device=dev)
            g_n2w, g_n2b = _ln_bwd(h_mid, D, bw["n2w"], d_y2, False, D, bw["eps"], dh, dh_mid, D, R, D)
            # h_mid = h_in + proj(attention(qkv(norm1(h_in))))
            d_att = H.gemm(dh_mid, _transpose_bf16(bw["proj_w"]))
            g_pw, g_pb = _linear_wgrad(dh_mid, att)
            d_qkv = torch.empty(R, 3 * D, dtype=dt, device=dev)
            H.check(lib.cvcl_attention_bwd(H.ptr(qkv), H.ptr(att), H.ptr(d_att), H.ptr(lse), H.ptr(d_qkv), B, T, bw["heads"], 64, bw["scale"], s),
                    "cvcl_attention_bwd")
            d_y1 = H.gemm(d_qkv, _transpose_bf16(bw["qkv_w"]))
            g_qw, g_qb = _linear_wgrad(d_qkv, y1)
            if bw["qkv_b"] is None:
                g_qb = None
            dh_in = torch.empty(R, D, dtype=dt, device=dev)
            g_n1w, g_n1b = _ln_bwd(h_in, D, bw["n1w"], d_y1, False, D, bw["eps"], dh_mid, dh_in, D, R, D)
            dh = dh_in
            grads_blocks.append([g_n1w, g_n1b, g_qw, g_qb, g_pw, g_pb, g_n2w, g_n2b, g_fc1w, g_fc1b, g_fc2w, g_fc2b])
        grads_blocks.reverse()
        # tokens: h[b][0] = cls + pos[0], h[b][1 + p] = patch_embed(x)[b][p] + pos[1 + p]
        d_tok = torch.empty(B * n_p, D, dtype=dt, device=dev)
        d_pos = torch.empty(T, D, dtype=_F, device=dev)
        H.check(lib.cvcl_vit_tokens_bwd(H.ptr(dh), H.ptr(d_tok), H.ptr(d_pos), B, T, D, s), "cvcl_vit_tokens_bwd")
        model = ctx.model
        g_pew, g_peb = _linear_wgrad(d_tok, ctx.cols, k_keep=Kpe)
        g_pew = g_pew.reshape(model.patch_embed.proj.weight.shape)
        grads = [g_pew, g_peb, d_pos[0].reshape(model.cls_token.shape).clone(), d_pos.reshape(model.pos_embed.shape)]
        for gb in grads_blocks:
            grads += gb
        grads += [g_nw, g_nb]
        grads = [g if (need and g is not None) else None for g, need in zip(grads, ctx.needs)]
        ctx.saved = ctx.h_last = ctx.cols = None
        return (None, None, *grads)


def vit_trunk_train(model, x: torch.Tensor) -> torch.Tensor:
    """Differentiable twin of vit_hip.vit_forward: cls token after the final LayerNorm, [B, D] fp32."""
    params = trunk_params(model)
    return VitTrunk.apply(model, x, *params)
