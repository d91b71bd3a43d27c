"""ViT forward on libcvcl_hip (reference multimodal/vision_transformer_dino_mugs.py:232-250).

prepare_tokens -> depth x [LN, qkv GEMM(+bias), attention, proj GEMM(+bias, +residual), LN, fc1 GEMM(+bias, GELU),
fc2 GEMM(+bias, +residual)] -> LN of the cls rows -> [B, D] fp32.  Weights are cast once per weight version into
the compute dtype (bf16 perf mode / fp32 parity mode).  Inference-only (the DINO ViT is frozen in every CVCL
configuration; ``--finetune_cnn`` with a ViT raises)."""
from __future__ import annotations

import os

import torch

from . import _hip as H


def _packed(model, dt, device):
    key = (str(dt), str(device), bool(getattr(model, "fp8_linears", False))) + tuple((p.data_ptr(), p._version) for p in model.parameters())
    hit = model._cache.get("w")
    if hit is not None and hit[0] == key:
        return hit[1]
    D, p = model.embed_dim, model.patch_size
    K = 3 * p * p
    Kpad = (K + 7) // 8 * 8
    w = {}
    pe = model.patch_embed.proj.weight.detach().reshape(D, K)
    wp = torch.zeros(D, Kpad, dtype=torch.float32, device=device)
    wp[:, :K] = pe
    w["pe_w"], w["Kpad"] = wp.to(dt).contiguous(), Kpad
    w["pe_b"] = model.patch_embed.proj.bias.detach().float().contiguous()
    w["cls"] = model.cls_token.detach().reshape(-1).float().contiguous()
    w["pos"] = model.pos_embed.detach().reshape(-1, D).float().contiguous()
    w["blocks"] = []
    for blk in model.blocks:
        w["blocks"].append({
            "n1w": blk.norm1.weight.detach().float().contiguous(), "n1b": blk.norm1.bias.detach().float().contiguous(),
            "qkv_w": blk.attn.qkv.weight.detach().to(dt).contiguous(),
            "qkv_b": None if blk.attn.qkv.bias is None else blk.attn.qkv.bias.detach().float().contiguous(),
            "proj_w": blk.attn.proj.weight.detach().to(dt).contiguous(), "proj_b": blk.attn.proj.bias.detach().float().contiguous(),
            "n2w": blk.norm2.weight.detach().float().contiguous(), "n2b": blk.norm2.bias.detach().float().contiguous(),
            "fc1_w": blk.mlp.fc1.weight.detach().to(dt).contiguous(), "fc1_b": blk.mlp.fc1.bias.detach().float().contiguous(),
            "fc2_w": blk.mlp.fc2.weight.detach().to(dt).contiguous(), "fc2_b": blk.mlp.fc2.bias.detach().float().contiguous(),
            "eps": blk.norm1.eps, "scale": float(blk.attn.scale), "heads": blk.attn.num_heads})
        if dt == torch.bfloat16:
            # LayerNorm folded into qkv / fc1 (gemm8w LNF, cvcl_hip.h): W' = W diag(gamma) rounded to bf16, s = row sums of THAT
            # matrix (what the MFMA multiplies, so rstd (x W'^T - mean s) is exact algebra), b' = b + W beta in fp64
            bw = w["blocks"][-1]
            for name, lin, norm in (("qkv", blk.attn.qkv, blk.norm1), ("fc1", blk.mlp.fc1, blk.norm2)):
                Wf = lin.weight.detach().double()
                g, be = norm.weight.detach().double(), norm.bias.detach().double()
                Wl = (Wf * g[None, :]).float().to(torch.bfloat16).contiguous()
                b0 = lin.bias.detach().double() if lin.bias is not None else torch.zeros(Wf.shape[0], dtype=torch.float64, device=Wf.device)
                bw[name + "_w_ln"] = Wl
                bw[name + "_s_ln"] = Wl.double().sum(1).float().contiguous()
                bw[name + "_b_ln"] = (b0 + Wf @ be).float().contiguous()
    w["nw"], w["nb"], w["neps"] = model.norm.weight.detach().float().contiguous(), model.norm.bias.detach().float().contiguous(), model.norm.eps
    if getattr(model, "fp8_linears", False):
        # BASELINE configs[4]: e4m3 weights with one scale per output channel (static), quantised once per weight version
        lib = H.lib()
        for blk, bw in zip(model.blocks, w["blocks"]):
            for name, lin in (("qkv", blk.attn.qkv), ("proj", blk.attn.proj), ("fc1", blk.mlp.fc1), ("fc2", blk.mlp.fc2)):
                wf = lin.weight.detach().float().contiguous()
                N, K = wf.shape
                q = torch.empty(N, K, dtype=torch.uint8, device=device)
                sc = torch.empty(N, dtype=torch.float32, device=device)
                H.check(lib.cvcl_quant_rows_fp8(H.F32, H.ptr(wf), K, None, None, 0.0, H.ptr(q), H.ptr(sc), N, K, H.stream_ptr()),
                        "cvcl_quant_rows_fp8")
                bw[name + "_q"], bw[name + "_s"] = q, sc
            # LayerNorm folded into qkv / fc1 (round 5; cvcl_gemm_fp8_ex): W' = e4m3(W diag(gamma)) with its own row scales, s = the row
            # sums of the DEQUANTISED matrix (what the MFMA multiplies: rstd (x W'^T - mean s) stays exact algebra), b' = b + W beta
            for name, lin, norm in (("qkv", blk.attn.qkv, blk.norm1), ("fc1", blk.mlp.fc1, blk.norm2)):
                Wf = lin.weight.detach().double()
                g, be = norm.weight.detach().double(), norm.bias.detach().double()
                wl = (Wf * g[None, :]).float().contiguous()
                N, K = wl.shape
                q = torch.empty(N, K, dtype=torch.uint8, device=device)
                sc = torch.empty(N, dtype=torch.float32, device=device)
                H.check(lib.cvcl_quant_rows_fp8(H.F32, H.ptr(wl), K, None, None, 0.0, H.ptr(q), H.ptr(sc), N, K, H.stream_ptr()),
                        "cvcl_quant_rows_fp8")
                b0 = lin.bias.detach().double() if lin.bias is not None else torch.zeros(N, dtype=torch.float64, device=Wf.device)
                deq = q.view(torch.float8_e4m3fn).double().sum(1) * sc.double()
                bw[name + "_q_ln"], bw[name + "_sw_ln"] = q, sc
                bw[name + "_cs_ln"] = deq.float().contiguous()
                bw[name + "_b8_ln"] = (b0 + Wf @ be).float().contiguous()
    if torch.device(device).type == "cuda":
        torch.cuda.current_stream(device).synchronize()      # packed once, then read by every stream that runs the trunk
    model._cache["w"] = (key, w)
    return w


def _quant(x, rows, K, q, sc, ln=None):
    """bf16 rows -> e4m3 rows + per-row scales, optionally through nn.LayerNorm first (ln = (gamma, beta, eps))."""
    g, b, eps = ln if ln is not None else (None, None, 0.0)
    H.check(H.lib().cvcl_quant_rows_fp8(H.BF16, H.ptr(x), K, H.ptr(g), H.ptr(b), eps, H.ptr(q), H.ptr(sc), rows, K, H.stream_ptr()),
            "cvcl_quant_rows_fp8")


def _gemm8(q, sc, wq, ws, out, bias, act=H.ACT_NONE, residual=None):
    M, K = q.shape
    N = wq.shape[0]
    H.check(H.lib().cvcl_gemm_fp8(H.ptr(q), H.ptr(sc), K, H.ptr(wq), H.ptr(ws), K, H.ptr(out), N, H.ptr(bias), act, H.ptr(residual), N,
                                  M, N, K, H.stream_ptr()), "cvcl_gemm_fp8")


def _gemm8_mx(q, sc, bs, wq, ws, out, out8, out_bs, bias, act=H.ACT_NONE, residual=None):
    """fp8 GEMM with MX (e8m0 per 32 elements) block scales on the input (bs) and/or the output (out8, out_bs)."""
    M, K = q.shape
    N = wq.shape[0]
    H.check(H.lib().cvcl_gemm_fp8_mx(H.ptr(q), H.ptr(sc), H.ptr(bs), K, H.ptr(wq), H.ptr(ws), K, H.ptr(out), N, H.ptr(out8), H.ptr(out_bs), N,
                                     H.ptr(bias), act, H.ptr(residual), N, M, N, K, H.stream_ptr()), "cvcl_gemm_fp8_mx")


def _gemm8_ex(q, bs, wq, ws, bias, *, out=None, out8=None, out_bs=None, act=H.ACT_NONE, residual=None, ln_stats=None, ln_colsum=None,
              row_part=None):
    """cvcl_gemm_fp8_ex with MX input: the LayerNorm-folded consumer (ln_stats, ln_colsum) / producer (row_part: bf16 ``out`` + residual
    AND the MX copy ``out8`` / ``out_bs`` of the stored rows AND their strip sums)."""
    import ctypes as C
    M, K = q.shape
    N = wq.shape[0]
    a = H.GemmFp8Args()
    a.A8, a.a_scale, a.a_block_scales, a.lda = H.ptr(q), None, H.ptr(bs), K
    a.W8, a.w_scale, a.ldw = H.ptr(wq), H.ptr(ws), K
    a.C, a.ldc, a.c8, a.c_block_scales, a.ldc8 = H.ptr(out), N, H.ptr(out8), H.ptr(out_bs), N
    a.bias, a.act, a.R, a.ldr = H.ptr(bias), act, H.ptr(residual), N
    a.M, a.N, a.K = M, N, K
    a.ln_stats, a.ln_colsum, a.row_part = H.ptr(ln_stats), H.ptr(ln_colsum), H.ptr(row_part)
    H.check(H.lib().cvcl_gemm_fp8_ex(C.byref(a), H.stream_ptr()), "cvcl_gemm_fp8_ex")


def ln_fold_mode(model):
    """None = automatic (fold when every block GEMM runs on the 8-wave kernel), True / False = forced (tests, A/B)."""
    m = model.__dict__.get("ln_fold")
    if m is None and os.environ.get("CVCL_LN_FOLD") in ("0", "1"):
        m = os.environ["CVCL_LN_FOLD"] == "1"
    return m


def _ln(cd, x, stride, g, b, eps, out, out_f32, rows, D):
    H.check(H.lib().cvcl_layernorm(cd, H.ptr(x), stride, H.ptr(g), H.ptr(b), eps, H.ptr(out), int(out_f32), rows, D,
                                   H.stream_ptr()), "cvcl_layernorm")


def vit_forward(model, x: torch.Tensor) -> torch.Tensor:
    if torch.is_grad_enabled() and any(p.requires_grad for n, p in model.named_parameters() if not n.startswith("head.")):
        from .vit_train import vit_trunk_train         # --finetune_cnn: differentiable twin (saves activations)
        return vit_trunk_train(model, x)
    ts = model.__dict__.get("_trunk_stream")
    cb = model.__dict__.get("_pre_head_callback")        # parallel.OverlappedUpdate: the previous step's all-reduce wait + optimizer
    if ts is not None and x.is_cuda:          # frozen ViT on its own stream: overlaps the previous step's text encoder / loss /
        x = x.contiguous()                    # backward / optimizer, which stay on the caller's stream (H.TrunkStream)
        handle = ts.launch(lambda slot: _vit_forward(model, x, slot), x)
        if cb is not None:
            cb()                              # ... enqueued on the caller's stream while the trunk runs on its own
        return ts.wait(handle)
    out = _vit_forward(model, x, None)
    if cb is not None:
        cb()
    return out


def enable_trunk_stream(model, device, inputs="caller", stream=None, n_streams=None):
    """n_streams = 2 ($CVCL_VIT_TRUNK_STREAMS): consecutive passes of the frozen ViT alternate between two streams and overlap each
    other (the forward keeps no state between passes: per-pass activations come from the stream's own allocator pool)."""
    if n_streams is None:                     # measured at B = 256 with two: fp8 linears 10.03 -> 9.33 ms/step, bf16 14.47 -> 14.28
        n_streams = 1 if stream is not None else int(os.environ.get("CVCL_VIT_TRUNK_STREAMS", "2"))
    model.__dict__["_trunk_stream"] = H.TrunkStream(device, inputs, stream, n_streams) if inputs else None
    return model.__dict__["_trunk_stream"]


def _vit_forward(model, x: torch.Tensor, slot) -> torch.Tensor:
    # two trunk passes in flight on two streams: each pass's 8-wave GEMMs fill HALF the chip so that the passes run side by side
    # instead of taking turns at whole-chip launches (cvcl_set_gemm_cu_share).  Measured at B = 256, same box, A/B: ViT-B/16 bf16
    # 11.98 -> 11.75 ms per step; NOT for the e4m3 linears (8.4 -> 9.7 ms) and not at patch 14 (65 792 token rows: 15.86 -> 16.41 ms,
    # two passes' activations no longer share the Infinity Cache) -- hence the policy below; $CVCL_VIT_CU_SHARE=0 / 1 forces it
    ts = model.__dict__.get("_trunk_stream")
    share = 0
    # (only while the OTHER stream's pass is really in flight: a validation loop with a host sync per batch, or a host-bound step,
    # runs one pass at a time, and half-chip grids would then leave half the CUs idle)
    if ts is not None and ts.n_streams == 2 and x.is_cuda and (ts.other_pass_in_flight() or os.environ.get("CVCL_VIT_CU_SHARE") == "1"):
        rows = x.shape[0] * ((x.shape[2] // model.patch_size) * (x.shape[3] // model.patch_size) + 1)
        auto = model.compute_dtype == torch.bfloat16 and not getattr(model, "fp8_linears", False) and rows <= 56 * 1024
        force = os.environ.get("CVCL_VIT_CU_SHARE")
        if (force == "1") or (force != "0" and auto):
            share = torch.cuda.get_device_properties(x.device).multi_processor_count // 2
    prev = H.lib().cvcl_set_gemm_cu_share(share)
    try:
        return _vit_forward_impl(model, x, slot)
    finally:
        H.lib().cvcl_set_gemm_cu_share(prev)


def _vit_forward_impl(model, x: torch.Tensor, slot) -> torch.Tensor:
    if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
        raise H.CvclError(f"expected NCHW fp32 images, got {tuple(x.shape)} {x.dtype}")
    x = x.contiguous()
    B, _, Hh, Ww = x.shape
    p, D = model.patch_size, model.embed_dim
    dt = model.compute_dtype
    cd = H.cvcl_dtype(dt)
    lib, s = H.lib(), H.stream_ptr()
    n_p = (Hh // p) * (Ww // p)
    T = n_p + 1
    with torch.no_grad():
        w = _packed(model, dt, x.device)
        pos = w["pos"]
        if T != model.pos_embed.shape[1] or Hh != Ww:           # non-native resolution: resampled table (reference :210-230), cached
            key = ("pos", Hh, Ww, model.pos_embed.data_ptr(), model.pos_embed._version)
            hit = model._cache.get("pos_interp")
            if hit is None or hit[0] != key:
                probe = torch.empty(1, T, 1, device="meta")
                hit = (key, model.interpolate_pos_encoding(probe, Hh, Ww).detach().reshape(-1, D).float().contiguous())
                model._cache["pos_interp"] = hit
            pos = hit[1]
        dev = x.device
        cols = torch.empty(B * n_p, w["Kpad"], dtype=dt, device=dev)
        H.check(lib.cvcl_im2col_patches(cd, H.ptr(x), H.ptr(cols), B, Hh, Ww, p, w["Kpad"], s), "cvcl_im2col_patches")
        tok = H.gemm(cols, w["pe_w"], bias=w["pe_b"])
        h = torch.empty(B * T, D, dtype=dt, device=dev)
        H.check(lib.cvcl_vit_assemble_tokens(cd, H.ptr(tok), H.ptr(w["cls"]), H.ptr(pos), H.ptr(h), B, T, D, s),
                "cvcl_vit_assemble_tokens")
        y = torch.empty_like(h)
        att = torch.empty_like(h)
        qkv = torch.empty(B * T, 3 * D, dtype=dt, device=dev)
        mid = torch.empty(B * T, w["blocks"][0]["fc1_w"].shape[0], dtype=dt, device=dev) if w["blocks"] else None
        fp8 = bool(getattr(model, "fp8_linears", False)) and dt == torch.bfloat16 and D % 128 == 0
        if fp8:
            # fp8 linears (BASELINE configs[4]): every GEMM operand is e4m3 with a per-token scale -- norm1 / norm2 are fused
            # with the quantisation; the attention kernel and fc1's (GELU) epilogue emit e4m3 with MX block scales (one e8m0
            # per 32 elements) that the scaled MFMAs of proj / fc2 consume directly -- no quantisation pass in between;
            # the residual stream, the attention maths and the statistics stay bf16 / fp32
            Dm = mid.shape[1]
            bw0 = w["blocks"][0]
            q_d = torch.empty(B * T, D, dtype=torch.uint8, device=dev)
            q_m = torch.empty(B * T, Dm, dtype=torch.uint8, device=dev)
            # (the 8-wave kernel fetches scales in 16-byte granules: a little slack behind each array, cvcl_hip.h)
            sc = torch.empty(B * T + 4, dtype=torch.float32, device=dev)[:B * T]
            bs_m = torch.empty(Dm // 128 * B * T * 4 + 16, dtype=torch.uint8, device=dev)[:Dm // 128 * B * T * 4].view(Dm // 128, B * T, 4)
            bs_d = torch.empty(D // 128 * B * T * 4 + 16, dtype=torch.uint8, device=dev)[:D // 128 * B * T * 4].view(D // 128, B * T, 4)
            mx_att = D // bw0["heads"] == 64 and bw0["heads"] % 2 == 0 and T > 32
            # LayerNorm folded into the e4m3 qkv / fc1 (round 5): the proj / fc2 epilogues leave the MX-quantised raw residual rows
            # and their strip sums; no LayerNorm + row-quantise pass between the linears (24 of them in a ViT-B)
            lib8 = H.lib()
            # OPT-IN (model.ln_fold = True / $CVCL_LN_FOLD=1): measured on one box (profiles/r05_ab_c5_fold.txt) the folded step is
            # 9.00 ms against 8.75 -- per block the two quantise passes it removes (2 x 28 us) are paid back by the MX-input kinds
            # of qkv / fc1 (+12 / +13 us) and the producers' second store stream (+18 us each), and the passes were hidden behind
            # the other trunk stream's GEMMs anyway.
            fold8 = (mx_att and ln_fold_mode(model) is True and D % 128 == 0 and D <= 1024 and
                     bool(lib8.cvcl_gemm_fp8_ln_supported(B * T, 3 * D, D)) and bool(lib8.cvcl_gemm_fp8_ln_supported(B * T, Dm, D)))
            if fold8:
                M8 = B * T
                q_x = torch.empty(M8, D, dtype=torch.uint8, device=dev)          # the residual rows, MX e4m3
                bs_x = torch.empty(D // 128 * M8 * 4 + 16, dtype=torch.uint8, device=dev)[:D // 128 * M8 * 4].view(D // 128, M8, 4)
                st8 = torch.empty(M8 + 1, 2, dtype=torch.float32, device=dev)[:M8]
                part8 = torch.empty(M8, D // 64, 2, dtype=torch.float32, device=dev)
                blocks = w["blocks"]
                H.check(lib.cvcl_quant_rows_mx(H.ptr(h), D, H.ptr(q_x), H.ptr(bs_x), M8, D, s), "cvcl_quant_rows_mx")
                H.check(lib.cvcl_row_stats(cd, H.ptr(h), D, H.ptr(st8), M8, D, blocks[0]["eps"], s), "cvcl_row_stats")
                for i, bw in enumerate(blocks):
                    _gemm8_ex(q_x, bs_x, bw["qkv_q_ln"], bw["qkv_sw_ln"], bw["qkv_b8_ln"], out=qkv, ln_stats=st8, ln_colsum=bw["qkv_cs_ln"])
                    H.check(lib.cvcl_attention_mx(H.ptr(qkv), H.ptr(q_d), H.ptr(bs_d), B, T, bw["heads"], 64, bw["scale"], s), "cvcl_attention_mx")
                    _gemm8_ex(q_d, bs_d, bw["proj_q"], bw["proj_s"], bw["proj_b"], out=h, residual=h, out8=q_x, out_bs=bs_x, row_part=part8)
                    H.check(lib.cvcl_row_stats_finalize(H.ptr(part8), D // 64, H.ptr(st8), M8, D, bw["eps"], s), "cvcl_row_stats_finalize")
                    _gemm8_ex(q_x, bs_x, bw["fc1_q_ln"], bw["fc1_sw_ln"], bw["fc1_b8_ln"], out8=q_m, out_bs=bs_m, act=H.ACT_GELU,
                              ln_stats=st8, ln_colsum=bw["fc1_cs_ln"])
                    if i + 1 < len(blocks):
                        _gemm8_ex(q_m, bs_m, bw["fc2_q"], bw["fc2_s"], bw["fc2_b"], out=h, residual=h, out8=q_x, out_bs=bs_x, row_part=part8)
                        H.check(lib.cvcl_row_stats_finalize(H.ptr(part8), D // 64, H.ptr(st8), M8, D, blocks[i + 1]["eps"], s),
                                "cvcl_row_stats_finalize")
                    else:
                        _gemm8_mx(q_m, None, bs_m, bw["fc2_q"], bw["fc2_s"], h, None, None, bw["fc2_b"], residual=h)
            for bw in (w["blocks"] if not fold8 else ()):
                _quant(h, B * T, D, q_d, sc, (bw["n1w"], bw["n1b"], bw["eps"]))
                _gemm8(q_d, sc, bw["qkv_q"], bw["qkv_s"], qkv, bw["qkv_b"])
                if mx_att:
                    H.check(lib.cvcl_attention_mx(H.ptr(qkv), H.ptr(q_d), H.ptr(bs_d), B, T, bw["heads"], 64, bw["scale"], s), "cvcl_attention_mx")
                    _gemm8_mx(q_d, None, bs_d, bw["proj_q"], bw["proj_s"], h, None, None, bw["proj_b"], residual=h)
                else:
                    H.check(lib.cvcl_attention(cd, H.ptr(qkv), None, H.ptr(att), B, T, bw["heads"], D // bw["heads"], bw["scale"], s),
                            "cvcl_attention")
                    _quant(att, B * T, D, q_d, sc)
                    _gemm8(q_d, sc, bw["proj_q"], bw["proj_s"], h, bw["proj_b"], residual=h)
                _quant(h, B * T, D, q_d, sc, (bw["n2w"], bw["n2b"], bw["eps"]))
                _gemm8_mx(q_d, sc, None, bw["fc1_q"], bw["fc1_s"], None, q_m, bs_m, bw["fc1_b"], act=H.ACT_GELU)
                _gemm8_mx(q_m, None, bs_m, bw["fc2_q"], bw["fc2_s"], h, None, None, bw["fc2_b"], residual=h)
        # bf16: nn.LayerNorm folded into the linear it feeds (reference :136-149).  qkv / fc1 multiply the RAW residual rows by
        # W diag(gamma) and apply (rstd, -mean rstd) per row and the column sums in their epilogue; proj / fc2 leave the row sums of
        # what they store (strip partials -> cvcl_row_stats_finalize): no normalised copy of the token matrix is written or read
        # (24 LayerNorm passes of a ViT-B gone), and the rows are rounded to bf16 once less.  Used when the dispatcher runs the
        # block's GEMMs on the 8-wave kernel (large B T); otherwise the LayerNorm kernel + plain GEMM below.
        M = B * T
        fold = False
        if not fp8 and dt == torch.bfloat16 and w["blocks"] and ln_fold_mode(model) is not False:
            bw0 = w["blocks"][0]
            st = torch.empty(M + 1, 2, dtype=torch.float32, device=dev)[:M]       # (16-byte granules: an even number of rows readable)
            part = torch.empty(M, D // 64, 2, dtype=torch.float32, device=dev) if D % 64 == 0 else None
            ok_c = part is not None and all(H.gemm(h, bw0[n + "_w_ln"], out=o, bias=bw0[n + "_b_ln"], ln_stats=st, ln_colsum=bw0[n + "_s_ln"],
                                                   act=a, query_ln=True) for n, o, a in (("qkv", qkv, H.ACT_NONE), ("fc1", mid, H.ACT_GELU)))
            ok_p = ok_c and all(H.gemm(i, bw0[n + "_w"], out=h, bias=bw0[n + "_b"], residual=h, row_part=part, query_ln=True)
                                for n, i in (("proj", att), ("fc2", mid)))
            fold = ok_c and (ok_p or ln_fold_mode(model) is True)
            if ln_fold_mode(model) is True and not ok_c:
                raise H.CvclError(f"ln_fold forced, but the qkv / fc1 GEMMs of this shape (M {M}, D {D}) do not run on the 8-wave kernel")
        if fold:
            def stats_of_h(eps):
                H.check(lib.cvcl_row_stats(cd, H.ptr(h), D, H.ptr(st), M, D, eps, s), "cvcl_row_stats")

            def finalize(eps):
                H.check(lib.cvcl_row_stats_finalize(H.ptr(part), D // 64, H.ptr(st), M, D, eps, s), "cvcl_row_stats_finalize")
            blocks = w["blocks"]
            stats_of_h(blocks[0]["eps"])                                  # norm1 of block 0 (the assembled tokens)
            for i, bw in enumerate(blocks):
                H.gemm(h, bw["qkv_w_ln"], out=qkv, bias=bw["qkv_b_ln"], ln_stats=st, ln_colsum=bw["qkv_s_ln"])
                H.check(lib.cvcl_attention(cd, H.ptr(qkv), None, H.ptr(att), B, T, bw["heads"], D // bw["heads"], bw["scale"], s),
                        "cvcl_attention")
                if ok_p:
                    H.gemm(att, bw["proj_w"], out=h, bias=bw["proj_b"], residual=h, row_part=part)      # h = h + proj(att)  (vit:146)
                    finalize(bw["eps"])
                else:
                    H.gemm(att, bw["proj_w"], out=h, bias=bw["proj_b"], residual=h)
                    stats_of_h(bw["eps"])
                H.gemm(h, bw["fc1_w_ln"], out=mid, bias=bw["fc1_b_ln"], act=H.ACT_GELU, ln_stats=st, ln_colsum=bw["fc1_s_ln"])
                last = i + 1 == len(blocks)
                if ok_p and not last:
                    H.gemm(mid, bw["fc2_w"], out=h, bias=bw["fc2_b"], residual=h, row_part=part)        # h = h + mlp(...)    (vit:147)
                    finalize(blocks[i + 1]["eps"])
                else:
                    H.gemm(mid, bw["fc2_w"], out=h, bias=bw["fc2_b"], residual=h)
                    if not last:
                        stats_of_h(blocks[i + 1]["eps"])
        for bw in (w["blocks"] if not (fp8 or fold) else ()):
            _ln(cd, h, D, bw["n1w"], bw["n1b"], bw["eps"], y, False, B * T, D)
            H.gemm(y, bw["qkv_w"], out=qkv, bias=bw["qkv_b"])
            H.check(lib.cvcl_attention(cd, H.ptr(qkv), None, H.ptr(att), B, T, bw["heads"], D // bw["heads"], bw["scale"], s),
                    "cvcl_attention")
            H.gemm(att, bw["proj_w"], out=h, bias=bw["proj_b"], residual=h)          # h = h + proj(att)   (vit:146)
            _ln(cd, h, D, bw["n2w"], bw["n2b"], bw["eps"], y, False, B * T, D)
            H.gemm(y, bw["fc1_w"], out=mid, bias=bw["fc1_b"], act=H.ACT_GELU)
            H.gemm(mid, bw["fc2_w"], out=h, bias=bw["fc2_b"], residual=h)            # h = h + mlp(...)     (vit:147)
        if slot is None:
            cls = torch.empty(B, D, dtype=torch.float32, device=dev)
        else:                                 # side-stream mode: a ring of persistent outputs (see H.TrunkStream.launch)
            ring = model.__dict__.setdefault("_trunk_out", {})
            key = (slot, B, D, str(dev))
            if key not in ring:
                ring[key] = torch.empty(B, D, dtype=torch.float32, device=dev)
            cls = ring[key]
        _ln(cd, h, T * D, w["nw"], w["nb"], w["neps"], cls, True, B, D)              # norm(x)[:, 0]        (vit:249-250)
    return cls
