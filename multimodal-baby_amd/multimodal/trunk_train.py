"""Fine-tuning path of the ResNeXt-50 trunk (``--finetune_cnn``): forward + backward as autograd-composed HIP kernels.

Reference behaviour: with ``finetune_cnn`` the CNN's parameters keep ``requires_grad`` (multimodal/multimodal.py:175-179),
so the contrastive loss back-propagates through torchvision's ``Bottleneck`` blocks (conv1x1 -> BN -> ReLU -> grouped
3x3 -> BN -> ReLU -> conv1x1 -> BN -> (+identity | downsample) -> ReLU) in train mode.

The frozen-CNN fast path (``ResNet.trunk`` -> ``cvcl_resnext50_fwd``) saves nothing for backward; this module is the
differentiable twin: one ``torch.autograd.Function`` per operator, each a thin wrapper over libcvcl_hip kernels, with
autograd only doing the bookkeeping (which tensors to keep, where gradients add).  Activations are NHWC, stored in the
trunk's compute dtype (fp32 parity mode / bf16); parameter gradients are returned in fp32.

  conv 1x1        fwd  cvcl_gemm                          dX  cvcl_gemm(dY, W^T)     dW  cvcl_gemm_tn(dY, X)
  grouped 3x3     fwd  cvcl_gconv3x3 (identity prologue)  dX  cvcl_gconv3x3 on (zero-stuffed) dY with the flipped,
                                                              group-transposed weight   dW  cvcl_gconv3x3_wgrad (bf16) / cvcl_conv_wgrad_direct (fp32)
  stem 7x7        fwd  cvcl_stem_conv7x7                  dW  cvcl_stem_im2col + cvcl_gemm_tn (bf16) / cvcl_conv_wgrad_direct (fp32)
  BatchNorm(+ReLU) fwd statistics from the conv epilogues -> bn_finalize (running stats) -> bn_apply;  bwd  cvcl_bn_bwd
  Bottleneck tail  fwd cvcl_bn_add_relu (BN3 + identity + ReLU in one pass);  bwd  cvcl_bn_bwd mode 2 (mask from the output)
  max / avg pool, residual add + ReLU: cvcl_maxpool3x3s2, cvcl_avgpool(_bwd), cvcl_bn_add_relu / cvcl_relu_mask

There is no PyTorch-op fallback: CPU tensors raise in ``_hip.ptr``.
"""
from __future__ import annotations

import os

import torch

from . import _hip as H

_F = torch.float32
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def _cd(t: torch.Tensor) -> int:
    return H.cvcl_dtype(t.dtype)


def _transpose(t2d: torch.Tensor) -> torch.Tensor:
    rows, cols = t2d.shape
    out = torch.empty(cols, rows, dtype=t2d.dtype, device=t2d.device)
    H.check(H.lib().cvcl_transpose(_cd(t2d), H.ptr(t2d), H.ptr(out), rows, cols, H.stream_ptr()), "cvcl_transpose")
    return out


def _pack(w: torch.Tensor, kind: int, dtype: torch.dtype) -> torch.Tensor:
    """fp32 OIHW master weight -> kernel operand layout in the compute dtype (cvcl_pack_conv_weight)."""
    cout, cing, k, _ = w.shape
    dt = H.cvcl_dtype(dtype)
    nb = H.lib().cvcl_packed_weight_bytes(dt, kind, cout, cing, k)
    buf = torch.empty(nb, dtype=torch.uint8, device=w.device)
    H.check(H.lib().cvcl_pack_conv_weight(dt, kind, H.ptr(w.detach().contiguous(), _F), H.ptr(buf), cout, cing, k,
                                          H.stream_ptr()), "cvcl_pack_conv_weight")
    return buf


def _gemm_tn(a2d: torch.Tensor, b2d: torch.Tensor, k_keep: int | None = None) -> torch.Tensor:
    """C[n][k] = sum_m a[m][n] * b[m][k] -> fp32 [N, k_keep] (cvcl_gemm_tn: weight-gradient GEMM, no transposed copies)."""
    M, N = a2d.shape
    K = b2d.shape[1]
    k_keep = K if k_keep is None else k_keep
    dt = _cd(a2d)
    lib = H.lib()
    nb = lib.cvcl_gemm_tn_workspace_bytes(dt, M, N, K)
    ws = torch.empty(nb, dtype=torch.uint8, device=a2d.device)
    out = torch.empty(N, k_keep, dtype=_F, device=a2d.device)
    H.check(lib.cvcl_gemm_tn(dt, H.ptr(a2d), N, H.ptr(b2d, a2d.dtype), K, M, N, K, H.ptr(out), k_keep, H.ptr(ws), nb, H.stream_ptr()),
            "cvcl_gemm_tn")
    return out


def _zero_stuff(dy: torch.Tensor) -> torch.Tensor:
    B, Ho, Wo, Cn = dy.shape
    z = torch.empty(B, 2 * Ho, 2 * Wo, Cn, dtype=dy.dtype, device=dy.device)
    H.check(H.lib().cvcl_zero_stuff2(_cd(dy), H.ptr(dy), H.ptr(z), B, Ho, Wo, Cn, H.stream_ptr()), "cvcl_zero_stuff2")
    return z


# ---- weight gradients on a stream of their own --------------------------------------------------------------------
# In the backward pass only the data gradients are on the critical path (dX of layer l feeds layer l-1); a weight gradient is
# needed by nobody before the optimizer step.  ``trunk_train`` therefore runs every weight-gradient kernel (TN GEMMs, the
# grouped-conv tap GEMMs, the stem's im2col + GEMM: MFMA work) on a second HIP stream, beside the data-gradient chain and the
# bandwidth-bound BatchNorm backward passes on the caller's stream.  The Function returns no gradient for the weight; the result
# is parked and, when the backward pass ends (autograd engine callback), the caller's stream waits for the side stream once and
# the gradients are stored / accumulated into ``param.grad`` -- the values autograd's own accumulation would have produced.
# $CVCL_WGRAD_STREAM=0 keeps everything on one stream.
_WGRAD_STREAMS: dict = {}
_WGRAD_PENDING: list = []
# parallel.DataParallelEngine.grad_ready when data-parallel: called with (param, dW) on the side stream right after dW is
# enqueued, so that the bucketed all-reduce of the trunk's gradients starts during the backward pass instead of after it
_WGRAD_LISTENER = None


def _wgrad_stream(device) -> torch.cuda.Stream:
    key = str(device)
    if key not in _WGRAD_STREAMS:
        _WGRAD_STREAMS[key] = torch.cuda.Stream(device=device)
    return _WGRAD_STREAMS[key]


def _wgrad_deferrable(weight) -> bool:
    return (os.environ.get("CVCL_WGRAD_STREAM", "1") != "0" and isinstance(weight, torch.nn.Parameter) and weight.is_cuda
            and weight.is_leaf and weight.requires_grad)


def _flush_wgrads():
    pending = list(_WGRAD_PENDING)
    _WGRAD_PENDING.clear()
    if not pending:
        return
    dev = pending[0][0].device
    main = torch.cuda.current_stream(dev)
    main.wait_stream(_wgrad_stream(dev))
    for param, dw in pending:
        dw.record_stream(main)                               # allocated on the side stream's pool, used here from now on
        dw = dw.view(param.shape)
        if param.grad is None:
            param.grad = dw
        else:
            param.grad.add_(dw)


def _defer_wgrad(param, compute, *reads):
    """compute() -> dW (fp32) enqueued on the weight-gradient stream; ``reads``: tensors of the caller's stream it consumes."""
    dev = param.device
    main, side = torch.cuda.current_stream(dev), _wgrad_stream(dev)
    ready = torch.cuda.Event()
    ready.record(main)
    side.wait_event(ready)
    with torch.cuda.stream(side):
        dw = compute()
        if _WGRAD_LISTENER is not None:
            _WGRAD_LISTENER(param, dw)
    for t in reads:
        t.record_stream(side)                                # their memory may be released on the caller's stream before the side stream is done
    _WGRAD_PENDING.append((param, dw))
    # one callback per deferral (the first to run flushes everything, the rest find nothing): a flag "already queued" would go
    # stale if a backward pass died half way
    torch.autograd.Variable._execution_engine.queue_callback(_flush_wgrads)


class Conv1x1(torch.autograd.Function):
    """nn.Conv2d(cin, cout, 1, stride, bias=False) on NHWC: x [B,H,W,K] -> raw [B,Ho,Wo,N]."""

    @staticmethod
    def forward(ctx, x, weight, stride: int, defer_wgrad: bool = False, centre=None):
        ctx.set_materialize_grads(False)         # the statistics output gets no gradient: None, not a zero tensor filled on the device
        B, Hh, Ww, K = x.shape
        N = weight.shape[0]
        ctx.param = weight if defer_wgrad and _wgrad_deferrable(weight) else None
        wq = _pack(weight, H.PACK_DENSE, x.dtype).view(x.dtype).view(N, K)
        Ho, Wo = (Hh - 1) // stride + 1, (Ww - 1) // stride + 1
        out = torch.empty(B, Ho, Wo, N, dtype=x.dtype, device=x.device)
        gather = (Ho, Wo, Hh, Ww, stride) if stride > 1 else None
        M = B * Ho * Wo
        st = torch.empty(H.gemm_stats_rows(_cd(x), M, N, K, gather), 2, N, dtype=_F, device=x.device)   # BN statistics from the epilogue
        H.gemm(x, wq, out=out, gather=gather, M=M, lda=K, stats=st, centre=centre)
        ctx.save_for_backward(x, wq)
        ctx.stride = stride
        ctx.mark_non_differentiable(st)
        return out, st

    @staticmethod
    def backward(ctx, dy, _dst):
        x, wq = ctx.saved_tensors
        B, Hh, Ww, K = x.shape
        N = wq.shape[0]
        dy = dy.contiguous()
        if ctx.stride > 1:                               # rows of the skipped pixels get zero gradient / contribute nothing
            dy = _zero_stuff(dy)
        M = B * Hh * Ww
        dy2, x2 = dy.view(M, N), x.view(M, K)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = H.gemm(dy2, _transpose(wq)).view(B, Hh, Ww, K)
        if ctx.needs_input_grad[1]:
            if ctx.param is not None:
                _defer_wgrad(ctx.param, lambda: _gemm_tn(dy2, x2), dy2, x2)
            else:
                dw = _gemm_tn(dy2, x2).view(N, K, 1, 1)
        return dx, dw, None, None, None


class Conv1x1Skip(torch.autograd.Function):
    """The stride-1 1x1 convolution at the head of a Bottleneck, which also hands its input on to the block's other consumer
    (the identity of the residual add, or the downsample convolution): -> (raw, stats, x).  Autograd would add the two gradients
    of the block input with an element-wise kernel of its own (one more read-read-write pass over the widest tensor of the
    block); here the gradient arriving through the second output rides the data-gradient GEMM as its residual operand:
    dX = round(round(dY W) + d_skip) -- the same two roundings as GEMM + add, one pass less."""

    @staticmethod
    def forward(ctx, x, weight, defer_wgrad: bool = False, centre=None):
        ctx.set_materialize_grads(False)         # the statistics output gets no gradient: None, not a zero tensor filled on the device
        B, Hh, Ww, K = x.shape
        N = weight.shape[0]
        ctx.param = weight if defer_wgrad and _wgrad_deferrable(weight) else None
        wq = _pack(weight, H.PACK_DENSE, x.dtype).view(x.dtype).view(N, K)
        out = torch.empty(B, Hh, Ww, N, dtype=x.dtype, device=x.device)
        M = B * Hh * Ww
        st = torch.empty(H.gemm_stats_rows(_cd(x), M, N, K), 2, N, dtype=_F, device=x.device)   # BN statistics from the epilogue
        H.gemm(x, wq, out=out, M=M, lda=K, stats=st, centre=centre)
        ctx.save_for_backward(x, wq)
        ctx.mark_non_differentiable(st)
        return out, st, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, _dst, d_skip):
        x, wq = ctx.saved_tensors
        B, Hh, Ww, K = x.shape
        N = wq.shape[0]
        M = B * Hh * Ww
        dy2, x2 = dy.contiguous().view(M, N), x.view(M, K)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            res = None if d_skip is None else d_skip.contiguous().view(M, K)
            dx = H.gemm(dy2, _transpose(wq), residual=res).view(B, Hh, Ww, K)
        if ctx.needs_input_grad[1]:
            if ctx.param is not None:
                _defer_wgrad(ctx.param, lambda: _gemm_tn(dy2, x2), dy2, x2)
            else:
                dw = _gemm_tn(dy2, x2).view(N, K, 1, 1)
        return dx, dw, None, None


class GroupedConv3x3(torch.autograd.Function):
    """nn.Conv2d(C, C, 3, stride, 1, groups=32, bias=False) on NHWC."""

    @staticmethod
    def forward(ctx, x, weight, stride: int, defer_wgrad: bool = False, centre=None):
        ctx.set_materialize_grads(False)         # the statistics output gets no gradient: None, not a zero tensor filled on the device
        B, Hh, Ww, Cn = x.shape
        ctx.param = weight if defer_wgrad and _wgrad_deferrable(weight) else None
        wp = _pack(weight, H.PACK_GCONV3, x.dtype)
        Ho, Wo = (Hh - 1) // stride + 1, (Ww - 1) // stride + 1
        out = torch.empty(B, Ho, Wo, Cn, dtype=x.dtype, device=x.device)
        srows = H.lib().cvcl_gconv3x3_stats_rows(_cd(x), B, Hh, Ww, Cn, stride)
        st = torch.empty(srows, 2, Cn, dtype=_F, device=x.device)
        H.check(H.lib().cvcl_gconv3x3(_cd(x), H.ptr(x), None, None, H.ptr(wp), H.ptr(out), H.ptr(st), srows, H.ptr(centre), B, Hh, Ww,
                                      Cn, 32, stride, H.stream_ptr()), "cvcl_gconv3x3")
        ctx.save_for_backward(x, weight)
        ctx.stride = stride
        ctx.mark_non_differentiable(st)
        return out, st

    @staticmethod
    def backward(ctx, dy, _dst):
        x, weight = ctx.saved_tensors
        B, Hh, Ww, Cn = x.shape
        cg = weight.shape[1]
        dy = dy.contiguous()
        dx = dw = None
        lib, s = H.lib(), H.stream_ptr()
        if ctx.needs_input_grad[1]:
            def wgrad():
                dw_ = torch.empty_like(weight, dtype=_F)
                s_ = H.stream_ptr()
                if x.dtype == torch.bfloat16:             # 9 tap-shifted TN GEMMs on the diagonal channel slabs (MFMA)
                    nb = lib.cvcl_gconv3x3_wgrad_workspace_bytes(B, Hh, Ww, Cn, ctx.stride)
                    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
                    H.check(lib.cvcl_gconv3x3_wgrad(H.ptr(x), H.ptr(dy), H.ptr(dw_), B, Hh, Ww, Cn, 32, ctx.stride, H.ptr(ws), nb, s_),
                            "cvcl_gconv3x3_wgrad")
                else:
                    H.check(lib.cvcl_conv_wgrad_direct(_cd(x), H.ptr(x), H.ptr(dy), H.ptr(dw_), B, Hh, Ww, Cn, Cn, cg, 3, ctx.stride, 1, 0,
                                                       s_), "cvcl_conv_wgrad_direct")
                return dw_
            if ctx.param is not None:
                _defer_wgrad(ctx.param, wgrad, x, dy)
            else:
                dw = wgrad()
        if ctx.needs_input_grad[0]:
            wf = torch.empty_like(weight, dtype=_F)
            H.check(lib.cvcl_gconv_weight_dgrad(H.ptr(weight.detach().contiguous(), _F), H.ptr(wf), Cn, cg, s), "cvcl_gconv_weight_dgrad")
            wp = _pack(wf, H.PACK_GCONV3, x.dtype)
            z = _zero_stuff(dy) if ctx.stride > 1 else dy
            dx = torch.empty_like(x)
            H.check(lib.cvcl_gconv3x3(_cd(x), H.ptr(z), None, None, H.ptr(wp), H.ptr(dx), None, 0, None, B, Hh, Ww, Cn, 32, 1, s),
                    "cvcl_gconv3x3")
        return dx, dw, None, None, None


class StemConv(torch.autograd.Function):
    """conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False) on the NCHW fp32 images -> raw NHWC."""

    @staticmethod
    def forward(ctx, x, weight, dtype, defer_wgrad: bool = False, centre=None):
        ctx.set_materialize_grads(False)         # the statistics output gets no gradient: None, not a zero tensor filled on the device
        B, _, Hh, Ww = x.shape
        ctx.param = weight if defer_wgrad and _wgrad_deferrable(weight) else None
        dt = H.cvcl_dtype(dtype)
        wp = _pack(weight, H.PACK_STEM7, dtype)
        out = torch.empty(B, Hh // 2, Ww // 2, 64, dtype=dtype, device=x.device)
        rows = H.lib().cvcl_stem_conv_stats_rows(dt, B, Hh, Ww)
        st = torch.empty(rows, 2, 64, dtype=_F, device=x.device)
        H.check(H.lib().cvcl_stem_conv7x7(dt, H.ptr(x, _F), H.ptr(wp), H.ptr(out), H.ptr(st), rows, H.ptr(centre), B, Hh, Ww,
                                          H.stream_ptr()), "cvcl_stem_conv7x7")
        ctx.save_for_backward(x)
        ctx.mark_non_differentiable(st)
        return out, st

    @staticmethod
    def backward(ctx, dy, _dst):
        (x,) = ctx.saved_tensors
        B, _, Hh, Ww = x.shape
        dy = dy.contiguous()
        def wgrad():
            if dy.dtype == torch.bfloat16:                # patch matrix (bf16, 147 -> 160 columns) + TN GEMM on MFMA
                P = B * (Hh // 2) * (Ww // 2)
                col = torch.empty(P, 160, dtype=torch.bfloat16, device=x.device)
                H.check(H.lib().cvcl_stem_im2col(H.ptr(x, _F), H.ptr(col), B, Hh, Ww, H.stream_ptr()), "cvcl_stem_im2col")
                return _gemm_tn(dy.view(P, 64), col, 147).view(64, 3, 7, 7)
            dw_ = torch.empty(64, 3, 7, 7, dtype=_F, device=x.device)
            H.check(H.lib().cvcl_conv_wgrad_direct(_cd(dy), H.ptr(x), H.ptr(dy), H.ptr(dw_), B, Hh, Ww, 3, 64, 3, 7, 2, 3, 1,
                                                   H.stream_ptr()), "cvcl_conv_wgrad_direct")
            return dw_
        if ctx.param is not None:
            _defer_wgrad(ctx.param, wgrad, x, dy)
            return None, None, None, None, None
        return None, wgrad(), None, None, None


def _bn_forward_stats(raw, stats, gamma, beta, running_mean, running_var, num_batches_tracked, centre=None):
    """statistics rows (from the producing conv's epilogue, or a cvcl_col_stats pass) -> scale/shift (+ running-stat EMA),
    batch mean and rstd -- all of the tensor AS STORED (raw = y - centre: include/cvcl_hip.h "Centred storage"); the running
    mean gets the centre added back, and ``centre`` is then moved to this batch's mean of y for the next step."""
    Cn = raw.shape[-1]
    rows = raw.numel() // Cn
    lib, s, dev = H.lib(), H.stream_ptr(), raw.device
    if stats is None:
        srows = lib.cvcl_col_stats_rows(rows)
        stats = torch.empty(srows, 2, Cn, dtype=_F, device=dev)
        H.check(lib.cvcl_col_stats(_cd(raw), H.ptr(raw), rows, Cn, H.ptr(stats), srows, s), "cvcl_col_stats")
    srows = stats.shape[0]
    vec = torch.empty(4, Cn, dtype=_F, device=dev)                 # scale, shift, mean, rstd
    scale, shift, mean, rstd = vec[0], vec[1], vec[2], vec[3]
    H.check(lib.cvcl_bn_finalize(H.ptr(stats), srows, rows, H.ptr(gamma.detach(), _F), H.ptr(beta.detach(), _F),
                                 H.ptr(running_mean, _F), H.ptr(running_var, _F), H.ptr(num_batches_tracked, torch.int64),
                                 BN_MOMENTUM, BN_EPS, H.ptr(scale), H.ptr(shift), H.ptr(centre), Cn, s), "cvcl_bn_finalize")
    H.check(lib.cvcl_bn_batch_moments(H.ptr(stats), srows, rows, BN_EPS, H.ptr(mean), H.ptr(rstd), H.ptr(centre), Cn, s),
            "cvcl_bn_batch_moments")
    return scale, shift, mean, rstd


def _bn_backward(mode, raw, out, dy, scale, shift, mean, rstd, gamma, want_g):
    Cn = raw.shape[-1]
    rows = raw.numel() // Cn
    dev, dt = raw.device, _cd(raw)
    lib = H.lib()
    prow = lib.cvcl_bn_bwd_partial_rows(dt, rows, Cn)
    scratch = torch.empty(prow * 2 + 5, Cn, dtype=_F, device=dev)  # partial rows | coef[3] | dgamma | dbeta
    partial, coef = scratch[:prow * 2], scratch[prow * 2:prow * 2 + 3]
    dgamma, dbeta = scratch[prow * 2 + 3], scratch[prow * 2 + 4]
    dx = torch.empty_like(raw)
    g = torch.empty_like(raw) if want_g else None
    H.check(lib.cvcl_bn_bwd(dt, mode, H.ptr(raw), H.ptr(out), H.ptr(dy), H.ptr(scale), H.ptr(shift), H.ptr(mean), H.ptr(rstd),
                            H.ptr(gamma.detach().contiguous(), _F), H.ptr(dgamma), H.ptr(dbeta), H.ptr(dx), H.ptr(g), rows, Cn,
                            H.ptr(partial), prow, H.ptr(coef), H.stream_ptr()), "cvcl_bn_bwd")
    return dx, dgamma, dbeta, g


class BatchNormTrain(torch.autograd.Function):
    """nn.BatchNorm2d in train mode (+ optional fused ReLU) on raw [.., C]; updates the running statistics in place.
    ``stats``: the per-channel (sum, sumsq) rows the producing convolution emitted, or None (computed here)."""

    @staticmethod
    def forward(ctx, raw, stats, gamma, beta, running_mean, running_var, num_batches_tracked, relu: bool, centre=None):
        Cn = raw.shape[-1]
        rows = raw.numel() // Cn
        scale, shift, mean, rstd = _bn_forward_stats(raw, stats, gamma, beta, running_mean, running_var, num_batches_tracked, centre)
        y = torch.empty_like(raw)
        H.check(H.lib().cvcl_bn_apply(_cd(raw), H.ptr(raw), H.ptr(scale), H.ptr(shift), H.ptr(y), rows, Cn, int(relu),
                                      H.stream_ptr()), "cvcl_bn_apply")
        ctx.save_for_backward(raw, scale, shift, mean, rstd, gamma)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        raw, scale, shift, mean, rstd, gamma = ctx.saved_tensors
        dx, dgamma, dbeta, _ = _bn_backward(1 if ctx.relu else 0, raw, None, dy.contiguous(), scale, shift, mean, rstd, gamma, False)
        return dx, None, dgamma, dbeta, None, None, None, None, None


class BnAddRelu(torch.autograd.Function):
    """Bottleneck tail: out = relu(bn3(raw) + identity) in one pass (cvcl_bn_add_relu); the backward masks with out > 0,
    hands g = dy * mask to the identity branch and runs the BatchNorm backward on g."""

    @staticmethod
    def forward(ctx, raw, stats, gamma, beta, running_mean, running_var, num_batches_tracked, identity, centre=None):
        Cn = raw.shape[-1]
        rows = raw.numel() // Cn
        scale, shift, mean, rstd = _bn_forward_stats(raw, stats, gamma, beta, running_mean, running_var, num_batches_tracked, centre)
        out = torch.empty_like(raw)
        H.check(H.lib().cvcl_bn_add_relu(_cd(raw), H.ptr(raw), H.ptr(scale), H.ptr(shift), H.ptr(identity.contiguous(), raw.dtype),
                                         None, None, H.ptr(out), rows, Cn, H.stream_ptr()), "cvcl_bn_add_relu")
        ctx.save_for_backward(raw, out, mean, rstd, gamma)
        return out

    @staticmethod
    def backward(ctx, dy):
        raw, out, mean, rstd, gamma = ctx.saved_tensors
        dx, dgamma, dbeta, g = _bn_backward(2, raw, out, dy.contiguous(), None, None, mean, rstd, gamma, True)
        return dx, None, dgamma, dbeta, None, None, None, g, None


class MaxPool3x3s2(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) on NHWC; the forward records the arg-max (one byte per output) for the backward."""

    @staticmethod
    def forward(ctx, x):
        B, Hh, Ww, Cn = x.shape
        Ho, Wo = (Hh - 1) // 2 + 1, (Ww - 1) // 2 + 1
        out = torch.empty(B, Ho, Wo, Cn, dtype=x.dtype, device=x.device)
        idx = torch.empty(B, Ho, Wo, Cn, dtype=torch.uint8, device=x.device)
        H.check(H.lib().cvcl_maxpool3x3s2_idx(_cd(x), H.ptr(x), None, H.ptr(out), H.ptr(idx), B, Hh, Ww, Cn, H.stream_ptr()),
                "cvcl_maxpool3x3s2_idx")
        ctx.save_for_backward(idx)
        ctx.in_shape, ctx.dt = (B, Hh, Ww, Cn), x.dtype
        return out

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, Hh, Ww, Cn = ctx.in_shape
        dx = torch.empty(ctx.in_shape, dtype=ctx.dt, device=dy.device)
        H.check(H.lib().cvcl_maxpool3x3s2_idx(H.cvcl_dtype(ctx.dt), None, H.ptr(dy.contiguous()), H.ptr(dx), H.ptr(idx), B, Hh, Ww, Cn,
                                              H.stream_ptr()), "cvcl_maxpool3x3s2_idx")
        return dx


class AddRelu(torch.autograd.Function):
    """out = relu(a + b): the Bottleneck tail (torchvision resnet.py: ``out += identity; out = self.relu(out)``)."""

    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty_like(a)
        n = a.numel()
        s = H.stream_ptr()
        H.check(H.lib().cvcl_add(_cd(a), H.ptr(a), H.ptr(b), H.ptr(out), n, 1, s), "cvcl_add")
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dy):
        (out,) = ctx.saved_tensors
        g = torch.empty_like(out)
        H.check(H.lib().cvcl_relu_mask(_cd(out), H.ptr(out), H.ptr(dy.contiguous()), H.ptr(g), out.numel(), H.stream_ptr()),
                "cvcl_relu_mask")
        return g, g


class AvgPool(torch.autograd.Function):
    """AdaptiveAvgPool2d((1,1)) + flatten: [B,H,W,C] -> [B,C] fp32."""

    @staticmethod
    def forward(ctx, x):
        B, Hh, Ww, Cn = x.shape
        out = torch.empty(B, Cn, dtype=_F, device=x.device)
        H.check(H.lib().cvcl_avgpool(_cd(x), H.ptr(x), H.ptr(out), B, Hh * Ww, Cn, H.stream_ptr()), "cvcl_avgpool")
        ctx.shape, ctx.dtype = tuple(x.shape), x.dtype
        return out

    @staticmethod
    def backward(ctx, d_pooled):
        B, Hh, Ww, Cn = ctx.shape
        dx = torch.empty(ctx.shape, dtype=ctx.dtype, device=d_pooled.device)
        H.check(H.lib().cvcl_avgpool_bwd(H.cvcl_dtype(ctx.dtype), H.ptr(d_pooled.contiguous(), _F), H.ptr(dx), B, Hh * Ww, Cn,
                                         H.stream_ptr()), "cvcl_avgpool_bwd")
        return dx


def _bn(raw_and_stats, bn, relu, centre=None):
    raw, st = raw_and_stats
    return BatchNormTrain.apply(raw, st, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, relu, centre)


def trunk_train(model, x: torch.Tensor):
    """Differentiable conv1 .. layer4 + avgpool of ``resnext.ResNet`` in train mode.
    -> (pooled [B,2048] fp32, layer4 map as a logical NCHW view)."""
    if not model.training:
        raise NotImplementedError("gradients through the ResNeXt trunk are implemented for train-mode BatchNorm only "
                                  "(the reference fine-tunes in train mode); eval-mode runs use the no-grad fast path")
    if x.dtype != _F or x.dim() != 4 or x.shape[1] != 3:
        raise H.CvclError(f"expected NCHW fp32 images, got {tuple(x.shape)} {x.dtype}")
    x = x.contiguous()
    cdt = model.compute_dtype
    _WGRAD_PENDING.clear()                  # leftovers of a backward pass that died half way must not reach the next one
    # centred storage (resnext.ResNet "centred storage"): layer l (torchvision state_dict order) stores round(y - cen[l]) and its
    # BatchNorm moves cen[l] to this batch's mean of y; the weights move every step, so the centres track instead of staying
    # calibrated.  The first step runs with whatever is there (zeros = plain storage).
    cen = model.tracking_centres(x.device)
    li_ = [0]

    def nxt():
        c = None if cen is None else cen[li_[0]]
        li_[0] += 1
        return c
    c0 = nxt()
    h = StemConv.apply(x, model.conv1.weight, cdt, True, None if c0 is None else c0[:64])
    h = _bn(h, model.bn1, True, None if c0 is None else c0[:64])
    h = MaxPool3x3s2.apply(h)
    for li in (1, 2, 3, 4):
        for blk in getattr(model, f"layer{li}"):
            width, outc = blk.conv1.weight.shape[0], blk.conv3.weight.shape[0]
            c1, c2, c3 = nxt(), nxt(), nxt()
            cd_ = nxt() if blk.downsample is not None else None
            cut = (lambda c, n: None if c is None else c[:n])
            raw1, st1, idn = Conv1x1Skip.apply(h, blk.conv1.weight, True, cut(c1, width))      # idn = h, its gradient folded into conv1's dX GEMM
            o = _bn((raw1, st1), blk.bn1, True, cut(c1, width))
            o = _bn(GroupedConv3x3.apply(o, blk.conv2.weight, blk.conv2.stride[0], True, cut(c2, width)), blk.bn2, True, cut(c2, width))
            if blk.downsample is not None:
                idn = _bn(Conv1x1.apply(idn, blk.downsample[0].weight, blk.downsample[0].stride[0], True, cut(cd_, outc)),
                          blk.downsample[1], False, cut(cd_, outc))
            raw3, st3 = Conv1x1.apply(o, blk.conv3.weight, 1, True, cut(c3, outc))
            b3 = blk.bn3
            h = BnAddRelu.apply(raw3, st3, b3.weight, b3.bias, b3.running_mean, b3.running_var, b3.num_batches_tracked, idn,
                                cut(c3, outc))
    pooled = AvgPool.apply(h)
    return pooled, h.permute(0, 3, 1, 2)
