"""Data-parallel sharding of the contrastive step: one process per GPU, torch.distributed over RCCL/xGMI.

The reference has no distributed code; under Lightning DDP it would compute the loss on each rank's local
B x B block and average gradients.  Two modes here:

* global negatives (default when world > 1): the L2-normalised image and text features of every rank are
  all-gathered in ONE collective ([2,B,E] fp32 per rank -> 2 x [world*B, E]); every rank evaluates the full
  symmetric InfoNCE on the replicated matrices (4.3 GFLOP at 2048 x 2048 x 512 -- cheaper than a second
  collective) and back-propagates only through its own rows (``global_sim_logits``: the two [B, N_g] x [N_g, E]
  gradient products of its shard, not the full N_g x N_g x E ones).  The full-batch gradient is then the SUM over ranks of the
  per-rank parameter gradients (no division by world); parameters whose gradient is computed identically
  on every rank from the replicated loss (the learned temperature) are flagged ``_cvcl_replicated_grad``
  and averaged instead.  Parity definition: the reference's ``calculate_contrastive_loss`` math applied to
  the concatenated feature matrices.
* local negatives (``--local_negatives``): Lightning-DDP behaviour -- local loss, gradients averaged.

Gradients are reduced in size-capped buckets launched from post-accumulate-grad hooks as soon as a bucket
is complete, so the RCCL all-reduce of early buckets runs on RCCL's stream while the rest of backward is
still executing; ``reduce_gradients`` waits and scatters the results back.  xGMI is point-to-point (7
links of ~153 GB/s per GPU): few large buckets keep the ring per-link efficient; the frozen-CNN step has
only ~9 MB of gradients, i.e. a single bucket.  With ``--finetune_cnn`` the trunk's weight gradients do not
come through autograd's accumulation (trunk_train computes them on a side stream and stores them when the
backward pass ends), so trunk_train reports each one as it is enqueued (``grad_ready``): it is copied into
its bucket on the stream that produced it and a complete bucket's all-reduce starts behind that stream --
the ~100 MB of ResNeXt gradients are reduced while the data-gradient chain is still running.

Terms of the loss that are rank-local means (the language-model cross entropy of ``lambda_lm > 0``) are
divided by the world size in global-negatives mode (``local_term_scale``), so that the SUM over ranks is
the mean over the global batch, like the replicated InfoNCE term.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


# collectives issued by this module since import: {"all_gather": n, "all_reduce": n, "broadcast": n} -- read by bench.py's multi-rank
# line (collectives per step) and by the tests (one feature all-gather per step)
COLLECTIVES = {"all_gather": 0, "all_reduce": 0, "broadcast": 0}

# Common per-word length of the spatial head's text rows under global negatives.  The collate pads every batch to ITS OWN longest
# utterance (multimodal_data_module.multiModalDataset_collate_fn; reference multimodal_data_module.py:92-110), so two ranks'
# [B, L, E] word rows generally differ in L while RCCL's all-gather wants one size on every rank.  A rank-local decision about a
# collective's size is a hang, so the size is a constant every rank knows without talking: the data module's bound on an
# utterance (MAX_LEN_UTTERANCE = 25 tokens).  A data module with longer utterances sets this before training, on every rank.
SPATIAL_TEXT_LEN = None


def is_distributed() -> bool:
    """A process group with more than one rank -- or, with $CVCL_FORCE_DIST=1, any initialised process group: a world-size-1
    ``nccl`` (= RCCL) group then drives the whole multi-GPU path (feature all-gather, bucket all-reduce from the hooks,
    ``OverlappedUpdate``, the trunk streams) on a single GPU, with results that must equal the plain single-process step bit
    for bit (tests/test_parallel_gpu.py::test_rccl_world1_*)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("CVCL_FORCE_DIST", "0") == "1"


def world_size() -> int:
    return dist.get_world_size() if is_distributed() else 1


def rank() -> int:
    return dist.get_rank() if is_distributed() else 0


def _all_gather_pair(fi: torch.Tensor, ft: torch.Tensor):
    """(image features, text features) [B, E] of every rank -> ([world*B, E], [world*B, E]), rank-major rows, in ONE collective:
    the two matrices travel stacked as [2, B, E] (SURVEY.md 8e asks for an all-gather of the features; two collectives of
    0.5 MB each are two latency-bound launches where one does)."""
    world = dist.get_world_size()
    x = torch.stack((fi, ft)).contiguous()                                      # [2, B, E]
    out = torch.empty((world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
    COLLECTIVES["all_gather"] += 1
    if dist.get_backend() == "nccl":                       # RCCL: one fused all-gather into the output buffer
        dist.all_gather_into_tensor(out, x)
    else:                                                  # gloo (CPU tests, several ranks sharing one GPU)
        dist.all_gather(list(out.unbind(0)), x)
    B, E = fi.shape
    return out[:, 0].reshape(world * B, E), out[:, 1].reshape(world * B, E)     # (copies: rank-major [world*B, E])


class _AllGatherPair(torch.autograd.Function):
    """(cat_r(fi_r), cat_r(ft_r)) along dim 0 from one collective; backward hands each rank the gradient rows of its own shard.
    (Every rank evaluates the same replicated loss, so no reduction is needed here.)"""

    @staticmethod
    def forward(ctx, fi, ft):
        ctx.rows = fi.shape[0]
        return _all_gather_pair(fi, ft)

    @staticmethod
    def backward(ctx, gi, gt):
        r = dist.get_rank()
        sl = slice(r * ctx.rows, (r + 1) * ctx.rows)
        return gi[sl].contiguous(), gt[sl].contiguous()


class _GlobalSimLogits(torch.autograd.Function):
    """all-gather of the (normalised) features + similarity logits of the global N_g x N_g batch, with the backward pass the
    sharding asks for ("back-props only its own 256 rows"): a rank needs d_loss / d_fi for ITS image rows and
    d_loss / d_ft for ITS text rows only, i.e. two [B, N_g] x [N_g, E] products (cvcl_sim_logits_bwd_rows) instead of the two
    N_g x N_g x E ones whose other world - 1 shards the gather's backward would throw away.  The temperature gradient is the
    full-matrix sum, identical on every rank (flagged ``_cvcl_replicated_grad`` -> averaged by the engine)."""

    @staticmethod
    def forward(ctx, fi, ft, neg_log_temp):
        from . import _hip as H
        gi, gt = _all_gather_pair(fi.contiguous(), ft.contiguous())
        Ng, E = gi.shape
        nlt = neg_log_temp.reshape(1).contiguous()
        logits = torch.empty(Ng, Ng, dtype=torch.float32, device=gi.device)
        H.check(H.lib().cvcl_sim_logits_fwd(H.ptr(gi, torch.float32), H.ptr(gt, torch.float32), H.ptr(nlt, torch.float32), H.ptr(logits),
                                           Ng, Ng, E, H.stream_ptr()), "cvcl_sim_logits_fwd")
        ctx.save_for_backward(gi, gt, nlt, logits)
        ctx.rows = fi.shape[0]
        ctx.temp_shape = neg_log_temp.shape
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        from . import _hip as H
        gi, gt, nlt, logits = ctx.saved_tensors
        Ng, E = gi.shape
        B, r = ctx.rows, dist.get_rank()
        need_i, need_t, need_s = ctx.needs_input_grad
        dev = gi.device
        d_fi = torch.empty(B, E, dtype=torch.float32, device=dev) if need_i else None
        d_ft = torch.empty(B, E, dtype=torch.float32, device=dev) if need_t else None
        d_s = torch.empty(1, dtype=torch.float32, device=dev) if need_s else None
        nb = H.lib().cvcl_sim_logits_bwd_workspace_bytes(Ng, Ng, E)
        ws = torch.empty(max(int(nb), 16), dtype=torch.uint8, device=dev)
        H.check(H.lib().cvcl_sim_logits_bwd_rows(H.ptr(gi), H.ptr(gt), H.ptr(nlt), H.ptr(logits), H.ptr(d_logits.contiguous(), torch.float32),
                                                H.ptr(d_fi), H.ptr(d_ft), H.ptr(d_s), Ng, Ng, E, r * B, B, r * B, B, H.ptr(ws), nb,
                                                H.stream_ptr()), "cvcl_sim_logits_bwd_rows")
        return d_fi, d_ft, (d_s.reshape(ctx.temp_shape) if need_s else None)


def global_sim_logits(image_features, text_features, neg_log_temp):
    """logits_per_image [N_g, N_g] of the all-gathered batch (reference multimodal/multimodal.py:755 applied to cat_r(features_r)):
    one collective forward, own-row gradient products backward.  Device tensors only (the HIP path)."""
    return _GlobalSimLogits.apply(image_features, text_features, neg_log_temp)


def local_term_scale(global_negatives: bool) -> float:
    """Factor for a loss term that is a mean over the RANK'S OWN samples when gradients are summed over ranks (global
    negatives): 1 / world, so that the summed gradient is the gradient of the mean over all ranks' samples.  1.0 otherwise
    (single process; local negatives, where the engine averages every gradient)."""
    return 1.0 / world_size() if (global_negatives and is_distributed()) else 1.0


class _AllGatherRows(torch.autograd.Function):
    """cat_r(x_r) along dim 0 (rank-major); backward hands each rank the gradient rows of its own shard (every rank evaluates the same
    replicated loss: no reduction).  The spatial head's exchange -- per-location image rows, per-word text rows and the utterance
    lengths differ in shape, so they travel one tensor per collective."""

    @staticmethod
    def forward(ctx, x):
        world = dist.get_world_size()
        x = x.contiguous()
        out = torch.empty((world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        COLLECTIVES["all_gather"] += 1
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(out, x)
        else:
            dist.all_gather(list(out.unbind(0)), x)
        ctx.rows = x.shape[0]
        return out.reshape((world * x.shape[0],) + tuple(x.shape[1:]))

    @staticmethod
    def backward(ctx, g):
        r = dist.get_rank()
        return g[r * ctx.rows:(r + 1) * ctx.rows].contiguous()


def gather_rows(x: torch.Tensor) -> torch.Tensor:
    """[B, ...] per rank -> [world*B, ...] on every rank (rank-major), differentiable for floating-point x."""
    if not is_distributed():
        return x
    return _AllGatherRows.apply(x)


def common_text_length(L: int) -> int:
    """The rank-invariant L the spatial head pads its per-word rows to before they are all-gathered (``SPATIAL_TEXT_LEN``, default
    MAX_LEN_UTTERANCE).  Raises on a batch longer than that -- loudly on this rank, instead of a mis-sized collective on all."""
    limit = SPATIAL_TEXT_LEN
    if limit is None:
        from .multimodal_data_module import MAX_LEN_UTTERANCE
        limit = MAX_LEN_UTTERANCE
    if L > limit:
        raise ValueError(f"spatial embeddings under data-parallel global negatives gather per-word rows padded to {limit} tokens, "
                         f"this rank's batch has L = {L}; set multimodal.parallel.SPATIAL_TEXT_LEN on every rank (or use --local_negatives)")
    return int(limit)


def check_spatial_global_bytes(n_loc_rows: int, n_word_rows: int, sim: str, device) -> None:
    """The replicated spatial match map of the GLOBAL batch is [N_g*HW, N_g*L] fp32, twice for sim='max' (the map and its
    gradient): 2 x 20 GB at 8 ranks x 256 pairs x 7x7 x 25.  Refuse with a message what would otherwise be an out-of-memory
    kill somewhere inside the step."""
    if sim != "max" or device.type != "cuda":
        return
    need = 2 * 4 * n_loc_rows * n_word_rows                                        # the map and its gradient
    total = torch.cuda.get_device_properties(device).total_memory
    if need > 0.6 * total:
        raise RuntimeError(f"the global-negatives spatial match map needs {need / 2**30:.1f} GiB of the device's {total / 2**30:.0f} GiB "
                           f"({n_loc_rows} location rows x {n_word_rows} word rows, fp32, forward + gradient): use --local_negatives, "
                           f"--sim mean or a smaller per-rank batch")


def broadcast_from_rank0(t: torch.Tensor) -> torch.Tensor:
    """In-place broadcast of rank 0's ``t`` (every rank must call it: see ResNet.request_centre_sync for the one caller)."""
    COLLECTIVES["broadcast"] += 1
    dist.broadcast(t, src=0)
    return t


def gather_features(image_features: torch.Tensor, text_features: torch.Tensor):
    """[B,E] per rank -> [world*B, E] on every rank (rank-major row order)."""
    if not is_distributed():
        return image_features, text_features
    return _AllGatherPair.apply(image_features, text_features)


class DataParallelEngine:
    """Bucketed gradient all-reduce over the trainable parameters of a module."""

    def __init__(self, device, bucket_bytes: int = 32 << 20, global_negatives: bool = True):
        self.device = device
        self.bucket_bytes = bucket_bytes
        self.global_negatives = global_negatives
        self.buckets = []            # list of dict(params, buf, pending, handle)
        self._hooks = []
        self._bucket_of = {}

    @classmethod
    def from_env(cls, device, **kw):
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1 and not dist.is_initialized():
            backend = os.environ.get("CVCL_DIST_BACKEND") or ("nccl" if device.type == "cuda" else "gloo")   # "nccl" = RCCL
            if device.type == "cuda":
                torch.cuda.set_device(device)
            dist.init_process_group(backend=backend)
        return cls(device, **kw)

    def attach(self, module: torch.nn.Module):
        for h in self._hooks:
            h.remove()
        self._hooks, self.buckets, self._bucket_of = [], [], {}
        self._listen(False)
        if not is_distributed():
            return self
        # attach() is called by every rank (it builds the rank's buckets): the one rank-synchronous point before the first step.
        # Trunks that keep rank-local calibration state adopt rank 0's at their next train-mode pass (ResNet.request_centre_sync).
        for m in module.modules():
            req = getattr(m, "request_centre_sync", None)
            if callable(req):
                req()
        params = [p for p in module.parameters() if p.requires_grad]
        params.reverse()                                   # roughly the order gradients become ready
        cur, cur_bytes = [], 0
        for p in params:
            nb = p.numel() * p.element_size()
            if cur and cur_bytes + nb > self.bucket_bytes:
                self._new_bucket(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self._new_bucket(cur)
        self._listen(True)
        return self

    def _listen(self, on: bool):
        """Receive the weight gradients trunk_train computes outside autograd's accumulation (fine-tuning path)."""
        try:
            from . import trunk_train
        except Exception:                                   # pragma: no cover - CPU-only host-logic tests without the library
            return
        if on:
            trunk_train._WGRAD_LISTENER = self.grad_ready
        elif getattr(trunk_train, "_WGRAD_LISTENER", None) is not None and \
                getattr(trunk_train._WGRAD_LISTENER, "__self__", None) is self:
            trunk_train._WGRAD_LISTENER = None

    def _new_bucket(self, params):
        total = sum(p.numel() for p in params)
        offs, off = {}, 0
        for p in params:
            offs[id(p)] = off
            off += p.numel()
        b = {"params": list(params), "buf": torch.zeros(total, dtype=params[0].dtype, device=params[0].device),
             "pending": len(params), "handle": None, "seen": set(), "offs": offs, "events": []}
        self.buckets.append(b)
        for p in params:
            self._bucket_of[id(p)] = b
            self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(b)))

    def _arrive(self, b, p, grad):
        """Gradient of p is final: into the bucket (on the current stream); the last arrival launches the all-reduce."""
        if id(p) in b["seen"]:
            return
        b["seen"].add(id(p))
        off = b["offs"][id(p)]
        b["buf"][off:off + p.numel()].copy_(grad.reshape(-1))
        # Gradients reach a bucket from more than one stream: autograd's hooks run on the backward (main) stream, trunk_train's
        # weight gradients on its side stream.  The collective is enqueued behind whichever stream makes the LAST arrival, so
        # every arrival leaves an event behind its copy and ``_launch`` makes the launching stream wait for all of them
        # (without this a bucket that closes on a BatchNorm hook reduced conv gradients still being written on the side stream).
        if b["buf"].is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(b["buf"].device))
            b["events"].append(ev)
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def _make_hook(self, b):
        def hook(p):
            self._arrive(b, p, p.grad)
        return hook

    def grad_ready(self, p, grad):
        """A gradient produced outside autograd's accumulation (trunk_train._defer_wgrad), called on the stream that
        computes it: the collective of a completed bucket is enqueued behind THAT stream, not behind the backward pass."""
        b = self._bucket_of.get(id(p))
        if b is None or not is_distributed():
            return
        if p.grad is not None:                               # accumulating over several backward passes: reduce the total
            b.setdefault("late", set()).add(id(p))           # at reduce_gradients() time instead
            return
        self._arrive(b, p, grad)

    def _launch(self, b):
        b["had_grad"] = [id(p) in b["seen"] or p.grad is not None for p in b["params"]]
        if b["events"]:                                      # behind every stream that wrote into the bucket
            cur = torch.cuda.current_stream(b["buf"].device)
            for ev in b["events"]:
                cur.wait_event(ev)
            b["events"] = []
        for p in b["params"]:                                # parameters without a gradient this step contribute zeros
            if id(p) not in b["seen"] or id(p) in b.get("late", ()):
                off = b["offs"][id(p)]
                if p.grad is not None:
                    b["buf"][off:off + p.numel()].copy_(p.grad.reshape(-1))
                else:
                    b["buf"][off:off + p.numel()].zero_()
        COLLECTIVES["all_reduce"] += 1
        b["handle"] = dist.all_reduce(b["buf"], op=dist.ReduceOp.SUM, async_op=True)

    def reduce_gradients(self):
        """Wait for (or launch) every bucket and write the reduced gradients back into ``.grad``."""
        if not is_distributed():
            return
        world = dist.get_world_size()
        for b in self.buckets:
            if b["handle"] is None:                         # some parameter received no gradient this step
                self._launch(b)
            b["handle"].wait()
            for p, had in zip(b["params"], b["had_grad"]):
                if had and p.grad is not None:   # a parameter no rank produced a gradient for keeps grad=None (the optimizer
                    off = b["offs"][id(p)]       # skips it, as in the single-process reference)
                    g = b["buf"][off:off + p.numel()].view_as(p)
                    averaged = (not self.global_negatives) or getattr(p, "_cvcl_replicated_grad", False)
                    p.grad.copy_(g / world if averaged else g)
            b["pending"], b["handle"] = len(b["params"]), None
            b["events"] = []
            b["seen"].clear()
            b.pop("late", None)


class OverlappedUpdate:
    """Hide the gradient all-reduce and the optimizer step of step k behind the frozen image trunk of step k+1.

    With a frozen CNN / ViT every trainable parameter (fc / head, text encoder, temperature) is used AFTER the trunk in
    the forward pass, and nothing the optimizer changes is read by the trunk.  So after ``loss.backward()`` (whose hooks
    have already launched the bucketed RCCL all-reduce) the next step's trunk can be enqueued immediately; the wait for
    the collective, the write-back and ``optimizer.step()`` run from a hook placed just before the trunk's output is
    consumed.  The sequence of parameter values is exactly the one of the sequential schedule; only the order in which
    independent work is enqueued changes.  ``flush()`` completes the last pending update.  No-op schedule change when a
    trunk parameter is trainable (``--finetune_cnn``): then the update is applied immediately.
    """

    def __init__(self, engine: DataParallelEngine, optimizer, vision_encoder):
        self.engine, self.opt = engine, optimizer
        self.pending = False
        trunk = getattr(vision_encoder, "model", None)
        head_names = ("fc.", "head.")
        self.can_defer = trunk is not None and hasattr(trunk, "_pre_head_callback") and not any(
            p.requires_grad for n, p in trunk.named_parameters() if not n.startswith(head_names))
        if self.can_defer:
            trunk._pre_head_callback = self.flush

    def flush(self):
        if self.pending:
            self.pending = False
            self.engine.reduce_gradients()
            self.opt.step()

    def step_done(self):
        """call right after loss.backward()"""
        self.pending = True
        if not self.can_defer:
            self.flush()

    def zero_grad(self):
        """call after the forward pass (which ran flush() through the hook) and before backward"""
        self.flush()
        self.opt.zero_grad(set_to_none=True)
