"""ResNeXt-50 32x4d with torchvision's module tree / state_dict keys, computed by libcvcl_hip.

The reference gets this network from ``torchvision.models.resnext50_32x4d``
(multimodal/multimodal.py:155-158, multimodal/utils.py:207-209 in the reference).  Here the
``nn.Conv2d`` / ``nn.BatchNorm2d`` children are *parameter containers only* (same names, shapes and
init as torchvision, so checkpoints interchange); ``forward`` hands their tensors to
``cvcl_resnext50_fwd`` which enqueues the whole trunk as hand-written HIP kernels.  There is no
PyTorch-op fallback: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch
import torch.nn as nn

from . import _hip as H
from . import ops

LAYERS = (3, 4, 6, 3)
GROUPS, WIDTH_PER_GROUP = 32, 4
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        width = int(planes * (WIDTH_PER_GROUP / 64.0)) * GROUPS
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, groups=GROUPS, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):  # pragma: no cover - blocks are never run one by one
        raise H.CvclError("ResNeXt blocks are parameter containers; run the whole trunk through ResNet.forward")


class _Stage(nn.Sequential):
    """layer{1..4}: holds the Bottlenecks; called with the already-computed stage output so that
    forward hooks registered on it (the reference's ``Hook(model.layer4)``, attention_maps.py:83-101)
    observe the feature map exactly as with torchvision."""

    def forward(self, feature_map):
        return feature_map


class ResNet(nn.Module):
    def __init__(self, num_classes: int = 1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inplanes = 64
        for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), LAYERS), start=1):
            mods = []
            for bi in range(blocks):
                stride = 2 if (li > 1 and bi == 0) else 1
                ds = None
                if bi == 0:
                    ds = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
                mods.append(Bottleneck(inplanes, planes, stride, ds))
                inplanes = planes * 4
            setattr(self, f"layer{li}", _Stage(*mods))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, num_classes)
        for m in self.modules():                       # torchvision's init (zero_init_residual=False)
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self.compute_dtype = torch.float32             # torch.bfloat16 = perf mode
        self._pack_cache = {}
        self._ws_cache = {}
        self._pre_head_callback = None                  # parallel.OverlappedUpdate: runs between the trunk and fc
        self._centres = None                            # storage centres of the 53 raw conv outputs (see storage_centres)

    # ---- (conv, bn) pairs in torchvision state_dict order -----------------------------------
    def conv_bn_pairs(self):
        pairs = [(self.conv1, self.bn1, H.PACK_STEM7)]
        for li in (1, 2, 3, 4):
            for blk in getattr(self, f"layer{li}"):
                pairs.append((blk.conv1, blk.bn1, H.PACK_DENSE))
                pairs.append((blk.conv2, blk.bn2, H.PACK_GCONV3))
                pairs.append((blk.conv3, blk.bn3, H.PACK_DENSE))
                if blk.downsample is not None:
                    pairs.append((blk.downsample[0], blk.downsample[1], H.PACK_DENSE))
        return pairs

    def _packed_layers(self, dt: int, device):
        """Re-lay-out conv weights for the kernels (once per weight version) and build the C array."""
        pairs = self.conv_bn_pairs()
        key = (dt, str(device)) + tuple((c.weight.data_ptr(), c.weight._version, b.running_mean.data_ptr()) for c, b, _ in pairs)
        hit = self._pack_cache.get("k")
        if hit is not None and hit[0] == key:
            return hit[1]
        lib = H.lib()
        arr = (H.ConvBnParams * len(pairs))()
        keep = []
        for i, (conv, bn, kind) in enumerate(pairs):
            w = conv.weight.detach()
            cout, cing, k, _ = w.shape
            nb = lib.cvcl_packed_weight_bytes(dt, kind, cout, cing, k)
            buf = torch.empty(nb, dtype=torch.uint8, device=device)
            H.check(lib.cvcl_pack_conv_weight(dt, kind, H.ptr(w.contiguous(), torch.float32), H.ptr(buf), cout, cing, k,
                                              H.stream_ptr()), "cvcl_pack_conv_weight")
            keep.append(buf)
            arr[i].w = buf.data_ptr()
            arr[i].gamma = H.ptr(bn.weight.detach(), torch.float32)
            arr[i].beta = H.ptr(bn.bias.detach(), torch.float32)
            arr[i].running_mean = H.ptr(bn.running_mean, torch.float32)
            arr[i].running_var = H.ptr(bn.running_var, torch.float32)
            arr[i].num_batches_tracked = H.ptr(bn.num_batches_tracked, torch.int64)
        if device.type == "cuda":
            torch.cuda.current_stream(device).synchronize()   # packed once, then read by every stream that runs the trunk
        self._pack_cache["k"] = (key, (arr, keep))
        return arr, keep

    # ---- centred storage of the raw convolution outputs (include/cvcl_hip.h "Centred storage") ------------------------
    # A raw conv output y whose per-channel batch mean is large next to its spread loses precision when it is rounded to bf16
    # before BatchNorm subtracts the mean.  BatchNorm(y - c) == BatchNorm(y), so the kernels store round(y - c) with c close to
    # the batch mean.  The frozen trunk takes c from ONE calibration pass -- the batch means of the first train-mode batch it
    # sees after its weights (re)appeared -- and keeps it: a forward pass stays a pure function of (input, weights, BatchNorm
    # buffers, centres), whatever the stream schedule.  c only has to be within ~sigma of the batch mean, which a stationary
    # input distribution guarantees; ``recalibrate_centres()`` drops it (e.g. after a change of data distribution).
    # Eval mode centres on the running means (the library's default for centres == NULL).  The fine-tuning twin
    # (trunk_train) tracks instead: its weights move every step, so each layer's c follows the previous step's batch mean.
    # $CVCL_CENTRED_STORAGE=0 switches the whole mechanism off (plain storage, the round-2 numerics).
    def recalibrate_centres(self):
        """Forget the calibrated (frozen path) and the tracked (fine-tuning path) centres: the next train-mode pass starts over.
        Data parallel: a COLLECTIVE call, like every other state change of a replicated model -- all ranks call it, and all ranks'
        next train-mode pass adopts rank 0's new calibration (``request_centre_sync``)."""
        self.__dict__["_centres"] = None
        self.__dict__["_track_centres"] = None
        self.__dict__["_centres_restored"] = None
        self.__dict__["_track_restored"] = None
        from . import parallel
        if parallel.is_distributed():
            self.request_centre_sync()

    def request_centre_sync(self):
        """Data parallel: ask this replica's NEXT train-mode pass to take part in one broadcast of rank 0's storage centres, whatever
        its own cache says (hit, restored from a checkpoint, freshly calibrated).  Raised only at rank-synchronous points --
        ``DataParallelEngine.attach`` and ``recalibrate_centres`` -- so either every rank enters the broadcast or none does: the
        decision never depends on a rank's local cache state (a checkpoint resumed on some ranks only, a weight re-allocated on one
        rank, a rank-local extra forward).  A cache miss WITHOUT a pending request recalibrates locally and talks to nobody."""
        self.__dict__["_centre_sync_pending"] = True

    def export_centres(self):
        """Checkpoint form of the calibrated / tracked centres (CPU tensors), or None when there are none yet."""
        out = {}
        hit = self.__dict__.get("_centres")
        if hit is not None:
            if hit[2] is not None:
                hit[2].synchronize()
            out["frozen"] = hit[1].detach().cpu()
        t = self.__dict__.get("_track_centres")
        if t is not None:
            out["tracking"] = t.detach().cpu()
        pend = self.__dict__.get("_centres_restored")
        if pend is not None and "frozen" not in out:
            out["frozen"] = pend.detach().cpu()
        return out or None

    def import_centres(self, state):
        """Adopt centres saved by ``export_centres``: the next train-mode pass uses them instead of calibrating (whatever the
        weights' new storage addresses are), so a resumed bf16 run evaluates the same forward function as the run that saved it.
        A checkpoint carries rank 0's centres (the ones every replica of a data-parallel run holds after the broadcast at its first
        step); on resume ``DataParallelEngine.attach`` requests the same broadcast again, so the replicas agree even when only some
        ranks restored a checkpoint."""
        self.__dict__["_centres"] = None
        self.__dict__["_centres_restored"] = state["frozen"].clone() if state.get("frozen") is not None else None
        self.__dict__["_track_centres"] = None
        self.__dict__["_track_restored"] = state["tracking"].clone() if state.get("tracking") is not None else None

    def centred_storage(self) -> bool:
        return os.environ.get("CVCL_CENTRED_STORAGE", "1") != "0" and self.compute_dtype == torch.bfloat16

    def tracking_centres(self, device):
        """[53, 2048] f32 centres of the fine-tuning path, updated in place by every BatchNorm forward (trunk_train)."""
        if not self.centred_storage():
            return None
        t = self.__dict__.get("_track_centres")
        if t is None or t.device != device:
            pend = self.__dict__.pop("_track_restored", None)
            t = pend.to(device) if pend is not None else torch.zeros(53, 2048, dtype=torch.float32, device=device)
            self.__dict__["_track_centres"] = t
        return t

    def _calibrated_centres(self, x, dt, arr, ws, nb, key):
        """-> [53, 2048] f32 centres for a train-mode pass over ``x``'s distribution (calibrating first if need be), or None."""
        if not self.centred_storage():
            return None
        from . import parallel
        sync = bool(self.__dict__.get("_centre_sync_pending")) and parallel.is_distributed()
        hit = self.__dict__.get("_centres")
        if hit is not None and hit[0] == key:
            if hit[2] is not None:
                torch.cuda.current_stream(x.device).wait_event(hit[2])     # (calibrated on another trunk stream)
            if not sync:
                return hit[1]
            centres = hit[1]
        else:
            pend = self.__dict__.pop("_centres_restored", None)
            if pend is not None:                                  # restored from a checkpoint: no calibration pass
                centres = pend.to(device=x.device, dtype=torch.float32).contiguous()
                if not sync:
                    self.__dict__["_centres"] = (key, centres, None)
                    return centres
            else:
                lib = H.lib()
                B, _, Hh, Ww = x.shape
                moments = torch.empty(lib.cvcl_resnext50_moments_floats(), dtype=torch.float32, device=x.device)
                fmap = torch.empty(B, Hh // 32, Ww // 32, 2048, dtype=self.compute_dtype, device=x.device)
                pooled = torch.empty(B, 2048, dtype=torch.float32, device=x.device)
                # a plain-storage train-mode pass that leaves every layer's batch mean behind and touches no BatchNorm buffer
                H.check(lib.cvcl_resnext50_fwd_deferred_stats(dt, B, Hh, Ww, H.ptr(x), arr, len(arr), H.ptr(ws), nb, H.ptr(fmap),
                                                              H.ptr(pooled), BN_EPS, H.ptr(moments), None, H.stream_ptr()),
                        "cvcl_resnext50_fwd_deferred_stats")
                centres = moments.view(53, 2, 2048)[:, 0, :].contiguous()
        if sync:
            # data parallel: every replica must evaluate the SAME bf16 forward function, so rank 0's centres are adopted by all (a
            # centre only has to be within ~sigma of a rank's batch mean; the ranks see shards of one distribution).  Every rank is
            # here: the request was raised at a rank-synchronous point (request_centre_sync), not by this rank's cache state
            self.__dict__["_centre_sync_pending"] = False
            centres = parallel.broadcast_from_rank0(centres.clone() if hit is not None and centres is hit[1] else centres)
        ready = None
        if x.is_cuda:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(x.device))
        self.__dict__["_centres"] = (key, centres, ready)
        return centres

    def trunk(self, x: torch.Tensor, defer_wait: bool = False):
        """conv1 .. layer4 + avgpool.  -> (pooled [B,2048] f32, layer4 map as a logical NCHW view).  With a trunk stream and
        ``defer_wait`` the result is a handle for ``self._trunk_stream.wait`` (the caller's stream has not waited yet)."""
        if torch.is_grad_enabled() and any(p.requires_grad for c, b, _ in self.conv_bn_pairs() for p in (c.weight, b.weight, b.bias)):
            from .trunk_train import trunk_train       # --finetune_cnn: differentiable twin (saves activations)
            return trunk_train(self, x)
        if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
            raise H.CvclError(f"expected NCHW fp32 images, got {tuple(x.shape)} {x.dtype}")
        x = x.contiguous()
        ts = self.__dict__.get("_trunk_stream")
        if ts is not None and x.is_cuda:                  # frozen trunk on its own stream (H.TrunkStream)
            handle = ts.launch(lambda slot: self._trunk_launch(x, slot), x)
            if defer_wait:
                return handle
            return ts.wait(handle)
        return self._trunk_launch(x)

    def enable_trunk_stream(self, device, inputs="caller", stream=None, n_streams=None):
        """Run the frozen trunk on a stream of its own so it overlaps the previous step's trainable tail (see H.TrunkStream).
        n_streams = 2 (the default; $CVCL_TRUNK_STREAMS overrides): consecutive passes alternate between two streams and also
        overlap each other -- each has its own workspace, and the BatchNorm running statistics are updated in pass order by
        ``cvcl_resnext50_apply_moments`` (same values as the one-stream schedule, bit for bit)."""
        if n_streams is None:
            n_streams = 1 if stream is not None else int(os.environ.get("CVCL_TRUNK_STREAMS", "2"))
        self.__dict__["_trunk_stream"] = H.TrunkStream(device, inputs, stream, n_streams) if inputs else None
        self.__dict__["_ema_done"] = None
        return self.__dict__["_trunk_stream"]

    def _trunk_launch(self, x, slot=None):
        B, _, Hh, Ww = x.shape
        dt = H.cvcl_dtype(self.compute_dtype)
        lib = H.lib()
        with torch.no_grad():
            arr, _keep = self._packed_layers(dt, x.device)
            nb = lib.cvcl_resnext50_workspace_bytes(dt, B, Hh, Ww)
            ts = self.__dict__.get("_trunk_stream")
            piped = slot is not None and ts is not None and ts.n_streams > 1    # passes on two streams: scratch per stream
            sidx = ts.stream_index if piped else None
            wkey = (nb, str(x.device), sidx)
            ws = self._ws_cache.get(wkey)
            if ws is None:
                for k in [k for k in self._ws_cache if isinstance(k[0], int) and k[:2] != wkey[:2]]:    # other batch shapes' scratch
                    del self._ws_cache[k]
                ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
                self._ws_cache[wkey] = ws
            if slot is None:
                fmap = torch.empty(B, Hh // 32, Ww // 32, 2048, dtype=self.compute_dtype, device=x.device)
                pooled = torch.empty(B, 2048, dtype=torch.float32, device=x.device)
            else:                                         # side-stream mode: a ring of persistent output sets (H.TrunkStream.launch)
                key = ("out", slot, B, Hh, Ww, self.compute_dtype, str(x.device))
                if key not in self._ws_cache:
                    self._ws_cache[key] = (torch.empty(B, Hh // 32, Ww // 32, 2048, dtype=self.compute_dtype, device=x.device),
                                           torch.empty(B, 2048, dtype=torch.float32, device=x.device))
                fmap, pooled = self._ws_cache[key]
            centres = self._calibrated_centres(x, dt, arr, ws, nb, self._pack_cache["k"][0]) if self.training else None
            if piped and self.training:
                # the pass leaves its batch moments behind; the 53 running-statistics updates run as one launch behind the
                # previous pass's (other stream), so they are applied in pass order with the one-stream arithmetic
                mkey = ("moments", sidx, str(x.device))
                if mkey not in self._ws_cache:
                    self._ws_cache[mkey] = torch.empty(lib.cvcl_resnext50_moments_floats(), dtype=torch.float32, device=x.device)
                moments = self._ws_cache[mkey]
                H.check(lib.cvcl_resnext50_fwd_deferred_stats(dt, B, Hh, Ww, H.ptr(x), arr, len(arr), H.ptr(ws), nb, H.ptr(fmap),
                                                              H.ptr(pooled), BN_EPS, H.ptr(moments), H.ptr(centres), H.stream_ptr()),
                        "cvcl_resnext50_fwd_deferred_stats")
                cur = torch.cuda.current_stream(x.device)
                prev = self.__dict__.get("_ema_done")
                if prev is not None:
                    cur.wait_event(prev)
                H.check(lib.cvcl_resnext50_apply_moments(arr, len(arr), H.ptr(moments), BN_MOMENTUM, H.stream_ptr()),
                        "cvcl_resnext50_apply_moments")
                done = torch.cuda.Event()
                done.record(cur)
                self.__dict__["_ema_done"] = done
            else:
                prev = self.__dict__.get("_ema_done")
                if prev is not None and x.is_cuda:           # this pass reads / updates the running statistics in place: behind
                    torch.cuda.current_stream(x.device).wait_event(prev)     # the last deferred update, whatever stream ran it
                # (eval mode: centres None = the library centres every stored tensor on its running mean)
                H.check(lib.cvcl_resnext50_fwd(dt, B, Hh, Ww, int(self.training), H.ptr(x), arr, len(arr), H.ptr(ws), nb,
                                               H.ptr(fmap), H.ptr(pooled), BN_MOMENTUM, BN_EPS, H.ptr(centres), H.stream_ptr()),
                        "cvcl_resnext50_fwd")
        return pooled, fmap.permute(0, 3, 1, 2)

    # caches hold ctypes arrays / device buffers: never pickled (save_hyperparameters pickles whole encoder modules)
    def __getstate__(self):
        d = dict(self.__dict__)
        d["_pack_cache"], d["_ws_cache"] = {}, {}
        d["_pre_head_callback"] = None
        d["_trunk_stream"] = None
        d["_ema_done"] = None
        d["_centres"] = None
        d["_track_centres"] = None
        d["_centres_restored"] = None
        d["_track_restored"] = None
        d["_centre_sync_pending"] = False
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self.__dict__.setdefault("compute_dtype", torch.float32)      # objects pickled by torchvision lack these
        self.__dict__.setdefault("_pack_cache", {})
        self.__dict__.setdefault("_ws_cache", {})
        self.__dict__.setdefault("_pre_head_callback", None)
        self.__dict__.setdefault("_trunk_stream", None)
        self.__dict__.setdefault("_centres", None)

    def forward(self, x):
        ts = self.__dict__.get("_trunk_stream")
        out = self.trunk(x, defer_wait=True)
        cb = self.__dict__.get("_pre_head_callback")
        if cb is not None:
            cb()                                         # deferred all-reduce wait + optimizer step of the previous step
        if ts is not None and isinstance(out, tuple) and len(out) == 2 and isinstance(out[1], torch.cuda.Event):
            out = ts.wait(out)                           # (with a trunk stream the update above overlapped the trunk)
        pooled, fmap = out
        # fire the forward hooks registered on layer4 (the reference's Hook(model.layer4)); done by hand so that it
        # also works when layer4 is a plain nn.Sequential unpickled from a torchvision-built checkpoint
        for hook in list(self.layer4._forward_hooks.values()):
            hook(self.layer4, (fmap,), fmap)
        if isinstance(self.fc, nn.Linear):
            return ops.linear_f32(pooled, self.fc.weight, self.fc.bias)
        return self.fc(pooled)                           # e.g. nn.Identity (utils.build_dino_mugs)


class _RowsToF32(torch.autograd.Function):
    """bf16 rows of the layer-4 map -> fp32 operand of the 1x1 projection; backward rounds the gradient to the trunk's bf16."""

    @staticmethod
    def forward(ctx, rows):
        rows = rows.contiguous()
        out = torch.empty(rows.shape, dtype=torch.float32, device=rows.device)
        H.check(H.lib().cvcl_bf16_to_f32(H.ptr(rows), H.ptr(out), rows.numel(), H.stream_ptr()), "cvcl_bf16_to_f32")
        ctx.dtype = rows.dtype
        return out

    @staticmethod
    def backward(ctx, d_out):
        d_out = d_out.contiguous()
        dx = torch.empty(d_out.shape, dtype=ctx.dtype, device=d_out.device)
        H.check(H.lib().cvcl_f32_to_bf16(H.ptr(d_out, torch.float32), H.ptr(dx), d_out.numel(), H.stream_ptr()), "cvcl_f32_to_bf16")
        return dx


class SpatialResNet(nn.Sequential):
    """embedding_type='spatial' vision model of the reference (multimodal.py:181-185):
    ``nn.Sequential(*list(resnet.children())[:-2], nn.Conv2d(2048, E, 1))`` -- same child indices, hence the same
    state_dict keys (``0.weight`` = conv1 ... ``7.*`` = layer4, ``8.*`` = the 1x1 projection).  The children ARE the
    ResNet's modules (shared parameters); the trunk runs through ``cvcl_resnext50_fwd`` and the projection as an fp32
    GEMM over the per-location rows.  Output: ([B, E, H/32, W/32] view of NHWC rows, layer4 map)."""

    def __init__(self, resnet: ResNet, embedding_dim: int):
        super().__init__(resnet.conv1, resnet.bn1, resnet.relu, resnet.maxpool, resnet.layer1, resnet.layer2, resnet.layer3,
                         resnet.layer4, nn.Conv2d(2048, embedding_dim, 1))
        object.__setattr__(self, "_resnet", resnet)          # not a registered child: its modules are already children 0..7

    @property
    def compute_dtype(self):
        return self._resnet.compute_dtype

    @compute_dtype.setter
    def compute_dtype(self, dt):
        self._resnet.compute_dtype = dt

    def train(self, mode: bool = True):
        super().train(mode)
        self._resnet.training = mode
        return self

    def forward(self, x):
        r = self._resnet
        # --finetune_cnn (reference multimodal.py:175-185: autograd through the whole nn.Sequential): r.trunk then returns
        # trunk_train's differentiable layer-4 map, and the gradient of the 1x1 projection flows back into it below
        _pooled, fmap = r.trunk(x)                           # fmap: NCHW view of the NHWC layer-4 map
        for hook in list(self[7]._forward_hooks.values()):   # the reference hooks model[-2] = layer4
            hook(self[7], (fmap,), fmap)
        B, Cc, Hh, Ww = fmap.shape
        rows = fmap.permute(0, 2, 3, 1).reshape(B * Hh * Ww, Cc)
        if rows.dtype != torch.float32:
            rows = _RowsToF32.apply(rows)
        proj = self[8]
        feat = ops.linear_f32(rows, proj.weight.view(proj.out_channels, Cc), proj.bias)       # [B*H*W, E]
        return feat.view(B, Hh, Ww, proj.out_channels).permute(0, 3, 1, 2)


def register_torchvision_alias():
    """Make ``torchvision.models.resnet.{ResNet,Bottleneck}`` resolvable when torchvision is absent, so that the
    reference's Lightning checkpoints (whose ``hyper_parameters`` pickle the whole VisionEncoder, i.e. a torchvision
    ResNet object: multimodal_lit.py:74,139 of the reference) can be unpickled onto these classes."""
    import sys
    import types
    try:
        import torchvision  # noqa: F401  (a real torchvision wins)
        return False
    except Exception:
        pass
    tv = sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    models = sys.modules.setdefault("torchvision.models", types.ModuleType("torchvision.models"))
    resnet = sys.modules.setdefault("torchvision.models.resnet", types.ModuleType("torchvision.models.resnet"))
    resnet.ResNet, resnet.Bottleneck = ResNet, Bottleneck
    models.resnet, models.resnext50_32x4d = resnet, resnext50_32x4d
    tv.models = models
    return True


def resnext50_32x4d(pretrained: bool = False, **kwargs) -> ResNet:
    if pretrained:
        raise H.CvclError("no network in this environment: ImageNet-pretrained torchvision weights are unavailable")
    return ResNet(**kwargs)
