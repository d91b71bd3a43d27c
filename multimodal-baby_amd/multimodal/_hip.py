"""ctypes binding of libcvcl_hip.so (the C ABI declared in include/cvcl_hip.h).

The library is loaded AFTER torch so that its DT_NEEDED ``libamdhip64.so.7`` resolves to the HIP
runtime torch already mapped (same soname) -- one runtime per process, so torch's stream handles
and device pointers are valid inside the library.  There is no CPU fallback: if the library is
missing or a tensor is not a contiguous device tensor the call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libcvcl_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
ABI_VERSION = 5
PACK_DENSE, PACK_STEM7, PACK_GCONV3 = 0, 1, 2
KERNEL_CLASSES = ("gemm", "gconv3x3", "stem7x7", "bn_finalize", "bn_add_relu", "bn_relu_maxpool", "avgpool", "head",
                  "other", "attention", "layernorm", "lstm", "gemm_f32", "bn_relu_apply", "bn_bwd", "wgrad", "gemm8w", "gemm_pro")


class CvclError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("lda", C.c_int), ("ldw", C.c_int), ("ldc", C.c_int),
        ("a_scale", C.c_void_p), ("a_shift", C.c_void_p), ("a_relu", C.c_int),
        ("gather_ho", C.c_int), ("gather_wo", C.c_int), ("gather_hi", C.c_int), ("gather_wi", C.c_int),
        ("gather_stride", C.c_int),
        ("exp_scale", C.c_void_p), ("bias", C.c_void_p), ("act", C.c_int),
        ("R", C.c_void_p), ("ldr", C.c_int),
        ("stats", C.c_void_p), ("stats_rows", C.c_int),
        ("c_scale", C.c_void_p), ("c_shift", C.c_void_p), ("r_scale", C.c_void_p), ("r_shift", C.c_void_p),
        ("C_pre", C.c_void_p), ("G", C.c_void_p), ("ldg", C.c_int),
        ("centre", C.c_void_p),
        ("A2", C.c_void_p), ("W2", C.c_void_p), ("K2", C.c_int), ("lda2", C.c_int), ("ldw2", C.c_int), ("centre2", C.c_void_p),
        ("ln_stats", C.c_void_p), ("ln_colsum", C.c_void_p), ("row_part", C.c_void_p),
        ("a_trans", C.c_int), ("w_trans", C.c_int), ("a_rowsum", C.c_void_p),
        ("f32_split", C.c_int),
    ]


class GemmFp8Args(C.Structure):
    _fields_ = [
        ("A8", C.c_void_p), ("a_scale", C.c_void_p), ("a_block_scales", C.c_void_p), ("lda", C.c_int),
        ("W8", C.c_void_p), ("w_scale", C.c_void_p), ("ldw", C.c_int),
        ("C", C.c_void_p), ("ldc", C.c_int), ("c8", C.c_void_p), ("c_block_scales", C.c_void_p), ("ldc8", C.c_int),
        ("bias", C.c_void_p), ("act", C.c_int), ("R", C.c_void_p), ("ldr", C.c_int),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("ln_stats", C.c_void_p), ("ln_colsum", C.c_void_p), ("row_part", C.c_void_p),
    ]


class ConvBnParams(C.Structure):
    _fields_ = [("w", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p)]


_P, _I, _F, _SZ = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); every symbol include/cvcl_hip.h declares
SIGNATURES = {
    "cvcl_abi_version": (_I, []),
    "cvcl_last_error": (C.c_char_p, []),
    "cvcl_prof_enable": (_I, [_I]),
    "cvcl_prof_collect": (_I, [_P, _P, _I]),
    "cvcl_prof_null_bracket_us": (_I, [_P, _I, _P]),
    "cvcl_embed_meanpool_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cvcl_embed_meanpool_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cvcl_l2norm_fwd": (_I, [_P, _P, _P, _I, _I, _F, _P]),
    "cvcl_l2norm_bwd": (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    "cvcl_sim_logits_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "cvcl_sim_logits_bwd_workspace_bytes": (_SZ, [_I, _I, _I]),
    "cvcl_sim_logits_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _SZ, _P]),
    "cvcl_sim_logits_bwd_rows": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "cvcl_infonce_workspace_bytes": (_SZ, [_I]),
    "cvcl_infonce_fwd": (_I, [_P, _I, _P, _P, _P, _P, _SZ, _P]),
    "cvcl_infonce_bwd": (_I, [_P, _P, _P, _P, _P, _I, _P]),
    "cvcl_row_entropy": (_I, [_P, _P, _I, _I, _P]),
    "cvcl_gemm_grid_m": (_I, [_I, _I, _I, _I]),
    "cvcl_gemm": (_I, [_I, C.POINTER(GemmArgs), _P]),
    "cvcl_transpose_f32": (_I, [_P, _P, _I, _I, _P]),
    "cvcl_colsum_f32": (_I, [_P, _P, _I, _I, _P]),
    "cvcl_bn_finalize": (_I, [_P, _I, C.c_long, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _I, _P]),
    "cvcl_bn_eval_affine": (_I, [_P, _P, _P, _P, _F, _P, _P, _P, _I, _P]),
    "cvcl_col_stats_rows": (_I, [C.c_long]),
    "cvcl_col_stats": (_I, [_I, _P, C.c_long, _I, _P, _I, _P]),
    "cvcl_packed_weight_bytes": (_SZ, [_I, _I, _I, _I, _I]),
    "cvcl_pack_conv_weight": (_I, [_I, _I, _P, _P, _I, _I, _I, _P]),
    "cvcl_stem_conv_stats_rows": (_I, [_I, _I, _I, _I]),
    "cvcl_stem_conv7x7": (_I, [_I, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "cvcl_stem_pool_supported": (_I, [_I, _I, _I]),
    "cvcl_stem_pool": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "cvcl_bn_relu_maxpool": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cvcl_gconv3x3_stats_rows": (_I, [_I, _I, _I, _I, _I, _I]),
    "cvcl_gconv3x3": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "cvcl_bn_add_relu": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, C.c_long, _I, _P]),
    "cvcl_bn_relu_apply": (_I, [_I, _P, _P, _P, _P, C.c_long, _I, _P]),
    "cvcl_avgpool": (_I, [_I, _P, _P, _I, _I, _I, _P]),
    "cvcl_im2col_patches": (_I, [_I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cvcl_vit_assemble_tokens": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _P]),
    "cvcl_layernorm": (_I, [_I, _P, C.c_long, _P, _P, _F, _P, _I, C.c_long, _I, _P]),
    "cvcl_attention": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "cvcl_embed_gather_pos": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cvcl_seq_sum_div": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "cvcl_lstm_cell": (_I, [_P, _P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "cvcl_dropout": (_I, [_P, _P, _P, C.c_long, _F, C.c_ulonglong, C.c_long, C.c_long, _P]),
    "cvcl_layernorm_bwd": (_I, [_P, _P, _P, _F, _P, _P, C.c_long, _I, _P]),
    "cvcl_relu_bwd": (_I, [_P, _P, _P, C.c_long, _P]),
    "cvcl_embed_rows_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "cvcl_seq_sum_div_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "cvcl_attention_small": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, C.c_ulonglong, _P]),
    "cvcl_lstm_cell_train": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "cvcl_lstm_cell_bwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _P]),
    "cvcl_bn_apply": (_I, [_I, _P, _P, _P, _P, C.c_long, _I, _I, _P]),
    "cvcl_bn_bwd_partial_rows": (_I, [_I, C.c_long, _I]),
    "cvcl_bn_bwd": (_I, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_long, _I, _P, _I, _P, _P]),
    "cvcl_bn_batch_moments": (_I, [_P, _I, C.c_long, C.c_float, _P, _P, _P, _I, _P]),
    "cvcl_gconv_weight_dgrad": (_I, [_P, _P, _I, _I, _P]),
    "cvcl_transpose": (_I, [_I, _P, _P, C.c_long, _I, _P]),
    "cvcl_add": (_I, [_I, _P, _P, _P, C.c_long, _I, _P]),
    "cvcl_relu_mask": (_I, [_I, _P, _P, _P, C.c_long, _P]),
    "cvcl_maxpool3x3s2": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cvcl_maxpool3x3s2_idx": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cvcl_avgpool_bwd": (_I, [_I, _P, _P, _I, _I, _I, _P]),
    "cvcl_zero_stuff2": (_I, [_I, _P, _P, _I, _I, _I, _I, _P]),
    "cvcl_conv_wgrad_direct": (_I, [_I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "cvcl_gemm_tn_workspace_bytes": (C.c_size_t, [_I, C.c_long, _I, _I]),
    "cvcl_gemm_tn": (_I, [_I, _P, _I, _P, _I, C.c_long, _I, _I, _P, _I, _P, C.c_size_t, _P]),
    "cvcl_gconv3x3_wgrad_workspace_bytes": (C.c_size_t, [_I, _I, _I, _I, _I]),
    "cvcl_gconv3x3_wgrad": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, C.c_size_t, _P]),
    "cvcl_stem_im2col": (_I, [_P, _P, _I, _I, _I, _P]),
    "cvcl_gemm_stats_rows": (_I, [_I, _P]),
    "cvcl_gemm_pro": (_I, [_P, _P]),
    "cvcl_gemm_pro_supported": (_I, [_P]),
    "cvcl_gemm_pro_stats_rows": (_I, [_I, _I]),
    "cvcl_gemm8w": (_I, [_I, _P, _P]),
    "cvcl_gemm8w_supported": (_I, [_I, _I, _I, _I, _I, _I]),
    "cvcl_gemm_ln_supported": (_I, [C.POINTER(GemmArgs)]),
    "cvcl_set_gemm_cu_share": (_I, [_I]),
    "cvcl_conv1x1_gram_workspace_bytes": (C.c_size_t, [_I]),
    "cvcl_conv1x1_gram": (_I, [_P, _I, C.c_long, _I, _P, _P, _I, _P, C.c_size_t, C.POINTER(C.c_void_p), _P]),
    "cvcl_bn_from_gram": (_I, [_P, _I, C.c_long, _P, _I, _I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _I, _P, _P]),
    "cvcl_row_stats": (_I, [_I, _P, C.c_long, _P, C.c_long, _I, _F, _P]),
    "cvcl_row_stats_finalize": (_I, [_P, _I, _P, C.c_long, _I, _F, _P]),
    "cvcl_gemm8w_tile_rows": (_I, [_I, _I]),
    "cvcl_gemm8w_stats_rows": (_I, [_I, _I]),
    "cvcl_bf16_to_f32": (_I, [_P, _P, C.c_long, _P]),
    "cvcl_f32_to_bf16": (_I, [_P, _P, C.c_long, _P]),
    "cvcl_spatial_max_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cvcl_spatial_max_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cvcl_lstm_add_dout": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "cvcl_seq_reverse": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "cvcl_scale_add_f32": (_I, [_P, _P, C.c_float, _P, C.c_long, _P]),
    "cvcl_augment_frames": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P]),
    "cvcl_cbow": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "cvcl_token_ce_fwd": (_I, [_P, _P, _P, _P, C.c_long, _I, _I, _P]),
    "cvcl_token_ce_bwd": (_I, [_P, _P, _P, _P, _P, C.c_long, _I, _I, _P]),
    "cvcl_lm_loss_summaries": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cvcl_quant_rows_fp8": (_I, [_I, _P, C.c_long, _P, _P, C.c_float, _P, _P, C.c_long, _I, _P]),
    "cvcl_attention_train": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "cvcl_attention_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "cvcl_layernorm_bwd_rows_partials": (_I, [C.c_long]),
    "cvcl_layernorm_bwd_rows": (_I, [_P, C.c_long, _P, _P, _I, C.c_long, _F, _P, _P, C.c_long, _P, C.c_long, _I, _P]),
    "cvcl_gelu_bf16": (_I, [_P, _P, _P, C.c_long, _P]),
    "cvcl_vit_tokens_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "cvcl_gemm_tn_colsum_workspace_bytes": (C.c_size_t, [C.c_long, _I, _I]),
    "cvcl_gemm_tn_colsum": (_I, [_P, _I, _P, _I, C.c_long, _I, _I, _P, _I, _P, _P, C.c_size_t, _P]),
    "cvcl_attention_mx": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "cvcl_gemm_fp8_mx": (_I, [_P, _P, _P, _I, _P, _P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P]),
    "cvcl_gemm_fp8_ex": (_I, [C.POINTER(GemmFp8Args), _P]),
    "cvcl_gemm_fp8_ln_supported": (_I, [_I, _I, _I]),
    "cvcl_quant_rows_mx": (_I, [_P, C.c_long, _P, _P, C.c_long, _I, _P]),
    "cvcl_gemm_fp8": (_I, [_P, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P]),
    "cvcl_resnext50_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "cvcl_resnext50_centres_floats": (_SZ, []),
    "cvcl_resnext50_fwd": (_I, [_I, _I, _I, _I, _I, _P, C.POINTER(ConvBnParams), _I, _P, _SZ, _P, _P, _F, _F, _P, _P]),
    "cvcl_resnext50_block_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I]),
    "cvcl_resnext50_block_fwd": (_I, [_I, _I, _I, _I, _I, _I, _I, _P, C.POINTER(ConvBnParams), _I, _P, _SZ, _P, _F, _F, _P, _P]),
    "cvcl_resnext50_moments_floats": (_SZ, []),
    "cvcl_resnext50_fwd_deferred_stats": (_I, [_I, _I, _I, _I, _P, C.POINTER(ConvBnParams), _I, _P, _SZ, _P, _P, _F, _P, _P, _P]),
    "cvcl_resnext50_apply_moments": (_I, [C.POINTER(ConvBnParams), _I, _P, _F, _P]),
}

_lib = None


def load(path: str | None = None):
    """Load the library (idempotent) and bind every declared symbol; raises CvclError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("CVCL_HIP_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise CvclError(f"libcvcl_hip.so not found at {path}: build it with `python multimodal-baby_amd/build.py` "
                        "(the CVCL hot path has no CPU fallback)")
    try:
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError as e:                                   # pragma: no cover
        raise CvclError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise CvclError(f"{path} does not export {name} (ABI mismatch)") from e
        fn.restype, fn.argtypes = res, args
    if lib.cvcl_abi_version() != ABI_VERSION:
        raise CvclError(f"ABI version {lib.cvcl_abi_version()} != {ABI_VERSION}")
    _lib = lib
    return lib


def lib():
    return load()


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().cvcl_last_error().decode(errors="replace")
        raise CvclError(f"{what} failed (rc={rc}): {msg}")


class TrunkStream:
    """Runs a frozen image trunk on its own HIP stream so that it overlaps the trainable tail of the PREVIOUS step
    (head GEMM, text encoder, loss, backward, optimizer, and in multi-GPU runs the feature all-gather / gradient all-reduce),
    which stay on the caller's stream.  The frozen trunk reads nothing the tail writes (its weights and BatchNorm buffers are
    touched on the trunk stream only), so the two streams need exactly one edge per step: the caller's stream waits for the
    trunk's event before it consumes the features.  With a host that runs ahead of the device (it does: the trunk is one
    enqueue), step k+1's trunk starts while step k's tail -- a few dozen latency-bound launches -- is still draining.

    ``n_streams=2``: consecutive steps alternate between two trunk streams, so step k+1's trunk also runs beside step k's
    TRUNK (consecutive passes of a frozen trunk are independent; ``fn`` must keep per-slot scratch and order whatever state
    the passes do share -- resnext.py chains the BatchNorm running-statistics updates with one event per pass).  Each pass
    fills the other's tail rounds, dependent-launch gaps and MFMA-bound phases: 6.40 -> 5.96 ms per ResNeXt-50 pass at B = 256.

    ``inputs='caller'``: the images were produced on the caller's stream; the trunk stream first waits for everything
    enqueued there (always correct, but it then also waits for the previous tail: no overlap).
    ``inputs='ready'``: the images are long-lived or were produced on the trunk stream itself (static benchmark batch;
    a data pipeline that runs its host-to-device copy and frame transform under ``with ts.context():``): no wait."""

    _pool = {}

    def __init__(self, device, inputs="caller", stream=None, n_streams=1):
        if inputs not in ("caller", "ready"):
            raise ValueError(inputs)
        self.device, self.inputs = torch.device(device), inputs
        if stream is not None:
            self.streams = [stream]
        else:
            # side streams are drawn from a per-device pool and REUSED by later TrunkStream objects: every new HIP stream is mapped
            # onto one of a few hardware queues, and a process that keeps creating streams (bench.py measures five configurations
            # in one process) ends up with two "parallel" trunk streams on one queue -- measured: the C4 sub-record of the default
            # line 13.06 ms against 12.26 ms for the same configuration run alone
            pool = TrunkStream._pool.setdefault((self.device.type, self.device.index), [])
            while len(pool) < max(1, int(n_streams)):
                pool.append(torch.cuda.Stream(device=self.device))
            self.streams = pool[:max(1, int(n_streams))]
        # launches so far: step k runs on stream k % n_streams (scratch is per stream: ``stream_index`` while fn runs) and writes
        # output slot k % n_slots.  One slot more than streams: with as many slots as streams, step k+2's trunk would have to wait
        # for step k's TAIL (the last reader of its slot) and each stream would idle for the length of a tail between its passes
        self._step = 0
        self.n_slots = len(self.streams) + 1
        self.stream_index = 0
        self._entries = []                                 # entry events of the last n_slots - 1 launches
        self._last_done = {}                               # stream index -> completion event of its latest pass

    @property
    def n_streams(self):
        return len(self.streams)

    @property
    def stream(self):
        """The stream the NEXT launch runs on (a pipeline that produces the batch there needs no extra edge)."""
        return self.streams[self._step % len(self.streams)]

    def context(self):
        return torch.cuda.stream(self.stream)

    def launch(self, fn, *inputs):
        """fn(slot) enqueues the trunk on the trunk stream and returns its output tensors; the caller's stream does NOT wait
        yet (``wait`` does), so work enqueued on it in between -- e.g. the deferred optimizer step of the previous batch --
        overlaps the trunk too.  slot cycles through n_slots persistent output buffer sets (a fresh allocation per step
        would rotate through allocator blocks: the caller's stream holds each one until its tail has run), so an output is
        valid until n_slots - 1 further steps have started -- the trunk stream waits, before reusing a slot, for the caller's
        stream to have passed the entry of the step after the slot's last reader, i.e. to have finished the tail that read
        it.  Scratch that must be private to a stream is keyed by ``self.stream_index`` (valid while fn runs)."""
        caller = torch.cuda.current_stream(self.device)
        stream = self.stream
        entry = torch.cuda.Event()
        entry.record(caller)                               # everything the caller enqueued for earlier steps precedes this
        # the last reader of this step's slot is the tail of step k - n_slots, which was enqueued before step k - n_slots + 1 entered
        free = self._entries[0] if len(self._entries) == self.n_slots - 1 else None
        self._entries = (self._entries + [entry])[-(self.n_slots - 1):]
        if self.inputs == "caller":
            stream.wait_stream(caller)
        elif free is not None:
            stream.wait_event(free)
        slot = self._step % self.n_slots
        self.stream_index = self._step % len(self.streams)
        self._step += 1
        with torch.cuda.stream(stream):
            outs = fn(slot)
            done = torch.cuda.Event()
            done.record(stream)
        self._last_done[self.stream_index] = done
        for t in inputs:                                   # allocated on the caller's pool, read on the trunk stream
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(stream)
        return outs, done

    def other_pass_in_flight(self) -> bool:
        """While ``fn`` runs (stream ``stream_index``): is a pass enqueued on ANOTHER trunk stream still unfinished?  True in a
        training loop (the host runs passes ahead of the device), False when every pass is awaited before the next starts
        (validation with a host sync per batch, host-bound steps)."""
        return any(i != self.stream_index and not ev.query() for i, ev in self._last_done.items())

    def wait(self, handle):
        outs, done = handle
        torch.cuda.current_stream(self.device).wait_event(done)
        return outs

    def run(self, fn, *inputs):
        return self.wait(self.launch(fn, *inputs))

    def join(self):
        """Make the caller's stream wait for everything on the trunk stream(s) (before reading BatchNorm buffers, saving)."""
        for s in self.streams:
            torch.cuda.current_stream(self.device).wait_stream(s)


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: torch.Tensor | None, dtype=None) -> int | None:
    """Device pointer of a contiguous CUDA(=HIP) tensor; loud failure otherwise."""
    if t is None:
        return None
    if not t.is_cuda:
        raise CvclError("the CVCL HIP path needs device tensors (got a CPU tensor); there is no CPU fallback")
    if not t.is_contiguous():
        raise CvclError("non-contiguous tensor passed to libcvcl_hip")
    if dtype is not None and t.dtype != dtype:
        raise CvclError(f"expected dtype {dtype}, got {t.dtype}")
    return t.data_ptr()


def torch_dtype(dt: int):
    return torch.float32 if dt == F32 else torch.bfloat16


def cvcl_dtype(t: torch.dtype) -> int:
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise CvclError(f"unsupported storage dtype {t}")


def gemm(A, W, out=None, *, bias=None, act=ACT_NONE, residual=None, a_scale=None, a_shift=None, a_relu=False,
         exp_scale=None, gather=None, stats=None, M=None, lda=None, pre_out=None, gelu_grad_of=None, centre=None,
         ln_stats=None, ln_colsum=None, row_part=None, query_ln=False, a_trans=False, w_trans=False, a_rowsum=None,
         split=False, stats_acc=None):
    """C = act(A' W^T * exp(*exp_scale) + bias) (+ residual).  A [M,K], W [N,K] row-major, same dtype.
    ``centre`` [N] f32 (convolution epilogues): C = round(A' W^T - centre), statistics of that (cvcl_hip.h "Centred storage").
    fp32 only: ``a_trans`` -- A is given as [K, M]; ``w_trans`` -- W is given as [K, N] (the operands of a gradient GEMM as they
    lie, no transposed copies); ``a_rowsum`` [M] f32 (with a_trans) receives sum_k A'[m][k] (the bias gradient beside dW)."""
    dt = cvcl_dtype(A.dtype)
    if W.dtype != A.dtype:
        raise CvclError("gemm operands must share a dtype")
    K, N = (W.shape[0], W.shape[1]) if w_trans else (W.shape[1], W.shape[0])
    ldw = W.shape[1]
    if a_trans:
        if A.dim() != 2 or A.shape[0] != K or M is not None or lda is not None:
            raise CvclError("a_trans: A must be a [K, M] matrix")
        M, lda = A.shape[1], A.shape[1]
    if M is None:
        M = A.numel() // A.shape[-1]
    lda = lda if lda is not None else A.shape[-1]
    if out is None:
        out = torch.empty((M, N), dtype=A.dtype, device=A.device)
    a = GemmArgs()
    a.A, a.W, a.C = ptr(A), ptr(W), ptr(out)
    a.M, a.N, a.K, a.lda, a.ldw, a.ldc = M, N, K, lda, ldw, N
    a.a_trans, a.w_trans, a.a_rowsum = int(a_trans), int(w_trans), ptr(a_rowsum, torch.float32)
    a.f32_split = int(bool(split) and dt == F32)           # (fp32 operands on the bf16 MFMA, hi / lo split: cvcl_hip.h)
    a.a_scale, a.a_shift, a.a_relu = ptr(a_scale, torch.float32), ptr(a_shift, torch.float32), int(a_relu)
    if gather is not None:
        a.gather_ho, a.gather_wo, a.gather_hi, a.gather_wi, a.gather_stride = gather
    a.exp_scale, a.bias, a.act = ptr(exp_scale, torch.float32), ptr(bias, torch.float32), act
    if residual is not None:
        if residual.dtype != A.dtype:
            raise CvclError("residual dtype mismatch")
        a.R, a.ldr = ptr(residual), N
    if stats is not None:
        a.stats, a.stats_rows = ptr(stats, torch.float32), stats.shape[0]
    if stats_acc is not None:                             # int64 [8, 2, N], zeroed by the caller: statistics ACCUMULATED (cvcl_hip.h)
        if stats is not None or stats_acc.dtype != torch.int64 or tuple(stats_acc.shape) != (8, 2, N) or not stats_acc.is_contiguous():
            raise CvclError("stats_acc: a contiguous int64 [8, 2, N] accumulator (and no stats rows)")
        a.stats, a.stats_rows = stats_acc.data_ptr(), STATS_ACCUMULATE
    if pre_out is not None:                               # act = GELU: also keep the pre-activation
        a.C_pre = ptr(pre_out, A.dtype)
    if gelu_grad_of is not None:                          # C = (A W^T) * gelu'(gelu_grad_of)
        a.G, a.ldg = ptr(gelu_grad_of, A.dtype), N
    a.centre = ptr(centre, torch.float32)
    # LayerNorm folded into the linear (cvcl_hip.h): consumer (ln_stats [M + (M & 1), 2], ln_colsum [N], bias = folded) / producer (row_part)
    a.ln_stats, a.ln_colsum, a.row_part = ptr(ln_stats, torch.float32), ptr(ln_colsum, torch.float32), ptr(row_part, torch.float32)
    if query_ln:                                          # would cvcl_gemm honour ln_stats / row_part for these arguments?
        return bool(lib().cvcl_gemm_ln_supported(C.byref(a)))
    check(lib().cvcl_gemm(dt, C.byref(a), stream_ptr()), "cvcl_gemm")
    return out


STATS_ACCUMULATE = -1          # cvcl_hip.h CVCL_STATS_ACCUMULATE


def gemm_grid_m(dtype: int, M: int, N: int, has_prologue: bool = False) -> int:
    return lib().cvcl_gemm_grid_m(dtype, M, N, int(has_prologue))


def gemm_stats_rows(dtype: int, M: int, N: int, K: int, gather=None, *, prologue=False, a_relu=False, bias=False, residual=False,
                    act=ACT_NONE) -> int:
    """BN-statistics rows ``gemm(..., stats=...)`` writes for an [M, K] x [N, K]^T product of this dtype with these options
    (which kernel the dispatcher picks decides: cvcl_gemm_stats_rows).  Size the statistics buffer by this, reduce exactly
    this many rows."""
    a = GemmArgs()
    a.M, a.N, a.K, a.lda, a.ldw, a.ldc = M, N, K, K, K, N
    if gather is not None:
        a.gather_ho, a.gather_wo, a.gather_hi, a.gather_wi, a.gather_stride = gather
    dummy = 16                                             # a non-null, 16-byte aligned stand-in: only null-ness / alignment is inspected
    if prologue:
        a.a_scale, a.a_shift, a.a_relu = dummy, dummy, int(a_relu)
    if bias:
        a.bias = dummy
    if residual:
        a.R, a.ldr = dummy, N
    a.act = act
    a.A, a.W, a.C = dummy, dummy, dummy
    return lib().cvcl_gemm_stats_rows(dtype, C.byref(a))


def prof_enable(on: bool):
    check(lib().cvcl_prof_enable(int(on)), "cvcl_prof_enable")


def prof_null_bracket_us(n: int = 256) -> float:
    """Event bracket of a kernel that does nothing (us): what every event-timed launch carries on top of its kernel."""
    v = C.c_double(0.0)
    check(lib().cvcl_prof_null_bracket_us(stream_ptr(), n, C.byref(v)), "cvcl_prof_null_bracket_us")
    return float(v.value)


def prof_collect():
    """-> {class_name: (total_ms, launches)} for the launches recorded since prof_enable(True)."""
    n = len(KERNEL_CLASSES)
    ms = (C.c_double * n)()
    cnt = (C.c_long * n)()
    check(lib().cvcl_prof_collect(ms, cnt, n), "cvcl_prof_collect")
    return {KERNEL_CLASSES[i]: (ms[i], cnt[i]) for i in range(n)}
