"""Informational: inference latency / throughput of encode_image + encode_text + logits (eval mode: BatchNorm on running
statistics, no autograd) -- the shape of the reference's evaluation callers (eval.py 4-way trials, feature extraction)."""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-baby_amd")); sys.path.insert(0, ROOT)
import torch
from multimodal.multimodal import TextEncoder, VisionEncoder
from multimodal.multimodal_data_module import read_vocab
from multimodal.multimodal_lit import MultiModalLitModel
from bench import c2_args, synthetic_batch_on_device
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(io.StringIO()):
    args = c2_args(); ve = VisionEncoder(args); te = TextEncoder(read_vocab(), ve.last_cnn_out_dim, args)
    lit = MultiModalLitModel(ve, te, args)
lit.to(dev); lit.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16"); lit.eval()
n_streams = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # > 0: consecutive batches' trunk passes pipelined on that many HIP streams
if n_streams:                                                 # (throughput of a feature-extraction loop, not the latency of one call)
    torch.cuda.synchronize(); ve.model.enable_trunk_stream(dev, inputs="ready", n_streams=n_streams)
for B in (1, 4, 16, 64, 256):
    x, y, yl = synthetic_batch_on_device(B, 0, dev)
    with torch.no_grad():
        for _ in range(5): lit.model(x, y, yl)
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 30 if B <= 64 else 10
        for _ in range(n): out = lit.model(x, y, yl)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"eval forward B={B} trunk streams={n_streams}: {dt*1e3:.3f} ms  ({B/dt:.0f} images/s)")
