"""Is the host ahead of the device in the C2 step?  Prints the host-side enqueue time of K steps next to the synchronised time,
and per-phase host times (training_step / backward / optimizer) with and without a device sync after each phase."""
import os, sys, time, io, contextlib
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multimodal-baby_amd"))
import bench
from multimodal.multimodal import TextEncoder, VisionEncoder
from multimodal.multimodal_data_module import read_vocab
from multimodal.multimodal_lit import MultiModalLitModel
dev = torch.device("cuda:0")
torch.manual_seed(0)
args = bench.c2_args()
with contextlib.redirect_stdout(io.StringIO()):
    ve = VisionEncoder(args); te = TextEncoder(read_vocab(), ve.last_cnn_out_dim, args); lit = MultiModalLitModel(ve, te, args)
lit.to(dev); lit.set_precision("bf16"); lit.train()
opt = lit.configure_optimizers()
batch = bench.synthetic_batch_on_device(256, seed=0, device=dev) + (None,)
mode = sys.argv[1] if len(sys.argv) > 1 else ""
if mode == "stream":
    ve.model.enable_trunk_stream(dev, inputs="ready")
if mode == "swap":       # trunk on the default stream, everything else on a side stream
    ve.model.enable_trunk_stream(dev, inputs="ready", stream=torch.cuda.default_stream(dev))
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    torch.cuda.set_stream(side)
def step(sync=False, acc=None):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    out = lit.training_step(batch, 0)
    if sync: torch.cuda.synchronize()
    t1 = time.perf_counter()
    out["loss"].backward()
    if sync: torch.cuda.synchronize()
    t2 = time.perf_counter()
    opt.step()
    if sync: torch.cuda.synchronize()
    t3 = time.perf_counter()
    if acc is not None:
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2
for _ in range(5): step()
torch.cuda.synchronize()
K = 20
acc = [0, 0, 0]
t0 = time.perf_counter()
for _ in range(K): step(acc=acc)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue {t_host / K * 1e3:.2f} ms/step, synchronised {t_all / K * 1e3:.2f} ms/step; host phases (no sync): "
      f"training_step {acc[0] / K * 1e3:.2f} backward {acc[1] / K * 1e3:.2f} optimizer {acc[2] / K * 1e3:.2f} ms")
acc = [0, 0, 0]
for _ in range(K): step(sync=True, acc=acc)
print(f"with a sync after each phase: training_step {acc[0] / K * 1e3:.2f} backward {acc[1] / K * 1e3:.2f} optimizer {acc[2] / K * 1e3:.2f} ms")
# trunk alone (same stream mode), back to back
x = batch[0]
with torch.no_grad():
    for _ in range(3): ve.model.trunk(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): ve.model.trunk(x)
    torch.cuda.synchronize()
    print(f"trunk alone: {(time.perf_counter() - t0) / K * 1e3:.2f} ms")
