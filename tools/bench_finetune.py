"""Informational (not the bench line): C2 with --finetune_cnn (the reference's *_finetune_cnn runner configs):
the whole ResNeXt-50 trains, forward + backward through multimodal/trunk_train.py."""
import argparse, contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-baby_amd"))
import torch
from multimodal import _hip as H
from multimodal.multimodal import TextEncoder, VisionEncoder
from multimodal.multimodal_data_module import read_vocab
from multimodal.multimodal_lit import MultiModalLitModel
sys.path.insert(0, ROOT)
from bench import c2_args, synthetic_batch_on_device

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=64); ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--dtype", default="bf16"); a = ap.parse_args()
dev = torch.device("cuda:0"); torch.manual_seed(0)
args = c2_args(); args.finetune_cnn = True
with contextlib.redirect_stdout(io.StringIO()):
    ve = VisionEncoder(args); te = TextEncoder(read_vocab(), ve.last_cnn_out_dim, args); lit = MultiModalLitModel(ve, te, args)
lit.to(dev); lit.set_precision(a.dtype); lit.train()
opt = lit.configure_optimizers()
opt = opt["optimizer"] if isinstance(opt, dict) else opt
batch = synthetic_batch_on_device(a.batch, 0, dev) + (None,)
def step():
    opt.zero_grad(set_to_none=True); out = lit.training_step(batch, 0); out["loss"].backward(); opt.step(); return out
out = step(); out = step(); out = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): out = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
H.prof_enable(True); step(); torch.cuda.synchronize(); prof = H.prof_collect(); H.prof_enable(False)
print(f"C2 --finetune_cnn B={a.batch} {a.dtype}: {dt*1e3:.1f} ms/step, {a.batch/dt:.0f} pairs/s, loss {float(out['loss']):.4f}, "
      f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
print({k: round(v[0], 2) for k, v in prof.items() if v[1] > 0})
if os.environ.get("FT_CPROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3): step()
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
