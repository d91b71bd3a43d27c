"""Informational (not the bench line): train-step throughput of BASELINE configs[3] on one GPU --
frozen DINO ViT-B/16 (bf16 MFMA) + trainable one-layer transformer text encoder (fp32, dropout 0.1) + head."""
import argparse, contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-baby_amd"))
import torch
import multimodal.multimodal as mm
from multimodal import _hip as H
from multimodal import vision_transformer_dino_mugs as vits
from multimodal.multimodal import TextEncoder, VisionEncoder
from multimodal.multimodal_data_module import read_vocab
from multimodal.multimodal_lit import MultiModalLitModel
sys.path.insert(0, ROOT)
from bench import synthetic_batch_on_device

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=256); ap.add_argument("--patch", type=int, default=16)
ap.add_argument("--steps", type=int, default=10); ap.add_argument("--precision", default="bf16")
ap.add_argument("--trunk-stream", action="store_true"); ap.add_argument("--trunk-streams", type=int, default=1); ap.add_argument("--finetune", action="store_true"); a = ap.parse_args()
dev = torch.device("cuda:0"); torch.manual_seed(0)
args = argparse.Namespace(embedding_type="flat", embedding_dim=512, pretrained_cnn=False, cnn_dino=False, vit_dino=True,
                          finetune_cnn=a.finetune, text_encoder="transformer", crange=1, dropout_i=0.0, dropout_o=0.0,
                          pos_embed_type="learned", normalize_features=True, sim="max", temperature=0.07, fix_temperature=True,
                          tie=True, bias=True, optimizer=torch.optim.AdamW, lr=1e-4, weight_decay=0.1, lr_scheduler=False,
                          lambda_mm=1.0, lambda_lm=0.0, lambda_ar=0.0, optimize_unused=True)
orig = mm.load_model
mm.load_model = lambda name, pretrained: vits.vit_base(patch_size=a.patch, num_classes=0)
with contextlib.redirect_stdout(io.StringIO()):
    ve = VisionEncoder(args); te = TextEncoder(read_vocab(), 768, args); lit = MultiModalLitModel(ve, te, args)
mm.load_model = orig
lit.to(dev); lit.set_precision(a.precision); lit.train()
opt = lit.configure_optimizers()
if a.trunk_stream:
    from multimodal import vit_hip
    vit_hip.enable_trunk_stream(ve.model, dev, inputs="ready", n_streams=a.trunk_streams)
batch = synthetic_batch_on_device(a.batch, 0, dev) + (None,)
def step():
    opt.zero_grad(set_to_none=True); out = lit.training_step(batch, 0); out["loss"].backward(); opt.step(); return out
for _ in range(3): out = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): out = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
H.prof_enable(True)
for _ in range(3): step()
torch.cuda.synchronize(); prof = H.prof_collect(); H.prof_enable(False)
flops = {16: 35.1e9, 14: 46.3e9}[a.patch] * a.batch
print(f"{'C5' if a.precision == 'fp8' else 'C4'} ViT-B/{a.patch} + transformer text, B={a.batch}, {a.precision}: {dt*1e3:.2f} ms/step, {a.batch/dt:.0f} pairs/s, "
      f"{flops/dt/1e12:.0f} TFLOP/s fwd-equivalent, loss {float(out['loss']):.4f}")
print({k: round(v[0] / 3, 3) for k, v in prof.items() if v[1] > 0})
