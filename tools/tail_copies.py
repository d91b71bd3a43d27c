"""Where do the device-to-device copies (__amd_rocclr_copyBuffer) and small torch kernels of a step's trainable tail come from?
torch.profiler over three tail-only steps (bench.frozen_trunk_cached): every aten::copy_ / clone / contiguous / to / fill_ / zero_
with its Python call stack, grouped.    python tools/tail_copies.py c2|c4"""
import collections
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multimodal-baby_amd"))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
dev = torch.device("cuda:0")
lit, ve, opt = bench.build_model(cfg, dev, "bf16")
batch = bench.synthetic_batch_on_device(256, seed=0, device=dev) + (None,)


def step():
    opt.zero_grad(set_to_none=True)
    out = lit.training_step(batch, 0)
    out["loss"].backward()
    opt.step()


with bench.frozen_trunk_cached(ve, cfg, batch[0]):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
names = ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::zeros", "aten::empty_like",
         "aten::mul", "aten::add", "aten::neg", "aten::exp", "aten::ones_like")
groups = collections.Counter()
for ev in prof.events():
    if ev.name in names:
        stack = [f for f in (ev.stack or []) if "multimodal" in f or "bench.py" in f or "optim" in f or "autograd" in f][:3]
        groups[(ev.name, str(ev.input_shapes)[:60], " <- ".join(s.split("/")[-1][:70] for s in stack))] += 1
for (name, shapes, stack), n in sorted(groups.items(), key=lambda kv: -kv[1]):
    print(f"{n / 3:5.1f}/step  {name:18s} {shapes:60s} {stack}")

print("---- device activities per step (kernels / memcpy / memset), by name ----")
dev_ev = collections.Counter()
for ev in prof.events():
    if str(ev.device_type).endswith("CUDA"):
        dev_ev[ev.name[:100]] += 1
for name, n in sorted(dev_ev.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n / 3:5.1f}/step  {name}")
print("---- CPU ops that launched a Memcpy ----")
try:
    ka = prof.profiler.kineto_results.events()
    byid = {}
    for e in ka:
        if e.device_type().name == "CPU":
            byid.setdefault(e.correlation_id(), []).append(e.name())
    c = collections.Counter()
    for e in ka:
        if e.device_type().name != "CPU" and "emcpy" in e.name():
            c[(e.name(), tuple(byid.get(e.correlation_id(), []))[:3])] += 1
    for (n, ops), k in sorted(c.items(), key=lambda kv: -kv[1]):
        print(f"{k / 3:5.1f}/step  {n}  <- {ops}")
except Exception as e:
    print("kineto correlation not available:", repr(e))
