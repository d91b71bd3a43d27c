"""The trainable tail of a step on its own (round 5): the frozen image trunk is replaced by its cached output
(bench.frozen_trunk_cached), so what runs is fc / head, the text encoder forward + backward, L2 normalise, similarity logits,
InfoNCE forward + backward and AdamW -- event-timed per step, with the library's per-class launch counts.  Under
`rocprofv3 --kernel-trace --stats -- python3 tools/tail_bench.py <cfg>` the kernel list is the tail's complete launch list
(torch's own kernels included): total dispatches / steps = launches per step.
    python tools/tail_bench.py c2|c4 [steps]"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multimodal-baby_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from multimodal import _hip as H  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
lit, ve, opt = bench.build_model(cfg, dev, "bf16")
batch = bench.synthetic_batch_on_device(256, seed=0, device=dev) + (None,)


def step():
    opt.zero_grad(set_to_none=True)
    out = lit.training_step(batch, 0)
    out["loss"].backward()
    opt.step()
    return out


with bench.frozen_trunk_cached(ve, cfg, batch[0]):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    H.prof_enable(True)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    prof = H.prof_collect()
    H.prof_enable(False)
print(json.dumps({"config": cfg, "tail_ms_per_step": round(ms, 4),
                  "library_launches_per_step": {k: v[1] / steps for k, v in prof.items() if v[1]},
                  "library_event_ms_per_step": {k: round(v[0] / steps, 4) for k, v in prof.items() if v[1]}}))
