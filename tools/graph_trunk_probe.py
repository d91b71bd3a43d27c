"""Probe (round 4): does replaying the train-mode ResNeXt trunk pass as a captured HIP graph shorten it?  Eager back-to-back passes vs
graph replays on one stream, and two graphs alternating on two streams (the bench's two-trunk-stream schedule).  Timing only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-baby_amd"))
import torch
import bench
dev = torch.device("cuda:0")
lit, ve, _ = bench.build_model("c2", dev, "bf16")
m = ve.model
x = bench.synthetic_batch_on_device(256, 0, dev)[0]
N = 30
def timed(fn, n=N):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    eager = timed(lambda: m.trunk(x))
    print(f"eager, one stream: {eager:.3f} ms per pass")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        m.trunk(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m.trunk(x)
    rep = timed(g.replay)
    print(f"graph replay, one stream: {rep:.3f} ms per pass")
    # two streams, eager (TrunkStream schedule) vs two graphs
    ts = m.enable_trunk_stream(dev, inputs="ready", n_streams=2)
    two = timed(lambda: m.trunk(x))
    print(f"eager, two trunk streams: {two:.3f} ms per pass")
