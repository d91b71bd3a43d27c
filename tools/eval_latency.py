"""Latency of the evaluation callers (batch-1..4 `encode_image`, the 4-way trial of eval.py:196-232) in eval mode,
eager launches against a captured HIP graph of the same launches.  Timing only; prints one line per case."""
import os, sys, time
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "multimodal-baby_amd"))
import bench

dev = torch.device("cuda:0")
N = int(os.environ.get("N", 200))


def lat(fn, n=N):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3, ts[int(len(ts) * 0.95)] * 1e3


def thr(fn, n=N):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for cfg, prec in (("c2", "bf16"), ("c2", "fp32"), ("c4", "bf16")):
    lit, ve, _ = bench.build_model(cfg, dev, precision=prec)
    lit.eval()
    for B in (1, 4, 64):
        img, tok, ln = bench.synthetic_batch_on_device(B, 0, dev)
        with torch.no_grad():
            def enc():
                return lit.model.encode_image(img)[0]
            def trial():                                   # one 4-way trial: B images against one label
                return lit(img, tok[:1], ln[:1])[1]
            ref = enc().clone()
            e_med, e_p95 = lat(enc)
            e_thr = thr(enc)
            t_med, _ = lat(trial)
            line = f"{cfg} {prec} B={B}: encode_image eager {e_med:.3f} ms median ({e_p95:.3f} p95), back-to-back {e_thr:.3f} ms; trial eager {t_med:.3f} ms"
            try:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    for _ in range(3):
                        enc()
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = enc()
                g.replay(); torch.cuda.synchronize()
                same = bool(torch.equal(out, ref))
                g_med, g_p95 = lat(g.replay)
                g_thr = thr(g.replay)
                line += f" | graph {g_med:.3f} ms median ({g_p95:.3f} p95), back-to-back {g_thr:.3f} ms, identical={same}"
            except Exception as e:                          # noqa: BLE001
                line += f" | graph capture failed: {type(e).__name__}: {str(e)[:200]}"
        print(line, flush=True)
    del lit, ve
