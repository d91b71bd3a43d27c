"""CVCL contrastive train-step benchmark on MI355X (driver contract: see the task prompt / DESIGN.md).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic input already resident in HBM: ResNeXt-50
forward (bf16 MFMA trunk, BN in train mode), embedding mean-pool, L2 normalise, [RCCL feature all-gather],
similarity logits, symmetric InfoNCE, backward of the trainable set (fc + embedding), [RCCL gradient
all-reduce], AdamW.  Workload = BASELINE.json configs[1] (C2), weak scaling (256 pairs per GPU).

Rank 0 prints ONE JSON line with the throughput, the roofline of the dominant kernel (timed live with
HIP events on the launch stream in a second pass of the same steps, so the events do not perturb the
headline number) and a CPU baseline (the oracle restatement on the host cores, bounded sample).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "multimodal-baby_amd"))

import torch                                   # noqa: E402
import torch.distributed as dist               # noqa: E402

METRIC = "image-text pairs/sec, CVCL ResNeXt+embed 224², bs256, 1/2/4/8 MI355X"
PER_GPU_BATCH = 256
EMBEDDING_DIM = 512
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16


def c2_args():
    import types
    return argparse.Namespace(
        embedding_type="flat", embedding_dim=EMBEDDING_DIM, pretrained_cnn=False, cnn_model="resnext50_32x4d",
        cnn_dino=False, vit_dino=False, finetune_cnn=False, text_encoder="embedding", captioning=False,
        attention=False, attention_gate=False, crange=1, dropout_i=0.5, dropout_o=0.0, pos_embed_type="no_pos_embed",
        normalize_features=True, sim="max", temperature=0.07, fix_temperature=True, tie=True, bias=True,
        optimizer=torch.optim.AdamW, lr=1e-4, weight_decay=0.1, lr_scheduler=False, lambda_mm=1.0, lambda_lm=0.0,
        lambda_ar=0.0, optimize_unused=True, local_negatives=False)


def synthetic_batch_on_device(batch, seed, device, vocab=2350):
    """rand -> ImageNet normalise; <sos> w1 w2 w3 <eos>, generated once on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    img = torch.rand(batch, 3, 224, 224, generator=g, device=device)
    mean = torch.tensor([0.485, 0.456, 0.406], device=device).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device=device).view(1, 3, 1, 1)
    img = ((img - mean) / std).contiguous()
    words = torch.randint(4, vocab, (batch, 3), generator=g, device=device)
    tok = torch.cat([torch.full((batch, 1), 2, device=device), words, torch.full((batch, 1), 3, device=device)], 1).long()
    ln = torch.full((batch,), 5, dtype=torch.long, device=device)
    return img, tok.contiguous(), ln


def gemm_algorithmic_work(B):
    """Algorithmic bytes / flops of the bf16 conv GEMM launches of one ResNeXt-50 forward at batch B (train mode):
    every operand element moved once, 2 bytes each; 2*M*N*K flops per launch.  Launch list = what cvcl_resnext50_fwd
    enqueues: conv1 and downsample (A + W + C); conv3 of layers 2-4 (A + W + C); conv3 of layer 1 twice -- a
    statistics-only pass (A + W) and the fused BN3 + identity + ReLU pass (A + W + residual + C).  (The number of leading
    stages that use the fused tail is the library's $CVCL_FUSED_TAIL_STAGES, default 1.)"""
    fused_stages = int(os.environ.get("CVCL_FUSED_TAIL_STAGES", "1"))
    nbytes = flops = launches = 0
    inplanes, h = 64, 56
    for stage, blocks in enumerate((3, 4, 6, 3)):
        planes = 64 << stage
        width, outc = planes * 2, planes * 4
        for bi in range(blocks):
            stride = 2 if (stage > 0 and bi == 0) else 1
            ho = h // stride
            m_in, m_out = B * h * h, B * ho * ho
            plain = [(m_in, width, inplanes)]                       # conv1
            if bi == 0:
                plain.append((m_out, outc, inplanes))               # downsample
            if stage >= fused_stages:
                plain.append((m_out, outc, width))                  # conv3, raw output materialised
            for (m, n, k) in plain:
                nbytes += 2 * (m * k + n * k + m * n)
                flops += 2 * m * n * k
                launches += 1
            if stage < fused_stages:                                # conv3 as statistics pass + fused tail pass
                m, n, k = m_out, outc, width
                nbytes += 2 * (m * k + n * k) + 2 * (m * k + n * k + 2 * m * n)
                flops += 2 * (2 * m * n * k)
                launches += 2
            h, inplanes = ho, outc
    return nbytes, flops, launches


def cpu_baseline(sample_batch=64, steps=2):
    """The oracle (CPU restatement of the reference step: fwd + InfoNCE + bwd(trainable) + AdamW) timed on the
    host cores on a bounded sample of the same workload: `steps` steps at batch `sample_batch`."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cvcl_oracle as O
    threads = torch.get_num_threads()
    p = O.cvcl_random_params(EMBEDDING_DIM, seed=0)
    step = O.CpuTrainStep(p, lr=1e-4, weight_decay=0.1, normalize_features=True)
    img, tok, ln = O.synthetic_batch(sample_batch, seed=0)
    step.step(img, tok, ln)                                   # untimed warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        step.step(img, tok, ln)
    dt = time.perf_counter() - t0
    return {"value": round(sample_batch * steps / dt, 2), "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"{steps} train steps at batch {sample_batch} (same per-pair work as the batch-256 step: "
                      f"ResNeXt-50 fwd with train-mode BN + embedding + InfoNCE + bwd + AdamW), fp32, torch "
                      f"{torch.__version__} CPU, {threads} threads of {os.cpu_count()} cpus, {dt:.1f}s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "32"])
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and not (a.gpus == 1 and world == 1):
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with --nproc-per-node {a.gpus} (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the CVCL hot path has no CPU fallback")
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; CVCL_DIST_BACKEND=gloo lets two ranks share one GPU to smoke-test the N>1 path
        dist.init_process_group(backend=os.environ.get("CVCL_DIST_BACKEND", "nccl"))

    from multimodal import _hip as H
    from multimodal import parallel
    from multimodal.multimodal import TextEncoder, VisionEncoder
    from multimodal.multimodal_data_module import read_vocab
    from multimodal.multimodal_lit import MultiModalLitModel

    torch.manual_seed(0)
    args = c2_args()
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        ve = VisionEncoder(args)
        te = TextEncoder(read_vocab(), ve.last_cnn_out_dim, args)
        lit = MultiModalLitModel(ve, te, args)
    lit.to(device)
    lit.set_precision(a.precision)
    lit.train()                                           # Lightning keeps .train(): BN uses batch statistics
    opt = lit.configure_optimizers()
    engine = parallel.DataParallelEngine(device, global_negatives=True).attach(lit)
    batch = synthetic_batch_on_device(PER_GPU_BATCH, seed=rank, device=device) + (None,)
    # the frozen trunk runs on its own HIP stream (H.TrunkStream; CVCL_TRUNK_STREAM=0 restores the single-stream schedule):
    # step k+1's trunk overlaps step k's trainable tail -- fc, text, loss, backward, AdamW, and with world > 1 the feature
    # all-gathers, the larger global-negatives loss and the deferred all-reduce wait + optimizer step -- which stays on the
    # main stream.  Every step does the same work with the same numbers (bit-identical, tests/test_train_entry_gpu.py), and
    # all of it is complete when the clock stops (torch.cuda.synchronize() waits for both streams).  One GPU: 7.42 -> 7.38
    # ms/step (the tail is only 0.26 ms there and the cross-stream events cost 0.1 ms).
    # With two trunk streams ($CVCL_TRUNK_STREAMS, default 2) consecutive trunk passes -- independent for a frozen trunk except
    # for the BatchNorm running statistics, which are still updated in step order -- also overlap each other: each fills the
    # other's tail rounds, dependent-launch gaps and MFMA-bound phases (6.40 -> 5.96 ms per pass).
    if os.environ.get("CVCL_TRUNK_STREAM", "1") != "0":
        torch.cuda.synchronize()
        ve.model.enable_trunk_stream(device, inputs="ready")          # the benchmark batch is resident and never rewritten

    # multi-GPU: the all-reduce + optimizer step of step k are enqueued behind the frozen trunk of step k+1
    # (parallel.OverlappedUpdate; same parameter sequence as the sequential schedule); flushed before the clock stops
    upd = parallel.OverlappedUpdate(engine, opt, ve) if world > 1 else None

    def step():
        if upd is None:
            opt.zero_grad(set_to_none=True)
            out = lit.training_step(batch, 0)
            out["loss"].backward()
            engine.reduce_gradients()
            opt.step()
            return out
        out = lit.training_step(batch, 0)          # trunk, then (hook) the previous step's update, then fc / text / loss
        upd.zero_grad()
        out["loss"].backward()
        upd.step_done()
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Schedule choice with more than one rank (untimed, before the warm-up): the two-trunk-stream schedule has only been
    # measured on one GPU; with RCCL's own streams in the process a trunk stream could end up sharing a hardware queue with a
    # collective.  So, unless $CVCL_TRUNK_STREAMS pins it, both schedules run a few steps and every rank keeps the faster one
    # (the slowest rank's time decides, all ranks agree through an all-reduce).  Same numbers either way.
    ts0 = ve.model.__dict__.get("_trunk_stream")
    trunk_streams = ts0.n_streams if ts0 is not None else 0
    if world > 1 and ts0 is not None and "CVCL_TRUNK_STREAMS" not in os.environ:
        trial = {}
        for n in (2, 1):
            torch.cuda.synchronize()
            ve.model.enable_trunk_stream(device, inputs="ready", n_streams=n)
            for _ in range(3):
                step()
            upd.flush()
            barrier()
            t0 = time.perf_counter()
            for _ in range(8):
                step()
            upd.flush()
            barrier()
            t = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            trial[n] = float(t.item())
        trunk_streams = 2 if trial[2] <= trial[1] else 1
        torch.cuda.synchronize()
        ve.model.enable_trunk_stream(device, inputs="ready", n_streams=trunk_streams)

    for _ in range(a.warmup):
        out = step()
    if upd is not None:
        upd.flush()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    if upd is not None:
        upd.flush()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(out["loss"].detach())
    pairs = world * PER_GPU_BATCH * a.steps
    value = pairs / elapsed

    roofline = None
    breakdown = None
    if not a.no_roofline:
        # further passes of the same steps with HIP events around every launch (on the launch stream).  The roofline figures come
        # from a pass with ONE trunk pass in flight: with two trunk streams a launch shares the CUs and HBM with the other pass's
        # kernels and its event-timed duration is no longer the kernel's own (in the rocprofv3 --kernel-trace run of this command
        # the per-kernel averages come out as the one-at-a-time durations too: profiles/r01_bench_c2_kernel_stats.csv); the durations seen under the
        # two-stream schedule are reported beside them (roofline.concurrent).
        def instrumented(n):
            for _ in range(2):
                step()
            if upd is not None:
                upd.flush()
            torch.cuda.synchronize()
            H.prof_enable(True)
            for _ in range(n):
                step()
            if upd is not None:
                upd.flush()
            torch.cuda.synchronize()
            out_ = H.prof_collect()
            H.prof_enable(False)
            return out_

        nprof = min(a.steps, 10)
        ts_now = ve.model.__dict__.get("_trunk_stream")
        conc, nconc = None, min(a.steps, 5)
        if ts_now is not None and ts_now.n_streams > 1:
            conc = instrumented(nconc)
            ve.model.enable_trunk_stream(device, inputs="ready", n_streams=1)
        prof = instrumented(nprof)
        breakdown = {k: round(v[0] / nprof, 4) for k, v in prof.items() if v[1] > 0}
        dom = max(prof.items(), key=lambda kv: kv[1][0])[0]
        nbytes, flops, launches = gemm_algorithmic_work(PER_GPU_BATCH)
        g_ms, g_n = prof["gemm"]
        avg_s = g_ms / max(g_n, 1) * 1e-3
        per_launch_bytes = nbytes / launches
        achieved = per_launch_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate rocprofv3
        # --pmc passes of this same command: tools/pmc_bench.sh -> profiles/r01_pmc_hbm_traffic.json)
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")) as f:
                pm = json.load(f)["gemm_glds_kernel"]
            traffic = int((pm["hbm_read_bytes_per_step_corrected_x2"] + pm["hbm_write_bytes_per_step"]) / pm["launches_per_step"])
        except Exception:
            pass
        roofline = {"kernel": "gemm_glds_kernel (bf16 1x1-conv MFMA GEMM, direct-to-LDS operand loads; epilogues: BN statistics / "
                              "fused BN3+identity+ReLU Bottleneck tail)",
                    "dominant_class_by_time": dom, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_note": "PMC HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/)",
                    "launches_per_step": g_n // nprof, "avg_launch_us": round(avg_s * 1e6, 2),
                    "algorithmic_bytes_per_launch": int(per_launch_bytes),
                    "mfma_tflops": round(flops / launches / avg_s / 1e12, 1) if avg_s > 0 else 0.0,
                    "mfma_frac_of_bf16_dense_peak": round(flops / launches / avg_s / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4) if avg_s > 0 else 0.0}
        if conc is not None:
            c_ms, c_n = conc["gemm"]
            c_avg = c_ms / max(c_n, 1) * 1e-3
            roofline["measured"] = ("one trunk pass in flight (CVCL_TRUNK_STREAMS=1 schedule): the kernel's own launch duration, "
                                    "which is also what the rocprofv3 --kernel-trace run of this command reports (profiles/r01_bench_c2_kernel_stats.csv)")
            roofline["concurrent"] = {"note": "the timed region keeps two trunk passes in flight on two HIP streams; event-timed "
                                              "there, a launch's duration includes the time it shares the GPU with the other "
                                              "pass's kernels -- per-kernel figures are not meaningful, the step time is",
                                      "avg_launch_us": round(c_avg * 1e6, 2),
                                      "kernel_ms_per_step": {k: round(v[0] / nconc, 4) for k, v in conc.items() if v[1] > 0}}
        if world > 1:
            dist.barrier()

    if rank == 0:
        line = {"metric": METRIC, "value": round(value, 1), "unit": "pairs/s", "n_gpus": world, "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if a.precision == "bf16" else "f32",
                "data": "synthetic",
                "config": {"workload": "C2 = BASELINE configs[1]: CVCL saycam_contrastive, frozen random-init ResNeXt-50 "
                                       "32x4d (BN train mode) + embedding mean-pool text encoder, E=512, L2-normalised, "
                                       "fixed tau 0.07, 224x224 frames + 3-word utterances; full step = fwd + InfoNCE + "
                                       "bwd(fc, embedding) + AdamW",
                           "per_gpu_batch": PER_GPU_BATCH, "global_batch": PER_GPU_BATCH * world,
                           "negatives": "global (RCCL all-gather)" if world > 1 else "local (single GPU)",
                           "parallelism": f"dp{world}", "trunk_streams": trunk_streams},
                "final_loss": round(loss, 5)}
        if roofline is not None:
            line["roofline"] = roofline
            line["kernel_ms_per_step"] = breakdown
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
