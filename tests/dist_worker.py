"""Worker for the multi-rank GPU tests (tests/test_parallel_gpu.py): started by torch.distributed.run with two ranks that
share the box's one GPU over the gloo backend (CVCL_DIST_BACKEND=gloo; RCCL refuses two ranks on one device).

    dist_worker.py bench_step OUT            one C2 step at 256 pairs per rank through DataParallelEngine + OverlappedUpdate
    dist_worker.py train OUT -- <train.py args>   Trainer.fit through train.py; dumps the trainable parameters

Not a test module itself (no test_ prefix)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "multimodal-baby_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch                                   # noqa: E402
import torch.distributed as dist               # noqa: E402


def bench_step(out_dir):
    import bench
    from multimodal import parallel
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    dist.init_process_group(backend=os.environ.get("CVCL_DIST_BACKEND", "gloo"))
    lit, ve, opt = bench.build_model("c2", device, "bf16")
    engine = parallel.DataParallelEngine(device, global_negatives=True).attach(lit)
    batch = bench.synthetic_batch_on_device(bench.PER_GPU_BATCH, seed=rank, device=device) + (None,)
    upd = parallel.OverlappedUpdate(engine, opt, ve)
    assert upd.can_defer
    with torch.no_grad():
        pooled = ve.model.trunk(batch[0])[0].clone()              # this rank's trunk output (train-mode BN over its own 256)
    before = {k: v.detach().clone() for k, v in lit.named_parameters() if v.requires_grad}
    losses = []
    for _ in range(2):                                            # step 2's trunk is enqueued before step 1's update is applied
        out = lit.training_step(batch, 0)
        upd.zero_grad()
        out["loss"].backward()
        upd.step_done()
        losses.append(float(out["loss"].detach()))
        if len(losses) == 1:
            upd.flush()
            grads = {k: v.grad.detach().clone().cpu() for k, v in lit.named_parameters() if v.grad is not None}
            after1 = {k: v.detach().clone().cpu() for k, v in lit.named_parameters() if v.requires_grad}
    upd.flush()
    torch.cuda.synchronize()
    torch.save({"pooled": pooled.cpu(), "tok": batch[1].cpu(), "len": batch[2].cpu(), "losses": losses, "grads": grads,
                "before": {k: v.cpu() for k, v in before.items()}, "after1": after1},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def train(out_dir, argv):
    import contextlib
    import io
    import train as train_entry
    with contextlib.redirect_stdout(io.StringIO()):
        trainer, lit = train_entry.main(argv)
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.synchronize()
    torch.save({"params": {k: v.detach().cpu() for k, v in lit.named_parameters() if v.requires_grad},
                "logged": {k: float(v) for k, v in trainer.logged_metrics.items() if isinstance(v, (int, float)) or (torch.is_tensor(v) and v.numel() == 1)},
                "global_step": trainer.global_step},
               os.path.join(out_dir, f"rank{rank}.pt"))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    mode, out = sys.argv[1], sys.argv[2]
    if mode == "bench_step":
        bench_step(out)
    elif mode == "train":
        train(out, sys.argv[sys.argv.index("--") + 1:])
    else:
        raise SystemExit(f"unknown mode {mode}")
