"""Worker for the multi-rank GPU tests (tests/test_parallel_gpu.py): started by torch.distributed.run with two (or eight) ranks that
share the box's one GPU over the gloo backend (CVCL_DIST_BACKEND=gloo; RCCL refuses two ranks on one device).

    dist_worker.py bench_step OUT            one C2 step at 256 pairs per rank through DataParallelEngine + OverlappedUpdate
    dist_worker.py spatial_step OUT          --embedding_type spatial under global negatives (sim max and mean), fp32 parity mode
    dist_worker.py train OUT -- <train.py args>   Trainer.fit through train.py; dumps the trainable parameters
    dist_worker.py rccl_w1 OUT               ONE rank, backend nccl (= RCCL), $CVCL_FORCE_DIST=1: the whole multi-GPU path on one GPU

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
    coll0 = dict(parallel.COLLECTIVES)
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
    centres = ve.model.export_centres()
    torch.save({"pooled": pooled.cpu(), "tok": batch[1].cpu(), "len": batch[2].cpu(), "losses": losses, "grads": grads,
                "before": {k: v.cpu() for k, v in before.items()}, "after1": after1,
                "centres": None if centres is None else centres["frozen"],
                "collectives_per_step": {k: (parallel.COLLECTIVES[k] - coll0[k]) / 2 for k in coll0},
                "broadcasts_total": parallel.COLLECTIVES["broadcast"]},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def spatial_step(out_dir):
    """--embedding_type spatial under data-parallel GLOBAL negatives: every rank runs the product model on its own shard (fp32 parity
    mode, frozen trunk, trainable 1x1 projection + word embeddings) through DataParallelEngine, and dumps what the oracle needs to
    rebuild the global step: its layer-4 maps, tokens, lengths, the loss and the reduced gradients."""
    import argparse
    import train as train_entry
    from multimodal import parallel
    from multimodal.multimodal_data_module import SyntheticDataModule
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    dist.init_process_group(backend=os.environ.get("CVCL_DIST_BACKEND", "gloo"))
    res = {}
    for sim in ("max", "mean"):
        argv = (f"--dataset synthetic --batch_size 4 --gpus 1 --text_encoder embedding --embedding_dim 32 --embedding_type spatial "
                f"--sim {sim} --normalize_features --lambda_lm 0 --optimize_unused --checkpoint_callback False --logger False "
                f"--precision 32 --seed 7").split()
        args = train_entry._setup_parser().parse_args(argv)
        torch.manual_seed(7)                                                   # replicas start identical
        dm = SyntheticDataModule(args)
        ve = train_entry.VisionEncoder(args=args)
        te = train_entry.TextEncoder(dm.read_vocab(), image_feature_map_dim=ve.last_cnn_out_dim, args=args)
        lit = train_entry.MultiModalLitModel(ve, te, args).to(device)
        lit.set_precision(32)
        lit.train()
        engine = parallel.DataParallelEngine(device, global_negatives=True).attach(lit)
        dm.setup()
        it = iter(dm.train_dataloader())
        for _ in range(rank + 1):                                              # a different shard per rank
            x, y, y_len, _ = next(it)
        # the collate pads a batch to ITS longest utterance, so two ranks' token matrices differ in width on real data (the synthetic
        # set's utterances all have one length): rank r's batch carries 2 r + 1 more PAD columns
        y = torch.nn.functional.pad(y, (0, 2 * rank + 1), value=0)
        fmaps = []
        hook = lit.vision_encoder.model[7].register_forward_hook(lambda m, i, o: fmaps.append(o.detach().float().cpu().clone()))
        coll0 = dict(parallel.COLLECTIVES)
        out = lit.training_step((x.to(device), y.to(device), y_len.to(device), None), 0)
        out["loss"].backward()
        engine.reduce_gradients()
        torch.cuda.synchronize()
        hook.remove()
        res[sim] = {"fmap": fmaps[0], "tok": y, "len": y_len, "loss": float(out["loss"].detach()),
                    "grads": {k: v.grad.detach().cpu().clone() for k, v in lit.named_parameters() if v.grad is not None},
                    "params": {k: v.detach().cpu().clone() for k, v in lit.named_parameters() if v.requires_grad},
                    "all_gathers": parallel.COLLECTIVES["all_gather"] - coll0["all_gather"]}
    torch.save(res, os.path.join(out_dir, f"rank{rank}.pt"))
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


def rccl_w1(out_dir):
    """World-size-1 RCCL process group driving global_sim_logits (the feature all-gather), DataParallelEngine (hooks, bucket launch behind the producing
    streams, handle.wait()), OverlappedUpdate and the two trunk streams; each schedule is run twice -- through the process group
    ($CVCL_FORCE_DIST=1) and as the plain single-process step -- from identical initial states, for the frozen C2 step and for
    --finetune_cnn (weight gradients arriving from trunk_train's side stream).  The caller compares bit for bit."""
    import bench
    from multimodal import parallel
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    res = {}

    def run(tag, forced, finetune, B, steps):
        os.environ["CVCL_FORCE_DIST"] = "1" if forced else "0"
        assert parallel.is_distributed() == forced
        lit, ve, _ = bench.build_model("c2", device, "bf16", seed=5)
        if finetune:
            for prm in ve.model.parameters():
                prm.requires_grad_(True)
        opt = lit.configure_optimizers()
        engine = parallel.DataParallelEngine(device, bucket_bytes=8 << 20, global_negatives=True).attach(lit)
        assert bool(engine.buckets) == forced
        batch = bench.synthetic_batch_on_device(B, seed=1, device=device) + (None,)
        if not finetune:
            ve.model.enable_trunk_stream(device, inputs="ready", n_streams=2)
        upd = parallel.OverlappedUpdate(engine, opt, ve) if forced else None
        assert upd is None or upd.can_defer == (not finetune)
        losses = []
        for _ in range(steps):
            if upd is None:
                opt.zero_grad(set_to_none=True)
                out = lit.training_step(batch, 0)
                out["loss"].backward()
                engine.reduce_gradients()
                opt.step()
            else:
                out = lit.training_step(batch, 0)
                upd.zero_grad()
                out["loss"].backward()
                upd.step_done()
            losses.append(out["loss"].detach().clone())
        if upd is not None:
            upd.flush()
        torch.cuda.synchronize()
        if not finetune:
            ve.model.enable_trunk_stream(device, inputs=None)
        res[tag] = {"losses": [float(v) for v in losses],
                    "state": {k: v.detach().cpu().clone() for k, v in lit.state_dict().items()}}
        del lit, ve, opt, engine, upd

    run("frozen_dist", True, False, 256, 4)
    run("frozen_plain", False, False, 256, 4)
    run("finetune_dist", True, True, 16, 3)
    run("finetune_plain", False, True, 16, 3)
    with open("/proc/self/maps") as f:
        res["librccl_mapped"] = any("librccl" in line for line in f)
    torch.save(res, os.path.join(out_dir, "w1.pt"))
    dist.destroy_process_group()


if __name__ == "__main__":
    mode, out = sys.argv[1], sys.argv[2]
    if mode == "rccl_w1":
        rccl_w1(out)
        raise SystemExit(0)
    if mode == "bench_step":
        bench_step(out)
    elif mode == "spatial_step":
        spatial_step(out)
    elif mode == "train":
        train(out, sys.argv[sys.argv.index("--") + 1:])
    else:
        raise SystemExit(f"unknown mode {mode}")
