"""GPU: the train.py entry (reference train.py:58-107 surface) runs BASELINE configs[0] end to end on synthetic data:
`run.sh:12` flags + --dataset synthetic; checkpoint save + resume; the C2 flag set in bf16."""
import os
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, ROOT)


def test_fast_dev_run_c1(dev, tmp_path, monkeypatch):
    import train
    monkeypatch.chdir(tmp_path)
    argv = ("--dataset synthetic --batch_size=8 --gpus=1 --num_workers=8 --checkpoint_callback=False --logger=False "
            "--multiple_frames --fast_dev_run --text_encoder=embedding --lambda_lm 0 --optimize_unused").split()
    trainer, lit = train.main(argv)
    assert trainer.global_step == 1
    logged = trainer.logged_metrics
    for k in ("train_infonce_loss", "train_image_accuracy", "train_text_accuracy", "train_image_entropy",
              "train_text_entropy", "train_loss", "temperature"):
        assert k in logged, k
    assert torch.isfinite(torch.as_tensor(float(logged["train_loss"])))
    assert int(lit.vision_encoder.model.bn1.num_batches_tracked) == 1            # BN trains its statistics even when frozen
    assert not (tmp_path / "checkpoints").exists()


def test_checkpoint_and_resume_c2_bf16(dev, tmp_path, monkeypatch):
    import train
    monkeypatch.chdir(tmp_path)
    base = ("--dataset synthetic --batch_size=16 --gpus=1 --precision bf16 --embedding_dim 512 --normalize_features "
            "--fix_temperature --temperature 0.07 --lr 1e-4 --weight_decay 0.1 --lambda_lm 0 --optimize_unused "
            "--logger=False --exp_name t --limit_train_batches 2 --drop_last").split()
    trainer, lit = train.main(base + ["--max_epochs", "1"])
    ck = tmp_path / "checkpoints" / "t" / "last.ckpt"
    assert ck.exists() and trainer.global_step == 2
    w1 = lit.vision_encoder.model.fc.weight.detach().clone()
    # the bf16 storage centres (calibrated on the first train batch; not in the state_dict) travel in the checkpoint ...
    saved = torch.load(ck, map_location="cpu", weights_only=False)
    c1 = lit.vision_encoder.model.export_centres()["frozen"]
    assert torch.equal(saved["cvcl_storage_centres"]["frozen"], c1) and c1.shape == (53, 2048) and float(c1.abs().max()) > 0
    assert "cvcl_storage_centres" not in saved["state_dict"] and not any("centre" in k for k in saved["state_dict"])
    trainer2, lit2 = train.main(base + ["--max_epochs", "2", "--resume_ckpt", "last"])
    # ... and the resumed run adopts them instead of recalibrating on whatever batch comes first
    assert torch.equal(lit2.vision_encoder.model.export_centres()["frozen"], c1)
    assert trainer2.global_step == 4                                              # resumed at epoch 1, ran one more epoch
    assert not torch.equal(lit2.vision_encoder.model.fc.weight.detach().cpu(), w1.cpu())
    assert int(lit2.vision_encoder.model.bn1.num_batches_tracked) == 4


def test_validation_and_four_way_trials(dev, tmp_path, monkeypatch):
    """Lightning's evaluation loop through train.py's objects: val pairs (eval-mode BN on running statistics, no grad)
    + one 4-way trial per batch (reference multimodal_lit.py:466-511: logits_per_text[0] over the 4 images, target at
    index 0), val_loss/val_accuracy logged, trial logits equal to an explicit encode_image/encode_text evaluation."""
    import contextlib, io
    import train
    monkeypatch.chdir(tmp_path)
    argv = ("--dataset synthetic --batch_size 4 --val_batch_size 4 --gpus 1 --text_encoder embedding --embedding_dim 32 "
            "--lambda_lm 0 --optimize_unused --max_epochs 1 --limit_train_batches 2 --normalize_features "
            "--checkpoint_callback True --logger False --exp_name evaltest").split()
    with contextlib.redirect_stdout(io.StringIO()):
        trainer, lit = train.main(argv)
    m = trainer.logged_metrics
    for k in ("val_loss", "val_infonce_loss", "val_image_accuracy", "val_accuracy", "val_entropy"):
        assert k in m, (k, sorted(m))
    assert 0.0 <= float(m["val_accuracy"]) <= 1.0 and float(m["val_loss"]) > 0
    assert (tmp_path / "checkpoints" / "evaltest" / "epoch=0.ckpt").exists()
    # a trial by hand: image features of the 4 candidates vs the label's text feature
    from multimodal.multimodal_data_module import SyntheticDataModule
    dm = SyntheticDataModule(train._setup_parser().parse_args(argv))
    dm.setup()
    x, y, y_len, raw = next(iter(dm.val_dataloader()[1]))
    lit.eval()
    with torch.no_grad():
        xi = x.view(-1, 3, 224, 224).to(dev)
        img = lit.encode_image(xi)
        txt = lit.encode_text(y.to(dev), y_len.to(dev))
        lpi, lpt = lit.model(xi, y.to(dev), y_len.to(dev))
    assert lpt.shape == (1, 4) and lpi.shape == (4, 1)
    scale = float(lit.model.logit_neg_log_temperature.exp())
    assert torch.allclose(lpt[0], (txt @ img.t())[0] * scale, rtol=1e-4, atol=1e-5)
    out = lit.validation_step((xi.view(1, 4, 3, 224, 224), y.to(dev), y_len.to(dev), raw), 0, dataloader_idx=1)
    assert out["accuracy"] == int(int(torch.argmax(lpt[0])) == 0)
    # test loop: same machinery under the test_ prefix
    with contextlib.redirect_stdout(io.StringIO()):
        res = trainer.test(lit, dm)[0]
    assert "test_loss" in res and "test_accuracy" in res


def test_device_frames_pipeline_through_train_entry(dev, tmp_path, monkeypatch):
    """--device_frames: the datasets yield uint8 frames and the data module's on_after_batch_transfer hook runs the
    reference's train transform (--augment_frames) or base transform on the device.  With the base
    transform the run is the float-frame run of the same frames, bit for bit; with augmentation it trains and validates."""
    import contextlib, io
    import train
    from multimodal.multimodal_data_module import SyntheticDataModule, IMAGENET_MEAN, IMAGENET_STD
    monkeypatch.chdir(tmp_path)
    base = ("--dataset synthetic --batch_size 4 --val_batch_size 4 --gpus 1 --text_encoder embedding --embedding_dim 32 "
            "--lambda_lm 0 --optimize_unused --max_epochs 1 --limit_train_batches 2 --normalize_features "
            "--checkpoint_callback False --logger False --exp_name devframes --device_frames").split()
    dm = SyntheticDataModule(train._setup_parser().parse_args(base))
    dm.setup()
    batch = next(iter(dm.train_dataloader()))
    assert batch[0].dtype == torch.uint8 and batch[0].shape == (4, 224, 224, 3)
    moved = tuple(b.to(dev) if torch.is_tensor(b) else b for b in batch)
    out = dm.on_after_batch_transfer(moved, 0, training=True)                    # no --augment_frames: base transform
    mean, std = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1), torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    want = (batch[0].permute(0, 3, 1, 2).float().div(255) - mean) / std
    assert out[0].dtype == torch.float32 and torch.equal(out[0].cpu(), want) and out[1] is moved[1]
    trial = next(iter(dm.val_dataloader()[1]))
    assert trial[0].shape == (1, 4, 224, 224, 3)
    tout = dm.on_after_batch_transfer(tuple(b.to(dev) if torch.is_tensor(b) else b for b in trial), 1, training=False)
    assert tout[0].shape == (1, 4, 3, 224, 224)
    with contextlib.redirect_stdout(io.StringIO()):
        trainer, lit = train.main(base + ["--augment_frames"])
    m = trainer.logged_metrics
    assert "val_loss" in m and "val_accuracy" in m and float(m["val_loss"]) > 0
    aug_dm = SyntheticDataModule(train._setup_parser().parse_args(base + ["--augment_frames"]))
    a1 = aug_dm.on_after_batch_transfer(moved, 0, training=True)[0]
    a2 = aug_dm.on_after_batch_transfer(moved, 0, training=False)[0]
    assert not torch.equal(a1, out[0]) and torch.equal(a2, out[0])               # augmentation only while training


def test_overlapped_update_matches_sequential_schedule(dev):
    """parallel.OverlappedUpdate defers the (all-reduce +) optimizer step of step k to a hook between the frozen trunk and
    fc of step k+1: losses and final parameters must be bit-identical to the sequential schedule."""
    import contextlib, io, copy
    sys.path.insert(0, ROOT)
    from bench import c2_args, synthetic_batch_on_device
    from multimodal import parallel
    from multimodal.multimodal import TextEncoder, VisionEncoder
    from multimodal.multimodal_data_module import read_vocab
    from multimodal.multimodal_lit import MultiModalLitModel

    def build():
        torch.manual_seed(0)
        args = c2_args()
        with contextlib.redirect_stdout(io.StringIO()):
            ve = VisionEncoder(args)
            te = TextEncoder(read_vocab(), ve.last_cnn_out_dim, args)
            lit = MultiModalLitModel(ve, te, args)
        lit.to(dev); lit.set_precision("bf16"); lit.train()
        return ve, lit

    batches = [synthetic_batch_on_device(16, seed, dev) + (None,) for seed in range(4)]
    ve, lit = build()
    opt = lit.configure_optimizers()
    seq_losses = []
    for bt in batches:
        opt.zero_grad(set_to_none=True)
        out = lit.training_step(bt, 0)
        out["loss"].backward()
        opt.step()
        seq_losses.append(float(out["loss"].detach()))
    seq_params = {k: v.detach().clone() for k, v in lit.state_dict().items()}

    ve2, lit2 = build()
    opt2 = lit2.configure_optimizers()
    eng = parallel.DataParallelEngine(dev).attach(lit2)
    upd = parallel.OverlappedUpdate(eng, opt2, ve2)
    assert upd.can_defer
    ov_losses = []
    for bt in batches:
        out = lit2.training_step(bt, 0)
        upd.zero_grad()
        out["loss"].backward()
        upd.step_done()
        ov_losses.append(float(out["loss"].detach()))
    assert upd.pending
    upd.flush()
    assert ov_losses == seq_losses
    sd2 = lit2.state_dict()
    assert all(torch.equal(seq_params[k], sd2[k]) for k in seq_params)
    # a trainable trunk disables the deferral
    ve2.model.layer4[0].conv1.weight.requires_grad_(True)
    assert not parallel.OverlappedUpdate(eng, opt2, ve2).can_defer


def test_eval_entry_batched_trials_match_reference_loop(dev, tmp_path, monkeypatch):
    """eval.py (reference eval.py:27-331): train.py writes a Lightning-layout checkpoint, eval.py reloads it (pickled encoder
    hyper-parameters), runs the 4-way trials and writes the reference's prediction records.  Encoding 64 trials per device
    pass gives the numbers of the reference's batch-1 loop (--trial_batch 1), for both evaluation types; a trial by hand
    through encode_image / encode_text agrees."""
    import contextlib, io, json
    import eval as ev
    import train
    monkeypatch.chdir(tmp_path)
    exp = "multimodal_text_encoder_embedding_pretrained_cnn_False_finetune_cnn_False_seed_0"
    argv = ("--dataset synthetic --batch_size 4 --val_batch_size 4 --gpus 1 --text_encoder embedding --embedding_dim 32 "
            "--lambda_lm 0 --optimize_unused --max_epochs 1 --limit_train_batches 2 --normalize_features "
            f"--checkpoint_callback True --logger False --exp_name {exp}").split()
    with contextlib.redirect_stdout(io.StringIO()):
        train.main(argv)
    for eval_type in ("image", "text"):
        runs = {}
        for tb in (1, 64):
            a = ev._parser().parse_args(["--checkpoint", exp, "--eval_dataset", "synthetic", "--eval_type", eval_type, "--n_trials", "10",
                                         "--trial_batch", str(tb), "--save_predictions"])
            out = io.StringIO()
            with contextlib.redirect_stdout(out):
                runs[tb] = ev.main(a)
            assert "Total accuracy:" in out.getvalue()
        assert len(runs[1]) == len(runs[64]) == 10
        ag = ev._parser().parse_args(["--checkpoint", exp, "--eval_dataset", "synthetic", "--eval_type", eval_type, "--n_trials", "10",
                                      "--trial_batch", "4", "--hip_graph"])        # batches of 4, 4, 2 trials: two graph shapes
        with contextlib.redirect_stdout(io.StringIO()):
            graphed = ev.main(ag)
        for r1, rg in zip(runs[1], graphed):
            assert r1["pred"] == rg["pred"] and torch.allclose(torch.tensor(r1["logits"]), torch.tensor(rg["logits"]), rtol=1e-4, atol=1e-6)
        for r1, r64 in zip(runs[1], runs[64]):
            assert r1["pred"] == r64["pred"] and r1["correct"] == r64["correct"] and r1["categories"] == r64["categories"]
            assert torch.allclose(torch.tensor(r1["logits"]), torch.tensor(r64["logits"]), rtol=1e-4, atol=1e-6)
        fn = tmp_path / "results" / "synthetic" / f"embedding_frozen_random_init_seed_0_{eval_type}_synthetic_test_eval_predictions.json"
        rec = json.loads(fn.read_text())["data"]
        assert len(rec) == 10
        assert list(rec[0]) == ["checkpoint", "model", "seed", "shuffle_utterances", "augment_frames", "multiple_frames", "cnn", "eval_type",
                                "eval_dataset", "stage", "trial_idx", "categories", "logits", "pred", "correct"]
        assert rec[3]["trial_idx"] == 3 and rec[3]["checkpoint"] == exp and rec[3]["cnn"] == "frozen_random_init" and rec[3]["seed"] == 0
        assert len(rec[3]["logits"]) == 4 and abs(sum(rec[3]["logits"]) - 1.0) < 1e-5 and rec[3]["correct"] == (rec[3]["pred"] == 0)
    # one image-type trial by hand
    from multimodal.multimodal_lit import MultiModalLitModel
    from multimodal.multimodal_data_module import SyntheticEvalTrials
    lit = MultiModalLitModel.load_from_checkpoint(ev.resolve_checkpoint(exp), map_location=dev).to(dev).eval()
    a = ev._parser().parse_args(["--checkpoint", exp, "--eval_dataset", "synthetic", "--n_trials", "10"])
    with contextlib.redirect_stdout(io.StringIO()):
        res = ev.main(a)
    imgs, label, n, raw = SyntheticEvalTrials(10, 2350, seed=0 + 4)[2]
    with torch.no_grad():
        fi = lit.encode_image(imgs.to(dev))
        ft = lit.encode_text(label.view(1, -1).to(dev), torch.tensor([n], device=dev))
        want = torch.softmax((ft @ fi.t())[0] * float(lit.model.logit_neg_log_temperature.exp().detach()), -1).cpu()
    assert torch.allclose(torch.tensor(res[2]["logits"]), want, rtol=1e-4, atol=1e-6)


def test_trunk_stream_overlap_is_bit_identical(dev):
    """H.TrunkStream: the frozen trunk on its own stream (overlapping the previous step's trainable tail), and consecutive trunk
    passes alternating between two streams (overlapping each other as well), give the same losses, parameters and BatchNorm
    buffers, bit for bit, as the single-stream schedule over several optimizer steps."""
    import contextlib, io
    import bench
    from multimodal.multimodal import TextEncoder, VisionEncoder
    from multimodal.multimodal_data_module import read_vocab
    from multimodal.multimodal_lit import MultiModalLitModel

    def run(stream_mode):
        torch.manual_seed(0)
        args = bench.c2_args()
        with contextlib.redirect_stdout(io.StringIO()):
            ve = VisionEncoder(args)
            te = TextEncoder(read_vocab(), ve.last_cnn_out_dim, args)
            lit = MultiModalLitModel(ve, te, args)
        lit.to(dev)
        lit.set_precision("bf16")
        lit.train()
        opt = lit.configure_optimizers()
        batches = [bench.synthetic_batch_on_device(16, seed=s, device=dev) + (None,) for s in range(3)]
        torch.cuda.synchronize()
        if stream_mode:
            ts = ve.model.enable_trunk_stream(dev, inputs="ready", n_streams=stream_mode)
            assert ts.n_streams == stream_mode
        losses = []
        for i in range(6):
            opt.zero_grad(set_to_none=True)
            out = lit.training_step(batches[i % 3], 0)
            out["loss"].backward()
            opt.step()
            losses.append(out["loss"].detach())
        torch.cuda.synchronize()
        # every BatchNorm buffer of the trunk: the two-stream schedule applies the running-statistics updates of consecutive
        # passes in pass order (cvcl_resnext50_apply_moments), so all 53 x (mean, var, count) must match too
        buffers = torch.cat([b.detach().double().flatten().cpu() for _, b in lit.vision_encoder.model.named_buffers()])
        return torch.stack(losses).cpu(), lit.vision_encoder.model.fc.weight.detach().cpu().clone(), buffers

    l0, w0, r0 = run(0)
    assert float(r0.abs().sum()) > 0
    for n_streams in (1, 2, 3):
        l1, w1, r1 = run(n_streams)
        assert torch.equal(l0, l1) and torch.equal(w0, w1) and torch.equal(r0, r1), n_streams


def test_trainer_trunk_stream_flag_bit_identical(dev, tmp_path, monkeypatch):
    """train.py --trunk_stream True (batch copy + device frame transform + frozen trunk on their own stream, overlapping the
    previous step's tail; validation back on the main stream) trains to the same parameters and validation metrics, bit for
    bit, as the single-stream run -- ResNeXt and ViT trunks, one and two trunk streams."""
    import contextlib, io
    import train
    import multimodal.multimodal as mm
    from multimodal import vision_transformer_dino_mugs as vits
    monkeypatch.chdir(tmp_path)
    base = ("--dataset synthetic --batch_size 4 --val_batch_size 4 --gpus 1 --text_encoder embedding --embedding_dim 32 "
            "--lambda_lm 0 --optimize_unused --max_epochs 2 --limit_train_batches 3 --normalize_features --device_frames "
            "--checkpoint_callback False --logger False --exp_name ts").split()
    for vit in (False, True):
        extra = ["--vit_dino"] if vit else []
        orig = mm.load_model
        if vit:
            mm.load_model = lambda name, pretrained: vits.vit_base(patch_size=16, num_classes=0)
        try:
            res = []
            for flag, n_streams in (("False", "1"), ("True", "1"), ("True", "2")):
                monkeypatch.setenv("CVCL_TRUNK_STREAMS", n_streams)          # two streams: consecutive trunk passes overlap too
                monkeypatch.setenv("CVCL_VIT_TRUNK_STREAMS", n_streams)
                torch.manual_seed(0)
                with contextlib.redirect_stdout(io.StringIO()):
                    trainer, lit = train.main(base + extra + ["--trunk_stream", flag])
                torch.cuda.synchronize()
                head = lit.vision_encoder.model.head if vit else lit.vision_encoder.model.fc
                res.append((head.weight.detach().cpu().clone(), lit.text_encoder.embedding.weight.detach().cpu().clone(),
                            float(trainer.logged_metrics["val_loss"])))
        finally:
            mm.load_model = orig
        for r in res[1:]:
            assert torch.equal(res[0][0], r[0]) and torch.equal(res[0][1], r[1]) and res[0][2] == r[2], vit


def test_vit_finetune_through_train_entry(dev, tmp_path, monkeypatch):
    """train.py --vit_dino --finetune_cnn: the ViT trunk trains through vit_train.VitTrunk (every trunk parameter receives a
    gradient and moves), the loss is finite, and a checkpoint round-trips."""
    import contextlib, io
    import train
    import multimodal.multimodal as mm
    from multimodal import vision_transformer_dino_mugs as vits
    monkeypatch.chdir(tmp_path)
    argv = ("--dataset synthetic --batch_size 4 --val_batch_size 4 --gpus 1 --text_encoder embedding --embedding_dim 32 --precision bf16 "
            "--lambda_lm 0 --optimize_unused --max_epochs 1 --limit_train_batches 2 --normalize_features --vit_dino --finetune_cnn "
            "--checkpoint_callback True --logger False --exp_name vitft").split()
    orig = mm.load_model
    mm.load_model = lambda name, pretrained: vits.VisionTransformer(img_size=[224], patch_size=16, embed_dim=768, depth=2, num_heads=12,
                                                                    mlp_ratio=4, qkv_bias=True, num_classes=0)
    try:
        torch.manual_seed(0)
        ref = mm.load_model("x", False).state_dict()
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            trainer, lit = train.main(argv)
    finally:
        mm.load_model = orig
    m = trainer.logged_metrics
    assert float(m["val_loss"]) > 0 and torch.isfinite(torch.tensor(float(m["val_loss"])))
    vit = lit.vision_encoder.model
    moved = [n for n, p in vit.named_parameters() if p.requires_grad]
    assert any(n.startswith("blocks.1.mlp.fc2") for n in moved) and "pos_embed" in moved and "patch_embed.proj.weight" in moved
    for n in ("blocks.0.attn.qkv.weight", "blocks.1.norm2.bias", "pos_embed", "cls_token", "patch_embed.proj.weight", "norm.weight"):
        p = dict(vit.named_parameters())[n]
        assert torch.isfinite(p).all() and not torch.equal(p.detach().cpu(), ref[n]), n        # AdamW moved it
    assert (tmp_path / "checkpoints" / "vitft" / "epoch=0.ckpt").exists()


def test_three_adamw_steps_match_the_oracle_step_fp32(dev):
    """Three full train steps (forward with train-mode BatchNorm, InfoNCE, backward of the trainable set, AdamW) in the fp32 parity
    mode against the oracle's CpuTrainStep on the same weights and batches: per-step losses, the updated fc / embedding
    parameters and the frozen trunk's BatchNorm running statistics (updated even though the trunk is frozen: SURVEY 0.4)."""
    import argparse, contextlib, io, math
    import cvcl_oracle as O
    from multimodal.multimodal import TextEncoder, VisionEncoder
    from multimodal.multimodal_data_module import read_vocab
    from multimodal.multimodal_lit import MultiModalLitModel
    args = argparse.Namespace(
        embedding_type="flat", embedding_dim=64, pretrained_cnn=False, cnn_model="resnext50_32x4d", cnn_dino=False,
        vit_dino=False, finetune_cnn=False, text_encoder="embedding", captioning=False, attention=False,
        attention_gate=False, crange=1, dropout_i=0.0, dropout_o=0.0, pos_embed_type="no_pos_embed",
        normalize_features=True, sim="max", temperature=0.07, fix_temperature=True, tie=True, bias=True, lr=1e-3,
        weight_decay=0.1, lambda_mm=1.0, lambda_lm=0.0, lambda_ar=0.0, optimize_unused=True, lr_scheduler=False,
        optimizer=torch.optim.AdamW)
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        ve = VisionEncoder(args)
        lit = MultiModalLitModel(ve, TextEncoder(read_vocab(), 2048, args), args)
    sd = {k: v.detach().clone() for k, v in lit.model.state_dict().items()}
    sd["logit_neg_log_temperature"] = torch.tensor(-math.log(0.07))
    lit.to(dev).train()
    lit.set_precision("32")
    opt = lit.configure_optimizers()
    cpu = O.CpuTrainStep(sd, lr=1e-3, weight_decay=0.1, normalize_features=True)
    for s in range(3):
        img, tok, ln = O.synthetic_batch(6, seed=10 + s)
        want = cpu.step(img, tok, ln)
        opt.zero_grad(set_to_none=True)
        out = lit.training_step((img.to(dev), tok.to(dev), ln.to(dev), None), 0)
        out["loss"].backward()
        opt.step()
        got = float(out["loss"].detach())
        assert abs(got - want) < 2e-4 * max(1.0, abs(want)), (s, got, want)
    now = {k: v.detach().float().cpu() for k, v in lit.model.state_dict().items()}
    for k in O.TRAINABLE_FROZEN_CNN:
        # AdamW's normalised step m / sqrt(v) is +-lr in the first steps whatever the gradient's size, so elements whose gradient
        # is ~0 amplify 1e-7 differences to lr: compare the three-step UPDATE as a whole (direction and size), not element-wise
        ua, ub = (now[k] - sd[k]).flatten().double(), (cpu.p[k].detach() - sd[k]).flatten().double()
        assert float(ub.norm()) > 0                                            # it did move
        cos = float(torch.nn.functional.cosine_similarity(ua, ub, dim=0))
        rel = float((ua - ub).norm() / ub.norm())
        assert cos > 0.9995 and rel < 3e-2, (k, cos, rel)
    for k in ("image_embed.model.bn1.running_mean", "image_embed.model.layer1.0.bn3.running_var",
              "image_embed.model.layer3.2.bn2.running_mean", "image_embed.model.layer4.2.bn3.running_var",
              "image_embed.model.layer2.0.downsample.1.running_mean"):
        a, b = now[k], cpu.p[k].detach()
        assert float((a - b).abs().max()) < 2e-4 * max(1e-3, float(b.abs().max())), k
        assert not torch.equal(b, sd[k])
    assert int(now["image_embed.model.layer4.2.bn3.num_batches_tracked"]) == 3


@pytest.mark.parametrize("config", ["c2", "c4"])
def test_eval_hip_graph_replay_is_bit_identical_to_eager_launches(dev, config):
    """Opt-in HIP-graph replay of the eval-mode image encoder (VisionEncoder.enable_hip_graphs; eval.py --hip_graph): one graph per
    input shape over a static input buffer, results bit-identical to the eager launches for new inputs, dropped by train() so that
    updated weights are seen, never pickled, and bypassed whenever gradients or train-mode BatchNorm are involved."""
    import pickle
    import bench
    lit, ve, _ = bench.build_model(config, dev, precision="bf16")
    lit.eval()
    g = torch.Generator(device=dev).manual_seed(5)
    xs = {B: [torch.randn(B, 3, 224, 224, generator=g, device=dev) for _ in range(3)] for B in (1, 4)}
    with torch.no_grad():
        eager = {B: [tuple(None if t is None else t.clone() for t in ve(x)) for x in v] for B, v in xs.items()}
        ve.enable_hip_graphs(True)
        for B, v in xs.items():
            for i, x in enumerate(v):
                f, m = ve(x)
                assert torch.equal(f, eager[B][i][0])
                assert (m is None and eager[B][i][1] is None) or torch.equal(m, eager[B][i][1])
        assert set(k[0][0] for k in ve._graphs) == {1, 4}
        f_keep, _ = ve(xs[1][0])
        ve(xs[1][1])                                          # a later replay must not overwrite results handed out earlier
        assert torch.equal(f_keep, eager[1][0][0])
        logits_graph = lit(xs[4][0], torch.tensor([[2, 9, 3]], device=dev), torch.tensor([3], device=dev))[1]
    pickle.dumps(ve)                                          # graphs are not part of a pickled encoder
    # weights change in train mode: the graphs go, the next eval call captures the new weights
    lit.train()
    assert ve._graphs == {}
    with torch.no_grad():
        (ve.model.head if config == "c4" else ve.model.fc).weight.mul_(2.0)
    lit.eval()
    with torch.no_grad():
        f2, _ = ve(xs[1][0])
        ve.enable_hip_graphs(False)
        f2_eager, _ = ve(xs[1][0])
        logits_eager = lit(xs[4][0], torch.tensor([[2, 9, 3]], device=dev), torch.tensor([3], device=dev))[1]
    assert torch.equal(f2, f2_eager) and not torch.equal(f2, eager[1][0][0])
    assert logits_graph.shape == logits_eager.shape == (1, 4)
    # a full-model checkpoint load in eval mode (nn.Module.load_state_dict on a PARENT never calls the child's load_state_dict),
    # and an in-place weight edit in eval mode: the replayed graph must see the new packed weights, not the captured ones
    ve.enable_hip_graphs(True)
    with torch.no_grad():
        ve(xs[1][0])
        assert len(ve._graphs) == 1
        sd = {k: (v * 0.5 if v.is_floating_point() and v.dim() == 4 else v.clone()) for k, v in lit.state_dict().items()}   # conv weights halved
        lit.load_state_dict(sd)
        assert ve._graphs == {}
        f3, _ = ve(xs[1][0])
        first = next(p for p in ve.model.parameters() if p.dim() == 4)
        first.mul_(-1.0)                                      # in place, eval mode, no load: only the fingerprint can see it
        f4, _ = ve(xs[1][0])
        ve.enable_hip_graphs(False)
        f4_eager, _ = ve(xs[1][0])
        first.mul_(-1.0)
        f3_eager, _ = ve(xs[1][0])
    assert torch.equal(f3, f3_eager) and torch.equal(f4, f4_eager) and not torch.equal(f3, f4) and not torch.equal(f3, f2)
    # a Parameter OBJECT replaced between eval calls -- the head swapped for a new nn.Linear, a weight re-registered through
    # load_state_dict(assign=True) on a parent: the cached parameter list would still hold the old objects; the
    # registration epoch drops it, the fingerprint of the NEW objects differs, the graph is re-captured
    import torch.nn as nn
    ve.enable_hip_graphs(True)
    with torch.no_grad():
        f5, _ = ve(xs[1][0])
        name = "head" if config == "c4" else "fc"
        old_head = getattr(ve.model, name)
        new_head = nn.Linear(old_head.in_features, old_head.out_features).to(dev)
        new_head.weight.copy_(old_head.weight * 3.0)
        new_head.bias.copy_(old_head.bias)
        for q in new_head.parameters():
            q.requires_grad_(False)
        setattr(ve.model, name, new_head)
        f6, _ = ve(xs[1][0])
        # (every 4-d weight: the encoder is registered twice in the lit module -- vision_encoder and model.image_embed -- and an
        # assigning load applies both aliases' entries in turn)
        sd = {k: (v * 0.25 if v.is_floating_point() and v.dim() == 4 else v.clone()) for k, v in lit.state_dict().items()}
        lit.load_state_dict(sd, assign=True)
        f7, _ = ve(xs[1][0])
        ve.enable_hip_graphs(False)
        f7_eager, _ = ve(xs[1][0])
    assert not torch.equal(f5, f6) and not torch.equal(f6, f7) and torch.equal(f7, f7_eager)
    # gradients enabled or train mode: never a graph
    ve.enable_hip_graphs(True)
    ve._graphs.clear()
    ve(xs[1][0])
    assert ve._graphs == {}
