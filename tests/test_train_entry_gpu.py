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
    trainer2, lit2 = train.main(base + ["--max_epochs", "2", "--resume_ckpt", "last"])
    assert trainer2.global_step == 4                                              # resumed at epoch 1, ran one more epoch
    assert not torch.equal(lit2.vision_encoder.model.fc.weight.detach().cpu(), w1.cpu())
    assert int(lit2.vision_encoder.model.bn1.num_batches_tracked) == 4
