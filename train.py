"""Training entry with the reference's CLI (reference train.py:18-107), MI355X-native underneath.

    python train.py --dataset synthetic --batch_size=8 --gpus=1 --checkpoint_callback=False --logger=False \
        --fast_dev_run --text_encoder=embedding --lambda_lm 0 --optimize_unused          # = run.sh:12 on synthetic data

Multi-GPU: ``python -m torch.distributed.run --nproc-per-node N train.py ... --gpus N`` (one process per GPU,
RCCL over xGMI; see multimodal/parallel.py)."""
import argparse
import os
import sys
from pathlib import Path

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "multimodal-baby_amd"))

from multimodal import lightning as pl                                            # noqa: E402
from multimodal.multimodal import LanguageModel, MultiModalModel, TextEncoder, VisionEncoder   # noqa: E402
from multimodal.multimodal_data_module import MultiModalDataModule, SyntheticDataModule        # noqa: E402
from multimodal.multimodal_lit import MultiModalLitModel                          # noqa: E402


def _setup_parser():
    """Trainer + data + model + lit-model arguments, same flag names as the reference."""
    parser = argparse.ArgumentParser()
    trainer_group = parser.add_argument_group("Trainer Args")
    pl.Trainer.add_argparse_args(trainer_group)
    data_group = parser.add_argument_group("Data Args")
    MultiModalDataModule.add_to_argparse(data_group)
    MultiModalDataModule.add_additional_to_argparse(data_group)
    model_group = parser.add_argument_group("Model Args")
    VisionEncoder.add_to_argparse(model_group)
    TextEncoder.add_to_argparse(model_group)
    MultiModalModel.add_to_argparse(model_group)
    LanguageModel.add_to_argparse(model_group)
    lit_group = parser.add_argument_group("LitModel Args")
    MultiModalLitModel.add_to_argparse(lit_group)
    parser.add_argument("--exp_name", type=str, default="multimodal_test")
    parser.add_argument("--dataset", type=str, choices=["saycam", "coco", "synthetic"], default="saycam")
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--save_top_k", type=int, default=1)
    parser.add_argument("--resume_ckpt", type=Path, default=None)
    return parser


def main(argv=None):
    args = _setup_parser().parse_args(argv)
    ckpt_dir = Path("checkpoints") / args.exp_name
    if str(args.resume_ckpt) == "last":
        args.resume_ckpt = ckpt_dir / "last.ckpt"
    pl.seed_everything(args.seed)
    if args.dataset != "synthetic":
        raise SystemExit(f"--dataset {args.dataset} reads a private dataset from hard-coded cluster paths in the "
                         "reference and is not available here; use --dataset synthetic")
    data = SyntheticDataModule(args)
    vocab = data.read_vocab()
    vision_encoder = VisionEncoder(args=args)
    text_encoder = TextEncoder(vocab, image_feature_map_dim=vision_encoder.last_cnn_out_dim, args=args)
    lit_model = MultiModalLitModel(vision_encoder, text_encoder, args)
    checkpoint_callback = pl.ModelCheckpoint(monitor="val_loss", save_last=True, save_top_k=args.save_top_k,
                                             dirpath=ckpt_dir, filename="{epoch}")
    trainer = pl.Trainer.from_argparse_args(args, enable_checkpointing=args.checkpoint_callback,
                                            callbacks=[checkpoint_callback])
    print(args)
    trainer.fit(lit_model, data, ckpt_path=args.resume_ckpt)
    return trainer, lit_model


if __name__ == "__main__":
    main()
