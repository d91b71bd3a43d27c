"""4-way forced-choice evaluation with the reference's CLI and output format (reference eval.py:27-331).

    python eval.py --checkpoint checkpoints/<exp>/epoch=0.ckpt --eval_dataset synthetic --stage test --save_predictions

Per trial the reference runs ``model(img, label, label_len)`` once (batch 1: four images vs one label for
``--eval_type image``, one image vs four labels for ``--eval_type text``), takes the soft-max of the 4 logits, the arg-max
as the prediction (target at index 0) and appends one record to ``results/<dataset>/<name>_predictions.json``
(``{"data": [{checkpoint, model, seed, shuffle_utterances, augment_frames, multiple_frames, cnn, eval_type, eval_dataset,
stage, trial_idx, categories, logits, pred, correct}, ...]}``).  Same records here; the difference is how the device is
used: ``--trial_batch T`` trials are encoded together (4T images + T labels in one pass through the HIP encoders, eval-mode
BatchNorm is per-sample so the numbers are those of the batch-1 calls) and each trial's 4 logits are read off the
block diagonal of the T x 4T logit matrix -- the batch-1 path is launch-latency bound (~0.6 ms per trial).

``--eval_dataset saycam | object_categories`` read the reference's private evaluation frames from hard-coded cluster paths
and are not available here; ``synthetic`` uses the synthetic trials of the data module (same item layout)."""
import argparse
import glob
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "multimodal-baby_amd"))

from multimodal.multimodal_data_module import SyntheticDataModule                 # noqa: E402
from multimodal.multimodal_lit import MultiModalLitModel                          # noqa: E402
from train import _setup_parser                                                   # noqa: E402


def config_from_checkpoint_name(name):
    """The run attributes the reference recovers from the checkpoint's directory name (eval.py:61-100)."""
    cfg = {"model": next((m for m in ("lstm", "transformer", "embedding") if m in name), None)}
    cfg["seed"] = next((i for i in (0, 1, 2) if f"seed_{i}" in name), None)
    cfg["shuffle_utterances"] = "shuffle_utterances" in name
    if "pretrained_cnn_True_finetune_cnn_True" in name:
        cfg["cnn"] = "finetune_pretrained"
    elif "pretrained_cnn_False_finetune_cnn_True" in name:
        cfg["cnn"] = "finetune_random_init"
    elif "pretrained_cnn_False_finetune_cnn_False" in name:
        cfg["cnn"] = "frozen_random_init"
    else:
        cfg["cnn"] = "frozen_pretrained"
    cfg["augment_frames"] = "augment_frames_False" not in name
    cfg["multiple_frames"] = "multiple_frames_False" not in name
    return cfg


def resolve_checkpoint(name, root="checkpoints"):
    """A .ckpt path is used as is; a run name resolves to its last.ckpt (shuffled-utterance runs) or its epoch*.ckpt
    (the best-val-loss checkpoint ModelCheckpoint kept), as eval.py:50-58 does under the reference's checkpoint root."""
    if name.endswith(".ckpt"):
        return name
    if "shuffle_utterances_True" in name:
        return os.path.join(root, name, "last.ckpt")
    found = sorted(glob.glob(os.path.join(root, name, "epoch*.ckpt")))
    if not found:
        raise FileNotFoundError(f"no epoch*.ckpt under {os.path.join(root, name)}")
    return found[0]


def results_filename(args, cfg):
    """eval.py:293-310."""
    d, m, c, s = args.eval_dataset, cfg["model"], cfg["cnn"], cfg["seed"]
    tail = f"{args.eval_type}_{d}_{args.stage}"
    if args.eval_metadata_filename == "eval_filtered_test.json":
        return f"results/{d}/{m}_{c}_seed_{s}_{tail}_eval_filtered_predictions.json"
    if args.eval_metadata_filename == "eval_manual_filtered_test.json":
        return f"results/{d}/{m}_{c}_seed_{s}_{tail}_eval_manual_filtered_predictions.json"
    if cfg["shuffle_utterances"]:
        return f"results/{d}/shuffle_{m}_{c}_seed_{s}_{tail}_eval_predictions.json"
    if not cfg["augment_frames"]:
        return f"results/{d}/{m}_{c}_augment_frames_{cfg['augment_frames']}_seed_{s}_{tail}_eval_predictions.json"
    if not cfg["multiple_frames"]:
        return f"results/{d}/{m}_{c}_multiple_frames_{cfg['multiple_frames']}_seed_{s}_{tail}_eval_predictions.json"
    return f"results/{d}/{m}_{c}_seed_{s}_{tail}_eval_predictions.json"


@torch.no_grad()
def evaluate_trials(model, trials, eval_type, device, datamodule=None):
    """trials: list of collated batch-1 items (img, label, label_len, raw_label).  Returns per trial (soft-max list, pred).
    All trials of the list are encoded in one pass; trial t's logits are the t-th diagonal block."""
    T = len(trials)
    if eval_type == "image":
        imgs = torch.cat([t[0].squeeze(0) for t in trials], 0).to(device)                    # [4T, ...]
        n_per = trials[0][0].shape[1]
        L = max(t[1].shape[1] for t in trials)
        labels = torch.zeros(T, L, dtype=torch.long)
        for i, t in enumerate(trials):
            labels[i, : t[1].shape[1]] = t[1][0]
        lens = torch.cat([t[2].reshape(1) for t in trials]).long()
    else:
        imgs = torch.cat([t[0].squeeze(0) for t in trials], 0).to(device)                    # [T, ...]
        n_per = trials[0][1].shape[1]
        L = max(t[1].shape[2] for t in trials)
        labels = torch.zeros(T * n_per, L, dtype=torch.long)
        for i, t in enumerate(trials):
            labels[i * n_per:(i + 1) * n_per, : t[1].shape[2]] = t[1][0]
        lens = torch.cat([t[2].reshape(-1) for t in trials]).long()
    if imgs.dtype == torch.uint8 and datamodule is not None:                                 # --device_frames: base transform on the GPU
        imgs = datamodule.on_after_batch_transfer((imgs,), 1, training=False)[0]
    logits_per_image, logits_per_text = model(imgs, labels.to(device), lens.to(device))
    out = []
    for i in range(T):
        if eval_type == "image":
            row = logits_per_text[i, i * n_per:(i + 1) * n_per]
        else:
            row = logits_per_image[i, i * n_per:(i + 1) * n_per]
        out.append((torch.softmax(row.float(), dim=-1).cpu().numpy().tolist(), int(torch.argmax(row))))
    return out


def main(args):
    if args.clip_eval:
        raise SystemExit("--clip_eval evaluates OpenAI CLIP ViT-L/14 through the `clip` package, which is not part of this path")
    if args.eval_dataset != "synthetic":
        raise SystemExit(f"--eval_dataset {args.eval_dataset} reads the reference's private evaluation frames from hard-coded "
                         "cluster paths and is not available here; use --eval_dataset synthetic")
    device = torch.device("cuda:0")
    checkpoint_name = args.checkpoint
    checkpoint = resolve_checkpoint(checkpoint_name, args.checkpoints_root)
    cfg = config_from_checkpoint_name(checkpoint_name)
    model = MultiModalLitModel.load_from_checkpoint(checkpoint, map_location=device)
    model.to(device)
    model.eval()
    if args.hip_graph:                                     # replay the image encoder's launches as a HIP graph per batch shape
        model.vision_encoder.enable_hip_graphs(True)

    data_args = _setup_parser().parse_args("")
    for key, value in model.args.items():
        setattr(data_args, key, value)
    data_args.augment_frames = False                       # deterministic frames (eval.py:115)
    data_args.eval_include_sos_eos = args.eval_include_sos_eos
    data_args.eval_type = args.eval_type
    data_args.eval_metadata_filename = args.eval_metadata_filename
    data_args.n_eval_trials = args.n_trials
    data = SyntheticDataModule(data_args)
    data.prepare_data()
    data.setup()
    loaders = {"dev": data.val_dataloader, "test": data.test_dataloader}[args.stage]()
    dataloader = loaders[1]                                # the second dataloader holds the evaluation trials
    eval_data = data.eval_sets["val" if args.stage == "dev" else "test"].metadata()
    classes = sorted({t["target_category"] for t in eval_data})
    correct_pred = {c: 0 for c in classes}
    total_pred = {c: 0 for c in classes}

    results, pending, first = [], [], 0

    def flush():
        nonlocal first
        for k, (logits_list, pred) in enumerate(evaluate_trials(model, pending, args.eval_type, device, data)):
            i = first + k
            class_label = pending[k][3][0][0]
            correct = pred == 0                            # the target is always at index 0
            correct_pred[class_label] += int(correct)
            total_pred[class_label] += 1
            trial = eval_data[i]
            results.append({
                "checkpoint": checkpoint_name, "model": cfg["model"], "seed": cfg["seed"],
                "shuffle_utterances": cfg["shuffle_utterances"], "augment_frames": cfg["augment_frames"],
                "multiple_frames": cfg["multiple_frames"], "cnn": cfg["cnn"], "eval_type": args.eval_type,
                "eval_dataset": args.eval_dataset, "stage": args.stage, "trial_idx": i,
                "categories": [trial["target_category"]] + trial["foil_categories"],
                "logits": logits_list, "pred": pred, "correct": bool(correct)})
        first += len(pending)
        pending.clear()

    for batch in dataloader:
        pending.append(batch)
        if len(pending) == max(args.trial_batch, 1):
            flush()
    if pending:
        flush()

    for classname, correct_count in correct_pred.items():
        print(f"Accuracy for class {classname:8s} is: {float(correct_count) / total_pred[classname]:.1%}")
    print(f"Total accuracy: {sum(correct_pred.values()) / sum(total_pred.values()):%}")

    if args.save_predictions:
        filename = results_filename(args, cfg)
        os.makedirs(os.path.dirname(filename), exist_ok=True)
        print(f"Saving predictions to {filename}")
        with open(filename, "w") as f:
            json.dump({"data": results}, f)
    return results


def _parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--checkpoint", type=str, help="path to checkpoint to use for evaluation")
    parser.add_argument("--clip_eval", action="store_true", help="Use CLIP model for evaluation")
    parser.add_argument("--stage", type=str, default="test", choices=["dev", "test"], help="which evaluation stage to use")
    parser.add_argument("--eval_include_sos_eos", action="store_true", help="include SOS/EOS tokens for eval labels")
    parser.add_argument("--eval_type", type=str, default="image", choices=["image", "text"],
                        help="Run evaluation using multiple images or multiple labels")
    parser.add_argument("--eval_dataset", type=str, default="saycam", choices=["saycam", "object_categories", "synthetic"],
                        help="Which evaluation dataset to use")
    parser.add_argument("--eval_metadata_filename", type=str, default="eval_test.json",
                        help="JSON file with metadata evaluation split to use")
    parser.add_argument("--use_kitty_label", action="store_true", help="replaces cat label with kitty")
    parser.add_argument("--save_predictions", action="store_true", help="save model predictions to JSON")
    # additions of this implementation
    parser.add_argument("--trial_batch", type=int, default=64, help="trials encoded per device pass (1 = the reference's loop)")
    parser.add_argument("--hip_graph", action="store_true", help="capture the eval-mode image encoder into a HIP graph per batch "
                                                                  "shape and replay it (removes the host's launch lead; same results)")
    parser.add_argument("--checkpoints_root", type=str, default="checkpoints", help="where run names resolve to checkpoints")
    parser.add_argument("--n_trials", type=int, default=32, help="number of synthetic trials")
    return parser


if __name__ == "__main__":
    main(_parser().parse_args())
