"""What PyTorch's own bf16 does on the benchmarked configuration -- the yardstick for ``logits_rel_vs_fp32``.

The oracle's functional restatement of the reference forward (oracle/cvcl_oracle.py, BatchNorm through F.batch_norm = what
nn.BatchNorm2d executes) runs on the GPU through torch's ATen / MIOpen ops twice on the benchmark's random-init weights and
batch: in fp32, and under ``torch.autocast(bfloat16)`` -- what the reference does when Lightning is given ``--precision bf16``.
Prints the same three numbers bench.py reports for the HIP path, plus the HIP path's own.  Test infrastructure only (it is
the oracle that runs here, on torch ops); nothing in the product calls it.

    python tools/torch_autocast_yardstick.py [B]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402  (puts the package on sys.path)
import cvcl_oracle as O  # noqa: E402


def deviation(a, b, la, lb):
    a, b = a.double(), b.double()
    return {"logits_rel_vs_fp32": float((a - b).abs().max() / b.abs().max()),
            "logits_cosine_vs_fp32": float(torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0)),
            "loss_abs_vs_fp32": abs(la - lb)}


def yardstick(lit, batch):
    """-> the three deviations of torch autocast(bf16) from torch fp32, oracle forward, same weights and batch."""
    p = {k: v.detach().clone() for k, v in lit.model.state_dict().items()}
    p["logit_neg_log_temperature"] = lit.model.logit_neg_log_temperature.detach().to(batch[0].device).float()
    kw = dict(normalize_features=True, training=True, bn_impl="torch")
    def loss(lpi):
        gt = torch.arange(lpi.shape[0], device=lpi.device)
        lpi = lpi.float()
        return float((torch.nn.functional.cross_entropy(lpi, gt) + torch.nn.functional.cross_entropy(lpi.t(), gt)) / 2)
    with torch.no_grad():
        ref = O.cvcl_forward(p, batch[0], batch[1], batch[2], **kw)[0]
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got = O.cvcl_forward(p, batch[0], batch[1], batch[2], **kw)[0]
    return deviation(got.float(), ref.float(), loss(got), loss(ref)), ref.float()


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda:0")
    lit, ve, _ = bench.build_model("c2", dev, "bf16")
    batch = bench.synthetic_batch_on_device(B, seed=0, device=dev)
    y, ref_logits = yardstick(lit, batch)
    print("torch autocast(bf16) vs torch fp32 :", {k: float(f"{v:.4g}") for k, v in y.items()}, flush=True)
    r = bench.logits_vs_fp32(lit, batch, "bf16")
    print("HIP bf16 vs HIP fp32               :", {k: float(f"{v:.4g}") for k, v in r.items()}, flush=True)
    lit.set_precision("32")
    with torch.no_grad():
        li, _ = lit.model(batch[0], batch[1], batch[2])
    print("HIP fp32 vs torch fp32 (oracle on the GPU): logits max-rel",
          float((li.double() - ref_logits.double()).abs().max() / ref_logits.double().abs().max()))


if __name__ == "__main__":
    main()
