"""A/B of centred storage on the benchmarked configuration (C2, B = 256, bf16, train-mode BatchNorm): logits of the bf16 mode
against the exact-fp32 parity mode with (a) plain storage, (b) centres calibrated on the evaluated batch itself, (c) centres
calibrated on ANOTHER batch of the same distribution (the steady state of a training run).  python tools/centre_ab.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda:0")
    lit, ve, _ = bench.build_model("c2", dev, "bf16")
    batch = bench.synthetic_batch_on_device(B, seed=0, device=dev)
    other = bench.synthetic_batch_on_device(B, seed=977, device=dev)
    net = ve.model
    for name in ("plain", "calibrated on the batch itself", "calibrated on another batch"):
        os.environ["CVCL_CENTRED_STORAGE"] = "0" if name == "plain" else "1"
        net.recalibrate_centres()
        if name.endswith("another batch"):
            keep = {k: v.clone() for k, v in lit.state_dict().items() if "running_" in k or "num_batches_tracked" in k}
            lit.set_precision("bf16")
            with torch.no_grad():
                lit.model(other[0], other[1], other[2])
            lit.load_state_dict(keep, strict=False)
        r = bench.logits_vs_fp32(lit, batch, "bf16")
        print(f"{name:32s}", {k: float(f"{v:.4g}") for k, v in r.items()}, flush=True)


if __name__ == "__main__":
    main()
