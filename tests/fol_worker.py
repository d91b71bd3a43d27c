"""Worker of tests/test_finalize_on_load_gpu.py: everything a bf16 train-mode trunk pass leaves behind, under the
$CVCL_FINALIZE_ON_LOAD setting of this process (the library reads it once) -> torch.save(argv[1])."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "multimodal-baby_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch                                   # noqa: E402


def main():
    import bench
    dev = torch.device("cuda", 0)
    out = {"env": os.environ.get("CVCL_FINALIZE_ON_LOAD", "")}
    for B, streams in ((32, 0), (256, 0), (256, 2)):
        lit, ve, _ = bench.build_model("c2", dev, "bf16", seed=5)
        lit.train()
        g = torch.Generator(device=dev).manual_seed(B)
        with torch.no_grad():
            for prm_name, prm in ve.model.named_parameters():              # non-trivial BatchNorm affines
                if prm.dim() == 1 and ("bn" in prm_name or "downsample.1" in prm_name):
                    prm.copy_(torch.rand(prm.shape, generator=g, device=dev) * 0.5 + 0.75 if prm_name.endswith("weight")
                              else torch.randn(prm.shape, generator=g, device=dev) * 0.1)
        if streams:
            ve.model.enable_trunk_stream(dev, n_streams=streams)
        x = torch.rand(B, 3, 224, 224, generator=g, device=dev)
        res = {}
        with torch.no_grad():
            for step in range(3):                                           # calibration pass + passes on both streams
                feats, fmap = ve(x + 0.01 * step)
                res[f"feats{step}"] = feats.float().cpu().clone()
                if fmap is not None:
                    res[f"fmap{step}"] = fmap.float().mean(dim=(2, 3)).cpu().clone()
        torch.cuda.synchronize()
        if streams:
            ve.model.enable_trunk_stream(dev, inputs=None)
        for k, v in ve.model.state_dict().items():
            if "running_" in k or "num_batches_tracked" in k:
                res[k] = v.detach().cpu().clone()
        out[f"B{B}_s{streams}"] = res
    torch.save(out, sys.argv[1])


if __name__ == "__main__":
    main()
