import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "multimodal-baby_amd")
import bench
dev = torch.device("cuda:0")
lit, ve, _ = bench.build_model("c2", dev, "bf16")
def bn3s():
    return [m.bn3 for n, m in ve.model.named_modules() if hasattr(m, "bn3")]
for g3 in (1.0, 0.25, 0.125):
    with torch.no_grad():
        for b in bn3s(): b.weight.fill_(g3)
    for name, gen in (("noise", bench.synthetic_batch_on_device), ("structured", bench.structured_batch_on_device)):
        keep = {k: v.clone() for k, v in lit.state_dict().items() if "running_" in k or "num_batches_tracked" in k}
        evalb, calib = gen(256, 4242, dev), gen(256, 1717, dev)
        ve.model.recalibrate_centres()
        gn, lit.model.global_negatives = lit.model.global_negatives, False
        with torch.no_grad(): lit.model(calib[0], calib[1], calib[2])
        lit.load_state_dict(keep, strict=False)
        r = bench.logits_vs_fp32(lit, evalb, "bf16")
        ty, _ = bench.torch_yardstick(lit, evalb)
        lit.model.global_negatives = gn
        lit.load_state_dict(keep, strict=False)
        print(f"gamma3 {g3} {name}: HIP {r['logits_rel_vs_fp32']:.4f} cos {r['logits_cosine_vs_fp32']:.5f} | autocast {ty['logits_rel']:.4f} cos {ty['logits_cosine']:.5f}", flush=True)
