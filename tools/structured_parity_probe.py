"""The well-conditioned parity point (bench.structured_parity) for the library selected with CVCL_HIP_LIB and the lab switches in the
environment: one line with HIP bf16 vs fp32 and torch autocast vs torch fp32."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
dev = torch.device("cuda:0")
lit, ve, _opt = bench.build_model("c2", dev, "bf16")
r = bench.structured_parity(lit, ve, bench.PER_GPU_BATCH, dev)
print(os.environ.get("TAG", ""), "HIP", round(r["logits_rel_vs_fp32"], 5), "cos", round(r["logits_cosine_vs_fp32"], 6),
      "| autocast", round(r["torch_autocast_bf16_vs_torch_fp32"]["logits_rel"], 5))
