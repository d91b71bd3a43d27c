"""Per-layer table of the ViT-B/16 trunk launches of the last profiled C4 / C5 step from a rocprofv3 --kernel-trace CSV
(tools/prof_layers.sh c4|c5): patch embedding, then per block qkv, attention, proj, fc1, fc2 (+ the LayerNorms)."""
import csv, sys
trace = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
opb = int(sys.argv[3]) if len(sys.argv) > 3 else 2              # operand bytes: 2 = bf16, 1 = e4m3
patch = int(sys.argv[4]) if len(sys.argv) > 4 else 16
B, T, D, MLP, DEPTH = 256, (224 // patch) ** 2 + 1, 768, 3072, 12
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def is_trunk_gemm(n):
    return ("gemm_glds" in n or "gemm8w" in n or "gemm8f" in n or "gemm_fp8" in n or "fp8" in n and "gemm" in n) and "f32_small" not in n
M = B * T
shapes = [("patch_embed", B * (T - 1), D, (3 * patch * patch + 7) // 8 * 8)]
for i in range(DEPTH):
    shapes += [(f"blocks.{i}.qkv", M, 3 * D, D), (f"blocks.{i}.proj", M, D, D), (f"blocks.{i}.fc1", M, MLP, D), (f"blocks.{i}.fc2", M, D, MLP)]
g = [r for r in rows if is_trunk_gemm(r["Kernel_Name"])][-len(shapes):]
assert len(g) == len(shapes), (len(g), len(shapes))
peak = 1.2e9 if opb == 2 else 2.4e9                              # flop per us at the 8-wave kernel's measured 4096^3 rate (bf16)
lines = ["layer,M,N,K,kernel,grid,duration_us,TFLOP_per_s,algorithmic_GB_per_s"]
tot = fl_tot = 0
for (nm, m, n, k), r in zip(shapes, g):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
    fl = 2 * m * n * k
    by = opb * (m * k + n * k) + 2 * m * n
    tot += d; fl_tot += fl
    kn = r["Kernel_Name"]
    kname = ("gemm8w" if "gemm8w" in kn else "gemm8f" if "gemm8f" in kn else "gemm_fp8" if "gemm_fp8" in kn else "gemm_glds" if "gemm_glds" in kn
             else kn.split("(")[0].split("::")[-1][:24])
    lines.append(f"{nm},{m},{n},{k},{kname},{int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']},{d:.1f},{fl/d/1e6:.0f},{by/d/1e3:.0f}")
print("\n".join(lines))
print(f"GEMM total {tot:.0f} us, {fl_tot/tot/1e6:.0f} TFLOP/s average")
step_start = int(g[0]["Start_Timestamp"])
agg = {}
for r in rows:
    if int(r["Start_Timestamp"]) >= step_start:
        n = r["Kernel_Name"].split("(")[0][-60:]
        a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {t:9.1f} us  x{c:4d}  {n}")
if out:
    open(out, "w").write("\n".join(lines) + "\n")
