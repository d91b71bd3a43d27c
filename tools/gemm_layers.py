"""Per-layer table of the conv GEMM launches of the last profiled C2 step from a rocprofv3 --kernel-trace CSV
(tools/prof_layers.sh): per block conv1, conv3 (layers 1-2: the Gram kernel that stands in for its statistics, csrc/bn_gram.hip),
[downsample (layer1.0: Gram kernel, the product is recomputed inside the tail)], [layers 1-2: fused tail pass]."""
import csv, os, sys
trace = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
fused = int(os.environ.get("CVCL_FUSED_TAIL_STAGES", "2"))
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def is_conv_gemm(n):
    return ("gemm_glds" in n or "gemm8w" in n or "gemm_pro" in n or "gram_pro" in n or ("gemm_kernel" in n and "DF16b" in n))
names = []
inpl, h = 64, 56
for stage, blocks in enumerate((3, 4, 6, 3)):
    planes = 64 << stage
    width, outc = planes * 2, planes * 4
    for bi in range(blocks):
        stride = 2 if (stage > 0 and bi == 0) else 1
        ho = h // stride
        B = 256
        ds_first = bi == 0                                       # the downsample branch (layer1.0: its Gram launch) runs at the top of the block
        if ds_first:
            names.append((f"layer{stage+1}.{bi}.downsample", B * ho * ho, outc, inpl, 1))
        names.append((f"layer{stage+1}.{bi}.conv1", B * h * h, width, inpl, 1))
        names.append((f"layer{stage+1}.{bi}.conv3" + (".stats" if stage < fused else ""), B * ho * ho, outc, width, 0 if stage < fused else 1))
        if bi == 0 and not ds_first:
            names.append((f"layer{stage+1}.{bi}.downsample", B * ho * ho, outc, inpl, 1))
        if stage < fused:
            names.append((f"layer{stage+1}.{bi}.conv3.tail", B * ho * ho, outc, width, 2))
        h, inpl = ho, outc
g = [r for r in rows if is_conv_gemm(r["Kernel_Name"])][-len(names):]
assert len(g) == len(names), (len(g), len(names))
lines = ["layer,M,N,K,kernel,grid,duration_us,algorithmic_GB_per_s,TFLOP_per_s,ideal_us(max(bytes/5.5TBps,flops/1.2PF))"]
tot = ideal_tot = 0
for (nm, M, N, K, outs), r in zip(names, g):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
    kn = r["Kernel_Name"]
    gram = "gram_pro" in kn                          # reads the operand once, upper-triangle tiles of A^T A (+ the column sums)
    by = 2 * M * K if gram else 2 * (M * K + N * K + outs * M * N)
    fl = M * K * (K + 32) if gram else 2 * M * N * K
    ideal = max(by / 5.5e6, fl / 1.2e9)
    tot += d; ideal_tot += ideal
    k = "gram_pro" if gram else "gemm8w" if "gemm8w" in kn else "gemm_pro" if "gemm_pro" in kn else "gemm_glds" if "gemm_glds" in kn else "gemm_reg"
    lines.append(f"{nm},{M},{N},{K},{k},{int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']},{d:.1f},{by/d/1e3:.0f},{fl/d/1e6:.0f},{ideal:.0f}")
print("\n".join(lines))
print(f"total {tot:.0f} us; ideal {ideal_tot:.0f} us")
# all kernel classes of that step, for the record
step_start = int(g[0]["Start_Timestamp"])
agg = {}
for r in rows:
    if int(r["Start_Timestamp"]) >= step_start:
        n = r["Kernel_Name"].split("(")[0][:60]
        a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {t:9.1f} us  x{c:4d}  {n}")
if out:
    open(out, "w").write("\n".join(lines) + "\n")
