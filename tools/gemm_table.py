"""Per-layer table of the 43 conv GEMM launches of the last profiled step (from tools/prof_bench.sh output):
per block conv1, conv3 (layers 1-2: statistics-only pass), [downsample], [layers 1-2: fused BN3+identity+ReLU tail pass]."""
import csv, sys
trace = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof1/r01_kernel_trace.csv"
out = sys.argv[2] if len(sys.argv) > 2 else None
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
g = [r for r in rows if "gemm_glds" in r["Kernel_Name"] or ("gemm_kernel" in r["Kernel_Name"] and ("DF16b" in r["Kernel_Name"] or "_Accum" in r["Kernel_Name"]))][-43:]
names = []
inpl, h = 64, 56
for stage, blocks in enumerate((3, 4, 6, 3)):
    planes = 64 << stage
    width, outc = planes * 2, planes * 4
    for bi in range(blocks):
        stride = 2 if (stage > 0 and bi == 0) else 1
        ho = h // stride
        B = 256
        names.append((f"layer{stage+1}.{bi}.conv1", B * h * h, width, inpl, 1))
        names.append((f"layer{stage+1}.{bi}.conv3" + (".stats" if stage < 2 else ""), B * ho * ho, outc, width, 0 if stage < 2 else 1))
        if bi == 0:
            names.append((f"layer{stage+1}.{bi}.downsample", B * ho * ho, outc, inpl, 1))
        if stage < 2:
            names.append((f"layer{stage+1}.{bi}.conv3.tail", B * ho * ho, outc, width, 2))
        h, inpl = ho, outc
lines = ["layer,M,N,K,grid_x,grid_y,duration_us,algorithmic_GB_per_s,TFLOP_per_s,ideal_us(max(bytes/5.5TBps,flops/1.2PF))"]
tot = ideal_tot = 0
assert len(names) == 43 and len(g) == 43, (len(names), len(g))
for (nm, M, N, K, outs), r in zip(names, g):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
    by = 2 * (M * K + N * K + outs * M * N)          # tail pass: residual read + output write
    fl = 2 * M * N * K
    ideal = max(by / 5.5e6, fl / 1.2e9)
    tot += d
    ideal_tot += ideal
    lines.append(f"{nm},{M},{N},{K},{int(r['Grid_Size_X'])//256},{r['Grid_Size_Y']},{d:.1f},{by/d/1e3:.0f},{fl/d/1e6:.0f},{ideal:.0f}")
print("\n".join(lines))
print(f"total {tot:.0f} us; ideal {ideal_tot:.0f} us")
if out:
    open(out, "w").write("\n".join(lines) + "\n")
