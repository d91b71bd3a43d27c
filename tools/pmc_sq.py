"""Summarise tools/pmc_sq.sh: mean duration and mean counter values per dispatch, per kernel (template arguments kept)."""
import collections, csv, glob, re, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "gemm_pro|gemm8w|gconv_mfma|gemm_glds"
tab = collections.defaultdict(dict)
dur = collections.defaultdict(lambda: [0, 0.0])
for f in sorted(glob.glob("gpurun_out/pmc_sq/g*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if not re.search(pat, name):
            continue
        m = re.search(r"(gemm_pro_kernel<[^>]*>|gemm8w_kernel<[^>]*>|gconv_mfma_kernel<[^>]*>|gemm_glds_kernel<[^>]*>|\w+_kernel)", name)
        key = (m.group(1) if m else name[:40]) + " g" + r.get("Grid_Size", "")
        a = agg[(key, r["Counter_Name"])]
        a[0] += 1; a[1] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU"):
            d = dur[key]; d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for (key, c), (n, v) in agg.items():
        tab[key][c] = v / n
for key in sorted(tab):
    n, us = dur[key]
    print(f"{key}  ({n // 4 if n else 0} launches x passes, mean {us / max(n, 1):.1f} us)")
    for c, v in sorted(tab[key].items()):
        print(f"    {c:32s} {v:16.0f}")
