"""Summarise tools/pmc_mem.sh: per run and kernel family, mean duration and mean counter values per dispatch."""
import collections, csv, glob, sys
FAM = ("gemm8w_kernel<7, 0", "gemm8w_kernel<8, 0", "gemm_pro", "gemm_glds", "gconv_mfma", "bn_add_relu", "bn_relu_apply", "lab_add")
only = sys.argv[1] if len(sys.argv) > 1 else None
for f in sorted(glob.glob("gpurun_out/pmc_mem/g*/*/**/*counter_collection.csv", recursive=True)):
    run = f.split("/")[3]
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
    for r in rows:
        name = r["Kernel_Name"]
        key = next((k for k in FAM if k in name), None)
        if key is None or (only and only not in key):
            continue
        if key.startswith("gemm8w") and run == "bench" and "50176" not in r.get("Grid_Size", "50176"):
            pass
        a = agg[key][r["Counter_Name"]]
        a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; a[2] += float(r["Counter_Value"])
    for k in sorted(agg):
        parts = []
        for c, (n, us, v) in sorted(agg[k].items()):
            parts.append(f"{c} {v/n:14.0f}")
        n, us, _ = next(iter(agg[k].values()))
        print(f"{run:6s} {k:20s} n {n:4d} mean {us/n:7.1f} us | " + " | ".join(parts))
