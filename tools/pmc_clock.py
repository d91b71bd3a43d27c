"""Summarise tools/pmc_clock.sh: per kernel family, mean duration, mean GRBM_GUI_ACTIVE and their ratio (cycles per us = MHz, up to the
number of XCDs the counter is summed over)."""
import collections, csv, glob
for run in ("bench", "lab0", "lab5"):
    fs = glob.glob(f"gpurun_out/pmc_clk/{run}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(run, "no counter file"); continue
    rows = list(csv.DictReader(open(fs[0])))
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in rows:
        if r.get("Counter_Name") != "GRBM_GUI_ACTIVE":
            continue
        name = r["Kernel_Name"]
        key = next((k for k in ("gemm8w_kernel<7, 0", "gemm8w_kernel<8, 0", "gemm_pro", "gemm_glds", "gconv_mfma", "bn_add_relu", "bn_relu_apply",
                                "lab_add") if k in name), None)
        if key is None:
            key = "gemm8w (mangled)" if "gemm8w" in name else None
        if key is None:
            continue
        if key.startswith("gemm8w") and run == "bench":
            pass
        a = agg[key]
        a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; a[2] += float(r["Counter_Value"])
    for k, (n, us, cyc) in sorted(agg.items()):
        print(f"{run:6s} {k:22s} n {n:4d}  mean {us/n:8.1f} us  GRBM_GUI_ACTIVE {cyc/n:12.0f}  -> {cyc/us:8.1f} cycles/us")
