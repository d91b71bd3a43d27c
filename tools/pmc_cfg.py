"""Summarise tools/pmc_cfg.sh <cfg>: per kernel (template arguments kept) the mean duration, the mean SQ counters per dispatch and
the HBM bytes per dispatch (FETCH_SIZE x 2: gfx950 counts a wide coalesced read at half its bytes, MI355X_MICROARCH.md; WRITE_SIZE as
is; both in KiB).  Derived lines: MFMA-pipe time = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / 2.4 GHz; VALU issue time = SQ_INSTS_VALU x 4
cycles / 1024 / 2.4 GHz; waiting = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES."""
import collections, csv, glob, json, re, sys
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
root = f"gpurun_out/pmc_{cfg}"
def key_of(r):
    name = r["Kernel_Name"]
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", name)
    return (m.group(1) if m else name[:48]) + " g" + r.get("Grid_Size", "")
tab = collections.defaultdict(dict)
dur = collections.defaultdict(lambda: [0, 0.0])
for f in sorted(glob.glob(f"{root}/*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    seen = set()
    for r in csv.DictReader(open(f)):
        key = key_of(r)
        a = agg[(key, r["Counter_Name"])]
        a[0] += 1; a[1] += float(r["Counter_Value"])
        did = r.get("Dispatch_Id")
        if (did, key) not in seen:
            seen.add((did, key))
            d = dur[key]; d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for (key, c), (n, v) in agg.items():
        tab[key][c] = v / n
out = {}
order = sorted(tab, key=lambda k: -dur[k][1])
for key in order:
    n, us = dur[key]
    t = tab[key]
    mean_us = us / max(n, 1)
    if us / 6 < 20:                    # (six passes) kernels under 20 us in total: noise
        continue
    print(f"{key}  ({n // 6} dispatches per pass, mean {mean_us:.1f} us under the counters)")
    for c, v in sorted(t.items()):
        print(f"    {c:32s} {v:16.0f}")
    d = {}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in t:
        d["mfma_pipe_us"] = t["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / 2400
    if "SQ_INSTS_VALU" in t:
        d["valu_issue_us_at_4cyc"] = t["SQ_INSTS_VALU"] * 4 / 1024 / 2400
        if t.get("SQ_INSTS_MFMA"):
            d["valu_per_mfma"] = t["SQ_INSTS_VALU"] / t["SQ_INSTS_MFMA"]
    if t.get("SQ_WAVE_CYCLES"):
        d["wait_inst_frac"] = t.get("SQ_WAIT_INST_ANY", 0) / t["SQ_WAVE_CYCLES"]
        d["wait_any_frac"] = t.get("SQ_WAIT_ANY", 0) / t["SQ_WAVE_CYCLES"]
    if t.get("SQ_VALU_MFMA_BUSY_CYCLES") and "SQ_VALU_MFMA_COEXEC_CYCLES" in t:
        d["coexec_of_mfma_busy"] = t["SQ_VALU_MFMA_COEXEC_CYCLES"] / t["SQ_VALU_MFMA_BUSY_CYCLES"]
    if "FETCH_SIZE" in t or "WRITE_SIZE" in t:
        d["hbm_read_MB_x2"] = t.get("FETCH_SIZE", 0) * 2048 / 1e6
        d["hbm_write_MB"] = t.get("WRITE_SIZE", 0) * 1024 / 1e6
    d["mean_us"] = mean_us
    d["dispatches_per_pass"] = n // 6
    print("    -> " + ", ".join(f"{k} {v:.3g}" for k, v in d.items()))
    out[key] = d
json.dump(out, open(f"{root}/summary.json", "w"), indent=1)
