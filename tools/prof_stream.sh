cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_stream
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/host_ahead.py stream > $O/log.txt 2>&1
python3 - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_stream/s_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(rows[0].keys())
stems = [i for i, r in enumerate(rows) if "stem_mfma" in r["Kernel_Name"]]
a, b = stems[10], stems[12]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    n = r["Kernel_Name"]
    if any(k in n for k in ("gemm_glds", "gconv", "bn_", "maxpool", "avgpool")) and "stem" not in n:
        continue
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:7.1f}  q{r.get("Queue_Id", "?")} {n[:70]}')
PY
