cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_c4
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_c4 -o c4 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_c4.py --steps 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_c4.log 2>&1
python3 - <<'PY'
import csv, os, collections
rows = list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_c4/c4_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
g = [r for r in rows if "gemm_glds" in r["Kernel_Name"]]
last = g[-49:]
agg = collections.OrderedDict()
for r in last:
    k = (r["Grid_Size_X"], r["Grid_Size_Y"], r["Kernel_Name"][-40:])
    agg.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    print(k, len(v), round(sum(v) / len(v), 1), "us")
PY
