#!/bin/bash
# which hipBLASLt kernels (tile configuration is in the kernel name) the vendor library picks for the MFMA-bound shapes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_blaslt
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o bl --output-format csv -- python3 $R/tools/blaslt_compare.py > $O/run.log 2>&1
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_blaslt/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "Cijk" in r["Name"] or "gemm" in r["Name"].lower():
        print(r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["Name"][:400])
PY
