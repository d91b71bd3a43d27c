#!/bin/bash
# SQ counters of the bf16 GEMM on one MFMA-bound shape (one --pmc pass per counter group; kernel-trace only).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm
rm -rf $OUT; mkdir -p $OUT
i=0
for shape in "50176 1024 512" "4096 4096 4096"; do
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gemm_one.py $shape 6 > $OUT/g$i.log 2>&1
  echo "$shape" > $OUT/g$i.shape
done
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_gemm"
for d in sorted(glob.glob(out + "/g*/"), key=lambda p: int(p.rstrip("/").split("g")[-1])):
    shape = open(d.rstrip("/") + ".shape").read().strip()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "gemm_glds" in r["Kernel_Name"]]
        agg = collections.defaultdict(list)
        for r in rows:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(shape, {c: round(sum(v) / len(v)) for c, v in agg.items()})
PY
