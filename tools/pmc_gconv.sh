#!/bin/bash
# SQ / memory counters for the grouped-conv kernel (tools/gconv_ablate.py shapes); one --pmc pass per counter group.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gconv
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gconv_ablate.py > $OUT/g$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_gconv"
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "gconv_mfma" in r["Kernel_Name"]]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[(r["Grid_Size"], r.get("LDS_Block_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(os.path.basename(os.path.dirname(os.path.dirname(f))), k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
