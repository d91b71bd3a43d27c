#!/bin/bash
# SQ counters of GEMM lab variants (one --pmc pass per counter group; kernel-trace only).  usage: pmc.sh "<variant M N K>" ...
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_lab
rm -rf $OUT; mkdir -p $OUT
i=0
for cfg in "$@"; do
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  (cd $ROOT && rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- $ROOT/tools/gemm_lab/lab $cfg 3 0 0 > $OUT/g$i.log 2>&1)
  echo "$cfg" > $OUT/g$i.cfg
done
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/pmc_lab"
res = collections.OrderedDict()
for d in sorted(glob.glob(out + "/g*/"), key=lambda p: int(p.rstrip("/").split("g")[-1])):
    cfg = open(d.rstrip("/") + ".cfg").read().strip()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"] and "ref_rows" not in r["Kernel_Name"]]
        agg = collections.defaultdict(list)
        for r in rows:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        res.setdefault(cfg, {}).update({c: round(sum(v) / len(v)) for c, v in agg.items()})
for cfg, c in res.items():
    print(cfg)
    wc = c.get("SQ_WAVE_CYCLES", 1)
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_VALU"):
        if k in c: print(f"   {k:24s} {c[k]:14d}  {c[k]/wc*100:6.1f} % of wave cycles")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
        print(f"   MFMA busy / SQ busy: {c['SQ_VALU_MFMA_BUSY_CYCLES']} / {c['SQ_BUSY_CYCLES']}  (GRBM_GUI_ACTIVE {c.get('GRBM_GUI_ACTIVE')})")
    for k in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VALU"):
        if k in c: print(f"   {k:24s} {c[k]:14d}")
PY
