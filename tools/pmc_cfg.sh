#!/bin/bash
# Round 4: SQ (issue / wait / pipe-busy) and HBM (FETCH_SIZE / WRITE_SIZE) counters per kernel of one bench configuration
# (c2 | c4 | c5 | c4p14), one trunk stream.  One --pmc pass per counter group; only --kernel-trace is combined with --pmc.
#   bash tools/pmc_cfg.sh c4     -> gpurun_out/pmc_c4/{g1..g4,FETCH_SIZE,WRITE_SIZE}, summary gpurun_out/pmc_c4/summary.txt
CFG=${1:-c4}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$CFG
rm -rf $OUT; mkdir -p $OUT
ARGS="--config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-extras"
export CVCL_TRUNK_STREAMS=1 CVCL_VIT_TRUNK_STREAMS=1
i=0
for grp in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/g$i.log 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/$c -o pmc --output-format csv -- python3 $R/bench.py $ARGS > $OUT/$c.log 2>&1
done
cd $R && python3 tools/pmc_cfg.py $CFG > $OUT/summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +6M -delete
tail -5 $OUT/summary.txt
