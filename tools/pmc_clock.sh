#!/bin/bash
# Effective shader clock of the 8-wave GEMM inside the C2 step and in the lab: GRBM_GUI_ACTIVE (busy cycles) per dispatch over the
# dispatch's duration.  One --pmc pass each; only --kernel-trace is combined with --pmc.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_clk
rm -rf $OUT; mkdir -p $OUT
CVCL_TRUNK_STREAMS=1 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/bench -o pmc --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-extras > $OUT/bench.log 2>&1
for c in 0 5; do
  LAB_COLD=$c rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $OUT/lab$c -o pmc --output-format csv -- $R/tools/gemm_lab/lab w7b 50176 512 1024 20 0 0 > $OUT/lab$c.log 2>&1
done
cd $R && python3 tools/pmc_clock.py
