#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py configuration (run on the GPU box from the repo root):
#   bash tools/prof_bench.sh c2|c4|c5 [extra bench.py flags]  ->  gpurun_out/prof_<cfg>/<cfg>_kernel_stats.csv + the bench line
# The summaries that are kept are copied to profiles/<round>_bench_<cfg>_kernel_stats.csv (tracked).
CFG=${1:-c2}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$CFG
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o $CFG --output-format csv -- python3 $R/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-extras "$@" > $O/bench.log 2>&1
find $O -name "*kernel_stats.csv" | head -3
tail -1 $O/bench.log | cut -c1-200
