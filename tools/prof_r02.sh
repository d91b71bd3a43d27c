#!/bin/bash
# round-2 profile refresh (run from the repo root on the GPU box): rocprofv3 --kernel-trace --stats of the three bench configs
# + per-layer GEMM table + PMC HBM traffic of the C2 step.  Summaries land under gpurun_out/; copy the ones to keep into profiles/.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/c2 -o c2 --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/c2.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/c4 -o c4 --output-format csv -- python3 $R/bench.py --config c4 --steps 6 --warmup 2 --no-parity > $O/c4.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/c5 -o c5 --output-format csv -- python3 $R/bench.py --config c5 --steps 6 --warmup 2 --no-parity > $O/c5.log 2>&1
cd $R && bash tools/prof_layers.sh c2 > $O/layers.log 2>&1
cd $R && bash tools/prof_layers.sh c4 > $O/layers_c4.log 2>&1
cd $R && bash tools/prof_layers.sh c5 > $O/layers_c5.log 2>&1
cd $R && GRAFT_REPO_ROOT=$R bash tools/pmc_bench.sh > $O/pmc.log 2>&1
cd $R && python3 tools/pmc_summary.py r02 3 > $O/pmc_summary.log 2>&1
tail -3 $O/c2.log | cut -c1-300; tail -50 $O/layers.log; tail -20 $O/layers_c4.log; tail -20 $O/layers_c5.log; cat $O/pmc_summary.log
