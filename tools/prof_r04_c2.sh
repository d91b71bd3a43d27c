#!/bin/bash
# C2-only part of tools/prof_r04.sh (after a change to the ResNeXt trunk): kernel stats with two / one trunk pass in flight, the
# per-layer GEMM table, the HBM-traffic PMC passes.  tools/keep_r04.sh copies the summaries into profiles/r04_*.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04
mkdir -p $O; rm -rf $O/c2 $O/c2_1s
cd /tmp && export TMPDIR=/tmp
B="--steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-extras"
rocprofv3 --kernel-trace --stats -d $O/c2 -o c2 --output-format csv -- python3 $R/bench.py $B > $O/c2.log 2>&1
CVCL_TRUNK_STREAMS=1 rocprofv3 --kernel-trace --stats -d $O/c2_1s -o c2 --output-format csv -- python3 $R/bench.py $B > $O/c2_1s.log 2>&1
cd $R
bash tools/prof_layers.sh c2 > $O/layers_c2.log 2>&1
cp gpurun_out/prof_layers_c2/gemm_per_layer.csv $O/ 2>/dev/null
GRAFT_REPO_ROOT=$R bash tools/pmc_bench.sh > $O/pmc.log 2>&1
python3 tools/pmc_summary.py r04 4 > $O/pmc_summary.log 2>&1
find $O -name "*kernel_trace.csv" -delete
tail -3 $O/pmc_summary.log; tail -3 $O/c2.log | cut -c1-400
