#!/bin/bash
# kernel trace of a short C2 bench run with one trunk pass in flight -> per-layer GEMM table (tools/gemm_layers.py)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_layers
rm -rf $O; mkdir -p $O
export CVCL_TRUNK_STREAMS=1
rocprofv3 --kernel-trace --stats -d $O -o c2 --output-format csv -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-parity --no-roofline > $O/c2.log 2>&1
python3 $R/tools/gemm_layers.py $(ls $O/*kernel_trace.csv $O/*/*kernel_trace.csv 2>/dev/null | head -1) $O/gemm_per_layer.csv
