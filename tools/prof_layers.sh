#!/bin/bash
# kernel trace of a short bench run with one trunk pass in flight -> per-layer GEMM table
#   tools/prof_layers.sh [c2|c4|c5]     (c2: tools/gemm_layers.py, c4 / c5: tools/vit_layers.py)
CFG=${1:-c2}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_layers_$CFG
rm -rf $O; mkdir -p $O
export CVCL_TRUNK_STREAMS=1 CVCL_VIT_TRUNK_STREAMS=1
rocprofv3 --kernel-trace --stats -d $O -o $CFG --output-format csv -- python3 $R/bench.py --config $CFG --steps 6 --warmup 3 --no-cpu-baseline --no-parity --no-roofline --no-extras > $O/$CFG.log 2>&1
T=$(ls $O/*kernel_trace.csv $O/*/*kernel_trace.csv 2>/dev/null | head -1)
if [ "$CFG" = c2 ]; then python3 $R/tools/gemm_layers.py $T $O/gemm_per_layer.csv
elif [ "$CFG" = c5 ]; then python3 $R/tools/vit_layers.py $T $O/gemm_per_layer_vit_fp8.csv 1
else python3 $R/tools/vit_layers.py $T $O/gemm_per_layer_vit.csv 2; fi
