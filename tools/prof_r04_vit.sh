#!/bin/bash
# ViT part of tools/prof_r04.sh alone (after a change to the ViT kernels): kernel stats of C4 / C4 at patch 14 / C5 with one trunk
# stream, per-layer GEMM tables, SQ + HBM counters per kernel (pmc_cfg.sh).  tools/keep_r04.sh copies the summaries into profiles/r04_*.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04
mkdir -p $O; rm -rf $O/c4 $O/c4p14 $O/c5
cd /tmp && export TMPDIR=/tmp
B="--steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-extras"
for c in c4 c4p14 c5; do
  CVCL_VIT_TRUNK_STREAMS=1 rocprofv3 --kernel-trace --stats -d $O/$c -o $c --output-format csv -- python3 $R/bench.py --config $c $B > $O/$c.log 2>&1
done
cd $R
for c in c4 c5; do bash tools/prof_layers.sh $c > $O/layers_$c.log 2>&1; done
cp gpurun_out/prof_layers_c4/gemm_per_layer_vit.csv gpurun_out/prof_layers_c5/gemm_per_layer_vit_fp8.csv $O/ 2>/dev/null
for c in c4 c5 c4p14; do GRAFT_REPO_ROOT=$R bash tools/pmc_cfg.sh $c > $O/pmc_$c.log 2>&1; cp gpurun_out/pmc_$c/summary.txt $O/pmc_${c}_summary.txt; cp gpurun_out/pmc_$c/summary.json $O/pmc_${c}_summary.json; done
find $O -name "*kernel_trace.csv" -delete
ls $O | head -30; tail -3 $O/layers_c4.log
