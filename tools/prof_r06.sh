#!/bin/bash
# Round-6 profile refresh (run from the repo root on the GPU box; everything lands under gpurun_out/r06/; the kept summaries are
# copied into profiles/r06_* by tools/keep_r06.sh on the build side):
#   * rocprofv3 --kernel-trace --stats of the default bench command (C2, two trunk passes in flight) and with ONE trunk pass in flight;
#   * the same (one trunk stream) for C4, C4 at patch 14, C5; per-layer GEMM tables;
#   * the trainable tail on its own (tools/tail_bench.py) for C2 and C4: the complete launch list;
#   * PMC: HBM traffic of the C2 step (pmc_bench.sh) and SQ + HBM counters per kernel for C4 / C5 (pmc_cfg.sh);
#   * the vendor-library comparison (tools/blaslt_compare.py).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-extras"
rocprofv3 --kernel-trace --stats -d $O/c2 -o c2 --output-format csv -- python3 $R/bench.py $B > $O/c2.log 2>&1
CVCL_TRUNK_STREAMS=1 rocprofv3 --kernel-trace --stats -d $O/c2_1s -o c2 --output-format csv -- python3 $R/bench.py $B > $O/c2_1s.log 2>&1
for c in c4 c4p14 c5; do
  CVCL_VIT_TRUNK_STREAMS=1 rocprofv3 --kernel-trace --stats -d $O/$c -o $c --output-format csv -- python3 $R/bench.py --config $c $B > $O/$c.log 2>&1
done
for c in c2 c4; do
  rocprofv3 --kernel-trace --stats -d $O/tail_$c -o tail --output-format csv -- python3 $R/tools/tail_bench.py $c 10 > $O/tail_$c.log 2>&1
done
cd $R
for c in c2 c4 c5; do bash tools/prof_layers.sh $c > $O/layers_$c.log 2>&1; done
cp gpurun_out/prof_layers_c2/gemm_per_layer.csv gpurun_out/prof_layers_c4/gemm_per_layer_vit.csv gpurun_out/prof_layers_c5/gemm_per_layer_vit_fp8.csv $O/ 2>/dev/null
GRAFT_REPO_ROOT=$R bash tools/pmc_bench.sh > $O/pmc.log 2>&1
python3 tools/pmc_summary.py r06 3 > $O/pmc_summary.log 2>&1
cp profiles/r06_pmc_hbm_traffic.json $O/ 2>/dev/null
for c in c4 c5; do GRAFT_REPO_ROOT=$R bash tools/pmc_cfg.sh $c > $O/pmc_$c.log 2>&1; cp gpurun_out/pmc_$c/summary.txt $O/pmc_${c}_summary.txt; cp gpurun_out/pmc_$c/summary.json $O/pmc_${c}_summary.json; done
python3 tools/blaslt_compare.py > $O/blaslt_compare.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
ls $O | head -40; tail -2 $O/pmc_summary.log; tail -14 $O/blaslt_compare.txt
