#!/bin/bash
# Round-3 profile refresh (run from the repo root on the GPU box; everything lands under gpurun_out/r03/, the summaries that
# are kept are then copied into profiles/r03_* by tools/keep_profiles.sh on the build side):
#   * rocprofv3 --kernel-trace --stats of the default bench command (C2, two trunk passes in flight) and of the same command
#     with ONE trunk pass in flight (CVCL_TRUNK_STREAMS=1: the kernels' own durations -- what bench.py's roofline pass times);
#   * the same for C4 and C5;  per-layer GEMM tables;  the bench lines printed under the tracer;
#   * PMC HBM traffic of the C2 step (FETCH_SIZE / WRITE_SIZE in separate passes);
#   * the RCCL world-1 worker under the tracer (which RCCL / copy kernels a one-rank process group launches).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-extras"
rocprofv3 --kernel-trace --stats -d $O/c2 -o c2 --output-format csv -- python3 $R/bench.py $B > $O/c2.log 2>&1
CVCL_TRUNK_STREAMS=1 rocprofv3 --kernel-trace --stats -d $O/c2_1s -o c2 --output-format csv -- python3 $R/bench.py $B > $O/c2_1s.log 2>&1
CVCL_VIT_TRUNK_STREAMS=1 rocprofv3 --kernel-trace --stats -d $O/c4 -o c4 --output-format csv -- python3 $R/bench.py --config c4 $B > $O/c4.log 2>&1
CVCL_VIT_TRUNK_STREAMS=1 rocprofv3 --kernel-trace --stats -d $O/c5 -o c5 --output-format csv -- python3 $R/bench.py --config c5 $B > $O/c5.log 2>&1
cd $R && bash tools/prof_layers.sh c2 > $O/layers_c2.log 2>&1; cp gpurun_out/prof_layers_c2/gemm_per_layer.csv $O/ 2>/dev/null
cd $R && bash tools/prof_layers.sh c4 > $O/layers_c4.log 2>&1; cp gpurun_out/prof_layers_c4/gemm_per_layer_vit.csv $O/ 2>/dev/null
cd $R && bash tools/prof_layers.sh c5 > $O/layers_c5.log 2>&1; cp gpurun_out/prof_layers_c5/gemm_per_layer_vit_fp8.csv $O/ 2>/dev/null
cd $R && GRAFT_REPO_ROOT=$R bash tools/pmc_bench.sh > $O/pmc.log 2>&1
cd $R && python3 tools/pmc_summary.py r03 4 > $O/pmc_summary.log 2>&1; cp profiles/r03_pmc_hbm_traffic.json $O/ 2>/dev/null
mkdir -p $O/w1out
cd /tmp && MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 rocprofv3 --kernel-trace --stats -d $O/rccl_w1 -o w1 --output-format csv -- python3 $R/tests/dist_worker.py rccl_w1 $O/w1out > $O/rccl_w1.log 2>&1
rm -rf $O/w1out
cd $R && python3 tools/torch_autocast_yardstick.py 256 > $O/yardstick.log 2>&1
cd $R && python3 tools/centre_ab.py 256 > $O/centre_ab.log 2>&1
find $O -name "*kernel_trace.csv" -size +8M -delete          # (the traces are large; the stats and the per-layer tables are what is kept)
ls -la $O $O/c2 | head -60
tail -2 $O/pmc_summary.log; tail -3 $O/yardstick.log; tail -3 $O/centre_ab.log
