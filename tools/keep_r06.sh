#!/bin/bash
# copy the summaries of tools/prof_r06.sh (gpurun_out/r06) into profiles/r06_* (tracked)
O=gpurun_out/r06; P=profiles
for c in c2 c2_1s c4 c4p14 c5 tail_c2 tail_c4; do f=$(ls $O/$c/*kernel_stats.csv $O/$c/*/*kernel_stats.csv 2>/dev/null | head -1); n=bench_$c; [ $c = c2_1s ] && n=bench_c2_1stream; [ $c = tail_c2 ] && n=tail_c2; [ $c = tail_c4 ] && n=tail_c4; [ -n "$f" ] && cp $f $P/r06_${n}_kernel_stats.csv; done
cp $O/gemm_per_layer.csv $P/r06_gemm_per_layer.csv 2>/dev/null
cp $O/gemm_per_layer_vit.csv $P/r06_gemm_per_layer_vit.csv 2>/dev/null
cp $O/gemm_per_layer_vit_fp8.csv $P/r06_gemm_per_layer_vit_fp8.csv 2>/dev/null
for c in c4 c5; do cp $O/pmc_${c}_summary.txt $P/r06_pmc_${c}_summary.txt 2>/dev/null; cp $O/pmc_${c}_summary.json $P/r06_pmc_${c}_summary.json 2>/dev/null; done
python3 tools/pmc_summary.py r06 > /dev/null 2>&1   # from gpurun_out/pmc (passes counted from the stem launches)
cp $O/blaslt_compare.txt $P/r06_blaslt_compare.txt 2>/dev/null
for c in c2 c4 c4p14 c5; do grep -h '^{"metric"' $O/$c.log | tail -1 > $P/r06_bench_line_$c.json 2>/dev/null; done   # (the line bench.py printed under rocprofv3: step times there include the profiler)
ls -la $P | grep r06
