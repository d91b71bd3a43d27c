#!/bin/bash
# copy the summaries of tools/prof_r04.sh (gpurun_out/r04) into profiles/r04_* (tracked)
O=gpurun_out/r04; P=profiles
for c in c2 c2_1s c4 c4p14 c5; do f=$(ls $O/$c/*kernel_stats.csv $O/$c/*/*kernel_stats.csv 2>/dev/null | head -1); n=$c; [ $c = c2_1s ] && n=c2_1stream; [ -n "$f" ] && cp $f $P/r04_bench_${n}_kernel_stats.csv; done
cp $O/gemm_per_layer.csv $P/r04_gemm_per_layer.csv 2>/dev/null
cp $O/gemm_per_layer_vit.csv $P/r04_gemm_per_layer_vit.csv 2>/dev/null
cp $O/gemm_per_layer_vit_fp8.csv $P/r04_gemm_per_layer_vit_fp8.csv 2>/dev/null
for c in c4 c5 c4p14; do cp $O/pmc_${c}_summary.txt $P/r04_pmc_${c}_summary.txt; cp $O/pmc_${c}_summary.json $P/r04_pmc_${c}_summary.json; done
cp $O/blaslt_compare.txt $P/r04_blaslt_compare.txt 2>/dev/null
cp $O/yardstick.log $P/r04_parity_yardstick.txt 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do f=$(ls gpurun_out/pmc/$c/*counter_collection.csv gpurun_out/pmc/$c/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python3 - "$f" $P/r04_pmc_${c}_counter_collection.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("gemm", "gconv", "bn_", "stem", "avgpool", "gram"))]
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=["Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"], extrasaction="ignore")
w.writeheader(); w.writerows(keep)
PY
done
ls -la $P | grep r04
