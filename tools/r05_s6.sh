#!/bin/bash
# round 5, GPU session 6: the 64 x 64 split kernel for the tail's GEMMs -- tests, tail time, step-time A/B (lab switch CVCL_SPLIT64)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_s6
mkdir -p $O
cd $R
( timeout 1500 python3 -m pytest tests/test_gemm_gpu.py tests/test_head_gpu.py tests/test_text_train_gpu.py tests/test_encoders_gpu.py tests/test_train_entry_gpu.py tests/test_lm_gpu.py tests/test_spatial_gpu.py tests/test_resnext_gpu.py -m gpu --maxfail=8 -q -p no:cacheprovider 2>&1 | tail -15 ) > $O/pytest.log 2>&1
echo "pytest: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"
LAB=$R/multimodal-baby_amd/lib/libcvcl_hip_lab.so
for v in 1 0; do CVCL_HIP_LIB=$LAB CVCL_SPLIT64=$v python3 tools/tail_bench.py c4 20 > $O/tail_c4_split64_$v.json 2>/dev/null; done
for rep in 1 2; do
  for cfg in c4 c4p14 c5; do
    for v in 1 0; do
      echo "$cfg CVCL_SPLIT64=$v: $(CVCL_HIP_LIB=$LAB CVCL_SPLIT64=$v python3 bench.py --config $cfg --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-roofline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('tail_ms_per_step'))")"
    done
  done
done > $O/ab_split64.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/p_tail_c4 -o x --output-format csv -- python3 $R/tools/tail_bench.py c4 10 > $O/p_tail_c4.log 2>&1 )
f=$(find $O/p_tail_c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/tail_c4_kernel_stats.csv
rm -rf $O/p_tail_c4
python3 bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras > $O/bench_c4_parity.json 2>/dev/null
ls -la $O
