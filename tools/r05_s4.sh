#!/bin/bash
# round 5, GPU session 4: the e4m3 kernels after the scratch fix -- tests, C5 fold A/B with per-kernel times, fp8 GEMM bench
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_s4
mkdir -p $O
cd $R
( timeout 1200 python3 -m pytest tests/test_resnext_gpu.py tests/test_c2_parity_gpu.py tests/test_gemm8f_gpu.py tests/test_gemm_gpu.py "tests/test_encoders_gpu.py::test_vit_fp8_linears_vs_emulation_and_bf16" "tests/test_encoders_gpu.py::test_vit_fp8_folded_layernorm_vs_emulation_and_unfolded" -m gpu --maxfail=8 -q -p no:cacheprovider 2>&1 | tail -15 ) > $O/pytest.log 2>&1
echo "pytest: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"
python3 tools/gemm_bench.py > $O/gemm_bench.txt 2>&1
for rep in 1 2; do
  for v in 1 0; do
    echo "c5 CVCL_LN_FOLD=$v: $(CVCL_LN_FOLD=$v python3 bench.py --config c5 --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('kernel_ms_per_step'))")"
  done
done > $O/ab_c5_fold.txt 2>&1
cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
  export CVCL_LN_FOLD=$v
  rocprofv3 --kernel-trace --stats -d $O/p_c5_fold$v -o x --output-format csv -- python3 $R/bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-roofline --no-extras > $O/p_c5_fold$v.log 2>&1
  f=$(find $O/p_c5_fold$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c5_fold${v}_kernel_stats.csv
  rm -rf $O/p_c5_fold$v
done
unset CVCL_LN_FOLD
cd $R
python3 bench.py --config c4 --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-roofline --no-extras > $O/bench_c4.json 2>$O/bench_c4.err
LAB=$R/multimodal-baby_amd/lib/libcvcl_hip_lab.so
for rep in 1 2 3; do
  for v in 1 0; do
    echo "c2 CVCL_STEM_POOL=$v: $(CVCL_HIP_LIB=$LAB CVCL_STEM_POOL=$v python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-roofline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
  done
done > $O/ab_c2_stem.txt 2>&1
( cd /tmp && CVCL_TRUNK_STREAMS=1 rocprofv3 --kernel-trace --stats -d $O/p_c2 -o x --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-roofline --no-extras > $O/p_c2.log 2>&1 )
f=$(find $O/p_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c2_1stream_kernel_stats.csv
rm -rf $O/p_c2
ls -la $O
