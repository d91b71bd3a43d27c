#!/bin/bash
# round 5, GPU session 2: new / failed tests, LayerNorm folded into the e4m3 linears (C5 A/B), the tail's copies
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_s2
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1200 python3 -m pytest tests/test_gemm_gpu.py tests/test_lnfold_gpu.py tests/test_head_gpu.py tests/test_gemm8f_gpu.py tests/test_bn_gram_gpu.py "tests/test_encoders_gpu.py::test_vit_fp8_linears_vs_emulation_and_bf16" "tests/test_encoders_gpu.py::test_vit_fp8_folded_layernorm_vs_emulation_and_unfolded" tests/test_text_train_gpu.py tests/test_c2_parity_gpu.py -m gpu --maxfail=12 -q -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -150 ) > $O/pytest.log 2>&1
echo "pytest: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"
for rep in 1 2; do
  for v in 1 0; do
    echo "c5 CVCL_LN_FOLD=$v: $(CVCL_LN_FOLD=$v python3 bench.py --config c5 --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('kernel_ms_per_step'))")"
  done
done > $O/ab_c5_fold.txt 2>&1
python3 bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras > $O/bench_c5.json 2>$O/bench_c5.err
python3 bench.py --config c2 --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-roofline --no-extras > $O/bench_c2_quick.json 2>$O/bench_c2_quick.err
python3 tools/tail_copies.py c2 > $O/tail_copies_c2.txt 2>&1
python3 tools/tail_bench.py c2 20 > $O/tail_c2.json 2>$O/tail_c2.err
python3 tools/tail_bench.py c4 20 > $O/tail_c4.json 2>$O/tail_c4.err
python3 bench.py --config c4 --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extras > $O/bench_c4.json 2>$O/bench_c4.err
python3 bench.py --config c4p14 --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-roofline --no-extras > $O/bench_c4p14.json 2>$O/bench_c4p14.err
ls -la $O
