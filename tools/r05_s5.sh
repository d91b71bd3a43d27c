#!/bin/bash
# round 5, GPU session 5: stem prefetch + maxpool fix (tests, per-kernel times), tail after the scatter-kernel / launch trims
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_s5
mkdir -p $O
cd $R
( timeout 1500 python3 -m pytest tests/test_resnext_gpu.py tests/test_c2_parity_gpu.py tests/test_head_gpu.py tests/test_text_train_gpu.py tests/test_encoders_gpu.py tests/test_train_entry_gpu.py tests/test_trunk_train_gpu.py tests/test_lm_gpu.py -m gpu --maxfail=8 -q -p no:cacheprovider 2>&1 | tail -15 ) > $O/pytest.log 2>&1
echo "pytest: $(grep -E 'passed|failed' $O/pytest.log | tail -1)"
for rep in 1 2 3; do
  echo "c2: $(python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-roofline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
done > $O/c2.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && CVCL_TRUNK_STREAMS=1 rocprofv3 --kernel-trace --stats -d $O/p_c2 -o x --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-roofline --no-extras > $O/p_c2.log 2>&1 )
f=$(find $O/p_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c2_1stream_kernel_stats.csv
rm -rf $O/p_c2
python3 tools/tail_bench.py c2 20 > $O/tail_c2.json 2>$O/tail_c2.err
python3 tools/tail_bench.py c4 20 > $O/tail_c4.json 2>$O/tail_c4.err
python3 bench.py --config c4 --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-extras > $O/bench_c4.json 2>$O/bench_c4.err
ls -la $O
