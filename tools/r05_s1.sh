#!/bin/bash
# round 5, GPU session 1: the full GPU test suite on the new tree, then same-box A/Bs of the supertile walk / remainder launch
# (lab library switches), the vendor comparison, and the trainable tail on its own (launch lists under rocprofv3).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_s1
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
LAB=$R/multimodal-baby_amd/lib/libcvcl_hip_lab.so
( time timeout 1500 python3 -m pytest tests -m gpu --maxfail=12 -q -p no:cacheprovider 2>&1 | tail -60 ) > $O/pytest.log 2>&1
echo "pytest done: $(tail -3 $O/pytest.log | head -1)"
# vendor comparison: product walk vs the column-fastest list of rounds 2-4 (lab: CVCL_G8_SUPERROW=1)
timeout 600 python3 tools/blaslt_compare.py > $O/blaslt_new.txt 2>&1
CVCL_HIP_LIB=$LAB CVCL_G8_SUPERROW=1 timeout 600 python3 tools/blaslt_compare.py > $O/blaslt_oldwalk.txt 2>&1
# step-time A/B (lab library, interleaved, two repeats)
for rep in 1 2; do
  for cfg in c4 c4p14; do
    echo "$cfg new: $(CVCL_HIP_LIB=$LAB python3 bench.py --config $cfg --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-roofline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
    echo "$cfg oldwalk: $(CVCL_HIP_LIB=$LAB CVCL_G8_SUPERROW=1 python3 bench.py --config $cfg --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-roofline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
    echo "$cfg noremainder: $(CVCL_HIP_LIB=$LAB CVCL_G8_REMAINDER=0 python3 bench.py --config $cfg --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-roofline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
    echo "$cfg oldwalk+noremainder: $(CVCL_HIP_LIB=$LAB CVCL_G8_SUPERROW=1 CVCL_G8_REMAINDER=0 python3 bench.py --config $cfg --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-roofline --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
  done
done > $O/ab_walk.txt 2>&1
python3 bench.py --config c2 --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-roofline --no-extras > $O/bench_c2_quick.json 2>$O/bench_c2_quick.err
# the trainable tail on its own
for cfg in c2 c4; do
  python3 tools/tail_bench.py $cfg 20 > $O/tail_$cfg.json 2>$O/tail_$cfg.err
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/tailprof_$cfg -o tail --output-format csv -- python3 $R/tools/tail_bench.py $cfg 10 > $O/tailprof_$cfg.log 2>&1 )
  f=$(find $O/tailprof_$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/tail_${cfg}_kernel_stats.csv
  rm -rf $O/tailprof_$cfg
done
ls -la $O
