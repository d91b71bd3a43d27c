#!/bin/bash
# same-box sweep of a lab switch: tools/ab_env.sh <ENV_NAME> <config> v1 v2 ...   (needs lib/libcvcl_hip_lab.so; two repeats, interleaved)
V=$1; CFG=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
export CVCL_HIP_LIB=$R/multimodal-baby_amd/lib/libcvcl_hip_lab.so
for rep in 1 2; do for x in "$@"; do
  env $V=$x python3 $R/bench.py --config $CFG --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-roofline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V=$x', '$CFG', d['value'], d['ms_per_step'])"
done; done
