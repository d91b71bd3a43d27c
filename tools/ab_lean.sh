#!/bin/bash
# same-box A/B of a lab switch on the C2 (and optionally C4) step: tools/ab_lean.sh <ENV_NAME> [config] -- needs lib/libcvcl_hip_lab.so
V=${1:-CVCL_LEAN_GRID}; CFG=${2:-c2}
R=${GRAFT_REPO_ROOT:-$PWD}
export CVCL_HIP_LIB=$R/multimodal-baby_amd/lib/libcvcl_hip_lab.so
for rep in 1 2 3; do for x in 0 1; do
  env $V=$x python3 $R/bench.py --config $CFG --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-roofline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V=$x', '$CFG', d['value'], d['ms_per_step'])"
done; done
