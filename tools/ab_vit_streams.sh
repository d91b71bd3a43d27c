#!/bin/bash
# C4 / C5 sweep: trunk streams x CU cap of the 8-wave kernels (lab library)
CFG=${1:-c4}
R=${GRAFT_REPO_ROOT:-$PWD}
export CVCL_HIP_LIB=$R/multimodal-baby_amd/lib/libcvcl_hip_lab.so
for s in 2 3 4; do for c in 0 128 96 64; do
  CVCL_VIT_TRUNK_STREAMS=$s CVCL_G8_CUS=$c python3 $R/bench.py --config $CFG --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-roofline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams $s cus $c', '$CFG', d['value'], d['ms_per_step'])"
done; done
