#!/bin/bash
# same-box A/B of a lab switch, reporting the step time AND one kernel class's event-timed ms per step:
#   tools/ab_kernel.sh <ENV_NAME> <config> <kernel_class> v1 v2 ...
V=$1; CFG=$2; K=$3; shift 3
R=${GRAFT_REPO_ROOT:-$PWD}
export CVCL_HIP_LIB=$R/multimodal-baby_amd/lib/libcvcl_hip_lab.so
for rep in 1 2; do for x in "$@"; do
  env $V=$x python3 $R/bench.py --config $CFG --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V=$x', '$CFG', d['ms_per_step'], '$K', d['kernel_ms_per_step'].get('$K'))"
done; done
