#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_lib.sh <config> <kernel_class> libA.so libB.so   (two repeats, interleaved)
CFG=$1; K=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do for L in "$@"; do
  CVCL_HIP_LIB=$R/multimodal-baby_amd/lib/$L python3 $R/bench.py --config $CFG --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', '$CFG', d['ms_per_step'], '$K', d['kernel_ms_per_step'].get('$K'))"
done; done
