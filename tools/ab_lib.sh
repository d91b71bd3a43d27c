#!/bin/bash
# same-box A/B of two BUILDS of the library: tools/ab_lib.sh <config> <suffixA> <suffixB> ...   ("" = the product build; e.g. _prev)
# (build the other tree with CVCL_LIB_SUFFIX=_prev python multimodal-baby_amd/build.py); three repeats, interleaved
CFG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2 3; do for s in "$@"; do
  [ "$s" = "-" ] && s=""
  CVCL_HIP_LIB=$R/multimodal-baby_amd/lib/libcvcl_hip$s.so python3 $R/bench.py --config $CFG --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('lib$s', '$CFG', d['ms_per_step'], {x: round(k[x],3) for x in k if 'gemm' in x or 'attention' in x})"
done; done
