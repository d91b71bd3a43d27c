#!/bin/bash
# trunk streams sweep (product library): tools/ab_streams.sh <config> n1 n2 ...
CFG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do for n in "$@"; do
  CVCL_TRUNK_STREAMS=$n CVCL_VIT_TRUNK_STREAMS=$n python3 $R/bench.py --config $CFG --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-roofline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams $n', '$CFG', d['value'], d['ms_per_step'])"
done; done
