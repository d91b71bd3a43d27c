#!/bin/bash
# same-box A/B of the ViT co-scheduling hint (product library): CVCL_VIT_CU_SHARE=0|1
R=${GRAFT_REPO_ROOT:-$PWD}
for cfg in c4 c5 c4p14; do for rep in 1 2; do for x in 0 1; do
  CVCL_VIT_CU_SHARE=$x python3 $R/bench.py --config $cfg --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-roofline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('share=$x', '$cfg', d['value'], d['ms_per_step'])"
done; done; done
