#!/bin/bash
# round 5: output rows per thread of bn_relu_maxpool (libraries built with -DCVCL_POOL_ROWS=1|4 as libcvcl_hip_r1.so / _r4.so; product = 2)
R=${GRAFT_REPO_ROOT:-$PWD}
for st in 1 2; do for rep in 1 2 3; do for s in _r1 "" _r4; do
  CVCL_TRUNK_STREAMS=$st CVCL_HIP_LIB=$R/multimodal-baby_amd/lib/libcvcl_hip$s.so python3 $R/bench.py --config c2 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('streams $st lib${s:-_r2}', d['ms_per_step'], 'maxpool', round(k['bn_relu_maxpool'],4))"
done; done; done
