#!/bin/bash
# round 6, session 2: finalize-on-load -- bit identity, the trunk / block parity tests, same-box A/B of the switch
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s2; mkdir -p $O; cd $R
python -m pytest tests/test_finalize_on_load_gpu.py tests/test_resnext_gpu.py tests/test_c2_parity_gpu.py tests/test_bn_gram_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
for rep in 1 2 3; do for v in 1 0; do
  CVCL_FINALIZE_ON_LOAD=$v python3 bench.py --config c2 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('FOL=$v', d['ms_per_step'], {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}, d['launches_per_step'])"
done; done > $O/ab_fol.txt 2>&1; cat $O/ab_fol.txt
for v in 1 0; do CVCL_TRUNK_STREAMS=1 CVCL_FINALIZE_ON_LOAD=$v python3 bench.py --config c2 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('1stream FOL=$v', d['ms_per_step'])"; done > $O/ab_fol_1stream.txt 2>&1; cat $O/ab_fol_1stream.txt
