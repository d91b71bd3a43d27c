#!/bin/bash
# round 6, session 4: accumulated BatchNorm statistics -- unit + trunk tests, same-box A/B of the switch and of its parts (lab library)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s4; mkdir -p $O; cd $R
python -m pytest tests/test_finalize_on_load_gpu.py tests/test_resnext_gpu.py tests/test_c2_parity_gpu.py tests/test_bn_gram_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log; grep -h "max-rel" $O/tests.log | head -3
export CVCL_HIP_LIB=$R/multimodal-baby_amd/lib/libcvcl_hip_lab.so
run() { env "$@" python3 bench.py --config c2 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$*', d['ms_per_step'], {x: round(k[x],3) for x in ('gemm8w','gemm','gconv3x3','bn_finalize','bn_add_relu','bn_relu_apply')}, d['launches_per_step']['bn_finalize'])"; }
for rep in 1 2 3; do
  run CVCL_FINALIZE_ON_LOAD=0
  run CVCL_FINALIZE_ON_LOAD=1
  run CVCL_FINALIZE_ON_LOAD=1 CVCL_FOL_TAIL=0
  run CVCL_FINALIZE_ON_LOAD=1 CVCL_FOL_TAIL=1
  run CVCL_FINALIZE_ON_LOAD=1 CVCL_FOL_CONV1=0
done > $O/ab_acc.txt 2>&1; cat $O/ab_acc.txt
for v in 0 1; do CVCL_TRUNK_STREAMS=1 CVCL_FINALIZE_ON_LOAD=$v python3 bench.py --config c2 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('1stream FOL=$v', d['ms_per_step'])"; done > $O/ab_acc_1stream.txt 2>&1; cat $O/ab_acc_1stream.txt
