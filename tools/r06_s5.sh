#!/bin/bash
# round 6, session 5: accumulated statistics incl. the Gram launch's BN2 -- tests; where the two-stream time of the finalize / Gram chain sits (lab skips)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s5; mkdir -p $O; cd $R
python -m pytest tests/test_finalize_on_load_gpu.py tests/test_resnext_gpu.py tests/test_c2_parity_gpu.py tests/test_bn_gram_gpu.py tests/test_gemm_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log; grep -h "max-rel" $O/tests.log | head -3
export CVCL_HIP_LIB=$R/multimodal-baby_amd/lib/libcvcl_hip_lab.so
run() { env "$@" python3 bench.py --config c2 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$*', d['ms_per_step'], {x: round(k[x],3) for x in ('gemm_pro','gconv3x3','bn_finalize','bn_add_relu','bn_relu_apply')}, d['launches_per_step']['bn_finalize'])"; }
for rep in 1 2; do
  run CVCL_FINALIZE_ON_LOAD=0
  run CVCL_FINALIZE_ON_LOAD=1
  run CVCL_FINALIZE_ON_LOAD=1 CVCL_FOL_GRAM=0
  run CVCL_FINALIZE_ON_LOAD=1 CVCL_FOL_TAIL=0
  run CVCL_FINALIZE_ON_LOAD=1 CVCL_SKIP_GRAM_REDUCE_AFTER=64
  run CVCL_FINALIZE_ON_LOAD=1 CVCL_SKIP_GRAM_REDUCE_AFTER=64 CVCL_SKIP_FROM_GRAM_AFTER=64
  run CVCL_FINALIZE_ON_LOAD=1 CVCL_SKIP_GRAM_REDUCE_AFTER=64 CVCL_SKIP_FROM_GRAM_AFTER=64 CVCL_SKIP_GRAM_PRO_AFTER=64
  run CVCL_FINALIZE_ON_LOAD=0 CVCL_SKIP_FINALIZE_AFTER=488
done > $O/ab.txt 2>&1; cat $O/ab.txt
for v in 0 1; do CVCL_TRUNK_STREAMS=1 CVCL_FINALIZE_ON_LOAD=$v python3 bench.py --config c2 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('1stream FOL=$v', d['ms_per_step'])"; done > $O/ab_1stream.txt 2>&1; cat $O/ab_1stream.txt
