#!/bin/bash
# round 6, session 3: atomic accumulation probe; finalize-on-load bit identity + A/B of its parts (lab library)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s3; mkdir -p $O; cd $R
( cd tools/probes; for a in "224 512 256" "224 256 256" "256 2048 256" "768 256 64" "512 128 128" "384 128 64"; do ./atomic_probe $a 2000; done ) > $O/atomic_probe.txt 2>&1; cat $O/atomic_probe.txt
python -m pytest tests/test_finalize_on_load_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
export CVCL_HIP_LIB=$R/multimodal-baby_amd/lib/libcvcl_hip_lab.so
run() { env "$@" python3 bench.py --config c2 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$*', d['ms_per_step'], {x: round(k[x],3) for x in ('gconv3x3','bn_finalize','bn_add_relu','bn_relu_apply')}, d['launches_per_step']['bn_finalize'])"; }
for rep in 1 2; do
  run CVCL_FINALIZE_ON_LOAD=0
  run CVCL_FINALIZE_ON_LOAD=1 CVCL_FOL_TAIL=0 CVCL_FOL_MAXROWS=128
  run CVCL_FINALIZE_ON_LOAD=1 CVCL_FOL_TAIL=0 CVCL_FOL_MAXROWS=256
  run CVCL_FINALIZE_ON_LOAD=1 CVCL_FOL_TAIL=1 CVCL_FOL_MAXROWS=0
  run CVCL_FINALIZE_ON_LOAD=1 CVCL_FOL_TAIL=3 CVCL_FOL_MAXROWS=0
  run CVCL_FINALIZE_ON_LOAD=1 CVCL_FOL_TAIL=3 CVCL_FOL_MAXROWS=128
done > $O/ab_fol_parts.txt 2>&1; cat $O/ab_fol_parts.txt
