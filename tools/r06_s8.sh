#!/bin/bash
# round 6, session 8 (lab): layer2.0's conv1 (M = 802816, K = 256, N = 256: 822 MB) on the W-in-registers streaming kernel with an identity operand affine
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s8; mkdir -p $O; cd $R
export CVCL_HIP_LIB=$R/multimodal-baby_amd/lib/libcvcl_hip_lab.so
run() { env "$@" python3 bench.py --config c2 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$*', d['ms_per_step'], {x: round(k[x],3) for x in ('gemm','gemm8w','gemm_pro','gconv3x3')}, d['final_loss'])"; }
for rep in 1 2 3; do run CVCL_CONV1_PRO=0; run CVCL_CONV1_PRO=1; done > $O/ab.txt 2>&1; cat $O/ab.txt
