#!/bin/bash
# round 6, session 7: does a streaming (nontemporal) read of raw3 / identity in bn_add_relu leave its OUTPUT in the Infinity Cache for the conv1 behind it?
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s7; mkdir -p $O; cd $R
export CVCL_HIP_LIB=$R/multimodal-baby_amd/lib/libcvcl_hip_lab.so
run() { env "$@" python3 bench.py --config c2 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$*', d['ms_per_step'], {x: round(k[x],3) for x in ('gemm8w','bn_add_relu','bn_relu_apply','gconv3x3')})"; }
for rep in 1 2 3; do for v in 0 1 2 3; do run CVCL_ADDRELU_NT=$v; done; done > $O/ab_nt.txt 2>&1; cat $O/ab_nt.txt
for v in 0 3; do run CVCL_TRUNK_STREAMS=1 CVCL_ADDRELU_NT=$v; done > $O/ab_nt_1stream.txt 2>&1; cat $O/ab_nt_1stream.txt
