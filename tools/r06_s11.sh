#!/bin/bash
# round 6, session 11 (lab): what the ViT's attention launches cost the C4 / C4p14 STEP with two trunk passes in flight (skipped after warm-up: timing only)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s11; mkdir -p $O; cd $R
export CVCL_HIP_LIB=$R/multimodal-baby_amd/lib/libcvcl_hip_lab.so
run() { c=$1; shift; env "$@" python3 bench.py --config $c --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$c $*', d['ms_per_step'], {x: round(k[x],3) for x in k})"; }
for rep in 1 2; do for c in c4 c4p14; do run $c CVCL_SKIP_ATTENTION_AFTER=0; run $c CVCL_SKIP_ATTENTION_AFTER=150; done; done > $O/ab.txt 2>&1; cat $O/ab.txt
