#!/bin/bash
# round 6, session 10 (lab): the K = 512 streaming kernel -- exact tests, then the step with it off / one strip only / up to two strips
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s10; mkdir -p $O; cd $R
python -m pytest tests/test_gemm_gpu.py -x -q -m gpu -k "plain_operand" > $O/tests.log 2>&1; tail -3 $O/tests.log
export CVCL_HIP_LIB=$R/multimodal-baby_amd/lib/libcvcl_hip_lab.so
run() { env "$@" python3 bench.py --config c2 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$*', d['ms_per_step'], {x: round(k[x],3) for x in ('gemm','gemm8w','gemm_pro')}, d['launches_per_step']['gemm8w'], d['launches_per_step']['gemm_pro'])"; }
for rep in 1 2 3; do run CVCL_STRM512_MAXN=0; run CVCL_STRM512_MAXN=256; run CVCL_STRM512_MAXN=512; done > $O/ab.txt 2>&1; cat $O/ab.txt
