#!/bin/bash
# round 6, session 9: plain-operand streaming kernel for conv1 of layer2.0 -- tests, then the step against the previous build (lib _prev)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s9; mkdir -p $O; cd $R
python -m pytest tests/test_gemm_gpu.py tests/test_resnext_gpu.py tests/test_c2_parity_gpu.py tests/test_trunk_train_gpu.py tests/test_finalize_on_load_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
bash tools/ab_lib.sh c2 _prev - > $O/ab.txt 2>&1; cat $O/ab.txt
