#!/bin/bash
# round 6, session 1: the tests behind the first failure, the finalize-chain upper bound (lab switch), a bench line of the tree
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s1; mkdir -p $O; cd $R
python -m pytest tests/test_train_entry_gpu.py tests/test_trunk_train_gpu.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
bash tools/ab_env.sh CVCL_SKIP_FINALIZE_AFTER c2 0 488 > $O/ab_finalize.txt 2>&1; cat $O/ab_finalize.txt
CVCL_TRUNK_STREAMS=1 bash tools/ab_env.sh CVCL_SKIP_FINALIZE_AFTER c2 0 488 > $O/ab_finalize_1stream.txt 2>&1; cat $O/ab_finalize_1stream.txt
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('c2', d['ms_per_step'], 'c4', d['c4']['ms_per_step'], 'c4p14', d['c4p14']['ms_per_step'], 'c5', d['c5']['ms_per_step'], 'ft', d['finetune_cnn']['ms_per_step'], 'fp32', d['fp32_parity_mode']['ms_per_step'])"
