#!/bin/bash
# round 6, session 6: the whole GPU suite on the tree with accumulated statistics (gconv + Gram), then the default bench line
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s6; mkdir -p $O; cd $R
python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; tail -4 $O/tests.log
for v in 0 1; do CVCL_TRUNK_STREAMS=1 CVCL_FINALIZE_ON_LOAD=$v python3 bench.py --config c2 --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('1stream FOL=$v', d['ms_per_step'], d['launches_per_step']['bn_finalize'])"; done > $O/ab_1stream.txt 2>&1; cat $O/ab_1stream.txt
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('c2', d['ms_per_step'], 'c4', d['c4']['ms_per_step'], 'c4p14', d['c4p14']['ms_per_step'], 'c5', d['c5']['ms_per_step'], 'ft', d['finetune_cnn']['ms_per_step'], 'fp32', d['fp32_parity_mode']['ms_per_step'])"
