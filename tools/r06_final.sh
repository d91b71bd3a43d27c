#!/bin/bash
# round 6: final validation of the tree -- the whole GPU suite, the driver's smoke, the default bench line
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_final; mkdir -p $O; cd $R
python -m pytest tests -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; python3 -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('c2', d['ms_per_step'], d['value'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], d['roofline']['traffic_source'][:40]); print('launches', d['launches_per_step']); print('c4', d['c4']['ms_per_step'], 'c4p14', d['c4p14']['ms_per_step'], 'c5', d['c5']['ms_per_step'], 'ft', d['finetune_cnn']['ms_per_step'], 'fp32', d['fp32_parity_mode']['ms_per_step']); print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
