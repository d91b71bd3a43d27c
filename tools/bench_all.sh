#!/bin/bash
# the four bench configurations on this box (step time only), appended to gpurun_out/boxes/bench_all.txt -- run in several gpurun calls to see the
# box-to-box spread the results table quotes
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out/boxes
echo "box $(hostname) $(date -u +%H:%M:%S)" >> $R/gpurun_out/boxes/bench_all.txt
for c in c2 c4 c4p14 c5; do
  python3 $R/bench.py --config $c --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-roofline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  $c', d['ms_per_step'], 'ms', d['value'], 'pairs/s', 'tail', d.get('tail_ms_per_step'), d.get('tail_launches_per_step'))" >> $R/gpurun_out/boxes/bench_all.txt
done
tail -5 $R/gpurun_out/boxes/bench_all.txt
