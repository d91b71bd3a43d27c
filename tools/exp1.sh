# A/B runs of bench.py under experiment switches: name, pairs/s, ms/step, single-stream kernel ms per class
run() { python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('$1', d['value'], d['ms_per_step'], {a:round(b,3) for a,b in k.items() if b>0.1})
"; }
run default
run default_again
