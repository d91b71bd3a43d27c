run() { python bench.py --config $2 --steps 20 --warmup 5 --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['ms_per_step'], {k:round(v,3) for k,v in d.get('kernel_ms_per_step',{}).items() if k in ('attention','layernorm','gemm8w','gemm')})
"; }
run c4 c4
run c5 c5
