run() { python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('$1', d['value'], d['ms_per_step'], {a:round(b,3) for a,b in k.items() if b>0.1})
"; }
export CVCL_FUSED_TAIL_STAGES=2
run base_fused2
CVCL_GCONV_LDS_KB=66 run gconv66
CVCL_GCONV_LDS_KB=80 run gconv80
CVCL_GCONV_LDS_KB=40 run gconv40
