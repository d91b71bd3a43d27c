run() { python bench.py --no-cpu-baseline --no-roofline --steps 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for r in 1 2; do
run base
CVCL_FUSED_TAIL_STAGES=0 run fused0
CVCL_FUSED_TAIL_STAGES=2 run fused2
CVCL_FUSED_TAIL_STAGES=3 run fused3
CVCL_FUSED_TAIL_STAGES=4 run fused4
CVCL_BN3_GRAM=1 run gram
CVCL_GEMM256=1 run gemm256
CVCL_TRUNK_STREAMS=3 run streams3
done
