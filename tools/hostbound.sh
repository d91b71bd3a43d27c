for b in 256 128 64 32; do python3 bench.py --config c2 --batch $b --steps 40 --warmup 10 --no-cpu-baseline --no-parity --no-roofline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c2 B=$b', d['ms_per_step'], d.get('tail_ms_per_step'), d.get('tail_launches_per_step'))"; done
for b in 256 64; do python3 bench.py --config c4 --batch $b --steps 30 --warmup 8 --no-cpu-baseline --no-parity --no-roofline --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c4 B=$b', d['ms_per_step'], d.get('tail_ms_per_step'), d.get('tail_launches_per_step'))"; done
