python -m pytest tests/test_train_entry_gpu.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-roofline --steps 40 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('3slots', d['value'], d['ms_per_step'], d['final_loss'])"
done
python tools/bench_c4.py --trunk-stream --trunk-streams 2 --steps 20 --precision fp8 2>&1 | grep -v amdgpu.ids | tail -2 | head -1
python tools/bench_c4.py --trunk-stream --trunk-streams 1 --steps 20 2>&1 | grep -v amdgpu.ids | tail -2 | head -1
