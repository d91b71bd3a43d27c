python -m pytest tests/test_trunk_train_gpu.py -m gpu -x -q -k "weight_gradient_stream or finetune or trunk_" 2>&1 | tail -3
CVCL_WGRAD_STREAM=0 python tools/bench_finetune.py --batch 256 --steps 5 2>&1 | grep -v amdgpu.ids | tail -2 | head -1
CVCL_WGRAD_STREAM=1 python tools/bench_finetune.py --batch 256 --steps 5 2>&1 | grep -v amdgpu.ids | tail -2 | head -1
