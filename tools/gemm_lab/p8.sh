#!/bin/bash
# 8-phase staggered K loop (tools/gemm_lab/gemm8p_kernel.h; w8p / w7p) against the library's kernel (w8b / w7b)
L=tools/gemm_lab/lab
echo "== exact checks (small integers, statistics) =="
for v in w8p w7p; do
  timeout 30 $L $v 1000 256 128 2 1 1 | grep -v tiles_m
  timeout 30 $L $v 4096 512 256 2 1 1 | grep -v tiles_m
  timeout 30 $L $v 777 768 384 2 1 0 | grep -v tiles_m
  timeout 30 $L $v 50176 512 1024 2 1 1 | grep -v tiles_m
done
echo "== timing =="
for shape in "4096 4096 4096" "8192 8192 8192" "50176 512 1024" "50176 1024 512" "12544 2048 1024" "12544 1024 2048" "200704 256 512"; do
  for v in w8b w8p w7b w7p; do timeout 60 $L $v $shape 20 0 0 | grep -v tiles_m; done
done
