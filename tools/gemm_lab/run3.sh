#!/bin/bash
L=tools/gemm_lab/lab
for cold in 0 1; do
export LAB_COLD=$cold
echo "== LAB_COLD=$cold =="
for shape in "50176 512 1024" "50176 1024 512" "12544 1024 2048" "200704 256 512" "50432 768 768"; do
  for v in w7b w8b; do $L $v $shape 12 0 0 | grep -v tiles_m; done
  CVCL_GEMM8W=0 $L old $shape 12 0 0 | grep -v tiles_m
done
done
