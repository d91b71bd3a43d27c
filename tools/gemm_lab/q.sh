#!/bin/bash
# 4-wave (gemm4w) vs 8-wave (gemm8w) lab run: exact checks first, then timing on the workload's MFMA-bound shapes
L=tools/gemm_lab/lab
VARS=${VARS:-"w8b w7b q7b q8b"}
echo "== exact checks (small integers, statistics) =="
for v in q7b q8b; do
  $L $v 1000 256 128 2 1 1 | tail -1
  $L $v 4096 512 256 2 1 1 | tail -1
  $L $v 777 768 384 2 1 0 | tail -1
  $L $v 50176 512 1024 2 1 1 | tail -1
done
echo "== timing =="
for shape in "4096 4096 4096" "8192 8192 8192" "50176 512 1024" "50176 1024 512" "12544 1024 2048" "12544 2048 1024" "200704 256 512" "50176 1024 1024" "50432 2304 768" "50432 768 768" "50432 3072 768" "50432 768 3072"; do
  for v in $VARS; do
    $L $v $shape 20 0 0 | grep -v tiles_m
  done
done
