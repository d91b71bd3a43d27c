#!/bin/bash
# stagger experiment (VAR bit 6): waves 4-7 half a stage behind waves 0-3, two barriers per stage, setprio around the pure-MFMA half
L=tools/gemm_lab/lab
echo "== exact checks =="
for v in w8s w7s; do
  $L $v 1000 256 128 2 1 1 | grep -v tiles_m
  $L $v 4096 512 256 2 1 1 | grep -v tiles_m
  $L $v 50176 512 1024 2 1 1 | grep -v tiles_m
done
echo "== timing =="
for shape in "4096 4096 4096" "8192 8192 8192" "50176 512 1024" "50176 1024 512" "12544 2048 1024" "12544 1024 2048"; do
  for v in w8b w8s w7b w7s; do $L $v $shape 20 0 0 | grep -v tiles_m; done
done
