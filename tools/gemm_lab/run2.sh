#!/bin/bash
L=tools/gemm_lab/lab
echo "== exact checks =="
for v in w8c w7c w8b; do $L $v 1000 256 128 2 1 1; $L $v 4096 512 256 2 1 1; done
echo "== timing =="
for shape in "4096 4096 4096" "50176 512 1024" "50176 1024 512" "12544 1024 2048" "12544 2048 1024" "200704 512 256" "802816 256 256" "200704 256 512" "50432 2304 768" "50432 768 3072"; do
  for v in old w8b w7b; do
    $L $v $shape 20 0 0 | grep -v tiles_m
  done
done
