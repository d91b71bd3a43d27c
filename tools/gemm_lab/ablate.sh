#!/bin/bash
# what is left of the 8-wave kernel's time when a pipeline component is removed (results are wrong by construction)
L=tools/gemm_lab/lab
for shape in "4096 4096 4096" "50432 2304 768" "50432 3072 768" "50176 512 1024" "50176 1024 512"; do
  for v in w8b w8n w8l w8r w8x w8e; do $L $v $shape 20 0 0 | grep -v tiles_m; done
done
