#!/bin/bash
# nontemporal (lab) vs ordinary write-back (lab_plain, -DCVCL_PLAIN_STORES) output stores of the 8-wave kernel
# build first: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCVCL_PLAIN_STORES -Iinclude -Imultimodal-baby_amd/csrc tools/gemm_lab/lab.hip -o tools/gemm_lab/lab_plain -ldl
for shape in "4096 4096 4096" "50432 2304 768" "50432 3072 768" "50432 768 3072" "50176 512 1024" "50176 1024 512" "12544 2048 1024" "200704 256 512"; do
  for v in w8b w7b; do
    for L in tools/gemm_lab/lab tools/gemm_lab/lab_plain; do echo -n "$(basename $L) "; $L $v $shape 20 0 0 | grep -v tiles_m; done
  done
done
