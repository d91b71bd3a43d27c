#!/bin/bash
# exact checks + the workload's shapes on the current 8-wave kernel (w8 / w7) next to the library dispatcher ("old" = cvcl_gemm)
L=tools/gemm_lab/lab
echo "== exact checks (small integers) =="
for v in w8 w7; do
  $L $v 1000 256 128 2 1 1 | tail -1
  $L $v 4096 512 256 2 1 1 | tail -1
  $L $v 50432 768 3072 2 1 0 | tail -1
done
echo "== timing =="
for shape in "4096 4096 4096" "8192 8192 8192" "50176 512 1024" "50176 1024 512" "12544 1024 2048" "12544 2048 1024" "200704 256 512" "50432 2304 768" "50432 768 768" "50432 3072 768" "50432 768 3072"; do
  for v in w8b w7b; do $L $v $shape 20 0 0 | grep -v tiles_m; done
done
bash tools/gemm_lab/ablate.sh
