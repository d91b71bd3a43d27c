#!/bin/bash
# round 5: two 4-wave workgroups per CU (gemm2wg_kernel.h) against the product's 8-wave kernel ("old" = cvcl_gemm), and the row-pitch probe
L=tools/gemm_lab/lab
echo "== exact checks (small integers) =="
for v in d8 d7 d7a; do
  $L $v 1000 256 128 2 1 | tail -1
  $L $v 4096 512 256 2 1 | tail -1
  $L $v 777 768 384 2 1 | tail -1
  $L $v 50432 768 768 2 1 | tail -1
done
echo "== timing =="
for shape in "4096 4096 4096" "8192 8192 8192" "50176 512 1024" "50176 1024 512" "12544 1024 2048" "12544 2048 1024" "200704 512 512" "50176 1024 1024" "50432 2304 768" "50432 768 768" "50432 3072 768" "50432 768 3072"; do
  for v in old d8 d8a d7 d7a; do
    $L $v $shape 20 0 | grep -v "tiles"
  done
done
echo "== GELU epilogue (fc1) =="
LAB_GELU=1 $L old 50432 3072 768 20 0 | tail -1
$L d7g 50432 3072 768 20 0 | tail -1
$L d7ag 50432 3072 768 20 0 | tail -1
echo "== super-row height / grid of the d kernels =="
for sr in 4 8 16; do LAB_SR=$sr $L d8 50432 3072 768 20 0 | tail -1; done
for g in 256 384 512; do LAB_GRID=$g $L d8 50432 3072 768 20 0 | tail -1; done
echo "== row pitch probe: lda = ldw = K + pad (product kernel) =="
for shape in "4096 4096 4096" "8192 8192 8192" "50432 768 3072" "12544 1024 2048" "50176 512 1024" "50176 1024 512" "50432 2304 768"; do
  for pad in 0 32 64 128; do
    echo -n "pad $pad: "; LAB_PAD=$pad $L old $shape 20 0 | tail -1
  done
done
