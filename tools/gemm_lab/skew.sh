#!/bin/bash
# round 5: start skew of the 8-wave kernel (CVCL_G8_SKEW = per mille of the estimated tile period), product kernel through cvcl_gemm.
# Result: profiles/r05_gemm_lab.txt (no setting gains).  The switch lived in gemm8w_kernel.h / gemm8w.hip of commit c475ccc and was removed
# afterwards: check that commit out to re-run this script.
L=tools/gemm_lab/lab
export CVCL_HIP_LIB=$PWD/multimodal-baby_amd/lib/libcvcl_hip_lab.so
for shape in "50432 2304 768" "50432 768 768" "50432 3072 768" "50432 768 3072" "50176 1024 512" "50176 512 1024" "200704 512 512" "12544 2048 1024" "8192 8192 8192"; do
  for sk in 0 125 250 500 750 1000 0; do
    echo -n "skew $sk: "; CVCL_G8_SKEW=$sk $L old $shape 20 0 | tail -1
  done
done
echo "== with the GELU epilogue (fc1) =="
for sk in 0 500 1000 0; do echo -n "skew $sk: "; CVCL_G8_SKEW=$sk LAB_GELU=1 $L old 50432 3072 768 20 0 | tail -1; done
echo "== statistics epilogue (conv) =="
for shape in "50176 1024 512" "50176 512 1024" "200704 512 512"; do
  for sk in 0 500 1000 0; do echo -n "skew $sk: "; CVCL_G8_SKEW=$sk $L old $shape 20 0 1 | tail -1; done
done
