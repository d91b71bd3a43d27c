#!/bin/bash
# round 5: what each linear epilogue of the 8-wave kernel costs on the ViT shapes (product kernel through cvcl_gemm, zero residual / bias)
L=tools/gemm_lab/lab
for shape in "50432 768 768" "50432 768 3072" "65792 768 768"; do
  for rep in 1 2; do
    echo -n "plain        : "; $L old $shape 20 0 | tail -1
    echo -n "+res         : "; LAB_RES=1 $L old $shape 20 0 | tail -1
    echo -n "+res+rowpart : "; LAB_RES=1 LAB_ROWPART=1 $L old $shape 20 0 | tail -1
  done
done
for rep in 1 2; do
  echo -n "fc1 plain : "; $L old 50432 3072 768 20 0 | tail -1
  echo -n "fc1 gelu  : "; LAB_GELU=1 $L old 50432 3072 768 20 0 | tail -1
done
