#!/bin/bash
# round 5: A/B of two library builds on the GEMM lab shapes and the bench configurations (libcvcl_hip_prev.so = the tree before the change)
L=tools/gemm_lab/lab
for rep in 1 2; do for s in _prev ""; do
  for shape in "50432 3072 768" "50432 2304 768" "50176 1024 512"; do
    echo -n "lib$s plain $shape: "; CVCL_HIP_LIB=$PWD/multimodal-baby_amd/lib/libcvcl_hip$s.so $L old $shape 20 0 | tail -1 | cut -c40-80
    echo -n "lib$s stats $shape: "; CVCL_HIP_LIB=$PWD/multimodal-baby_amd/lib/libcvcl_hip$s.so $L old $shape 20 0 1 | tail -1 | cut -c40-80
  done
  echo -n "lib$s gelu : "; CVCL_HIP_LIB=$PWD/multimodal-baby_amd/lib/libcvcl_hip$s.so LAB_GELU=1 $L old 50432 3072 768 20 0 | tail -1 | cut -c40-80
done; done
timeout 900 python3 -m pytest tests/test_lnfold_gpu.py tests/test_gemm_gpu.py tests/test_encoders_gpu.py tests/test_c2_parity_gpu.py tests/test_resnext_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
for c in ${CFGS:-c2 c4 c4p14}; do bash tools/ab_lib.sh $c _prev -; done
