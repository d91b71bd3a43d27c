L=tools/gemm_lab/lab
for rep in 1 2; do for s in _prev ""; do
  echo -n "lib$s plain: "; CVCL_HIP_LIB=$PWD/multimodal-baby_amd/lib/libcvcl_hip$s.so $L old 50432 3072 768 20 0 | tail -1 | cut -c1-80
  echo -n "lib$s gelu : "; CVCL_HIP_LIB=$PWD/multimodal-baby_amd/lib/libcvcl_hip$s.so LAB_GELU=1 $L old 50432 3072 768 20 0 | tail -1 | cut -c1-80
done; done
timeout 600 python3 -m pytest tests/test_lnfold_gpu.py tests/test_gemm_gpu.py tests/test_encoders_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
for c in c4 c4p14; do bash tools/ab_lib.sh $c _prev -; done
