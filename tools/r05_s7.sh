#!/bin/bash
# round 5, GPU session 7: attention phase-skew probe (lab switches CVCL_ATT_SKEW_BIT / CVCL_ATT_SKEW_SLEEPS)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_s7
mkdir -p $O
cd $R
export CVCL_HIP_LIB=$R/multimodal-baby_amd/lib/libcvcl_hip_lab.so
for bit in 8 0 3 5 7; do for sl in 0 3 6 9 12; do
  echo "bit $bit sleeps $sl: $(CVCL_ATT_SKEW_BIT=$bit CVCL_ATT_SKEW_SLEEPS=$sl python3 tools/att_bench.py 2>/dev/null | tr '\n' ' ')"
  [ $sl = 0 ] && [ $bit != 8 ] && continue
done; done > $O/att_skew.txt 2>&1
cat $O/att_skew.txt
