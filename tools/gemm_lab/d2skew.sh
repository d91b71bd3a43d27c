#!/bin/bash
# round 5: do two co-resident 4-wave workgroups overlap their epilogues once they are put out of phase?  (gemm2wg_kernel.h VAR bit 1:
# the second dispatch wave starts LAB_SKEW_US late; a tile of fc1 takes ~25 us)
L=tools/gemm_lab/lab
for rep in 1 2; do
  echo -n "8-wave plain : "; $L old 50432 3072 768 20 0 | tail -1
  echo -n "8-wave gelu  : "; LAB_GELU=1 $L old 50432 3072 768 20 0 | tail -1
  echo -n "d7 plain     : "; $L d7 50432 3072 768 20 0 | tail -1
  echo -n "d7 gelu      : "; $L d7g 50432 3072 768 20 0 | tail -1
  for sk in 4 8 12 16 24; do echo -n "d7 gelu skew $sk us: "; LAB_SKEW_US=$sk $L d7g 50432 3072 768 20 0 | tail -1; done
  for sk in 8 12; do echo -n "d7 plain skew $sk us: "; LAB_SKEW_US=$sk $L d7 50432 3072 768 20 0 | tail -1; done
done
