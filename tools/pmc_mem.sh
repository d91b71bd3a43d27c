#!/bin/bash
# L2 / fabric counters of the 8-wave GEMM inside the C2 step (one trunk stream) and in the lab (back to back / behind a bn_add_relu-like
# pass): L2 hit rate, average fabric read latency (RDREQ_LEVEL / RDREQ), write stalls.  One --pmc pass per counter group; only
# --kernel-trace is combined with --pmc.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_mem
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"; do
  i=$((i+1))
  CVCL_TRUNK_STREAMS=1 rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i/bench -o pmc --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-extras > $OUT/g${i}_bench.log 2>&1
  for c in 0 5; do
    LAB_COLD=$c rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i/lab$c -o pmc --output-format csv -- $R/tools/gemm_lab/lab w7b 50176 512 1024 20 0 1 > $OUT/g${i}_lab$c.log 2>&1
  done
done
cd $R && python3 tools/pmc_mem.py
