#!/bin/bash
# HBM traffic counters for the bench step, one rocprofv3 --pmc pass per counter (FETCH_SIZE takes 3 of the 4 TCC
# slots, WRITE_SIZE 2: they cannot share a pass).  Only --kernel-trace is combined with --pmc.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/$c -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-extras > $OUT/$c.log 2>&1
done
ls -R $OUT | head -30
