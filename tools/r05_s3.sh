#!/bin/bash
# round 5, GPU session 3: per-kernel times of C5 with / without the folded LayerNorm, of the C4 tail (split-bf16 GEMMs), copies
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_s3
mkdir -p $O
cd $R
prof() { # name, env..., -- cmd
  name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --stats -d $O/p_$name -o x --output-format csv -- python3 $R/bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-roofline --no-extras > $O/p_$name.log 2>&1 )
  f=$(find $O/p_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${name}_kernel_stats.csv
  rm -rf $O/p_$name
}
cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
  export CVCL_LN_FOLD=$v
  rocprofv3 --kernel-trace --stats -d $O/p_c5_fold$v -o x --output-format csv -- python3 $R/bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-roofline --no-extras > $O/p_c5_fold$v.log 2>&1
  f=$(find $O/p_c5_fold$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c5_fold${v}_kernel_stats.csv
  rm -rf $O/p_c5_fold$v
done
unset CVCL_LN_FOLD
rocprofv3 --kernel-trace --stats -d $O/p_tail_c4 -o x --output-format csv -- python3 $R/tools/tail_bench.py c4 10 > $O/p_tail_c4.log 2>&1
f=$(find $O/p_tail_c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/tail_c4_kernel_stats.csv
rm -rf $O/p_tail_c4
cd $R
python3 tools/tail_copies.py c2 > $O/tail_copies_c2.txt 2>&1
ls -la $O
