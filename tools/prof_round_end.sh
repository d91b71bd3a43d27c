#!/bin/bash
# round-end profile refresh: rocprofv3 --kernel-trace --stats of bench.py (C2; under the tracer the per-kernel averages come
# out as one-at-a-time durations although the command keeps two trunk passes in flight), the C5 fp8 probe and the frame transform
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_end
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/c2 -o c2 --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/c2.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/c5 -o c5 --output-format csv -- python3 $R/tools/bench_c4.py --precision fp8 --steps 3 > $O/c5.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/aug -o aug --output-format csv -- python3 $R/tools/bench_augment.py --cpu-frames 4 > $O/aug.log 2>&1
ls -R $O | head -40
tail -2 $O/c2.log | cut -c1-400
