#!/bin/bash
# round 5, what prof_r06.sh leaves out: SQ + HBM counters per kernel for C2 and C4 at patch 14, kernel stats of the --finetune_cnn step.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06x; mkdir -p $O
for c in c2 c4p14; do GRAFT_REPO_ROOT=$R bash $R/tools/pmc_cfg.sh $c > $O/pmc_$c.log 2>&1; cp $R/gpurun_out/pmc_$c/summary.txt $O/pmc_${c}_summary.txt; cp $R/gpurun_out/pmc_$c/summary.json $O/pmc_${c}_summary.json; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/finetune -o ft --output-format csv -- python3 $R/tools/bench_finetune.py > $O/finetune.log 2>&1
tail -3 $O/finetune.log; ls $O
