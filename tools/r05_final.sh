#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r05_final
( time timeout 1500 python3 -m pytest tests -m gpu --maxfail=10 -q -p no:cacheprovider 2>&1 | tail -12 ) > gpurun_out/r05_final/pytest.log 2>&1
echo "pytest: $(grep -E 'passed|failed' gpurun_out/r05_final/pytest.log | tail -1)"
bash tools/prof_r05.sh > gpurun_out/r05_final/prof.log 2>&1
tail -20 gpurun_out/r05_final/prof.log
