#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 > $OUT/r03j_suite.log 2>&1; grep -E "passed|failed" $OUT/r03j_suite.log | tail -3 | cut -c1-300; grep -A 14 "slowest" $OUT/r03j_suite.log | cut -c1-160
timeout 900 bash tools/profile_round.sh r03k > /dev/null 2>&1
tail -c 1200 $OUT/r03k_bench.json; echo; head -22 $OUT/r03k_kernel_stats.txt | cut -c1-150
timeout 1500 bash tools/record_artifacts.sh r03 > /dev/null 2>&1
ls $OUT | grep "^r03_" | head -40
