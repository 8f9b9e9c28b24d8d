#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/suite_repeat.sh 10 r03_suite_repeat_final_build
bash tools/profile_round.sh r03k > /dev/null 2>&1
tail -c 1500 gpurun_out/r03k_bench.json; echo; head -24 gpurun_out/r03k_kernel_stats.txt | cut -c1-150
