#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03f_suite.log 2>&1; tail -15 $OUT/r03f_suite.log | cut -c1-300
# configs[2] (quadruped) and one SF learner: kernel table, step timeline, three PMC passes
bash tools/profile_cmd.sh r03_quadruped "python bench.py --workload quadruped --steps 640 --warmup 64 --repeats 3" "python bench.py --workload quadruped --steps 128 --warmup 32 --repeats 1"
bash tools/profile_cmd.sh r03_sf_icm "python tools/sf_bench.py --learner icm --steps 640 --warmup 64" "python tools/sf_bench.py --learner icm --steps 128 --warmup 32 --no-cpu-baseline"
