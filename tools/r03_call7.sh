#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -p no:cacheprovider -k "p3" > $OUT/r03_p3_kernel_tests.log 2>&1; tail -5 $OUT/r03_p3_kernel_tests.log
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/r03_p3_suite_mode1.log 2>&1; tail -15 $OUT/r03_p3_suite_mode1.log
FBHIP_P3=2 timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03_p3_suite_mode2.log 2>&1; tail -30 $OUT/r03_p3_suite_mode2.log
python bench.py --steps 640 --warmup 64 --repeats 3 --no-cpu-baseline > $OUT/r03_p3_bench.json 2> $OUT/r03_p3_bench.err; cat $OUT/r03_p3_bench.json | cut -c1-400; tail -3 $OUT/r03_p3_bench.err
FBHIP_P3=0 python bench.py --steps 640 --warmup 64 --repeats 3 --no-cpu-baseline > $OUT/r03_p3off_bench.json 2> $OUT/r03_p3off_bench.err; cat $OUT/r03_p3off_bench.json | cut -c1-400
