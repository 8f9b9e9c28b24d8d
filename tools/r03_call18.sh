#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1700 python -m pytest tests/test_distributed_gpu.py -m gpu -q -p no:cacheprovider -x -k "bench" > $OUT/r03i_bench_tests.log 2>&1; tail -30 $OUT/r03i_bench_tests.log | cut -c1-400
timeout 600 python -m pytest tests/test_update_parity_gpu.py -m gpu -q -p no:cacheprovider -k "legacy" 2>&1 | tail -3 | cut -c1-300
