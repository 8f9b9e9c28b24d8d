#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03o_suite.log 2>&1; grep -E "passed|failed" $OUT/r03o_suite.log | tail -3 | cut -c1-300
timeout 200 python bench.py --steps 960 --warmup 64 --repeats 3 --no-cpu-baseline --no-single-update-probe 2>/dev/null | cut -c1-160
