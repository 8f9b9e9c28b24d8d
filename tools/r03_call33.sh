#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03l_suite.log 2>&1; grep -E "passed|failed" $OUT/r03l_suite.log | tail -3 | cut -c1-300
