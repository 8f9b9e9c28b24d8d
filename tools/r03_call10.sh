#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03d_suite_default.log 2>&1; tail -4 $OUT/r03d_suite_default.log
FBHIP_P3=2 timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03d_suite_p3_forced.log 2>&1; tail -12 $OUT/r03d_suite_p3_forced.log
timeout 900 python tools/tolerance_probe.py > $OUT/r03_tolerance_probe.txt 2>&1; tail -24 $OUT/r03_tolerance_probe.txt
