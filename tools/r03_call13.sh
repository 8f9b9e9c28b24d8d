#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03g_suite.log 2>&1; tail -8 $OUT/r03g_suite.log | cut -c1-300
for st in 128 320 640 960; do
  echo "== sf icm steps $st"; python tools/sf_bench.py --learner icm --steps $st --warmup 64 --no-cpu-baseline 2>/dev/null | cut -c1-330
done
for st in 128 640; do
  echo "== sf icm steps $st explicit stream"; python tools/sf_bench.py --learner icm --steps $st --warmup 64 --no-cpu-baseline --explicit-stream 2>/dev/null | cut -c1-330
done
echo "== fb bench 640 / 1920"; python bench.py --steps 640 --warmup 64 --repeats 3 --no-cpu-baseline --no-single-update-probe 2>/dev/null | cut -c1-200
python bench.py --steps 1920 --warmup 64 --repeats 3 --no-cpu-baseline --no-single-update-probe 2>/dev/null | cut -c1-200
