#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for m in 5 9 11; do
  echo "== sf icm legacy stream, nullsync mode $m"; FBHIP_DBG_NULLSYNC=$m python tools/sf_bench.py --learner icm --steps 320 --warmup 64 --no-cpu-baseline 2>&1 | grep -E "^\{|Error|error" | cut -c1-230
done
echo "== fb bench legacy stream modes 3 / 1"
FBHIP_BENCH_LEGACY_STREAM=1 python bench.py --steps 640 --warmup 64 --repeats 3 --no-cpu-baseline --no-single-update-probe 2>&1 | grep -E "^\{|Error|error" | cut -c1-160
FBHIP_DBG_NULLSYNC=1 FBHIP_BENCH_LEGACY_STREAM=1 python bench.py --steps 640 --warmup 64 --repeats 3 --no-cpu-baseline --no-single-update-probe 2>&1 | grep -E "^\{|Error|error" | cut -c1-160
