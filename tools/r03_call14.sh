#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for m in 0 1 2 3; do
  echo "== sf icm legacy stream, nullsync mode $m"; FBHIP_DBG_NULLSYNC=$m python tools/sf_bench.py --learner icm --steps 320 --warmup 64 --no-cpu-baseline 2>/dev/null | cut -c1-230
done
echo "== pipeline off, legacy"; FBHIP_UPDATE_PIPELINE=0 python tools/sf_bench.py --learner icm --steps 320 --warmup 64 --no-cpu-baseline 2>/dev/null | cut -c1-230
echo "== pipeline off, explicit"; FBHIP_UPDATE_PIPELINE=0 python tools/sf_bench.py --learner icm --steps 320 --warmup 64 --no-cpu-baseline --explicit-stream 2>/dev/null | cut -c1-230
echo "== quadruped P3 0/1/2"
for m in 0 1 2; do FBHIP_P3=$m python bench.py --workload quadruped --steps 320 --warmup 64 --repeats 3 2>/dev/null | cut -c1-120; done
