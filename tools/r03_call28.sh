#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
B="python bench.py --steps 8 --warmup 4 --repeats 1 --no-cpu-baseline --no-single-update-probe --steps-per-launch 4"
FBHIP_GRAPH_DOT=$OUT/g_rccl.dot FBHIP_FORCE_PHASE_SPLIT=1 $B > /dev/null 2>&1
FBHIP_GRAPH_DOT=$OUT/g_plain.dot $B > /dev/null 2>&1
for f in g_rccl g_plain; do
  echo "== $f: nodes $(grep -c 'label=' $OUT/$f.dot) edges $(grep -c -- '->' $OUT/$f.dot)"
  grep -o 'label="[^"]*"' $OUT/$f.dot | sed -E 's/[0-9]+//g' | cut -c1-60 | sort | uniq -c | sort -rn | head -30
done
