#!/bin/bash
# Run ON THE GPU BOX: configs[4] rehearsal (tools/online_bench.py) as a same-box A/B of deferred batching -- the default (run-length
# rule: a lone update() is launched at once) against FBHIP_UPDATE_DEFER=0 (every update() eager), env 0 and 300 us, 2 alternating
# pairs.    tools/online_ab.sh <tag>  -> gpurun_out/<tag>_online_ab.txt
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_online_ab.txt
mkdir -p $ROOT/gpurun_out
cd $ROOT
: > $OUT
for pair in 1 2; do
  for us in 0 300; do
    for defer in 1 0; do
      echo "## pair $pair  --env-us $us  FBHIP_UPDATE_DEFER=$defer" >> $OUT
      FBHIP_UPDATE_DEFER=$defer python tools/online_bench.py --frames 12000 --env-us $us 2>&1 | grep "online loop" >> $OUT
    done
  done
done
cat $OUT
