#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
p() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(round(d['value'],1), d['repeats']['steps_per_s']['min'], d['repeats']['steps_per_s']['max'])"; }
B="python bench.py --steps 960 --warmup 64 --repeats 3 --no-cpu-baseline --no-single-update-probe"
for K in FBHIP_NOP=1 HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=2000 "HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=2000" FBHIP_NOP=2; do
  echo "== ROC_CPU_WAIT_FOR_SIGNAL=1 + $K"; timeout 100 env $K $B 2>/dev/null | p
done
