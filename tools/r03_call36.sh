#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
p() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(round(d['value'],1), 'probe us', round(d['roofline']['dominant_kernel']['us'],2))"; }
B="python bench.py --steps 960 --warmup 64 --repeats 3 --no-cpu-baseline --no-single-update-probe"
for k in 1 0 1 0; do echo "== ROC_CPU_WAIT_FOR_SIGNAL=$k"; timeout 120 env ROC_CPU_WAIT_FOR_SIGNAL=$k $B 2>/dev/null | p; done
echo "== steps-per-launch 64"; timeout 120 $B --steps-per-launch 64 2>/dev/null | p
echo "== steps-per-launch 16"; timeout 120 $B --steps-per-launch 16 2>/dev/null | p
