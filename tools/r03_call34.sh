#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
B="python bench.py --steps 960 --warmup 64 --repeats 3 --no-cpu-baseline --no-single-update-probe"
p() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(round(d['value'],1), 'host_enqueue', round(d['config']['host_enqueue_ms_per_step'],3))"; }
echo "== N=1 default"; timeout 120 $B 2>/dev/null | p
echo "== N=1 ROC_CPU_WAIT_FOR_SIGNAL=1"; timeout 120 env ROC_CPU_WAIT_FOR_SIGNAL=1 $B 2>/dev/null | p
echo "== dp slow case + nccl group + ROC_CPU_WAIT_FOR_SIGNAL=1"; timeout 120 env ROC_CPU_WAIT_FOR_SIGNAL=1 FBHIP_FORCE_PHASE_SPLIT=1 FBHIP_UPDATE_PIPELINE=1 $B --nccl-world1 2>/dev/null | p
