#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
B="python bench.py --steps 640 --warmup 64 --repeats 3 --no-cpu-baseline --no-single-update-probe"
p() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(round(d['value'],1), 'host_enqueue', round(d['config']['host_enqueue_ms_per_step'],3))"; }
echo "== gloo group world 1 + library rccl"; FBHIP_BENCH_WORLD1_BACKEND=gloo FBHIP_FORCE_PHASE_SPLIT=1 $B --nccl-world1 2>/dev/null | p
echo "== nccl group world 1 + library rccl"; FBHIP_FORCE_PHASE_SPLIT=1 $B --nccl-world1 2>/dev/null | p
for n in 1 2 3 4 6 8; do echo "== no group + $n extra streams"; FBHIP_BENCH_EXTRA_STREAMS=$n FBHIP_FORCE_PHASE_SPLIT=1 $B 2>/dev/null | p; done
echo "== plain single-GPU bench + 3 extra streams"; FBHIP_BENCH_EXTRA_STREAMS=3 $B 2>/dev/null | p
echo "== plain single-GPU bench + 6 extra streams"; FBHIP_BENCH_EXTRA_STREAMS=6 $B 2>/dev/null | p
