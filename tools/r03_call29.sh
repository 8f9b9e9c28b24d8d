#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --steps 640 --warmup 64 --repeats 3 --no-cpu-baseline --no-single-update-probe"
p() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(round(d['value'],1), 'host_enqueue', round(d['config']['host_enqueue_ms_per_step'],3), (d.get('data_parallel') or {}).get('transport'), [a['transport']+':'+a['ranks'][0]['outcome'] for a in (d.get('data_parallel') or {}).get('attempts', [])])"; }
echo "== plain single-GPU path, nccl group world 1 in the process"; $B --nccl-world1 2>/dev/null | p
echo "== plain single-GPU path"; $B 2>/dev/null | p
R="python bench.py --gpus 2 --rehearse-on-one-gpu --peer-allreduce --steps 320 --warmup 32 --repeats 3 --episodes 1000 --no-cpu-baseline --no-fallback-transports"
echo "== two ranks on one GPU, peer kernels"; $R 2>/dev/null | p
echo "== same, pipeline off"; FBHIP_UPDATE_PIPELINE=0 $R 2>/dev/null | p
echo "== same, AMD_DIRECT_DISPATCH=0"; AMD_DIRECT_DISPATCH=0 $R 2>/dev/null | p
