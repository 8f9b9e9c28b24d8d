#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
B="python bench.py --steps 640 --warmup 64 --repeats 3 --no-cpu-baseline --no-single-update-probe"
p() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(round(d['value'],1), 'host_enqueue', round(d['config']['host_enqueue_ms_per_step'],3), (d.get('data_parallel') or {}).get('transport'))"; }
echo "== phase split forced, library rccl world 1, plain"; FBHIP_FORCE_PHASE_SPLIT=1 $B 2>/dev/null | p
echo "== same, OMP_NUM_THREADS=1"; OMP_NUM_THREADS=1 FBHIP_FORCE_PHASE_SPLIT=1 $B 2>/dev/null | p
echo "== same, NCCL_DEBUG=INFO"; NCCL_DEBUG=INFO FBHIP_FORCE_PHASE_SPLIT=1 $B 2>/dev/null | p
echo "== same, transport c10d (no process group)"; FBHIP_FORCE_PHASE_SPLIT=1 $B --transport c10d 2>/dev/null | p
echo "== torchrun world 1 + nccl group"; FBHIP_FORCE_PHASE_SPLIT=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 640 --warmup 64 --repeats 3 --no-cpu-baseline --no-single-update-probe --nccl-world1 2>/dev/null | p
echo "== plain, nccl group world 1 without torchrun"; FBHIP_FORCE_PHASE_SPLIT=1 $B --nccl-world1 2>/dev/null | p
for m in gate event; do
  echo "== four-rank peer test, legacy order $m"; FBHIP_LEGACY_STREAM_ORDER=$m timeout 600 python -m pytest tests/test_distributed_gpu.py -m gpu -q -p no:cacheprovider -k "four_ranks or peer_allreduce_inside" --durations=3 2>&1 | grep -E "passed|failed|s call" | cut -c1-200
done
