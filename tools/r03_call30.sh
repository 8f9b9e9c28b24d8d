#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_distributed_gpu.py -m gpu -q -p no:cacheprovider -x -k "bench or peer or rccl" 2>&1 | grep -E "passed|failed|Error|assert" | head -20 | cut -c1-300
timeout 600 python -m pytest tests/test_update_parity_gpu.py -m gpu -q -p no:cacheprovider -x -k "pipelined or update_many or legacy" 2>&1 | grep -E "passed|failed|Error|assert" | head | cut -c1-300
p() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(round(d['value'],1), (d.get('data_parallel') or {}).get('transport'), (d.get('data_parallel') or {}).get('graph_form'))"; }
echo "== two ranks on one GPU, peer kernels, calibrated"; python bench.py --gpus 2 --rehearse-on-one-gpu --peer-allreduce --steps 320 --warmup 32 --repeats 3 --episodes 1000 --no-cpu-baseline --no-fallback-transports 2>/dev/null | p
echo "== N=1 bench"; python bench.py --steps 640 --warmup 64 --repeats 3 --no-cpu-baseline --no-single-update-probe 2>/dev/null | p
