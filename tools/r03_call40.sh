#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== driver-style short run"; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['steps'], d['warmup'], d.get('flags'), d['roofline']['frac'], d['cpu_baseline']['value'])"
echo "== torchrun form, 2 rehearsal ranks, default transport"; timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --rehearse-on-one-gpu --steps 64 --warmup 8 --repeats 2 --episodes 400 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['n_gpus'], d['replicas']['identical'], d['data_parallel']['transport'][:60], [(a['transport'], a['ROC_CPU_WAIT_FOR_SIGNAL'], a['ranks'][0]['outcome']) for a in d['data_parallel']['attempts']])"
