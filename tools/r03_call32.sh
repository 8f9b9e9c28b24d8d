#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
B="python bench.py --steps 640 --warmup 64 --repeats 3 --no-cpu-baseline --no-single-update-probe"
p() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(round(d['value'],1), 'host_enqueue', round(d['config']['host_enqueue_ms_per_step'],3))"; }
export FBHIP_FORCE_PHASE_SPLIT=1 FBHIP_UPDATE_PIPELINE=1
echo "== slow case (dp graph pipelined, library rccl, no group)"; $B 2>/dev/null | p
for K in ROC_CPU_WAIT_FOR_SIGNAL=0 ROC_CPU_WAIT_FOR_SIGNAL=1 ROC_SIGNAL_POOL_SIZE=4096 ROC_SIGNAL_POOL_SIZE=16 ROC_SYSTEM_SCOPE_SIGNAL=0 ROC_AQL_QUEUE_SIZE=65536 GPU_STREAMOPS_CP_WAIT=1 DEBUG_CLR_MAX_BATCH_SIZE=1 DEBUG_CLR_BATCH_CPU_SYNC_SIZE=1 AMD_SERIALIZE_KERNEL=0 HIP_LAUNCH_BLOCKING=0; do
  echo "== $K"; timeout 120 env $K $B 2>/dev/null | p    # (ROC_SYSTEM_SCOPE_SIGNAL=0 hangs: never without a timeout)
done
unset FBHIP_FORCE_PHASE_SPLIT FBHIP_UPDATE_PIPELINE
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03l_suite.log 2>&1; grep -E "passed|failed" $OUT/r03l_suite.log | tail -3 | cut -c1-300
