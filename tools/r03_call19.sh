#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
B="python bench.py --steps 960 --warmup 64 --repeats 3 --no-cpu-baseline --no-single-update-probe"
run() { echo "== $1"; env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['steps_per_s'], 'host_enqueue', d['config']['host_enqueue_ms_per_step'])"; }
{
run "FBHIP_NOP=1"
run "HIP_FORCE_DEV_KERNARG=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "ROC_USE_FGS_KERNARG=0"
run "ROC_USE_FGS_KERNARG=1"
run "GPU_MAX_HW_QUEUES=8"
run "GPU_MAX_HW_QUEUES=2"
run "DEBUG_HIP_FORCE_GRAPH_QUEUES=1"
run "DEBUG_HIP_DYNAMIC_QUEUES=1"
run "AMD_DIRECT_DISPATCH=0"
run "FBHIP_NOP=2"
} > $OUT/r03_runtime_knobs.txt 2>&1
cat $OUT/r03_runtime_knobs.txt | cut -c1-200
