#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
TL=/usr/local/lib/python3.10/dist-packages/torch/lib
export LD_LIBRARY_PATH=$TL:$LD_LIBRARY_PATH
{
echo "== torch's bundled runtime (LD_PRELOAD $TL/libamdhip64.so)"
LD_PRELOAD=$TL/libamdhip64.so timeout 120 $ROOT/tools/scratch/stream_queue_probe
LD_PRELOAD=$TL/libamdhip64.so timeout 900 $ROOT/tools/scratch/graph_queue_collision 3000
} > $OUT/r03_graph_queue_collision_torchrt.txt 2>&1
cat $OUT/r03_graph_queue_collision_torchrt.txt
