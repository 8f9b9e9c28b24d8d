#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
timeout 600 $ROOT/tools/scratch/graph_queue_collision 3000 > $OUT/r03_graph_queue_collision.txt 2>&1
cat $OUT/r03_graph_queue_collision.txt
P=$ROOT/tools/scratch/gemm3_probe
{
for cfg in 0 1 3; do for k in 0 1 7; do $P single 4096 4096 1024 3 1 $cfg 10 $k; done; done
for cfg in 0 1 5 3; do $P single 1024 2048 1024 3 1 $cfg 20 0; $P single 1024 2048 1024 3 4 $cfg 20 0; done
for cfg in 0 1 3; do $P single 1024 1024 2048 2 4 $cfg 20 0; $P single 2048 1024 1024 0 4 $cfg 20 0; $P single 1024 512 1024 3 4 $cfg 20 0;  $P single 1024 1024 96 3 4 $cfg 20 0; done
} > $OUT/r03_g3_epi2.txt 2>&1
cat $OUT/r03_g3_epi2.txt
timeout 600 $P 5 2>&1 | grep -E "correctness|FAIL" | head -20
