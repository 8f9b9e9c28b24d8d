#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
P=$ROOT/tools/scratch/gemm3_probe
timeout 900 $P 5 2>&1 | grep -E "correctness|FAIL" | head -20 > $OUT/r03c_probe_check.txt; cat $OUT/r03c_probe_check.txt
{
# forward (weights image as B), dgrad (weights image as B, k-strided), wgrad (both fp32) -- the update's operand modes
for cfg in 0 1 3; do
  $P single 1024 2048 1024 3 1 $cfg 20 0 0 1; $P single 1024 2048 1024 3 4 $cfg 20 0 0 1
  $P single 1024 1024 2048 2 2 $cfg 20 0 0 1; $P single 2048 1024 1024 0 2 $cfg 20 0 0 0
  $P single 1024 512 1024 3 4 $cfg 20 0 0 1; $P single 1024 1024 1024 3 2 $cfg 20 0 0 1
done
$P single 4096 4096 1024 3 1 0 10 0 0 1; $P single 4096 4096 1024 3 1 0 10 0 1 1; $P single 4096 4096 4096 3 1 0 5 0 0 1
} > $OUT/r03c_probe_time.txt 2>&1; cat $OUT/r03c_probe_time.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -p no:cacheprovider -k "p3" > $OUT/r03c_p3_kernel_tests.log 2>&1; tail -3 $OUT/r03c_p3_kernel_tests.log
for m in 1 0; do FBHIP_P3=$m python bench.py --steps 640 --warmup 64 --repeats 3 --no-cpu-baseline --no-single-update-probe 2> $OUT/r03c_bench_p3_$m.err | tee $OUT/r03c_bench_p3_$m.json | cut -c1-200; done
FBHIP_GEMM_LOG=1 python bench.py --steps 32 --warmup 32 --repeats 1 --no-cpu-baseline --no-single-update-probe > /dev/null 2> $OUT/r03c_gemmlog.txt
FBHIP_P3=2 timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/r03c_suite_mode2.log 2>&1; tail -8 $OUT/r03c_suite_mode2.log
