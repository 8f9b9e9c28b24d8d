#!/bin/bash
# gpurun call 2 of round 3: P3 GEMM probe + lifecycle stress variants (each in its own process, HIP error log on)
mkdir -p gpurun_out
timeout 900 tools/scratch/gemm3_probe 20 > gpurun_out/r03_gemm3_probe.txt 2>&1; echo "probe rc=$?" >> gpurun_out/r03_gemm3_probe.txt
for v in "400 0" "400 1" "200 0 pickle" "400 50"; do
  tag=$(echo $v | tr ' ' '_')
  AMD_LOG_LEVEL=1 timeout 600 python tools/lifecycle_stress.py $v > gpurun_out/r03_stress_$tag.txt 2>&1; echo "rc=$?" >> gpurun_out/r03_stress_$tag.txt
  tail -3 gpurun_out/r03_stress_$tag.txt
done
tail -40 gpurun_out/r03_gemm3_probe.txt
