#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
TL=/usr/local/lib/python3.10/dist-packages/torch/lib
X=tools/scratch/cross_queue_dep_cost
{
echo "# tools/cross_queue_dep_cost.hip on torch 2.10's bundled HIP runtime (ROCm 7.0, LD_PRELOAD): a two-branch graph of tiny kernels against the same kernels on one stream"
for CW in "" 1; do for EX in 0 8 24; do for HI in 1 0; do
  if [ -z "$CW" ]; then LD_PRELOAD=$TL/libamdhip64.so LD_LIBRARY_PATH=$TL timeout 60 $X 200 $EX $HI 2>&1 | grep -v amdgpu.ids
  else ROC_CPU_WAIT_FOR_SIGNAL=$CW LD_PRELOAD=$TL/libamdhip64.so LD_LIBRARY_PATH=$TL timeout 60 $X 200 $EX $HI 2>&1 | grep -v amdgpu.ids; fi
done; done; done
echo "# the same against /opt/rocm (ROCm 7.2)"
timeout 60 $X 200 0 1 2>&1 | grep -v amdgpu.ids
ROC_CPU_WAIT_FOR_SIGNAL=1 timeout 60 $X 200 0 1 2>&1 | grep -v amdgpu.ids
} > $OUT/r03_cross_queue_dep_cost.txt 2>&1
cat $OUT/r03_cross_queue_dep_cost.txt | cut -c1-260
