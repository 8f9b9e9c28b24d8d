#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03e_suite.log 2>&1; tail -25 $OUT/r03e_suite.log | cut -c1-300
# the library's own RCCL transport under torchrun, one rank (all this box holds), NCCL_DEBUG=INFO kept
NCCL_DEBUG=INFO FBHIP_FORCE_PHASE_SPLIT=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 \
   bench.py --gpus 1 --steps 640 --warmup 64 --repeats 3 --no-cpu-baseline --no-single-update-probe --nccl-world1 > $OUT/r03e_library_rccl_world1.log 2>&1
grep -E "^\{" $OUT/r03e_library_rccl_world1.log | cut -c1-300; grep -c "NCCL INFO" $OUT/r03e_library_rccl_world1.log
