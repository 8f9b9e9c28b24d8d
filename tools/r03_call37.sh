#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 700 python -m pytest tests/test_distributed_gpu.py -m gpu -q -p no:cacheprovider -x -k "bench" 2>&1 | grep -E "passed|failed|Error|assert" | head -20 | cut -c1-400
