#!/bin/bash
# Run ON THE GPU BOX: the whole -m gpu suite against the bounds-checked debug build (make debug: -O1 -g -DFBHIP_DEBUG, device-side
# asserts on every replay gather index).  The debug library is built there (it is not shipped) and swapped in for this run only.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
make -C controllable_agent_amd/csrc debug > /dev/null 2>&1 || exit 1
cp controllable_agent_amd/libfbhip.so /tmp/libfbhip_release.so
cp controllable_agent_amd/libfbhip_debug.so controllable_agent_amd/libfbhip.so
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^  File\|^Extension\|^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -5 > gpurun_out/${1:-r02}_pytest_gpu_debug_build.txt
cp /tmp/libfbhip_release.so controllable_agent_amd/libfbhip.so
cat gpurun_out/${1:-r02}_pytest_gpu_debug_build.txt
