#!/bin/bash
# Round-3 hunt for the release-build segfault: the whole -m gpu suite with the 37 surface cases in the MAIN
# process (FBHIP_SURFACE_INNER=1 lifts their skip), N times under rocgdb so that a SIGSEGV / SIGABRT leaves a native backtrace.
N=${1:-4}
mkdir -p gpurun_out
for i in $(seq 1 $N); do
  FBHIP_SURFACE_INNER=1 PYTHONFAULTHANDLER=1 timeout 900 /opt/rocm/bin/rocgdb -batch -ex "handle SIGSEGV stop print" -ex run -ex bt -ex "info sharedlibrary fbhip" -ex "thread apply all bt 25" \
      --args python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/segv_hunt_$i.log 2>&1
  echo "run $i rc=$?" >> gpurun_out/segv_hunt_summary.txt
  grep -E "passed|failed|SIGSEGV|SIGABRT|Segmentation" gpurun_out/segv_hunt_$i.log | tail -3 >> gpurun_out/segv_hunt_summary.txt
done
cat gpurun_out/segv_hunt_summary.txt
