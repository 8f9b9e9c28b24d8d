#!/bin/bash
# N consecutive runs of the whole -m gpu suite in ONE pytest process each (release build), summary lines to gpurun_out/<tag>.txt
N=${1:-10}; TAG=${2:-r03_suite_repeat}
mkdir -p gpurun_out
: > gpurun_out/$TAG.txt
for i in $(seq 1 $N); do
  timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/${TAG}_last.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(grep -E '^(=+ )?[0-9]+ (passed|failed)|^[0-9]+ failed' gpurun_out/${TAG}_last.log | tail -1) [$(date +%H:%M:%S)]" >> gpurun_out/$TAG.txt
  if [ $rc -ne 0 ]; then cp gpurun_out/${TAG}_last.log gpurun_out/${TAG}_fail_$i.log; fi
done
cat gpurun_out/$TAG.txt
