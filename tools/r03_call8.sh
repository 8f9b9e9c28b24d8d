#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
TAG=r03b
cd /tmp && export TMPDIR=/tmp
FBHIP_GEMM_LOG=1 python $ROOT/bench.py --steps 32 --warmup 32 --repeats 1 --no-cpu-baseline --no-single-update-probe > /dev/null 2> $OUT/${TAG}_gemmlog.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o t -- python $ROOT/bench.py --steps 320 --warmup 64 --repeats 1 --no-cpu-baseline --no-single-update-probe > $OUT/${TAG}_bench_under_rocprof.log 2>&1
DB=$(ls $OUT/${TAG}_trace/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then
  python $ROOT/tools/prof_summary.py $DB $OUT/${TAG}_kernel_stats.txt > /dev/null
  python $ROOT/tools/prof_timeline.py $DB > $OUT/${TAG}_step_timeline.txt
  rm -f $OUT/${TAG}_trace/*.db
fi
head -30 $OUT/${TAG}_kernel_stats.txt | cut -c1-160
cd $ROOT; timeout 600 python tools/tolerance_probe.py > $OUT/r03_tolerance_probe.txt 2>&1; tail -24 $OUT/r03_tolerance_probe.txt
