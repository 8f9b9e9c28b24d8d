#!/bin/bash
# Run ON THE GPU BOX: kernel trace of a short bench.py run -> gpurun_out/<tag>_kernel_stats.txt, <tag>_step_timeline.txt (no PMC passes)
#   tools/trace_step.sh <tag> [extra bench.py flags]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o t -- python bench.py --steps 320 --warmup 64 --repeats 1 --no-cpu-baseline --no-single-update-probe "$@" > $OUT/${TAG}_bench_under_rocprof.log 2>&1 )
DB=$(ls $OUT/${TAG}_trace/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then
  python $ROOT/tools/prof_summary.py $DB $OUT/${TAG}_kernel_stats.txt > /dev/null
  python $ROOT/tools/prof_timeline.py $DB > $OUT/${TAG}_step_timeline.txt
  rm -rf $OUT/${TAG}_trace
fi
head -70 $OUT/${TAG}_step_timeline.txt
