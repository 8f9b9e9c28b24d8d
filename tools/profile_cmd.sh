#!/bin/bash
# Run ON THE GPU BOX (through gpurun): bench line, kernel trace, one step's timeline and the three PMC passes of ANY measurement script
# (bench.py with another --workload, tools/sf_bench.py, ...):
#   tools/profile_cmd.sh <tag> "<long command: the bench line>" "<short command: a few hundred steps for the traces>"
# -> gpurun_out/<tag>_*  (copy the summaries you want judged into profiles/)
set -u
TAG=$1; LONG=$2; SHORT=$3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
eval "$LONG" > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
( cd $ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o t -- $SHORT > $OUT/${TAG}_bench_under_rocprof.log 2>&1 )
DB=$(ls $OUT/${TAG}_trace/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then
  python $ROOT/tools/prof_summary.py $DB $OUT/${TAG}_kernel_stats.txt > /dev/null
  python $ROOT/tools/prof_timeline.py $DB > $OUT/${TAG}_step_timeline.txt
  rm -f $OUT/${TAG}_trace/*.db
fi
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  N=$(echo $C | cut -d' ' -f1)
  ( cd $ROOT && timeout 420 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$N -o p -- $SHORT > $OUT/${TAG}_pmc_$N.log 2>&1 )
  python $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc_$N $OUT/${TAG}_pmc_$N.txt > /dev/null
  rm -rf $OUT/${TAG}_pmc_$N
done
tail -c 600 $OUT/${TAG}_bench.json; echo; head -12 $OUT/${TAG}_kernel_stats.txt | cut -c1-150
