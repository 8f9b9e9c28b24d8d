#!/bin/bash
# Run ON THE GPU BOX: kernel-trace stats + a few PMC passes of tools/pairwise_bench.py for one shape.
#   tools/pairwise_pmc.sh <tag> <BxD>     -> gpurun_out/<tag>_pw_*
set -u
TAG=$1; SHAPE=${2:-2048x100}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $ROOT && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_pw_trace -o t -- python tools/pairwise_bench.py $SHAPE > $OUT/${TAG}_pw.log 2>&1 )
DB=$(ls $OUT/${TAG}_pw_trace/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $ROOT/tools/prof_summary.py $DB $OUT/${TAG}_pw_kernel_stats.txt > /dev/null && rm -f $OUT/${TAG}_pw_trace/*.db
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  N=$(echo $C | cut -d' ' -f1)
  ( cd $ROOT && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pwpmc_$N -o p -- python tools/pairwise_bench.py $SHAPE > $OUT/${TAG}_pwpmc_$N.log 2>&1 )
  python $ROOT/tools/pmc_summary.py $OUT/${TAG}_pwpmc_$N $OUT/${TAG}_pwpmc_$N.txt > /dev/null
  rm -rf $OUT/${TAG}_pwpmc_$N
done
head -6 $OUT/${TAG}_pw_kernel_stats.txt | cut -c1-160
for f in $OUT/${TAG}_pwpmc_*.txt; do grep -i "pairwise_kernel\|^kernel\|name" $f | head -4 | cut -c1-220; done
