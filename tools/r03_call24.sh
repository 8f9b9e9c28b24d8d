#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for P in 0 1; do
FBHIP_GEMM_LOG=1 FBHIP_UPDATE_PIPELINE=$P timeout 600 rocprofv3 --kernel-trace -d $OUT/glr$P -o t -- python $ROOT/bench.py --steps-per-launch 1 --steps 64 --warmup 16 --repeats 1 --no-cpu-baseline --no-single-update-probe > $OUT/glr$P.out 2> $OUT/glr$P.log
DB=$(ls $OUT/glr$P/*.db | head -1)
python $ROOT/tools/gemm_launch_report.py $DB $OUT/glr$P.log > $OUT/r03_gemm_launch_report_pipeline$P.txt 2>&1
rm -rf $OUT/glr$P
done
cat $OUT/r03_gemm_launch_report_pipeline0.txt | cut -c1-260
