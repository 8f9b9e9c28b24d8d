#!/bin/bash
# Run ON THE GPU BOX (through gpurun): bench line, kernel trace and the three PMC passes of one round.
#   tools/profile_round.sh <tag>      -> gpurun_out/<tag>_*  (copy the summaries you want judged into profiles/)
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py --steps 1920 --warmup 192 --repeats 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o t -- python $ROOT/bench.py --steps 320 --warmup 64 --repeats 1 --no-cpu-baseline --no-single-update-probe > $OUT/${TAG}_bench_under_rocprof.log 2>&1
DB=$(ls $OUT/${TAG}_trace/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then
  python $ROOT/tools/prof_summary.py $DB $OUT/${TAG}_kernel_stats.txt > /dev/null
  python $ROOT/tools/prof_timeline.py $DB > $OUT/${TAG}_step_timeline.txt
  rm -f $OUT/${TAG}_trace/*.db
fi
# the GEMM kernels inside the benchmarked step: GFLOP (launch log) and kernel time (trace) per update and tile configuration
FBHIP_GEMM_LOG=1 timeout 600 rocprofv3 --kernel-trace -d $OUT/${TAG}_gtrace -o t -- python $ROOT/bench.py --steps 320 --warmup 64 --repeats 1 --no-cpu-baseline --no-single-update-probe --no-dominant-probe > /dev/null 2> $OUT/${TAG}_gemmlog.txt
DB=$(ls $OUT/${TAG}_gtrace/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then
  python $ROOT/tools/dominant_in_step.py $DB $OUT/${TAG}_gemmlog.txt $OUT/${TAG}_dominant_in_step.json > $OUT/${TAG}_gemm_in_step.txt
  python - <<PY
import json
f = "$OUT/${TAG}_dominant_in_step.json"
d = json.load(open(f)); d["tag"] = "$TAG"; json.dump(d, open(f, "w"), indent=1)
PY
  rm -rf $OUT/${TAG}_gtrace
fi
grep -c GEMMLOG $OUT/${TAG}_gemmlog.txt > /dev/null && grep GEMMLOG $OUT/${TAG}_gemmlog.txt | head -120 > $OUT/${TAG}_gemm_launch_log.txt; rm -f $OUT/${TAG}_gemmlog.txt
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 420 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$N -o p -- python $ROOT/bench.py --steps 64 --warmup 32 --repeats 1 --no-cpu-baseline --no-single-update-probe > $OUT/${TAG}_pmc_$N.log 2>&1
  python $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc_$N $OUT/${TAG}_pmc_$N.txt > /dev/null
  rm -rf $OUT/${TAG}_pmc_$N
done
cat $OUT/${TAG}_bench.json
