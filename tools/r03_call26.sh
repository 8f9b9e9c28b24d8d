#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
B="python bench.py --steps 640 --warmup 64 --repeats 3 --no-cpu-baseline --no-single-update-probe"
p() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(round(d['value'],1), 'host_enqueue', round(d['config']['host_enqueue_ms_per_step'],3))"; }
echo "== slow case as is"; FBHIP_FORCE_PHASE_SPLIT=1 $B 2>/dev/null | p
echo "== HSA_ENABLE_INTERRUPT=0"; HSA_ENABLE_INTERRUPT=0 FBHIP_FORCE_PHASE_SPLIT=1 $B 2>/dev/null | p
echo "== ROC_ACTIVE_WAIT_TIMEOUT=1000000"; ROC_ACTIVE_WAIT_TIMEOUT=1000000 FBHIP_FORCE_PHASE_SPLIT=1 $B 2>/dev/null | p
echo "== AMD_DIRECT_DISPATCH=0"; AMD_DIRECT_DISPATCH=0 FBHIP_FORCE_PHASE_SPLIT=1 $B 2>/dev/null | p
echo "== steps-per-launch 8"; FBHIP_FORCE_PHASE_SPLIT=1 $B --steps-per-launch 8 2>/dev/null | p
echo "== pipeline off"; FBHIP_UPDATE_PIPELINE=0 FBHIP_FORCE_PHASE_SPLIT=1 $B 2>/dev/null | p
echo "== peer transport cannot run at world 1; c10d no group:"; FBHIP_FORCE_PHASE_SPLIT=1 $B --transport c10d 2>/dev/null | p
cd /tmp && export TMPDIR=/tmp
FBHIP_FORCE_PHASE_SPLIT=1 timeout 600 rocprofv3 --hip-trace --stats -d $OUT/w1api -o t -- python $ROOT/bench.py --steps 64 --warmup 32 --repeats 1 --no-cpu-baseline --no-single-update-probe > $OUT/w1api.log 2>&1
python - "$(ls $OUT/w1api/*.db | head -1)" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, count(*), sum(end-start)/1e3 from regions group by name order by 1").fetchall()
print(" | ".join(f"{r[0]}:{r[1]}" for r in rows))
PY
rm -rf $OUT/w1api
