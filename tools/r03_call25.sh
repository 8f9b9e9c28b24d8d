#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 64 --warmup 32 --repeats 1 --no-cpu-baseline --no-single-update-probe"
for CASE in slow fast; do
  EXTRA=""; [ $CASE = fast ] && EXTRA="--nccl-world1"
  FBHIP_FORCE_PHASE_SPLIT=1 timeout 600 rocprofv3 --kernel-trace --hip-trace --stats -d $OUT/w1$CASE -o t -- $B $EXTRA > $OUT/w1$CASE.log 2>&1
  DB=$(ls $OUT/w1$CASE/*.db | head -1)
  echo "=== $CASE: $(grep '^{' $OUT/w1$CASE.log | cut -c1-120)"
  python - "$DB" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
views = [r[0] for r in con.execute("select name from sqlite_master where type='view'").fetchall()]
print("views:", views[:40])
for r in con.execute("select name, count(*), sum(end-start)/1e3 from kernels group by name order by 2 desc").fetchall()[:40]:
    if "fbhip" not in r[0] and "at::native" not in r[0]: print(f"kernel {r[1]:7d} {r[2]:12.1f} us  {r[0][:100]}")
for v in views:
    if "region" in v.lower() or "api" in v.lower():
        try:
            rows = con.execute(f"select name, count(*), sum(end-start)/1e3 from {v} group by name order by 3 desc").fetchall()
            print("--", v)
            for r in rows[:25]: print(f"   {r[1]:8d} {r[2]:12.1f} us  {r[0][:80]}")
        except Exception as e:
            print(v, "->", e)
PY
  rm -rf $OUT/w1$CASE
done
