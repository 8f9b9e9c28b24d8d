#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 64 --warmup 32 --repeats 1 --no-cpu-baseline --no-single-update-probe"
FBHIP_FORCE_PHASE_SPLIT=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/w1slow -o t -- $B > $OUT/w1slow.log 2>&1
DB=$(ls $OUT/w1slow/*.db | head -1)
python - "$DB" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
print([r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')").fetchall()][:60])
rows = con.execute("select name, count(*), sum(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
for r in rows[:30]: print(f"{r[1]:7d} {r[2]:12.1f} us  {r[0][:110]}")
try:
    rows = con.execute("select name, count(*), sum(end-start)/1e3, sum(size) from memory_copies group by name").fetchall()
    for r in rows: print("memcpy", r)
except Exception as e:
    print("memcpy table:", e)
PY
grep "^{" $OUT/w1slow.log | cut -c1-200
rm -rf $OUT/w1slow
