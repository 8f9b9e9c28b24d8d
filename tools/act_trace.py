"""Timeline of the LAST few batch-1 calls (kernels + memory copies) from a rocprofv3 rocpd database:
    rocprofv3 --kernel-trace --memory-copy-trace -d out -o t -- python tools/act_bench.py ;  python tools/act_trace.py out/t_results.db"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
names = [r[0] for r in con.execute("select name from sqlite_master where type in ('view','table')")]
ev = [(r[1], r[2], r[0].replace("fbhip::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48])
      for r in con.execute("select name, start, end from kernels")]
mc = [n for n in names if "memory_cop" in n and not n.startswith("rocpd_")]
if mc:
    cols = [r[1] for r in con.execute(f"pragma table_info({mc[0]})")]
    nm = "name" if "name" in cols else cols[0]
    for r in con.execute(f"select {nm}, start, end from {mc[0]}"):
        ev.append((r[1], r[2], f"COPY {r[0]}"))
ev.sort()
tail = ev[-40:]
t0 = tail[0][0]
prev_end = t0
for s, e, n in tail:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:6.1f} us  {n}")
    prev_end = e
