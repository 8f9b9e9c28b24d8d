"""Print the kernel timeline of ONE update step (the last complete graph replay) from a rocprofv3 rocpd database."""
import sqlite3, sys
db = sys.argv[1]
con = sqlite3.connect(db)
rows = con.execute("select name, start, end, stream_id, queue_id from kernels order by start").fetchall()
# a step starts at the sampler's draw_kernel
starts = [i for i, r in enumerate(rows) if "draw_kernel" in r[0]]
if len(starts) < 3:
    print("no step boundary found"); sys.exit(0)
a, b = starts[-2], starts[-1]
t0 = rows[a][1]
periods = [(rows[starts[i + 1]][1] - rows[starts[i]][1]) / 1e3 for i in range(len(starts) // 2, len(starts) - 1)]
spans = [(rows[starts[i + 1] - 1][2] - rows[starts[i]][1]) / 1e3 for i in range(len(starts) // 2, len(starts) - 1)]
print(f"step of {b - a} kernels, span {(rows[b - 1][2] - t0) / 1e3:.1f} us; steady state: mean period {sum(periods) / len(periods):.1f} us, "
      f"mean span {sum(spans) / len(spans):.1f} us (gap between graph launches {sum(periods) / len(periods) - sum(spans) / len(spans):.1f} us)")
busy = 0
for r in rows[a:b]:
    n = r[0].replace("fbhip::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0][:40]
    print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f}  q{r[4]}  {n}")
