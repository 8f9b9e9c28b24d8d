"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel table (calls, total/avg us, %)."""
import sqlite3
import sys


def main(db, out=None, skip_first=0):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    rows = cur.execute(f"select {name_col}, start, end, grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count from kernels order by start").fetchall() \
        if {"grid_x", "workgroup_x"} <= set(cols) else cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    agg = {}
    for r in rows:
        n = r[0]
        d = (r[2] - r[1]) / 1e3
        a = agg.setdefault(n, [0, 0.0, 1e30, 0.0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = [f"{'kernel':<90} {'calls':>8} {'total_us':>12} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6}"]
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = n if len(n) <= 90 else n[:87] + "..."
        lines.append(f"{short:<90} {a[0]:>8} {a[1]:>12.1f} {a[1] / a[0]:>9.2f} {a[2]:>9.2f} {a[3]:>9.2f} {100 * a[1] / tot:>6.2f}")
    lines.append(f"TOTAL kernel time {tot:.1f} us over {sum(a[0] for a in agg.values())} dispatches; "
                 f"wall span {(rows[-1][2] - rows[0][1]) / 1e3:.1f} us")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
