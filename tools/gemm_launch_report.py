"""Per-launch efficiency of the grouped GEMM launches of ONE update (not part of the product).

  FBHIP_GEMM_LOG=1 FBHIP_UPDATE_PIPELINE=0 rocprofv3 --kernel-trace -d DIR -o t -- python bench.py --steps-per-launch 1 ... 2> log
  python tools/gemm_launch_report.py DIR/*.db log

libfbhip prints one GEMMLOG line per grouped launch while the update graph is captured (cfg, workgroups, GFLOP, problems as
MxNxK/kslices); the kernel trace of a later replay has the same launches in the same order.
"""
import sqlite3, sys
db, log = sys.argv[1], sys.argv[2]
lines = [l.strip() for l in open(log, errors="ignore") if l.startswith("GEMMLOG")]
rows = sqlite3.connect(db).execute("select name, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "draw_kernel" in r[0]]
a, b = starts[-2], starts[-1]
gemms = [r for r in rows[a:b] if "gemm_kernel" in r[0] or "gemm_dma_kernel" in r[0] or "gemm_thin_kernel" in r[0]]
n = len(gemms)
lines = lines[:n]          # the update graph is the first thing captured (bench.py --steps-per-launch 1); later lines: probes
print(f"{n} GEMM launches per update, {len(lines)} log lines; total GEMM time {sum(r[2]-r[1] for r in gemms)/1e3:.1f} us of step span {(rows[b-1][2]-rows[a][1])/1e3:.1f} us")
tot = 0.0
for r, l in zip(gemms, lines):
    us = (r[2] - r[1]) / 1e3
    head, probs = l.split(":", 1)
    kv = dict(x.split("=") for x in head.split()[1:])
    gf = float(kv["gflop"]); tot += gf
    print(f"{us:7.1f} us {gf / us * 1e3:7.1f} TF/s  cfg {kv['cfg']} wgs {kv['wgs']:>5} {gf:7.3f} GF |{probs}")
print(f"executed GEMM GFLOP per update: {tot:.2f}")
