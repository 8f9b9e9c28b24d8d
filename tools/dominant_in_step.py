"""The grouped GEMM kernels INSIDE the benchmarked (pipelined, 32 updates per graph) step: GFLOP and kernel time per update of
every tile configuration, i.e. the in-step average rate of the dominant kernel that bench.py reports beside its launched-alone
figure (not part of the product).

  FBHIP_GEMM_LOG=1 rocprofv3 --kernel-trace -d DIR -o t -- python bench.py --steps 320 --warmup 64 --repeats 1 \
      --no-cpu-baseline --no-single-update-probe --no-dominant-probe 2> log
  python tools/dominant_in_step.py DIR/*.db log [out.json]

libfbhip prints one GEMMLOG line per grouped launch while the 32-update graph is captured (cfg, workgroups, GFLOP, problems); every
update of the graph holds the same launches, so GFLOP per update and configuration = the capture's total / 32.  Kernel time per
update = the trace's total per kernel name / the number of updates traced (draw_kernel launches)."""
import json, sqlite3, sys
from collections import defaultdict
db, log = sys.argv[1], sys.argv[2]
PEAK = 157.3
NAME = {0: "gemm_kernel<2, 2, 1, 32", 1: "gemm_kernel<2, 1, 2, 32", 2: "gemm_kernel<1, 2, 2, 32", 3: "gemm_kernel<1, 1, 4, 16",
        4: "gemm_kernel<4, 1, 1, 32", 5: "gemm_dma_kernel"}
gf, launches = defaultdict(float), defaultdict(int)
for l in open(log, errors="ignore"):
    if l.startswith("GEMMLOG"):
        kv = dict(x.split("=") for x in l.split(":", 1)[0].split()[1:])
        gf[int(kv["cfg"])] += float(kv["gflop"]); launches[int(kv["cfg"])] += 1
con = sqlite3.connect(db)
rows = con.execute("select name, sum(end - start) / 1e3, count(*) from kernels group by name").fetchall()
updates = sum(r[2] for r in rows if "draw_kernel" in r[0])
steps_in_graph = 32
out = {"updates_traced": updates, "steps_per_graph": steps_in_graph, "kernels": []}
print(f"{updates} updates traced; GEMMLOG lines of one {steps_in_graph}-update graph capture")
for cfg in sorted(gf):
    us = sum(r[1] for r in rows if NAME[cfg] in r[0]) / updates
    n = sum(r[2] for r in rows if NAME[cfg] in r[0]) / updates
    g = gf[cfg] / steps_in_graph
    rec = {"kernel": "fbhip::" + NAME[cfg] + ", *>", "launches_per_update": round(n, 2), "gflop_per_update": round(g, 3),
           "us_per_update": round(us, 1), "tflops_in_step": round(g / us * 1e3, 1), "frac_of_peak_in_step": round(g / us * 1e3 / PEAK, 3)}
    out["kernels"].append(rec)
    print(f"{rec['kernel']:42s} {n:5.1f} launches  {g:7.3f} GFLOP  {us:7.1f} us per update  {rec['tflops_in_step']:6.1f} TF/s = {rec['frac_of_peak_in_step']:.3f} of peak (kernel times overlap across the graph's two branches)")
tot_g = sum(gf.values()) / steps_in_graph
print(f"executed GEMM GFLOP per update: {tot_g:.2f}")
out["gemm_gflop_per_update"] = round(tot_g, 2)
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
