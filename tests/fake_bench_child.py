"""Stand-in for a bench.py rank in the CPU test of bench.py::supervise_ranks (no GPU, no torch): writes the heartbeat file the
supervisor watches, then -- by FAKE_CHILD_PLAN = "<transport>:<cpu_wait>:<rank>:<crash|hang>,..." -- exits non-zero, sleeps, or
prints the one JSON line (rank 0) and exits 0."""
import json
import os
import sys
import time

rank = int(os.environ["RANK"])
transport = sys.argv[sys.argv.index("--transport") + 1]
cpu_wait = os.environ.get("ROC_CPU_WAIT_FOR_SIGNAL", "")
hb = os.environ["FBHIP_BENCH_HEARTBEAT"]
open(hb, "w").write(f"{time.time():.3f} started\n")
for item in filter(None, os.environ.get("FAKE_CHILD_PLAN", "").split(",")):
    t, cw, r, how = item.split(":")
    if t == transport and cw in ("*", cpu_wait) and int(r) == rank:
        if how == "hang":
            time.sleep(600)
        sys.exit(7)
time.sleep(0.5)
open(hb, "w").write(f"{time.time():.3f} timed repeat 1 of 1 done\n")
if rank == 0:
    print("some library banner on stdout")
    print(json.dumps({"metric": "fake", "value": 1.0, "n_gpus": int(os.environ["WORLD_SIZE"]), "data_parallel": {"transport": transport}}), flush=True)
