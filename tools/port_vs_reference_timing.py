"""BUILD CONTAINER ONLY (needs /root/reference): is the CPU port that bench.py times on the GPU box (oracle/fb_oracle.py,
``cpu_baseline.kind = "port"``) as fast / slow as the REAL reference on the same cores?  BASELINE.md section 4 promised
"timing within +-10 % of the reference on the same 8 cores"; this script is that check.

    python tools/port_vs_reference_timing.py [--threads 8] [--updates 30] > profiles/r02_port_vs_reference_timing.json

Both sides run walker dims (obs 24, action 6, z_dim 50, hidden 1024) at batch 256 (configs[0]) and 1024 (configs[1]),
metrics on (``use_tb=True``), same torch thread count, interleaved A/B/A/B so that host noise hits both alike:
* reference: ``FBDDPGAgent.update(replay_loader, step)`` of /root/reference imported with the stub modules of
  tests/golden/make_golden.py, sampling its own ``ReplayBuffer`` (its numpy / torch RNG);
* port: ``OracleAgent.update(gather_batch(...), draws)`` with host-side draws, like bench.py::_cpu_port_rate.
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests" / "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--updates", type=int, default=30)
    args = ap.parse_args()
    import make_golden as G
    from oracle import fb_oracle as fo
    R = G.import_reference()
    torch.set_num_threads(args.threads)
    out = {"threads": args.threads, "updates_per_arm_per_round": args.updates, "rows": []}
    for B in (256, 1024):
        cfg = fo.OracleConfig(obs_dim=24, action_dim=6, goal_dim=24, z_dim=50, batch_size=B)
        rng = np.random.default_rng(1)
        nets = {n: fo.synthetic_params(rng, fo.NET_SHAPES[n](cfg)) for n in ("actor", "forward_net", "backward_net")}
        n_eps, T = 20, 200
        storage, lengths = fo.synthetic_storage(rng, n_eps, T, cfg.obs_dim, cfg.action_dim)
        ref = G.make_ref_agent(R, cfg)
        G.load_nets(ref, nets)
        rb = G.fill_ref_buffer(R, storage, lengths, cfg.discount)
        port = fo.OracleAgent(cfg, nets)

        def ref_one(step):
            ref.update(rb, step)

        def port_one(step):
            d = fo.make_draws(rng, cfg, n_eps, lengths)
            port.update(fo.gather_batch(storage, d.ep_idx, d.step_idx, cfg.discount), d)

        for f in (ref_one, port_one):                 # warm-up
            for i in range(3):
                f(i)
        t = {"reference": [], "port": []}
        for rnd in range(3):                          # A/B/A/B/A/B
            for name, f in (("reference", ref_one), ("port", port_one)):
                t0 = time.perf_counter()
                for i in range(args.updates):
                    f(rnd * args.updates + i)
                t[name].append((time.perf_counter() - t0) / args.updates)
        r, p = float(np.median(t["reference"])), float(np.median(t["port"]))
        out["rows"].append({"batch": B, "reference_ms_per_update": 1e3 * r, "port_ms_per_update": 1e3 * p,
                            "port_over_reference": p / r, "reference_rounds_ms": [1e3 * x for x in t["reference"]],
                            "port_rounds_ms": [1e3 * x for x in t["port"]]})
    out["within_10_percent"] = all(abs(row["port_over_reference"] - 1) <= 0.10 for row in out["rows"])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
