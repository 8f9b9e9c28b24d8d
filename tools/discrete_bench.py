#!/usr/bin/env python
"""Measurement for SURVEY section 8 row n4 (DiscreteFBAgent on the HIP path) -- NOT the bench line (bench.py is).

One step = one ``DiscreteFBHipAgent.update()`` (on-device sample + FB step + EMA; the agent has no actor), replayed as one
hipGraph, 32 steps per launch.  Dims: the walker networks with a 6-way discrete action (obs 24, A 6, z_dim 50, hidden 1024,
backward hidden 526, batch 1024) -- the reference publishes no benchmark configuration for this agent.  Prints one JSON line
with the same roofline / cpu_baseline objects as bench.py (cpu_baseline = oracle/discrete_fb_oracle.py on the host cores).

    python tools/discrete_bench.py [--steps 2000] [--warmup 100] [--actions 6] [--boltzmann] [--no-cpu-baseline]
"""
import os
os.environ.setdefault("ROC_CPU_WAIT_FOR_SIGNAL", "1")      # before the HIP runtime loads; see bench.py
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def gflop_per_update(o, A, g, d, H, Hb, B):
    """Minimal algorithm, FLOP = 2 MAC: target ForwardMap pass 1x, online 3x (forward + both backward products), BackwardMap
    0.5x (z-mix rows) + 1x (target) + 3x (online), pairwise loss and gradients 11 B d per row."""
    Ff = (o + d) * H + 2 * H * H + 2 * (H * H + H * d * A)
    Fb = g * Hb + Hb * Hb + Hb * d
    return 2 * (4 * Ff + 4.5 * Fb + 11 * B * d) * B / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--actions", type=int, default=6)
    ap.add_argument("--boltzmann", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    W = dict(bench.WALKER, action_dim=args.actions)
    dev = "cuda:0"
    torch.cuda.set_device(0)
    from controllable_agent_amd.agent import DiscreteFBHipAgent
    torch.manual_seed(1)
    agent = DiscreteFBHipAgent(obs_type="states", obs_shape=(W["obs_dim"],), action_shape=(W["action_dim"],), device=dev,
                               num_expl_steps=0, update_every_steps=1, batch_size=W["batch_size"], z_dim=W["z_dim"],
                               hidden_dim=W["hidden_dim"], feature_dim=W["feature_dim"],
                               backward_hidden_dim=W["backward_hidden_dim"], boltzmann=args.boltzmann,
                               use_tb=False, use_wandb=False, use_hiplog=False)
    rb = bench.make_replay(5000, 1000, W["obs_dim"], 1, dev, seed=100)
    rb._storage["action"] = torch.randint(0, args.actions, rb._storage["action"].shape, device=dev).float()
    rb._touch()
    spl = 32

    def run(first, n):
        done = 0
        while done < n:
            k = min(spl, n - done)
            agent.update_many(rb, first + done, k) if k > 1 else agent.update(rb, first + done)
            done += k

    run(0, args.warmup)
    for sz in {spl} | ({args.steps % spl} if args.steps % spl else set()):
        run(args.warmup, sz)
    agent.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.warmup, args.steps)
    agent.flush()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rate = args.steps / dt
    gf = gflop_per_update(W["obs_dim"], args.actions, W["goal_dim"], W["z_dim"], W["hidden_dim"], W["backward_hidden_dim"], W["batch_size"])
    out = {"metric": "DiscreteFB update-steps/sec (batch=1024, z_dim=50)", "value": rate, "unit": "update-steps/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"discrete_fb offline: obs 24, {args.actions} actions, z_dim 50, hidden 1024, backward hidden 526, batch 1024; "
                                  f"{'softmax' if args.boltzmann else 'greedy'} targets; 5000 x 1000 synthetic replay in HBM; metrics off",
                      "steps_per_graph_launch": spl},
           "roofline": {"bound": "mfma", "achieved": gf * rate / 1e3, "peak": bench.PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": gf * rate / 1e3 / bench.PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                        "what": f"whole update step: {gf:.2f} algorithmic GFLOP/update x measured updates/s vs the exact-fp32 MFMA peak"}}
    if not args.no_cpu_baseline:
        from oracle import discrete_fb_oracle as do
        from oracle import fb_oracle as fo
        cfg = fo.OracleConfig(**W, preprocess=False, boltzmann=args.boltzmann)
        rng = np.random.default_rng(1)
        nets = {n: fo.synthetic_params(rng, do.NET_SHAPES[n](cfg)) for n in do.NET_SHAPES}
        storage, lengths = fo.synthetic_storage(rng, 20, 100, cfg.obs_dim, cfg.action_dim)
        do.synthetic_actions(rng, storage, cfg.action_dim)
        torch.set_num_threads(16)
        ag = do.DiscreteOracleAgent(cfg, nets)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 12.0 and n < 120:
            d = fo.make_draws(rng, cfg, 20, lengths)
            ag.update(fo.gather_batch(storage, d.ep_idx, d.step_idx, cfg.discount, d.future_idx), d)
            n += 1
        el = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n / el, "unit": "update-steps/s", "cores": 16, "kind": "port",
                               "sample": f"{n} updates of the same workload with oracle/discrete_fb_oracle.py (torch-CPU fp32, autograd), {el:.1f} s"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
