#!/usr/bin/env python
"""Measurement for SURVEY section 8 row n4, second sibling (SFAgent on the HIP path) -- NOT the bench line (bench.py is).

One step = one ``SFHipAgent.update()`` (on-device sample + successor-feature TD step + feature-learner step + actor step +
target EMA) replayed as one hipGraph, 32 steps per launch.  Dims: the walker inputs with the reference's SFAgent defaults
(sf.py:37-82: z_dim 100, hidden 1024, feature 512, backward hidden 512, batch 1024, lr_coef 5, q_loss on) -- the reference
publishes no benchmark configuration for this agent.  Prints one JSON line with the roofline / cpu_baseline objects of bench.py
(cpu_baseline = oracle/sf_oracle.py on the host cores).

    python tools/sf_bench.py [--steps 2000] [--warmup 100] [--learner icm|lap|random|autoencoder|transition|svd_p|latent|svd_sr|svd_srv2|contrastive|contrastivev2] [--no-cpu-baseline]
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


def gflop_per_update(o, a, g, d, H, Fd, Hb, B, learner):
    """Minimal algorithm, FLOP = 2 MAC.  Actor: 1x (next_obs, no grad) + 3x (trained pass); successor_net: 1x target + 3x online
    + 1x forward in the actor phase + its action-path data gradient; feature_net on 2B rows trained (3x each); icm: the inverse-
    dynamics mlp trained (3x); lap: the orthonormality products (3 B d per row: Cov, and two gradient contractions)."""
    Ff = (o + a) * H + H * Fd + (o + d) * H + H * Fd + 2 * (2 * Fd * H + H * d)
    Fa = o * H + H * Fd + (o + d) * H + H * Fd + 2 * Fd * H + H * a
    Fphi = g * Hb + Hb * Hb + Hb * d
    Fdg = 2 * (H * d + 2 * Fd * H) + H * Fd + a * H
    head = {"icm": (2 * d, a), "autoencoder": (d, g), "transition": (d + a, g), "latent": (d + a, d)}.get(learner)
    extra = 3 * (head[0] * Hb + Hb * Hb + Hb * head[1]) if head else (3 * B * d if learner == "lap" else 0)
    # feature_net passes: forward on [goal ; next_goal] always; trained through both (icm, lap), through goal only (autoencoder,
    # transition) or not at all (random)
    nphi = {"icm": 6, "lap": 6, "autoencoder": 4, "transition": 4, "random": 2, "svd_p": 4, "latent": 5, "svd_sr": 5, "svd_srv2": 5, "contrastive": 4, "contrastivev2": 5}[learner]   # latent / svd_sr: + target nets on next_goal
    if learner in ("svd_sr", "svd_srv2"):    # mu_net trained (3x) + target_mu_net (1x) + the low-rank and orthonormality products
        extra = 4 * (g * Hb + Hb * Hb + Hb * d) + 9 * B * d
    if learner in ("contrastive", "contrastivev2"):    # mu_net on the hindsight goal trained (3x) + the logits and the two gradient contractions
        extra = 3 * (g * Hb + Hb * Hb + Hb * d) + 3 * B * d
    if learner == "svd_p":        # mu_net on [goal | action] trained (3x) + the low-rank and orthonormality products (P, Cov and their contractions)
        extra = 3 * ((g + a) * Hb + Hb * Hb + Hb * d) + 6 * B * d
    return 2 * (4 * Fa + 5 * Ff + Fdg + nphi * Fphi + extra) * B / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--learner", choices=("icm", "lap", "random", "autoencoder", "transition", "svd_p", "latent", "svd_sr", "svd_srv2", "contrastive", "contrastivev2"), default="icm")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--explicit-stream", action="store_true", help="enqueue from an explicit torch stream (like bench.py) instead of the legacy default stream")
    ap.add_argument("--mix-ratio", type=float, default=0.0, help="SFAgent.mix_ratio (sf.py:725-739); the reference default is 0")
    args = ap.parse_args()
    W = dict(obs_dim=24, action_dim=6, goal_dim=24, z_dim=100, hidden_dim=1024, feature_dim=512, backward_hidden_dim=512, batch_size=1024)
    dev = "cuda:0"
    torch.cuda.set_device(0)
    from controllable_agent_amd.agent import SFHipAgent
    torch.manual_seed(1)
    agent = SFHipAgent(obs_type="states", obs_shape=(W["obs_dim"],), action_shape=(W["action_dim"],), device=dev, num_expl_steps=0,
                       update_every_steps=1, feature_learner=args.learner, mix_ratio=args.mix_ratio, use_tb=False, use_wandb=False, use_hiplog=False)
    rb = bench.make_replay(5000, 1000, W["obs_dim"], W["action_dim"], dev, seed=100)
    spl = 32

    def run(first, n):
        done = 0
        while done < n:
            k = min(spl, n - done)
            agent.update_many(rb, first + done, k) if k > 1 else agent.update(rb, first + done)
            done += k

    if args.explicit_stream:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    run(0, args.warmup)
    for sz in {spl} | ({args.steps % spl} if args.steps % spl else set()):
        run(args.warmup, sz)
    rates = []
    for rep in range(3):
        agent.flush()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.warmup, args.steps)
        agent.flush()
        torch.cuda.synchronize()
        rates.append(args.steps / (time.perf_counter() - t0))
    rate = sorted(rates)[1]
    agent.cfg.use_tb = True
    m = agent.update(rb, 0)
    gf = gflop_per_update(W["obs_dim"], W["action_dim"], W["goal_dim"], W["z_dim"], W["hidden_dim"], W["feature_dim"],
                          W["backward_hidden_dim"], W["batch_size"], args.learner)
    out = {"metric": "SF update-steps/sec (batch=1024, z_dim=100)", "value": rate, "unit": "update-steps/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 / rate, "higher_is_better": True, "vs_baseline": None,
           "dtype": "f32", "data": "synthetic", "repeats": rates,
           "config": {"workload": f"sf offline (SFAgent defaults, feature_learner={args.learner}, q_loss on{f', mix_ratio {args.mix_ratio}' if args.mix_ratio else ''}): obs 24, action 6, z_dim 100, "
                                  "hidden 1024, feature 512, backward hidden 512, batch 1024; 5000 x 1000 synthetic replay in HBM; "
                                  "metrics off in the timed loop", "steps_per_graph_launch": spl,
                      "metrics_after": {k: m[k] for k in ("sf_loss", "phi_loss", "actor_loss", "phi_norm", "z_norm") if k in m}},
           "roofline": {"bound": "mfma", "achieved": gf * rate / 1e3, "peak": bench.PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": gf * rate / 1e3 / bench.PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                        "what": f"whole update step: {gf:.2f} algorithmic GFLOP/update x measured updates/s vs the exact-fp32 MFMA peak"}}
    if not args.no_cpu_baseline:
        from oracle import fb_oracle as fo
        from oracle import sf_oracle as so
        cfg = fo.OracleConfig(**W, lr_coef=5.0, mix_ratio=args.mix_ratio)
        rng = np.random.default_rng(1)
        shapes = so.net_shapes(cfg, args.learner)
        nets = {n: fo.synthetic_params(rng, shapes[n]) for n in shapes}
        storage, lengths = fo.synthetic_storage(rng, 20, 100, cfg.obs_dim, cfg.action_dim)
        torch.set_num_threads(16)
        ag = so.SFOracleAgent(cfg, nets, args.learner, True)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 12.0 and n < 120:
            d = fo.make_draws(rng, cfg, 20, lengths)
            ag.update(fo.gather_batch(storage, d.ep_idx, d.step_idx, cfg.discount), d)
            n += 1
        el = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n / el, "unit": "update-steps/s", "cores": 16, "kind": "port",
                               "sample": f"{n} updates of the same workload with oracle/sf_oracle.py (torch-CPU fp32, autograd), {el:.1f} s"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
