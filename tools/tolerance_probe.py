"""Round 3, VERDICT item 5: how far is the HIP step from the EXACT (fp64) evaluation of the same update, next to how far the
fp32 oracle is -- per gradient tensor class, over the 124 random configurations of tests/test_update_parity_gpu.py.  Prints the
distribution the tightened test bounds are derived from."""
import dataclasses
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import fb_oracle as fo                     # noqa: E402
from tests import helpers as H                         # noqa: E402
from tests.test_update_parity_gpu import _buffer, _random_case     # noqa: E402

rows = []
for seed in list(range(300, 400)) + list(range(400, 424)):
    cfg, goal_space = _random_case(seed)
    if seed >= 400:
        cfg = dataclasses.replace(cfg, debug=True, z_dim=cfg.goal_dim, batch_size=max(cfg.batch_size, 3 * cfg.goal_dim))
    rng = np.random.default_rng(1000 + seed)
    nets = {n: fo.synthetic_params(rng, fo.NET_SHAPES[n](cfg)) for n in ("actor", "forward_net", "backward_net")}
    storage, lengths = fo.synthetic_storage(rng, 6, 9, cfg.obs_dim, cfg.action_dim, cfg.goal_dim if cfg.use_goal else None)
    draws = fo.make_draws(rng, cfg, 6, lengths)
    batch = fo.gather_batch(storage, draws.ep_idx, draws.step_idx, cfg.discount, draws.future_idx)
    o64 = fo.OracleAgent(cfg, nets, torch.float64)
    o64.update(batch, draws, keep=True)
    if cfg.q_loss:
        Bm = o64.last["Bm"]
        if float(torch.linalg.cond(Bm.T @ Bm / Bm.shape[0])) > 1e6:
            cfg = dataclasses.replace(cfg, q_loss=False)
            o64 = fo.OracleAgent(cfg, nets, torch.float64)
            o64.update(batch, draws, keep=True)
    o32 = fo.OracleAgent(cfg, nets)
    o32.update(batch, draws, keep=True)
    agent = H.make_hip_agent(cfg, nets, goal_space)
    rb = _buffer(storage, lengths, cfg.discount, cfg.future)
    agent.update_injected(rb, 0, H.draws_dict(draws))
    cond = 1.0
    if cfg.q_loss:
        Bm = o64.last["Bm"]
        cond = float(torch.linalg.cond(Bm.T @ Bm / Bm.shape[0]))
    # smallest |Q1 - Q2| of the actor phase relative to |Q|: torch.min routes a row's whole gradient to one head
    q1 = (o64.last["aF1"] * o64.last["z"]).sum(1); q2 = (o64.last["aF2"] * o64.last["z"]).sum(1)
    gap = float(((q1 - q2).abs() / (q1.abs() + q2.abs() + 1e-30)).min())
    for net, key in (("forward_net", "grads_forward"), ("backward_net", "grads_backward"), ("actor", "grads_actor")):
        worst = (0.0, 0.0, 0.0, "")
        for k, g in agent._grad_views[net].state_dict().items():
            ref = o64.last[key][k]
            if float(ref.abs().max()) == 0.0:
                continue
            eh, eo = H.rel_err(g.cpu().double(), ref), H.rel_err(o32.last[key][k].double(), ref)
            if eh > worst[0]:
                worst = (eh, eo, eh / max(eo, 1e-12), k)
        rows.append((seed, net, worst[0], worst[1], worst[2], cond, gap, cfg.batch_size))
    del agent
for net in ("forward_net", "backward_net", "actor"):
    r = [x for x in rows if x[1] == net]
    eh = np.array([x[2] for x in r]); eo = np.array([x[3] for x in r]); ratio = np.array([x[4] for x in r])
    print(f"{net}: HIP err vs fp64  median {np.median(eh):.2e}  p90 {np.quantile(eh, 0.9):.2e}  max {eh.max():.2e} | fp32 oracle err  median "
          f"{np.median(eo):.2e}  max {eo.max():.2e} | ratio HIP/oracle  median {np.median(ratio):.1f}  p90 {np.quantile(ratio, 0.9):.1f}  max {ratio.max():.1f}")
    for x in sorted(r, key=lambda x: -x[4])[:6]:
        print(f"    seed {x[0]}  hip {x[2]:.2e}  oracle32 {x[3]:.2e}  ratio {x[4]:.1f}  cond {x[5]:.1e}  min rel |Q1-Q2| {x[6]:.1e}  batch {x[7]}")
