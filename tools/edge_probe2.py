"""Diagnostic (not part of the product): per-tensor differences of one edge-dimension case against the oracle (which
activation / gradient first leaves the tolerance).  python tools/edge_probe2.py"""
import sys, os, subprocess
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "dump":
    from oracle import fb_oracle as fo
    from tests import helpers as H
    from tests.test_update_parity_gpu import _buffer
    dims = dict(obs_dim=40, action_dim=20, goal_dim=40, z_dim=128, hidden_dim=2048, feature_dim=1024, backward_hidden_dim=2048, batch_size=96)
    cfg = fo.OracleConfig(lr=1e-3, **dims)
    rng = np.random.default_rng(41)
    nets = {n: fo.synthetic_params(rng, fo.NET_SHAPES[n](cfg)) for n in ("actor", "forward_net", "backward_net")}
    storage, lengths = fo.synthetic_storage(rng, 6, 9, cfg.obs_dim, cfg.action_dim)
    agent = H.make_hip_agent(cfg, nets)
    rb = _buffer(storage, lengths, cfg.discount, cfg.future)
    draws = fo.make_draws(rng, cfg, 6, lengths)
    # stop after the FB backward: patch dp_update to run only those phases
    from controllable_agent_amd import distributed as D
    import controllable_agent_amd.agent as A
    def only_fb(run_phases, fb, ac, exchange=None, early=None):
        run_phases(D.PHASE_SAMPLE | D.PHASE_FB_FWD | D.PHASE_FB_BWD | D.PHASE_ACTOR_FWD)
    D.dp_update = only_fb
    agent.update_injected(rb, 0, H.draws_dict(draws))
    out = {v: agent.workspace_view(v).cpu().numpy() for v in ("mu", "pi_action", "dp", "dh", "dt1a", "actor_p", "actor_h", "actor_premu", "online_p", "online_h", "z", "obs", "F1", "tF1", "dF1", "dBm")}
    for net in ("forward_net",):
        for k, g in agent._grad_views[net].state_dict().items():
            out["g_" + k] = g.cpu().numpy().reshape(g.shape[0], -1) if g.dim() > 1 else g.cpu().numpy().reshape(1, -1)
    np.savez(sys.argv[2], **out)
else:
    a, b = np.load("/tmp/m1.npz"), np.load("/tmp/m0.npz")
    for k in a.files:
        d = np.abs(a[k].astype(np.float64) - b[k])
        rows = np.where(d.max(axis=1) > 1e-4 * max(np.abs(b[k]).max(), 1e-30))[0]
        print(f"{k:12s} rel {np.linalg.norm(a[k].astype(np.float64)-b[k])/max(np.linalg.norm(b[k]),1e-30):.2e} bad rows {rows[:10].tolist()} shape {a[k].shape}")
