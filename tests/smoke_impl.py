"""__graft_entry__.smoke(): one tiny injected update on cuda:0 through libfbhip.so, checked against the oracle."""
import numpy as np
import torch

from oracle import fb_oracle as fo
from tests import helpers as H


def run_smoke() -> None:
    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    cfg = fo.OracleConfig(obs_dim=24, action_dim=6, goal_dim=24, z_dim=50, hidden_dim=128, feature_dim=64,
                          backward_hidden_dim=62, batch_size=64)
    rng = np.random.default_rng(0)
    nets = {n: fo.synthetic_params(rng, fo.NET_SHAPES[n](cfg)) for n in ("actor", "forward_net", "backward_net")}
    storage, lengths = fo.synthetic_storage(rng, 8, 30, cfg.obs_dim, cfg.action_dim)
    from controllable_agent_amd.replay import DeviceReplayBuffer
    rb = DeviceReplayBuffer.from_arrays(storage, lengths, cfg.discount, device="cuda:0")
    agent = H.make_hip_agent(cfg, nets)
    oracle = fo.OracleAgent(cfg, nets)
    for s in range(2):
        d = fo.make_draws(rng, cfg, 8, lengths)
        want = oracle.update(fo.gather_batch(storage, d.ep_idx, d.step_idx, cfg.discount), d)
        got = agent.update_injected(rb, s, H.draws_dict(d))
        for k in ("fb_loss", "fb_offdiag", "orth_loss", "actor_loss", "q"):
            assert abs(got[k] - want[k]) <= 1e-4 * max(1.0, abs(want[k])), (s, k, got[k], want[k])
    agent.update(rb, 2)                          # production mode: on-device sampler + hipGraph (queued: metrics are off ...)
    agent.flush()                                # ... and launched here
    torch.cuda.synchronize()
    assert agent.step_counts()[0] >= 1
    print("smoke ok:", {k: round(got[k], 5) for k in ("fb_loss", "actor_loss")})
