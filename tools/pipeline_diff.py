"""Diagnostic (not part of the product): n single updates vs update_many(n) at walker dims -- per-tensor differences of the
last step's gradients and of the parameters.  python tools/pipeline_diff.py <n>"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from oracle import fb_oracle as fo
from tests import helpers as H
from tests.test_update_parity_gpu import _buffer

n = int(sys.argv[1])
cfg = fo.OracleConfig(obs_dim=24, action_dim=6, goal_dim=24, z_dim=50, batch_size=1024)
rng = np.random.default_rng(5)
nets = {k: fo.synthetic_params(rng, fo.NET_SHAPES[k](cfg)) for k in ("actor", "forward_net", "backward_net")}
storage, lengths = fo.synthetic_storage(rng, 20, 100, cfg.obs_dim, cfg.action_dim)
rb = _buffer(storage, lengths, cfg.discount)
a1, a2 = (H.make_hip_agent(cfg, nets) for _ in range(2))
for s in range(n):
    a1.update(rb, s)
a2.update_many(rb, 0, n)
torch.cuda.synchronize()
for net in ("forward_net", "backward_net", "actor"):
    for (k, g1), (_, g2) in zip(a1._grad_views[net].state_dict().items(), a2._grad_views[net].state_dict().items()):
        print(f"grad {net}/{k:24s} rel-L2 {H.rel_err(g2.cpu(), g1.cpu()):.2e}  |g| {float(g1.abs().mean()):.2e}")
s1, s2 = H.get_agent_state(a1), H.get_agent_state(a2)
for k in s1:
    if not k.startswith("adam_"):
        d = np.abs(s1[k].astype(np.float64) - s2[k].astype(np.float64))
        print(f"param {k:40s} max {d.max():.2e}  frac>1e-6 {float((d > 1e-6).mean()):.4f}")
