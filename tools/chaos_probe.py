"""Calibration (not part of the product): how fast do two runs that differ ONLY in fp32 summation order drift apart?
  python tools/chaos_probe.py dump <n> <out.npz>     # n single updates at walker dims, dump last gradients + parameters
  python tools/chaos_probe.py cmp a.npz b.npz
Run the two dumps with different FBHIP_SMALL_SPLIT_MAX_BLOCKS (different split-K factors, same math)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np

if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    worst = sorted(((float(np.linalg.norm(a[k].astype(np.float64) - b[k]) / max(np.linalg.norm(b[k]), 1e-30)), k) for k in a.files), reverse=True)
    for e, k in worst[:6]:
        print(f"{k:50s} rel-L2 {e:.2e}")
    sys.exit(0)
import torch
from oracle import fb_oracle as fo
from tests import helpers as H
from tests.test_update_parity_gpu import _buffer
n, out = int(sys.argv[2]), sys.argv[3]
cfg = fo.OracleConfig(obs_dim=24, action_dim=6, goal_dim=24, z_dim=50, batch_size=1024)
rng = np.random.default_rng(5)
nets = {k: fo.synthetic_params(rng, fo.NET_SHAPES[k](cfg)) for k in ("actor", "forward_net", "backward_net")}
storage, lengths = fo.synthetic_storage(rng, 20, 100, cfg.obs_dim, cfg.action_dim)
rb = _buffer(storage, lengths, cfg.discount)
torch.manual_seed(3)
a = H.make_hip_agent(cfg, nets)
for s in range(n):
    a.update(rb, s)
torch.cuda.synchronize()
d = {f"grad/{net}/{k}": g.cpu().numpy() for net in ("forward_net", "backward_net", "actor") for k, g in a._grad_views[net].state_dict().items()}
d.update({f"param/{k}": v for k, v in H.get_agent_state(a).items() if not k.startswith("adam_")})
np.savez(out, **d)
