"""HIP free-running against a reference metric-curve fixture: max relative deviation of the loss keys per step
(python tools/curve_probe.py walker_b256_50)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from oracle import fb_oracle as fo
from tests import helpers as H
from tests.test_update_parity_gpu import _buffer

name = sys.argv[1]
meta = H.load_meta(name)
cfg, nets, storage, lengths, rng = H.regenerate_inputs(meta)
agent = H.make_hip_agent(cfg, nets, meta["goal_space"])
rb = _buffer(storage, lengths, cfg.discount)
for s in range(meta["n_steps"]):
    d = fo.make_draws(rng, cfg, meta["n_eps"], lengths)
    m = agent.update_injected(rb, s, H.draws_dict(d))
    ref = meta["metrics"][s]
    scale = max(1.0, abs(ref["fb_offdiag"]))
    dev = {k: abs(m[k] - ref[k]) / max(abs(ref[k]), 1e-6) for k in H.LOSS_KEYS}
    dev_abs = max(abs(m[k] - ref[k]) / scale for k in H.LOSS_KEYS)
    if s in (0, 1, 2, 5, 10, 20, 30, 40, 49):
        print(s, "max rel %.1e (%s)  max abs/scale %.1e" % (max(dev.values()), max(dev, key=dev.get), dev_abs))
    if str(s + 1) in meta["checksums"]:
        refc = meta["checksums"][str(s + 1)]
        cs = H.checksums(H.get_agent_state(agent))
        print("   checksum step", s + 1, "max l2 rel %.1e" % max(abs(l2 - refc[k][1]) / max(refc[k][1], 1e-30) for k, (_, l2) in cs.items()))
