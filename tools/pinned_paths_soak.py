"""Soak of the two results that leave a running graph through pinned host memory + a sequence number (round 6):
  * metrics: after every metrics-on update() the dict from fbhip_wait_metrics (returned while the update's tail still runs) must
    equal what fbhip_read_metrics copies after the stream has drained -- n_updates times;
  * batch-1 act / compute_z_correl: an agent on the direct path and a twin on the copy + synchronise path (FBHIP_INFER_DIRECT=0 at
    construction) with the same weights must return the same bits for n_calls random observations, updates interleaved.
    python tools/pinned_paths_soak.py [n_updates] [n_calls]"""
import ctypes as C
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import bench
import torch
from controllable_agent_amd import _lib
from controllable_agent_amd.agent import FBHipAgent

n_updates = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
W = bench.WALKER
kw = dict(obs_type="states", obs_shape=(W["obs_dim"],), action_shape=(W["action_dim"],), device="cuda", num_expl_steps=0,
          update_every_steps=1, batch_size=W["batch_size"], z_dim=W["z_dim"], use_wandb=False, use_hiplog=False)
rb = bench.make_replay(200, 1000, W["obs_dim"], W["action_dim"], "cuda", seed=3)
lib = _lib.load()

torch.manual_seed(2)
a = FBHipAgent(use_tb=True, **kw)
bad = 0
buf = (C.c_float * _lib.NUM_METRICS)()
with torch.cuda.stream(torch.cuda.Stream(priority=-1)):
    for s in range(n_updates):
        m = a.update(rb, s)
        _lib.check(lib.fbhip_read_metrics(a.__dict__["_ctx"], buf, _lib.stream_ptr()))          # (drains the stream first)
        for k in ("fb_loss", "actor_loss", "q", "B_norm", "orth_linf", "F1", "actor_logprob", "fb_offdiag"):
            if m[k] != float(buf[_lib.METRIC_INDEX[k]]) or not np.isfinite(m[k]):
                bad += 1
    torch.cuda.synchronize()
print(f"metrics: {n_updates} metrics-on updates, published dict vs drained copy: {bad} mismatches; last fb_loss {m['fb_loss']:.4f}")

torch.manual_seed(2)
d1 = FBHipAgent(use_tb=False, **kw)
os.environ["FBHIP_INFER_DIRECT"] = "0"
torch.manual_seed(2)
d0 = FBHipAgent(use_tb=False, **kw)
os.environ.pop("FBHIP_INFER_DIRECT")
d0.defer_updates = d1.defer_updates = False
rng = np.random.default_rng(0)
meta = {"z": (rng.standard_normal(W["z_dim"]) * 3).astype(np.float32)}
import types
bad_act = bad_zc = 0
for i in range(n_calls):
    obs = rng.standard_normal(W["obs_dim"]).astype(np.float32)
    if i % 500 == 0:                               # weights move under both twins the same way
        d1.update(rb, i)
        d0.update(rb, i)
    x1, x0 = d1.act(obs, meta, i, eval_mode=True), d0.act(obs, meta, i, eval_mode=True)
    bad_act += int(not np.array_equal(x1, x0))
    if i % 4 == 0:
        ts = types.SimpleNamespace(observation=obs, goal=None)
        bad_zc += int(d1.compute_z_correl(ts, meta) != d0.compute_z_correl(ts, meta))
print(f"batch-1: {n_calls} act calls + {n_calls // 4} compute_z_correl calls, direct path vs copy + synchronise: {bad_act} / {bad_zc} mismatches")
sys.exit(1 if (bad or bad_act or bad_zc) else 0)
