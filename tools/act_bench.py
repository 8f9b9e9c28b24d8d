"""Latency of the batch-1 inference calls the online loop makes every env step (pretrain.py:628-632, 651-652)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from controllable_agent_amd.agent import FBHipAgent

agent = FBHipAgent(obs_type="states", obs_shape=(24,), action_shape=(6,), device="cuda", num_expl_steps=0,
                   use_tb=False, use_wandb=False, use_hiplog=False, goal_space=None)
rng = np.random.default_rng(0)
obs = rng.standard_normal(24).astype(np.float32)
meta = agent.init_meta()
import types
ts = types.SimpleNamespace(observation=obs, goal=None)
for name, fn in (("act(eval)", lambda: agent.act(obs, meta, 0, eval_mode=True)),
                 ("act(explore)", lambda: agent.act(obs, meta, 0, eval_mode=False)),
                 ("compute_z_correl", lambda: agent.compute_z_correl(ts, meta))):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 500
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{name:>18}: {(time.perf_counter() - t0) / n * 1e6:8.1f} us per call")

# raw C-ABI latency (no Python wrapper work)
from controllable_agent_amd import _lib
lib = _lib.load()
g = np.ascontiguousarray(obs); z = np.ascontiguousarray(meta["z"], np.float32); out = np.empty(8, np.float32)
for name, call in (("fbhip_act raw", lambda: lib.fbhip_act(agent._ctx, g.ctypes.data, z.ctypes.data, None, 0.2, 1, out.ctypes.data, agent._stream.cuda_stream)),
                   ("fbhip_z_correl raw", lambda: lib.fbhip_z_correl(agent._ctx, g.ctypes.data, z.ctypes.data, out.ctypes.data, agent._stream.cuda_stream))):
    for _ in range(50):
        call()
    t0 = time.perf_counter()
    for _ in range(1000):
        call()
    print(f"{name:>18}: {(time.perf_counter() - t0) / 1000 * 1e6:8.1f} us per call")
