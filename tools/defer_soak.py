"""Deferred batching at the benchmarked dims (not part of the product): agent A takes N updates as update_many(32) calls, agent B as
N single ``agent.update()`` calls whose queue is flushed at random points (a state read every 1..200 calls, so the n-step graphs
it launches have every size 1..32).  Same seed, same replay: the two must end in the same state -- bit for bit if the grouping of
steps into graphs does not change any GEMM's K-slicing, else to fp32 summation order.
python tools/defer_soak.py [n_updates]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import bench
from controllable_agent_amd.agent import FBHipAgent

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
W = bench.WALKER
rb = bench.make_replay(500, 1000, W["obs_dim"], W["action_dim"], "cuda", seed=3)


def make():
    torch.manual_seed(0)
    return FBHipAgent(obs_type="states", obs_shape=(W["obs_dim"],), action_shape=(W["action_dim"],), device="cuda", num_expl_steps=0,
                      update_every_steps=1, batch_size=W["batch_size"], z_dim=W["z_dim"], use_tb=False, use_wandb=False, use_hiplog=False)


a, b = make(), make()
t0 = time.time()
for s in range(0, n, 32):
    a.update_many(rb, s, min(32, n - s))
torch.cuda.synchronize()
ta = time.time() - t0
rng = np.random.default_rng(1)
sizes = {}
t0 = time.time()
next_read = int(rng.integers(1, 200))
for s in range(n):
    assert b.update(rb, s) == {}
    if s + 1 == next_read:
        k = b.__dict__["_pending"][3] if b.__dict__.get("_pending") else 0
        sizes[k] = sizes.get(k, 0) + 1
        b.step_counts()                          # an observer: the queue goes out
        next_read += int(rng.integers(1, 200))
b.flush()
torch.cuda.synchronize()
tb = time.time() - t0
assert a.step_counts() == b.step_counts() == (n, n), (a.step_counts(), b.step_counts())
worst, same = 0.0, True
for net in ("actor", "forward_net", "backward_net", "forward_target_net", "backward_target_net"):
    for (k, x), (_, y) in zip(getattr(a, net).state_dict().items(), getattr(b, net).state_dict().items()):
        assert torch.isfinite(x).all() and torch.isfinite(y).all(), (net, k)
        if not torch.equal(x, y):
            same = False
            worst = max(worst, float((x - y).abs().max()))
print(f"{n} updates: update_many(32) {n / ta:.0f}/s, update() with {sum(sizes.values())} random flushes (tail graph sizes {sorted(sizes)[:5]}..{sorted(sizes)[-3:]}) "
      f"{n / tb:.0f}/s; final states {'IDENTICAL bit for bit' if same else f'differ, max |delta| = {worst:.3e}'}; rng counts {a.rng_counts()} {b.rng_counts()}")
