"""Soak: N updates back to back (graph replays, update_many), then check that every parameter / Adam moment is finite
and that the loss metrics are sane.  python tools/soak.py [n_steps]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import bench
from controllable_agent_amd.agent import FBHipAgent

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
torch.manual_seed(0)            # (the agent's Philox key and initial weights come from torch's seed: fixed, so checksums compare across runs)
W = bench.WALKER
agent = FBHipAgent(obs_type="states", obs_shape=(W["obs_dim"],), action_shape=(W["action_dim"],), device="cuda",
                   num_expl_steps=0, update_every_steps=1, batch_size=W["batch_size"], z_dim=W["z_dim"], use_tb=False,
                   use_wandb=False, use_hiplog=False)
rb = bench.make_replay(2000, 1000, W["obs_dim"], W["action_dim"], "cuda", seed=3)
t0 = time.time()
done = 0
while done < n:
    k = min(8, n - done)
    agent.update_many(rb, done, k) if k > 1 else agent.update(rb, done)
    done += k
agent.flush()
torch.cuda.synchronize()
dt = time.time() - t0
agent.cfg.use_tb = True
m = agent.update(rb, n)
bad = [k for net in ("actor", "forward_net", "backward_net", "forward_target_net", "backward_target_net")
       for k, v in getattr(agent, net).state_dict().items() if not torch.isfinite(v).all()]
print(f"{n} updates in {dt:.1f} s ({n / dt:.0f}/s); step counts {agent.step_counts()}; non-finite tensors: {bad}; "
      f"fb_loss {m['fb_loss']:.3f} actor_loss {m['actor_loss']:.3f} B_norm {m['B_norm']:.4f} orth_linf {m['orth_linf']:.4f}")
assert not bad and agent.step_counts() == (n + 1, n + 1) and np.isfinite(m["fb_loss"])
# run-to-run determinism of the whole trajectory (graph replays, two-branch pipelining, fixed-order reductions): a checksum
ck = float(sum(v.double().sum() for net in ("actor", "forward_net", "backward_net") for v in getattr(agent, net).state_dict().values()))
print(f"checksum {ck!r}")
