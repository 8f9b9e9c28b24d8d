"""update-steps/s with the reference's metric dict ON (use_tb=True: every update computes the 18 metrics of
fb_ddpg.py:356-377,413-418 and the caller reads them back each step, like train_offline.py:118-119).
    python tools/metrics_on_bench.py [--off] [-n 1500]
--off: the same loop of single update() calls with metrics OFF and deferred batching disabled (every call launches its own
single-update graph at once): the GPU-side yardstick the metrics-on rate is compared with."""
import argparse, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
ap = argparse.ArgumentParser()
ap.add_argument("--off", action="store_true")
ap.add_argument("-n", type=int, default=1500)
a = ap.parse_args()
if a.off:
    os.environ["FBHIP_UPDATE_DEFER"] = "0"
import bench                                   # (sets ROC_CPU_WAIT_FOR_SIGNAL before torch is imported, like the bench line)
import torch

W = bench.WALKER
from controllable_agent_amd.agent import FBHipAgent
agent = FBHipAgent(obs_type="states", obs_shape=(W["obs_dim"],), action_shape=(W["action_dim"],), device="cuda",
                   num_expl_steps=0, update_every_steps=1, batch_size=W["batch_size"], z_dim=W["z_dim"], use_tb=not a.off,
                   use_wandb=False, use_hiplog=False)
rb = bench.make_replay(1000, 1000, W["obs_dim"], W["action_dim"], "cuda", seed=1)
with torch.cuda.stream(torch.cuda.Stream(priority=-1)):
    for s in range(100):
        m = agent.update(rb, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = a.n
    for s in range(n):
        m = agent.update(rb, 100 + s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
if a.off:
    print(f"metrics OFF, one single-update graph per call (FBHIP_UPDATE_DEFER=0): {n / dt:.1f} update-steps/s")
else:
    print(f"metrics ON, read back every step: {n / dt:.1f} update-steps/s; last metrics: "
          f"fb_loss {m['fb_loss']:.3f} actor_loss {m['actor_loss']:.3f} ({len(m)} keys)")
