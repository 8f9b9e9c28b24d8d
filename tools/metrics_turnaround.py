"""Where the time of a metrics-ON update() goes (README.md:50: use_tb=1 use_hiplog=1): per call, on the host,
  launch  = the library call that enqueues the single-update graph (hipGraphLaunch inside)
  wait    = fbhip_wait_metrics: how long the host spun for the step's published metrics (long = the GPU is the pace-maker, ~0 = the
            host arrived late and the GPU had already drained: idle time)
  python  = everything else of update(): key, hyper-parameters, dict
and the rate.    python tools/metrics_turnaround.py [n]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench                                   # (sets ROC_CPU_WAIT_FOR_SIGNAL before torch is imported, like the bench line)
import torch
from controllable_agent_amd import _lib
from controllable_agent_amd.agent import FBHipAgent

W = bench.WALKER
agent = FBHipAgent(obs_type="states", obs_shape=(W["obs_dim"],), action_shape=(W["action_dim"],), device="cuda", num_expl_steps=0,
                   update_every_steps=1, batch_size=W["batch_size"], z_dim=W["z_dim"], use_tb=True, use_wandb=False, use_hiplog=True)
rb = bench.make_replay(1000, 1000, W["obs_dim"], W["action_dim"], "cuda", seed=1)
lib = _lib.load()
acc = {"launch": 0.0, "wait": 0.0}
real_u, real_w = lib.fbhip_update, lib.fbhip_wait_metrics


def timed(name, fn):
    def f(*a):
        t = time.perf_counter()
        r = fn(*a)
        acc[name] += time.perf_counter() - t
        return r
    return f


n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
for label, stream in (("high-priority stream", torch.cuda.Stream(priority=-1)), ("normal stream", torch.cuda.Stream())):
    with torch.cuda.stream(stream):
        for s in range(100):
            agent.update(rb, s)
        torch.cuda.synchronize()
        lib.fbhip_update, lib.fbhip_wait_metrics = timed("launch", real_u), timed("wait", real_w)
        acc["launch"] = acc["wait"] = 0.0
        t0 = time.perf_counter()
        for s in range(n):
            m = agent.update(rb, 100 + s)
        total = time.perf_counter() - t0
        torch.cuda.synchronize()
        lib.fbhip_update, lib.fbhip_wait_metrics = real_u, real_w
    us = lambda x: 1e6 * x / n
    print(f"metrics ON on a {label}: {n / total:.1f} update-steps/s = {us(total):.0f} us per call: launch {us(acc['launch']):.0f} us, "
          f"wait {us(acc['wait']):.0f} us, python {us(total - acc['launch'] - acc['wait']):.0f} us ({len(m)} keys)")
