import sys, faulthandler
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
faulthandler.enable()
import numpy as np, torch
from oracle import fb_oracle as fo
from tests import helpers as H
from controllable_agent_amd import _lib
par = int(sys.argv[1]); graph = int(sys.argv[2])
meta = H.load_meta("tiny_trace")
cfg, nets, storage, lengths, rng = H.regenerate_inputs(meta)
from controllable_agent_amd.replay import DeviceReplayBuffer
rb = DeviceReplayBuffer.from_arrays(storage, lengths, cfg.discount, device="cuda")
agent = H.make_hip_agent(cfg, nets)
_lib.check(_lib.load().fbhip_set_parallel(agent._ctx, par))
agent._use_graph = bool(graph)
for s in range(3):
    agent.update(rb, s)
    torch.cuda.synchronize()
    print("step", s, "ok", flush=True)
print("DONE par", par, "graph", graph)
