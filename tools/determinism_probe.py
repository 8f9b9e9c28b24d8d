"""Run-to-run determinism probe (not part of the product): checksum of all parameters after n updates, single launches or
update_many.  python tools/determinism_probe.py <n> <single|many>"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from controllable_agent_amd.agent import FBHipAgent

n, mode = int(sys.argv[1]), sys.argv[2]
W = bench.WALKER
torch.manual_seed(1)
agent = FBHipAgent(obs_type="states", obs_shape=(W["obs_dim"],), action_shape=(W["action_dim"],), device="cuda",
                   num_expl_steps=0, update_every_steps=1, batch_size=W["batch_size"], z_dim=W["z_dim"], use_tb=False,
                   use_wandb=False, use_hiplog=False)
rb = bench.make_replay(200, 1000, W["obs_dim"], W["action_dim"], "cuda", seed=3)
ck = lambda: float(sum(v.double().sum() for net in ("actor", "forward_net", "backward_net") for v in getattr(agent, net).state_dict().values()))
print("init", repr(ck()), "replay", repr(float(rb._storage["observation"].double().sum())))
done = 0
while done < n:
    k = min(8, n - done) if mode == "many" else 1
    agent.update_many(rb, done, k) if k > 1 else agent.update(rb, done)
    done += k
agent.flush()
torch.cuda.synchronize()
print(n, mode, repr(ck()))
