import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from tests import helpers as H
from oracle import fb_oracle as fo
from tests.test_update_parity_gpu import _buffer
name = sys.argv[1]; gs = sys.argv[2] if len(sys.argv) > 2 else None
meta = H.load_meta(name); cfg = H.cfg_from_meta(meta)
z = np.load(H.GOLDEN / f"{name}.npz")
storage = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith("storage/")}
lengths = z["lengths"]
nets = {n: {k.split("/", 2)[2]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"init/{n}/")} for n in ("actor", "forward_net", "backward_net")}
agent = H.make_hip_agent(cfg, nets, gs)
rb = _buffer(storage, lengths, cfg.discount, cfg.future)
oracle = fo.OracleAgent(cfg, nets)
for s in range(meta["n_steps"]):
    draws = fo.Draws(**{f: z[f"draws/{s}/{f}"] for f in fo.Draws.__dataclass_fields__ if f"draws/{s}/{f}" in z.files})
    if s > 0:
        prev = {k.split("/", 2)[2]: z[k] for k in z.files if k.startswith(f"state/{s - 1}/")}
        H.set_agent_state(agent, prev, s, s)
    om = oracle.update(fo.gather_batch(storage, draws.ep_idx, draws.step_idx, cfg.discount, draws.future_idx), draws, keep=True)
    m = agent.update_injected(rb, s, H.draws_dict(draws))
    for k, v in meta["metrics"][s].items():
        print(s, k, m[k], v, om.get(k), abs(m[k]-v)/max(abs(v),1e-12))
    for view, ref in (("z", oracle.last["z"]), ("next_action", oracle.last["next_action"]), ("F1", oracle.last["aF1"]),
                      ("F2", oracle.last["aF2"]), ("tF1", oracle.last["tF1"]), ("tF2", oracle.last["tF2"]), ("Bm", oracle.last["Bm"]), ("tB", oracle.last["tB"]),
                      ("pi_action", oracle.last["pi_action"]), ("mu", oracle.last["mu"])):
        print(s, view, H.rel_err(agent.workspace_view(view).cpu(), ref))
