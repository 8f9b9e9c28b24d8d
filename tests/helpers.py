"""Shared helpers for the parity tests (fixture loading; oracle replay)."""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import torch

from oracle import fb_oracle as fo
from oracle import discrete_fb_oracle as do

GOLDEN = Path(__file__).resolve().parent / "golden"


def load_meta(name: str) -> dict:
    return json.loads((GOLDEN / f"{name}.json").read_text())


def cfg_from_meta(meta: dict) -> fo.OracleConfig:
    return fo.OracleConfig(**meta["cfg"])


def regenerate_inputs(meta: dict):
    """Recreate (nets, storage, lengths, rng) exactly as tests/golden/make_golden.py::trace_fixture did."""
    cfg = cfg_from_meta(meta)
    rng = np.random.default_rng(meta["seed"])
    shapes = do.NET_SHAPES if meta.get("discrete") else fo.NET_SHAPES
    nets = {n: fo.synthetic_params(rng, shapes[n](cfg)) for n in shapes}
    lengths = None
    n_eps, T = meta["n_eps"], meta["T"]
    if meta["variable_len"]:
        lengths = rng.integers(max(2, T // 2), T + 1, size=n_eps).astype(np.int32)
        lengths[0] = T
    storage, lengths = fo.synthetic_storage(rng, n_eps, T, cfg.obs_dim, cfg.action_dim,
                                            cfg.goal_dim if cfg.use_goal else None, lengths)
    if meta.get("discrete"):
        do.synthetic_actions(rng, storage, cfg.action_dim)
    return cfg, nets, storage, lengths, rng


def checksums(state: dict) -> dict:
    return {k: [float(np.sum(v, dtype=np.float64)), float(np.sqrt(np.sum(np.asarray(v, np.float64) ** 2)))]
            for k, v in state.items() if not k.startswith("adam_")}


def rel_err(a, b) -> float:
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


LOSS_KEYS = ("fb_loss", "fb_offdiag", "fb_diag", "orth_loss", "orth_loss_offdiag", "actor_loss", "q")


# ------------------------------------------------------------------------------------------------ HIP agent glue
def agent_kwargs(cfg: fo.OracleConfig, goal_space=None, metrics=True, **extra):
    kw = dict(obs_type="states", obs_shape=(cfg.obs_dim,), action_shape=(cfg.action_dim,), device="cuda",
              num_expl_steps=0, use_tb=metrics, use_wandb=False, use_hiplog=False, goal_space=goal_space,
              lr=cfg.lr, lr_coef=cfg.lr_coef, fb_target_tau=cfg.fb_target_tau, hidden_dim=cfg.hidden_dim,
              backward_hidden_dim=cfg.backward_hidden_dim, feature_dim=cfg.feature_dim, z_dim=cfg.z_dim,
              stddev_schedule=str(cfg.stddev), stddev_clip=cfg.stddev_clip, batch_size=cfg.batch_size,
              ortho_coef=cfg.ortho_coef, mix_ratio=cfg.mix_ratio, q_loss=cfg.q_loss, q_loss_coef=cfg.q_loss_coef,
              future_ratio=cfg.future_ratio, norm_z=cfg.norm_z, rand_weight=cfg.rand_weight,
              add_trunk=cfg.add_trunk, preprocess=cfg.preprocess, boltzmann=cfg.boltzmann, temp=cfg.temp,
              log_std_bounds=(cfg.log_std_min, cfg.log_std_max), debug=cfg.debug, update_every_steps=1)
    kw.update(extra)
    return kw


def make_hip_agent(cfg: fo.OracleConfig, nets, goal_space=None, metrics=True, discrete=False):
    from controllable_agent_amd.agent import DiscreteFBHipAgent, FBHipAgent
    ag = (DiscreteFBHipAgent if discrete else FBHipAgent)(**agent_kwargs(cfg, goal_space, metrics))
    ag.load_nets({n: {k: v for k, v in p.items()} for n, p in nets.items()})
    return ag


NETS5 = ("actor", "forward_net", "backward_net", "forward_target_net", "backward_target_net")


def get_agent_state(agent) -> dict:
    out = {}
    has_actor = hasattr(agent, "actor")          # DiscreteFBHipAgent has none
    for n in NETS5[0 if has_actor else 1:]:
        for k, v in getattr(agent, n).state_dict().items():
            out[f"{n}/{k}"] = v.detach().cpu().numpy().copy()
    for n in ("actor", "forward_net", "backward_net")[0 if has_actor else 1:]:
        for mv in ("m", "v"):
            for k, v in agent._adam_views[n][mv].items():
                out[f"adam_{mv}/{n}/{k}"] = v.detach().cpu().numpy().copy()
    return out


def set_agent_state(agent, state: dict, fb_steps: int, actor_steps: int) -> None:
    """state: {'net/param': array, 'adam_m/net/param': array, ...} (the layout of the golden traces)"""
    has_actor = hasattr(agent, "actor")
    for n in NETS5[0 if has_actor else 1:]:
        sd = {k.split("/", 1)[1]: torch.from_numpy(np.asarray(v)) for k, v in state.items() if k.startswith(n + "/")}
        getattr(agent, n).load_state_dict(sd)
    for n in ("actor", "forward_net", "backward_net")[0 if has_actor else 1:]:
        for mv in ("m", "v"):
            for k, view in agent._adam_views[n][mv].items():
                key = f"adam_{mv}/{n}/{k}"
                if key in state:
                    view.copy_(torch.from_numpy(np.asarray(state[key])))
                else:
                    view.zero_()
    agent.set_step_counts(fb_steps, actor_steps)


def draws_dict(d: fo.Draws) -> dict:
    return {f: getattr(d, f) for f in d.__dataclass_fields__ if getattr(d, f) is not None}
