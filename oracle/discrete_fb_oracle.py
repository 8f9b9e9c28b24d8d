"""CPU restatement of ``DiscreteFBAgent.update()`` (url_benchmark/agent/discrete_fb.py:383-468) -- TEST INFRASTRUCTURE ONLY.

The sibling of fb_oracle.py for the discrete-action FB agent: the same BackwardMap, z
sampling / mixing, pairwise FB + orthonormality loss, Adam and target EMA, around a ForwardMap WITHOUT an action input
whose heads emit ``[B, z_dim, A]`` (one successor embedding per action), no actor:

  * target embedding = the greedy action's column of the TARGET ForwardMap on next_obs (``boltzmann``: the
    softmax(next_Q / temp)-weighted mix of the columns)                              discrete_fb.py:289-303
  * online embedding = the column of the action actually taken                        discrete_fb.py:309-311

Same import rule as fb_oracle.py: tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` only.

PARITY STATUS: PINNED -- tests/test_oracle_golden.py replays the traces that tests/golden/make_golden.py recorded from
the real ``url_benchmark.agent.discrete_fb.DiscreteFBAgent`` (tiny_discrete_trace: greedy targets, goal space, hindsight;
tiny_discrete_boltz_trace: softmax targets, q_loss, norm_z=False): every parameter, target, Adam moment and metric per step.
"""
from __future__ import annotations

import math
import typing as tp

import numpy as np
import torch
import torch.nn.functional as F

from . import fb_oracle as fo

Params = fo.Params


def forward_map_shapes(cfg: fo.OracleConfig):
    """discrete_fb.ForwardMap with preprocess=False (discrete_fb.py:74-83; the preprocess=True branch of its forward, :91-94,
    reads an ``obs_action_net`` the constructor never builds, so only this one can run): trunk on cat([obs, z]), two heads
    ``mlp(H, H, "irelu", z_dim * A)``.  ``cfg.action_dim`` is A, the NUMBER of actions (discrete_fb.py:110)."""
    assert not cfg.preprocess, "discrete_fb.ForwardMap.forward only works with preprocess=False"
    o, A, d, H = cfg.obs_dim, cfg.action_dim, cfg.z_dim, cfg.hidden_dim
    out = fo._single_trunk_shapes(o + d, H)
    for head in ("F1", "F2"):
        out += [(f"{head}.0.weight", (H, H)), (f"{head}.0.bias", (H,)),
                (f"{head}.2.weight", (d * A, H)), (f"{head}.2.bias", (d * A,))]
    return out


NET_SHAPES = {"forward_net": forward_map_shapes, "backward_net": fo.backward_map_shapes}


def forward_map(p: Params, obs, z, n_actions: int) -> tp.Tuple[torch.Tensor, torch.Tensor]:
    """discrete_fb.ForwardMap.forward (discrete_fb.py:87-101): [B, d, A] per head."""
    h = fo._single_trunk(p, torch.cat([obs, z], dim=-1))
    outs = []
    for head in ("F1", "F2"):
        t = torch.relu(F.linear(h, p[f"{head}.0.weight"], p[f"{head}.0.bias"]))
        outs.append(F.linear(t, p[f"{head}.2.weight"], p[f"{head}.2.bias"]).reshape(-1, z.shape[-1], n_actions))
    return outs[0], outs[1]


def greedy_action(p: Params, obs, z, n_actions: int) -> torch.Tensor:
    """DiscreteFBAgent.act without exploration (discrete_fb.py:263-268)."""
    F1, F2 = forward_map(p, obs, z, n_actions)
    Q1, Q2 = [torch.einsum('sda, sd -> sa', Fi, z) for Fi in (F1, F2)]
    return torch.min(Q1, Q2).max(1)[1]


def synthetic_actions(rng: np.random.Generator, storage: tp.Dict[str, np.ndarray], n_actions: int) -> None:
    """Replace the continuous synthetic actions of ``fo.synthetic_storage`` by action indices stored as float32 [.., 1]
    (the in-memory buffer stores whatever the environment returned; update() casts to int64, discrete_fb.py:395)."""
    shp = storage["action"].shape[:2] + (1,)
    storage["action"] = rng.integers(0, n_actions, size=shp).astype(np.float32)


class DiscreteOracleAgent:
    """State of one DiscreteFBAgent (discrete_fb.py:103-165) and its ``update`` (:383-468)."""

    NETS = ("forward_net", "backward_net")

    def __init__(self, cfg: fo.OracleConfig, nets: tp.Dict[str, Params]) -> None:
        self.cfg = cfg
        self.forward_net = {k: v.clone() for k, v in nets["forward_net"].items()}
        self.backward_net = {k: v.clone() for k, v in nets["backward_net"].items()}
        self.forward_target_net = {k: v.clone() for k, v in self.forward_net.items()}      # discrete_fb.py:149-150
        self.backward_target_net = {k: v.clone() for k, v in self.backward_net.items()}
        z = lambda d: {k: torch.zeros_like(v) for k, v in d.items()}
        self.adam = {n: {"m": z(getattr(self, n)), "v": z(getattr(self, n))} for n in self.NETS}
        self.fb_steps = 0
        self.last: tp.Dict[str, tp.Any] = {}

    # the z-mix / hindsight code of discrete_fb.py:428-452 is the text of fb_ddpg.py:467-491
    mix_z = fo.OracleAgent.mix_z

    def update(self, batch: tp.Dict[str, np.ndarray], draws: fo.Draws, keep: bool = False) -> tp.Dict[str, float]:
        cfg = self.cfg
        A, d = cfg.action_dim, cfg.z_dim
        t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32)
        obs, next_obs = t(batch["obs"]), t(batch["next_obs"])
        action = torch.as_tensor(np.ascontiguousarray(batch["action"])).reshape(-1, 1).type(torch.int64)   # :395
        discount = t(batch["discount"]).reshape(-1, 1)
        next_goal, backward_input = next_obs, obs
        if cfg.use_goal:                                           # discrete_fb.py:398-400, 419-424
            next_goal, backward_input = t(batch["next_goal"]), t(batch["goal"])
        future_goal = None
        if cfg.future_ratio > 0:
            future_goal = t(batch["future_goal"] if cfg.use_goal else batch["future_obs"])
        z = fo.sample_z_from_gauss(t(draws.z_gauss), d, None if cfg.norm_z else t(draws.z_uniform))        # :409
        z = self.mix_z(z, backward_input, draws, future_goal)
        metrics: tp.Dict[str, float] = {}

        # ---------------- update_fb (discrete_fb.py:277-381) ---------------- #
        with torch.no_grad():
            tF1a, tF2a = forward_map(self.forward_target_net, next_obs, z, A)
            next_Q1, next_Q2 = [torch.einsum('sda, sd -> sa', Fi, z) for Fi in (tF1a, tF2a)]
            next_Q = torch.min(next_Q1, next_Q2)
            if cfg.boltzmann:                                      # :294-297
                pi = F.softmax(next_Q / cfg.temp, dim=-1)
                tF1, tF2 = [torch.einsum("sa, sda -> sd", pi, Fi) for Fi in (tF1a, tF2a)]
                next_Qv = torch.einsum("sa, sa -> s", pi, next_Q)
            else:                                                  # :299-302
                next_action = next_Q.max(1)[1]
                next_idx = next_action[:, None].repeat(1, d)[:, :, None]
                tF1, tF2 = [Fi.gather(-1, next_idx).squeeze(-1) for Fi in (tF1a, tF2a)]
                next_Qv = next_Q.max(1)[0]
            tB = fo.backward_map(self.backward_target_net, next_goal, d, cfg.norm_z)
        fp, bp = fo.OracleAgent._req(self.forward_net), fo.OracleAgent._req(self.backward_net)
        idxs = action.repeat(1, d)[:, :, None]                     # :309
        F1a, F2a = forward_map(fp, obs, z, A)
        F1, F2 = F1a.gather(-1, idxs).squeeze(-1), F2a.gather(-1, idxs).squeeze(-1)
        if cfg.debug:                                              # IdentityMap (discrete_fb.py:134-136): B = next_goal
            y = Bm = next_goal.clone().requires_grad_(keep)
        else:
            y = fo.backward_map_raw(bp, next_goal)
            Bm = math.sqrt(d) * F.normalize(y, dim=1) if cfg.norm_z else y * 1.0
        if keep:
            for x in (F1, F2) + (() if cfg.debug else (Bm, y)):
                x.retain_grad()
        L = fo.fb_loss_terms(F1, F2, Bm, tF1, tF2, tB, discount, 0.0)      # fb_loss = offdiag + diag first (:317) ...
        fb_loss = L["fb_offdiag"] + L["fb_diag"]
        if cfg.q_loss:                                             # :321-333, with pinv instead of inverse
            with torch.no_grad():
                cov = torch.matmul(Bm.T, Bm) / Bm.shape[0]
                inv_cov = torch.linalg.pinv(cov)
                implicit_reward = (torch.matmul(Bm, inv_cov) * z).sum(dim=1)
                target_Q = implicit_reward.detach() + discount.squeeze(1) * next_Qv
            Q1, Q2 = [torch.einsum('sd, sd -> s', Fi, z) for Fi in (F1, F2)]
            q_loss = F.mse_loss(Q1, target_Q) + F.mse_loss(Q2, target_Q)
            fb_loss = fb_loss + cfg.q_loss_coef * q_loss
            metrics["q_loss"] = q_loss.item()
        fb_loss = fb_loss + cfg.ortho_coef * L["orth_loss"]        # ... then the orthonormality term (:337-341)
        metrics.update({                                            # :349-370
            "target_M": L["target_M"].mean().item(), "M1": L["M1"].mean().item(), "F1": F1.mean().item(),
            "B": Bm.mean().item(), "B_norm": torch.norm(Bm, dim=-1).mean().item(),
            "z_norm": torch.norm(z, dim=-1).mean().item(), "fb_loss": fb_loss.item(),
            "fb_diag": L["fb_diag"].item(), "fb_offdiag": L["fb_offdiag"].item(),
            "orth_loss": L["orth_loss"].item(), "orth_loss_diag": L["orth_loss_diag"].item(),
            "orth_loss_offdiag": L["orth_loss_offdiag"].item()})
        with torch.no_grad():
            eye_diff = torch.matmul(Bm.T, Bm) / Bm.shape[0] - torch.eye(Bm.shape[1])
            metrics["orth_linf"] = torch.max(torch.abs(eye_diff)).item()
            metrics["orth_l2"] = eye_diff.norm().item() / math.sqrt(Bm.shape[1])
        metrics["fb_opt_lr"] = cfg.lr
        fb_loss.backward()
        gF = {k: v.grad for k, v in fp.items()}
        gB = {k: v.grad for k, v in bp.items()}
        self.fb_steps += 1
        fo.adam_step(self.forward_net, gF, self.adam["forward_net"]["m"], self.adam["forward_net"]["v"], self.fb_steps, cfg.lr)
        fo.adam_step(self.backward_net, gB, self.adam["backward_net"]["m"], self.adam["backward_net"]["v"], self.fb_steps,
                     cfg.lr_coef * cfg.lr)
        fo.soft_update(self.forward_net, self.forward_target_net, cfg.fb_target_tau)       # :462-465
        fo.soft_update(self.backward_net, self.backward_target_net, cfg.fb_target_tau)
        if keep:
            dd = lambda x: x.detach().clone()
            self.last = dict(z=dd(z), tF1=dd(tF1), tF2=dd(tF2), tB=dd(tB), F1=dd(F1), F2=dd(F2), Bm=dd(Bm), y=dd(y),
                             next_Q=dd(next_Qv), dF1=dd(F1.grad), dF2=dd(F2.grad), dBm=dd(Bm.grad),
                             grads_forward={k: dd(v) for k, v in gF.items()}, grads_backward={k: dd(v) for k, v in gB.items()})
        return metrics

    def state_tensors(self) -> tp.Dict[str, np.ndarray]:
        out = {}
        for n in ("forward_net", "backward_net", "forward_target_net", "backward_target_net"):
            for k, v in getattr(self, n).items():
                out[f"{n}/{k}"] = v.detach().numpy().copy()
        for n in self.NETS:
            for k in getattr(self, n):
                out[f"adam_m/{n}/{k}"] = self.adam[n]["m"][k].numpy().copy()
                out[f"adam_v/{n}/{k}"] = self.adam[n]["v"][k].numpy().copy()
        return out
