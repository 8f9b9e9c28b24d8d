"""CPU restatement of ``SFAgent.update()`` (url_benchmark/agent/sf.py:700-768) -- TEST INFRASTRUCTURE ONLY.

The second sibling of fb_oracle.py: the successor-feature agent.  Same ``Actor`` and
``ForwardMap`` modules (the latter under the name ``successor_net``), same actor phase, Adam and target EMA; what differs:

  * the critic loss is a TD regression on successor features instead of the pairwise FB loss       sf.py:594-664
        target_F = phi(next_goal) + discount * next_F[argmin_i next_F_i . z]
        q_loss (default):  mse(F1 . z, target_F . z) + mse(F2 . z, target_F . z);   else  mse(F1, target_F) + mse(F2, target_F)
  * the embedding phi is ``feature_learner.feature_net`` = mlp(g, Hb, "ntanh", Hb, "relu", d, "L2") -- the BackwardMap
    architecture with the projection as its last module (sf.py:84-88, fb_modules.py:33-40) -- trained by its own loss with
    its own optimiser at ``lr_coef * lr`` (sf.py:461-463, 649-653):
        icm  (sf.py:194-213):  mean((action - tanh-mlp(cat[phi(goal), phi(next_goal)]))^2)      inverse dynamics
        lap  (sf.py:100-116):  mean((phi(goal) - phi(next_goal))^2) + mean_{s!=t} Cov^2 - 2 mean_s Cov_ss,  Cov = phi phi^T
  * z is ``sqrt(d) normalize(randn)``; with ``mix_ratio > 0`` the rows drawn by the mix uniform take the whitened feature of a permuted next goal (sf.py:724-739)

Same import rule as fb_oracle.py: tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` only.

PARITY STATUS: PINNED -- tests/test_oracle_golden.py replays the traces that tests/golden/make_golden.py recorded from the
real ``url_benchmark.agent.sf.SFAgent`` (tiny_sf_icm_trace: icm + q_loss; tiny_sf_lap_trace: lap + feature-space mse + goal
space + variable lengths): every parameter, target, Adam moment and metric per step.
"""
from __future__ import annotations

import math
import typing as tp

import numpy as np
import torch
import torch.nn.functional as F

from . import fb_oracle as fo

Params = fo.Params


def feature_learner_shapes(cfg: fo.OracleConfig, learner: str):
    """``feature_learner.state_dict()`` names and shapes (sf.py:84-88 feature_net; :198 inverse_dynamic_net for icm)."""
    g, d, Hb, a = cfg.goal_dim, cfg.z_dim, cfg.backward_hidden_dim, cfg.action_dim
    if learner == "identity":     # feature_net = nn.Identity() (sf.py:94-98): no parameters
        return []
    out = [("feature_net.0.weight", (Hb, g)), ("feature_net.0.bias", (Hb,)), ("feature_net.1.weight", (Hb,)),
           ("feature_net.1.bias", (Hb,)), ("feature_net.3.weight", (Hb, Hb)), ("feature_net.3.bias", (Hb,)),
           ("feature_net.5.weight", (d, Hb)), ("feature_net.5.bias", (d,))]
    if learner in HEADS:          # e.g. icm: mlp(2 z_dim, Hb, 'irelu', Hb, 'irelu', action_dim, 'tanh')
        name, fin, fout = HEADS[learner](cfg)
        out += [(f"{name}.0.weight", (Hb, fin)), (f"{name}.0.bias", (Hb,)), (f"{name}.2.weight", (Hb, Hb)), (f"{name}.2.bias", (Hb,)),
                (f"{name}.4.weight", (fout, Hb)), (f"{name}.4.bias", (fout,))]
    if learner == "latent":       # + target_feature_net: feature_net's architecture, own weights, no gradients (sf.py:234)
        out += [("target_" + n, shp) for n, shp in out[:8]]
    elif learner in ("contrastive", "contrastivev2"):     # mu_net = feature_net's architecture (sf.py:121, 162)
        out += [("mu_net" + n[len("feature_net"):], shp) for n, shp in out[:8]]
    elif learner in ("svd_sr", "svd_srv2"):     # mu_net on the goal alone, then target copies of both nets (sf.py:265-269)
        mu = [("mu_net.0.weight", (Hb, g)), ("mu_net.0.bias", (Hb,)), ("mu_net.1.weight", (Hb,)), ("mu_net.1.bias", (Hb,)),
              ("mu_net.3.weight", (Hb, Hb)), ("mu_net.3.bias", (Hb,)), ("mu_net.5.weight", (d, Hb)), ("mu_net.5.bias", (d,))]
        out += mu + [("target_" + n, shp) for n, shp in out[:8] + mu]
    elif learner == "svd_p":      # mu_net = mlp(goal_dim + action_dim, Hb, "ntanh", Hb, "relu", z_dim)   (sf.py:340)
        out += [("mu_net.0.weight", (Hb, g + a)), ("mu_net.0.bias", (Hb,)), ("mu_net.1.weight", (Hb,)), ("mu_net.1.bias", (Hb,)),
                ("mu_net.3.weight", (Hb, Hb)), ("mu_net.3.bias", (Hb,)), ("mu_net.5.weight", (d, Hb)), ("mu_net.5.bias", (d,))]
    elif learner not in ("lap", "random") and learner not in HEADS:
        raise NotImplementedError(learner)
    return out


# the head mlp(in, Hb, 'irelu', Hb, 'irelu', out) a feature learner trains feature_net through: (module name, in, out)
HEADS = {"icm": lambda c: ("inverse_dynamic_net", 2 * c.z_dim, c.action_dim),              # sf.py:198 (+ 'tanh')
         "latent": lambda c: ("forward_dynamic_net", c.z_dim + c.action_dim, c.z_dim),     # sf.py:233
         "autoencoder": lambda c: ("decoder", c.z_dim, c.goal_dim),                        # sf.py:253
         "transition": lambda c: ("forward_dynamic_net", c.z_dim + c.action_dim, c.goal_dim)}   # sf.py:219


def net_shapes(cfg: fo.OracleConfig, learner: str):
    return {"actor": fo.actor_shapes(cfg), "successor_net": fo.forward_map_shapes(cfg),
            "feature_learner": feature_learner_shapes(cfg, learner)}


def feature_net(p: Params, goal: torch.Tensor, z_dim: int) -> torch.Tensor:
    """feature_learner.feature_net (sf.py:87): Linear, LayerNorm, Tanh, Linear, ReLU, Linear, _L2  (no parameters: nn.Identity, sf.py:97)"""
    if not p:
        return goal
    h = F.linear(goal, p["feature_net.0.weight"], p["feature_net.0.bias"])
    h = torch.tanh(F.layer_norm(h, (h.shape[-1],), p["feature_net.1.weight"], p["feature_net.1.bias"], fo.LN_EPS))
    h = torch.relu(F.linear(h, p["feature_net.3.weight"], p["feature_net.3.bias"]))
    y = F.linear(h, p["feature_net.5.weight"], p["feature_net.5.bias"])
    return math.sqrt(z_dim) * F.normalize(y, dim=1)


def inverse_dynamics(p: Params, phi: torch.Tensor, next_phi: torch.Tensor) -> torch.Tensor:
    x = torch.cat([phi, next_phi], dim=-1)
    h = torch.relu(F.linear(x, p["inverse_dynamic_net.0.weight"], p["inverse_dynamic_net.0.bias"]))
    h = torch.relu(F.linear(h, p["inverse_dynamic_net.2.weight"], p["inverse_dynamic_net.2.bias"]))
    return torch.tanh(F.linear(h, p["inverse_dynamic_net.4.weight"], p["inverse_dynamic_net.4.bias"]))


def head_mlp(p: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    h = torch.relu(F.linear(x, p[f"{name}.0.weight"], p[f"{name}.0.bias"]))
    h = torch.relu(F.linear(h, p[f"{name}.2.weight"], p[f"{name}.2.bias"]))
    return F.linear(h, p[f"{name}.4.weight"], p[f"{name}.4.bias"])


def phi_loss_terms(p: Params, learner: str, goal, action, next_goal, z_dim: int, future_goal=None) -> tp.Dict[str, tp.Any]:
    phi, next_phi = feature_net(p, goal, z_dim), feature_net(p, next_goal, z_dim)
    if learner in ("contrastive", "contrastivev2"):                # sf.py:125-143 / :166-186
        assert future_goal is not None
        mup = {"feature_net" + k[len("mu_net"):]: v for k, v in p.items() if k.startswith("mu_net.")}
        if learner == "contrastive":
            mu = feature_net(mup, future_goal, z_dim)
            logits = torch.einsum('sd, td-> st', F.normalize(phi, dim=1), F.normalize(mu, dim=1))
        else:                                                      # v2: mu on the goal, the features of the hindsight goal
            mu = feature_net(mup, goal, z_dim)
            future_phi = feature_net(p, future_goal, z_dim)
            logits = torch.einsum('sd, td-> st', F.normalize(mu, dim=1), F.normalize(future_phi, dim=1))
        off = ~torch.eye(*logits.size()).bool()
        lod = logits[off].reshape(logits.shape[0], logits.shape[0] - 1)
        return {"phi_loss": (-logits.diag() + torch.logsumexp(lod, dim=1)).mean(), "phi": phi, "next_phi": next_phi, "mu": mu}
    if learner == "icm":                                           # sf.py:203-213
        pred = inverse_dynamics(p, phi, next_phi)
        return {"phi_loss": (action - pred).pow(2).mean(), "phi": phi, "next_phi": next_phi, "pred": pred}
    if learner in ("random", "identity"):                          # FeatureLearner.forward returns None, sf.py:91-92
        return {"phi_loss": None, "phi": phi, "next_phi": next_phi}
    if learner == "autoencoder":                                   # sf.py:256-262
        pred = head_mlp(p, "decoder", phi)
        return {"phi_loss": (pred - goal).pow(2).mean(), "phi": phi, "next_phi": next_phi, "pred": pred}
    if learner == "svd_p":                                         # sf.py:344-362 (the features are those of NEXT_goal)
        x = torch.cat([goal, action], dim=1)
        h = F.linear(x, p["mu_net.0.weight"], p["mu_net.0.bias"])
        h = torch.tanh(F.layer_norm(h, (h.shape[-1],), p["mu_net.1.weight"], p["mu_net.1.bias"], fo.LN_EPS))
        h = torch.relu(F.linear(h, p["mu_net.3.weight"], p["mu_net.3.bias"]))
        mu = F.linear(h, p["mu_net.5.weight"], p["mu_net.5.bias"])
        P = torch.einsum("sd, td -> st", mu, next_phi)
        off = ~torch.eye(*P.size()).bool()
        loss = -2 * P.diag().mean() + P[off].pow(2).mean()
        Cov = torch.matmul(next_phi, next_phi.T)
        orth = Cov[off].pow(2).mean() - 2 * Cov.diag().mean()
        return {"phi_loss": loss + orth, "phi": phi, "next_phi": next_phi, "mu": mu, "orth_loss": orth}
    if learner in ("svd_sr", "svd_srv2"):                          # sf.py:271-292 / :311-332 (the caller moves both target nets afterwards)
        mlp3 = lambda q, x: F.linear(torch.relu(F.linear(torch.tanh(F.layer_norm(F.linear(x, p[q + "0.weight"], p[q + "0.bias"]),
                                                                                  (p[q + "0.bias"].shape[0],), p[q + "1.weight"], p[q + "1.bias"], fo.LN_EPS)),
                                                         p[q + "3.weight"], p[q + "3.bias"])), p[q + "5.weight"], p[q + "5.bias"])
        v2 = learner == "svd_srv2"                                 # v2: mu on the goal, the features of next_goal, 0.98
        mu = mlp3("mu_net.", goal if v2 else next_goal)
        feat = next_phi if v2 else phi
        SR = torch.einsum("sd, td -> st", mu, feat) if v2 else torch.einsum("sd, td -> st", feat, mu)
        with torch.no_grad():
            tphi = feature_net({k[len("target_"):]: v for k, v in p.items() if k.startswith("target_feature_net.")}, next_goal, z_dim)
            tmu = mlp3("target_mu_net.", next_goal)
            tSR = torch.einsum("sd, td -> st", tmu, tphi) if v2 else torch.einsum("sd, td -> st", tphi, tmu)
        off = ~torch.eye(*SR.size()).bool()
        loss = -2 * SR.diag().mean() + (SR - (0.98 if v2 else 0.99) * tSR)[off].pow(2).mean()
        Cov = torch.matmul(feat, feat.T)
        orth = Cov[off].pow(2).mean() - 2 * Cov.diag().mean()
        return {"phi_loss": loss + orth, "phi": phi, "next_phi": next_phi, "mu": mu, "orth_loss": orth}
    if learner == "latent":                                        # sf.py:238-246 (the caller moves the target net afterwards)
        with torch.no_grad():
            tgt = feature_net({k[len("target_"):]: v for k, v in p.items() if k.startswith("target_feature_net.")}, next_goal, z_dim)
        pred = head_mlp(p, "forward_dynamic_net", torch.cat([phi, action], dim=-1))
        return {"phi_loss": (pred - tgt).pow(2).mean(), "phi": phi, "next_phi": next_phi, "pred": pred, "target_phi": tgt}
    if learner == "transition":                                    # sf.py:222-227
        pred = head_mlp(p, "forward_dynamic_net", torch.cat([phi, action], dim=-1))
        return {"phi_loss": (pred - next_goal).pow(2).mean(), "phi": phi, "next_phi": next_phi, "pred": pred}
    loss = (phi - next_phi).pow(2).mean()                          # lap, sf.py:101-116
    Cov = torch.matmul(phi, phi.T)
    off = ~torch.eye(*Cov.size()).bool()
    orth = Cov[off].pow(2).mean() - 2 * Cov.diag().mean()
    return {"phi_loss": loss + orth, "phi": phi, "next_phi": next_phi, "orth_loss": orth}


def _grad_or_zero(x: torch.Tensor, retained: bool) -> torch.Tensor:
    g = x.grad if retained else None
    return g.detach().clone() if g is not None else torch.zeros_like(x).detach()


class SFOracleAgent:
    """State of one SFAgent (sf.py:383-474) and its ``update`` (:700-768) for ``feature_learner`` in {"icm", "lap", "random",
    "autoencoder", "transition", "svd_p", "latent", "svd_sr", "svd_srv2", "contrastive", "contrastivev2", "identity"}."""

    NETS = ("actor", "successor_net", "feature_learner")

    def __init__(self, cfg: fo.OracleConfig, nets: tp.Dict[str, Params], learner: str = "icm", sf_q_loss: bool = True) -> None:
        self.cfg, self.learner, self.sf_q_loss = cfg, learner, sf_q_loss
        for n in self.NETS:
            setattr(self, n, {k: v.clone() for k, v in nets[n].items()})
        self.successor_target_net = {k: v.clone() for k, v in self.successor_net.items()}          # sf.py:451
        z = lambda d: {k: torch.zeros_like(v) for k, v in d.items()}
        self.adam = {n: {"m": z(getattr(self, n)), "v": z(getattr(self, n))} for n in self.NETS}
        self.sf_steps = self.actor_steps = 0
        self.last: tp.Dict[str, tp.Any] = {}

    _req = staticmethod(fo.OracleAgent._req)

    def update(self, batch: tp.Dict[str, np.ndarray], draws: fo.Draws, keep: bool = False) -> tp.Dict[str, float]:
        cfg = self.cfg
        t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32)
        obs, action, next_obs = t(batch["obs"]), t(batch["action"]), t(batch["next_obs"])
        discount = t(batch["discount"]).reshape(-1, 1)
        goal, next_goal = obs, next_obs
        if cfg.use_goal:                                           # sf.py:716-721
            goal, next_goal = t(batch["goal"]), t(batch["next_goal"])
        z = fo.sample_z_from_gauss(t(draws.z_gauss), cfg.z_dim)    # sf.py:570-573, 723
        if cfg.mix_ratio > 0:                                      # sf.py:725-739: whitened features of permuted next goals as tasks
            with torch.no_grad():
                phi_p = feature_net(self.feature_learner, next_goal[torch.as_tensor(draws.perm, dtype=torch.long)], cfg.z_dim)
                cov = torch.matmul(phi_p.T, phi_p) / phi_p.shape[0]
                inv_cov = torch.linalg.pinv(cov)
                mix_idxs = np.where(draws.mix_uniform < cfg.mix_ratio)[0]
                new_z = math.sqrt(cfg.z_dim) * F.normalize(torch.matmul(phi_p[mix_idxs], inv_cov), dim=1)
                z = z.clone()
                z[mix_idxs] = new_z
        metrics: tp.Dict[str, float] = {}

        # ---------------- update_sf (sf.py:594-664) ---------------- #
        with torch.no_grad():
            mu_n = fo.actor_mu(self.actor, next_obs, z)
            next_action = fo.truncated_normal_sample(mu_n, cfg.stddev, cfg.stddev_clip, t(draws.eps_next))
            nF1, nF2 = fo.forward_map(self.successor_target_net, next_obs, z, next_action)
            target_phi = feature_net(self.feature_learner, next_goal, cfg.z_dim)
            nQ1, nQ2 = [torch.einsum('sd, sd -> s', Fi, z) for Fi in (nF1, nF2)]
            next_F = torch.where((nQ1 < nQ2).reshape(-1, 1), nF1, nF2)
            target_F = target_phi + discount * next_F
        sp = self._req(self.successor_net)
        F1, F2 = fo.forward_map(sp, obs, z, action)
        if not self.sf_q_loss:
            sf_loss = F.mse_loss(F1, target_F) + F.mse_loss(F2, target_F)
        else:
            Q1, Q2 = [torch.einsum('sd, sd -> s', Fi, z) for Fi in (F1, F2)]
            target_Q = torch.einsum('sd, sd -> s', target_F, z)
            sf_loss = F.mse_loss(Q1, target_Q) + F.mse_loss(Q2, target_Q)
        pp = self._req(self.feature_learner)
        future_goal = None
        if self.learner in ("contrastive", "contrastivev2"):        # sf.py:713, 719
            future_goal = t(batch["future_goal"] if cfg.use_goal else batch["future_obs"])
        L = phi_loss_terms(pp, self.learner, goal, action, next_goal, cfg.z_dim, future_goal)
        if keep:
            for x in (F1, F2) + ((L["phi"], L["next_phi"]) if L["phi_loss"] is not None else ()):
                x.retain_grad()
        metrics.update({                                            # sf.py:627-636
            "target_F": target_F.mean().item(), "F1": F1.mean().item(), "phi": target_phi.mean().item(),
            "phi_norm": torch.norm(target_phi, dim=-1).mean().item(), "z_norm": torch.norm(z, dim=-1).mean().item(),
            "sf_loss": sf_loss.item(), "sf_opt_lr": cfg.lr})
        if L["phi_loss"] is not None:                               # sf.py:634-635
            metrics["phi_loss"] = L["phi_loss"].item()
        sf_loss.backward()
        gS = {k: v.grad for k, v in sp.items()}
        self.sf_steps += 1
        fo.adam_step(self.successor_net, gS, self.adam["successor_net"]["m"], self.adam["successor_net"]["v"], self.sf_steps, cfg.lr)
        gP: tp.Dict[str, tp.Any] = {}
        if L["phi_loss"] is not None:                               # sf.py:447-449, 657-660: "random" has no phi_opt
            if self.learner in ("latent", "svd_sr", "svd_srv2"):                # sf.py:245 / :294-295: inside the learner's forward(), i.e. BEFORE phi_opt.step()
                with torch.no_grad():
                    for k in [k for k in self.feature_learner if k.startswith(("feature_net.", "mu_net.") if self.learner != "latent" else "feature_net.")]:
                        self.feature_learner["target_" + k].mul_(0.99).add_(self.feature_learner[k], alpha=0.01)
            L["phi_loss"].backward()
            gP = {k: v.grad for k, v in pp.items() if v.grad is not None}    # (target_feature_net never has gradients: Adam skips it)
            sub = lambda dct: {k: dct[k] for k in gP}
            fo.adam_step(sub(self.feature_learner), gP, sub(self.adam["feature_learner"]["m"]), sub(self.adam["feature_learner"]["v"]),
                         self.sf_steps, cfg.lr_coef * cfg.lr)

        # ---------------- update_actor (sf.py:666-694) ------------- #
        ap = self._req(self.actor)
        snew = self._req(self.successor_net)
        mu = fo.actor_mu(ap, obs, z)
        act = fo.truncated_normal_sample(mu, cfg.stddev, cfg.stddev_clip, t(draws.eps_actor))
        log_prob = fo.normal_log_prob(mu, cfg.stddev, act).sum(-1, keepdim=True)
        aF1, aF2 = fo.forward_map(snew, obs, z, act)
        Q = torch.min(torch.einsum('sd, sd -> s', aF1, z), torch.einsum('sd, sd -> s', aF2, z))
        actor_loss = -Q.mean()
        if keep:
            mu.retain_grad()
        actor_loss.backward()
        gA = {k: v.grad for k, v in ap.items()}
        self.actor_steps += 1
        fo.adam_step(self.actor, gA, self.adam["actor"]["m"], self.adam["actor"]["v"], self.actor_steps, cfg.lr)
        metrics.update(actor_loss=actor_loss.item(), actor_logprob=log_prob.mean().item())

        fo.soft_update(self.successor_net, self.successor_target_net, cfg.fb_target_tau)            # sf.py:751-752 (sf_target_tau)
        if keep:
            d = lambda x: x.detach().clone()
            self.last = dict(z=d(z), next_action=d(next_action), nF1=d(nF1), nF2=d(nF2), target_phi=d(target_phi),
                             target_F=d(target_F), F1=d(F1), F2=d(F2), dF1=d(F1.grad), dF2=d(F2.grad), phi=d(L["phi"]),
                             next_phi=d(L["next_phi"]),
                             # (only icm / lap read next_phi; "random" has no feature loss at all)
                             dphi=_grad_or_zero(L["phi"], L["phi_loss"] is not None),
                             dnext_phi=_grad_or_zero(L["next_phi"], L["phi_loss"] is not None),
                             grads_successor={k: d(v) for k, v in gS.items()}, grads_feature={k: d(v) for k, v in gP.items()},
                             grads_actor={k: d(v) for k, v in gA.items()}, mu=d(mu), pi_action=d(act),
                             d_premu=d(mu.grad) * (1 - d(mu) ** 2))
        return metrics

    def state_tensors(self) -> tp.Dict[str, np.ndarray]:
        out = {}
        for n in self.NETS + ("successor_target_net",):
            for k, v in getattr(self, n).items():
                out[f"{n}/{k}"] = v.detach().numpy().copy()
        for n in self.NETS:
            for mv in ("m", "v"):
                for k, v in self.adam[n][mv].items():
                    out[f"adam_{mv}/{n}/{k}"] = v.numpy().copy()
        return out
