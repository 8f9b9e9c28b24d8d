"""CPU restatement of the FB-DDPG update step -- TEST INFRASTRUCTURE ONLY.

Every function cites the reference lines it restates (paths relative to
/root/reference).  Arithmetic is torch-CPU fp32 (the reference's own arithmetic
substrate), random draws are *injected* (``Draws``) because the reference mixes
three RNG streams (numpy global, torch CPU, torch device) that cannot be
reproduced bit-for-bit on a GPU (SURVEY.md section 7 "hard parts").

Parameters are kept in plain dicts keyed by the reference's ``state_dict``
names (``obs_action_net.0.weight`` ...), so reference weights load unchanged.

PARITY STATUS: PINNED.  tests/test_oracle_golden.py replays this file against outputs of the REAL reference
(``url_benchmark.agent.fb_ddpg.FBDDPGAgent`` + ``in_memory_replay_buffer.ReplayBuffer``, imported from /root/reference by
tests/golden/make_golden.py in the development container; only the vectors are committed): after every step of 13
tiny-dimension traces every parameter, target, Adam moment and all 18 metrics (default config, goal space + q_loss +
variable episode lengths, hindsight replay, norm_z=False, rand_weight, add_trunk, preprocess=False, boltzmann), the
metric curves and parameter checksums of three full-dimension runs, the sampler's index -> row mapping with the real
numpy RNG, the constructor's weight init under a torch seed, and the inference entry points.

WHO MAY IMPORT THIS: tests/, ``__graft_entry__.smoke()`` (as the checker) and ``bench.py``'s ``cpu_baseline`` leg (as the
thing timed on the host cores).  The product (controllable_agent_amd/) never does: it has no CPU fallback.
"""
from __future__ import annotations

import dataclasses
import math
import typing as tp

import numpy as np
import torch
import torch.nn.functional as F

Params = tp.Dict[str, torch.Tensor]

ADAM_BETA1 = 0.9
ADAM_BETA2 = 0.999
ADAM_EPS = 1e-8
LN_EPS = 1e-5


# --------------------------------------------------------------------------- #
# configuration (defaults = url_benchmark/agent/fb_ddpg.py:37-82)
# --------------------------------------------------------------------------- #
@dataclasses.dataclass
class OracleConfig:
    obs_dim: int
    action_dim: int
    goal_dim: int                 # == obs_dim when goal_space is None (fb_ddpg.py:111-114)
    z_dim: int = 50
    hidden_dim: int = 1024
    feature_dim: int = 512
    backward_hidden_dim: int = 526
    batch_size: int = 1024
    lr: float = 1e-4
    lr_coef: float = 1.0
    fb_target_tau: float = 0.01
    stddev: float = 0.2           # utils.schedule("0.2", step) (utils.py:235-237)
    stddev_clip: float = 0.3
    ortho_coef: float = 1.0
    mix_ratio: float = 0.5
    q_loss: bool = False
    q_loss_coef: float = 0.01
    use_goal: bool = False        # goal_space is not None
    discount: float = 0.99        # ReplayBuffer._discount (in_memory_replay_buffer.py:171)
    future_ratio: float = 0.0     # hindsight replay: z[u < future_ratio] = B(future_goal)  (fb_ddpg.py:487-491)
    future: float = 1.0           # ReplayBuffer._future; < 1 => future_idx = step + Geometric(1 - future) (:157-161)
    norm_z: bool = True           # False: B unprojected, z = sqrt(d) U g/|g| (fb_ddpg.py:227-231, fb_modules.py:228-229)
    rand_weight: bool = False     # mixed rows = (u * normalize(rand[B])) @ B(backward_input)  (fb_ddpg.py:475-482)
    add_trunk: bool = False       # extra Linear(2Fd, H)+ReLU "trunk" after the two preprocess nets (fb_modules.py:93-98,168-173)
    preprocess: bool = True       # False: ONE trunk mlp(in, H, "ntanh", H, "irelu", H, "irelu") on the concatenated input
                                  # instead of the two preprocess nets (fb_modules.py:99-103, 174-178); add_trunk is then moot
    boltzmann: bool = False       # DiagGaussianActor + SquashedNormal, actor_loss = (temp log_prob - Q).mean()
                                  # (fb_ddpg.py:118-120, 304-306, 391-393, 406; fb_modules.py:129-151)
    temp: float = 1.0             # fb_ddpg.py:71
    log_std_min: float = -5.0     # fb_ddpg.py:70  log_std_bounds
    log_std_max: float = 2.0
    debug: bool = False           # backward_net = backward_target_net = IdentityMap(): B(goal) = goal, no parameters, no projection
                                  # (fb_ddpg.py:128-130, fb_modules.py:202-208; needs goal_dim == z_dim)


@dataclasses.dataclass
class Draws:
    """All random draws one ``update()`` consumes, in the order the reference draws them."""
    ep_idx: np.ndarray        # int64 [B]   in_memory_replay_buffer.py:147-151
    step_idx: np.ndarray      # int64 [B]   in_memory_replay_buffer.py:155  (already +1)
    z_gauss: np.ndarray       # f32 [B,d]   fb_ddpg.py:225   (torch.randn)
    perm: np.ndarray          # int64 [B]   fb_ddpg.py:467    (torch.randperm)
    mix_uniform: np.ndarray   # f64 [B]     fb_ddpg.py:471    (np.random.uniform)
    eps_next: np.ndarray      # f32 [B,a]   utils.py:178 inside update_fb (fb_ddpg.py:310)
    eps_actor: np.ndarray     # f32 [B,a]   utils.py:178 inside update_actor (fb_ddpg.py:397)
    future_idx: tp.Optional[np.ndarray] = None      # int64 [B]  in_memory_replay_buffer.py:157-161 (only if future < 1)
    future_uniform: tp.Optional[np.ndarray] = None  # f64 [B]    fb_ddpg.py:490 (only if future_ratio > 0)
    z_uniform: tp.Optional[np.ndarray] = None       # f32 [B,d]  fb_ddpg.py:230 (torch.rand, only if not norm_z)
    rand_weight: tp.Optional[np.ndarray] = None     # f32 [B,B]  fb_ddpg.py:477: row i = the weights of batch row i
    rand_weight_u: tp.Optional[np.ndarray] = None   # f32 [B]    fb_ddpg.py:479  (both only if cfg.rand_weight; the reference
                                                    #            draws them for the mixed rows only, in mix_idxs order)


def make_draws(rng: np.random.Generator, cfg: OracleConfig, n_episodes: int,
               episode_lengths: np.ndarray) -> Draws:
    """Deterministic draws from a numpy Generator (PCG64 stream is stable across platforms).

    Index distribution follows in_memory_replay_buffer.py:146-155: uniform episode
    (length-proportional if lengths vary), then uniform step in [1, len]."""
    B = cfg.batch_size
    lens = np.asarray(episode_lengths[:n_episodes], dtype=np.int64)
    if (lens == lens[0]).all():
        ep = rng.integers(0, n_episodes, size=B)
    else:
        ep = rng.choice(np.arange(n_episodes), size=B, p=lens / lens.sum())
    step = rng.integers(0, lens[ep]) + 1
    fut_idx = fut_u = None
    if cfg.future < 1:                                           # in_memory_replay_buffer.py:157-161
        fut_idx = np.clip(step + rng.geometric(p=1 - cfg.future, size=B), 0, lens[ep]).astype(np.int64)
    return _with_future(cfg, rng, fut_idx, Draws(
        ep_idx=ep.astype(np.int64), step_idx=step.astype(np.int64),
        z_gauss=rng.standard_normal((B, cfg.z_dim)).astype(np.float32),
        perm=rng.permutation(B).astype(np.int64),
        mix_uniform=rng.uniform(size=B),
        eps_next=rng.standard_normal((B, cfg.action_dim)).astype(np.float32),
        eps_actor=rng.standard_normal((B, cfg.action_dim)).astype(np.float32)))


def _with_future(cfg: OracleConfig, rng: np.random.Generator, fut_idx, d: Draws) -> Draws:
    d.future_idx = fut_idx
    if cfg.future_ratio > 0:
        d.future_uniform = rng.uniform(size=cfg.batch_size)
    if not cfg.norm_z:
        d.z_uniform = rng.uniform(size=(cfg.batch_size, cfg.z_dim)).astype(np.float32)
    if cfg.rand_weight:
        d.rand_weight = rng.uniform(size=(cfg.batch_size, cfg.batch_size)).astype(np.float32)
        d.rand_weight_u = rng.uniform(size=cfg.batch_size).astype(np.float32)
    return d


# --------------------------------------------------------------------------- #
# parameter construction
# --------------------------------------------------------------------------- #
def _trunk_shapes(prefix: str, in_dim: int, hidden: int, feat: int) -> tp.List[tp.Tuple[str, tp.Tuple[int, ...]]]:
    # mlp(in, hidden, "ntanh", feat, "irelu")  (fb_modules.py:60-78, 91-92, 166-167)
    return [(f"{prefix}.0.weight", (hidden, in_dim)), (f"{prefix}.0.bias", (hidden,)),
            (f"{prefix}.1.weight", (hidden,)), (f"{prefix}.1.bias", (hidden,)),
            (f"{prefix}.3.weight", (feat, hidden)), (f"{prefix}.3.bias", (feat,))]


def _single_trunk_shapes(in_dim: int, H: int):
    # mlp(in, H, "ntanh", H, "irelu", H, "irelu")  (fb_modules.py:100-102, 175-177): Linear, LayerNorm, Tanh, Linear, ReLU, Linear, ReLU
    return [("trunk.0.weight", (H, in_dim)), ("trunk.0.bias", (H,)), ("trunk.1.weight", (H,)), ("trunk.1.bias", (H,)),
            ("trunk.3.weight", (H, H)), ("trunk.3.bias", (H,)), ("trunk.5.weight", (H, H)), ("trunk.5.bias", (H,))]


def forward_map_shapes(cfg: OracleConfig):
    """ForwardMap parameter list in ``parameters()`` order (fb_modules.py:165-182)."""
    o, a, d, H, Fd = cfg.obs_dim, cfg.action_dim, cfg.z_dim, cfg.hidden_dim, cfg.feature_dim
    if not cfg.preprocess:
        out, feat = _single_trunk_shapes(o + d + a, H), H
    else:
        out = _trunk_shapes("obs_action_net", o + a, H, Fd) + _trunk_shapes("obs_z_net", o + d, H, Fd)
        feat = 2 * Fd
        if cfg.add_trunk:                                      # mlp(2 * feature_dim, hidden_dim, "irelu")
            out += [("trunk.0.weight", (H, 2 * Fd)), ("trunk.0.bias", (H,))]
            feat = H
    for head in ("F1", "F2"):
        out += [(f"{head}.0.weight", (H, feat)), (f"{head}.0.bias", (H,)),
                (f"{head}.2.weight", (d, H)), (f"{head}.2.bias", (d,))]
    return out


def actor_shapes(cfg: OracleConfig):
    """Actor parameter list (fb_modules.py:91-105)."""
    o, a, d, H, Fd = cfg.obs_dim, cfg.action_dim, cfg.z_dim, cfg.hidden_dim, cfg.feature_dim
    if cfg.boltzmann:             # DiagGaussianActor.policy = mlp(o + d, H, "ntanh", H, "relu", 2a)  (fb_modules.py:138)
        return [("policy.0.weight", (H, o + d)), ("policy.0.bias", (H,)), ("policy.1.weight", (H,)), ("policy.1.bias", (H,)),
                ("policy.3.weight", (H, H)), ("policy.3.bias", (H,)), ("policy.5.weight", (2 * a, H)), ("policy.5.bias", (2 * a,))]
    if not cfg.preprocess:
        out, feat = _single_trunk_shapes(o + d, H), H
    else:
        out = _trunk_shapes("obs_net", o, H, Fd) + _trunk_shapes("obs_z_net", o + d, H, Fd)
        feat = 2 * Fd
        if cfg.add_trunk:
            out += [("trunk.0.weight", (H, 2 * Fd)), ("trunk.0.bias", (H,))]
            feat = H
    return out + [("policy.0.weight", (H, feat)), ("policy.0.bias", (H,)),
                  ("policy.2.weight", (a, H)), ("policy.2.bias", (a,))]


def backward_map_shapes(cfg: OracleConfig):
    """BackwardMap parameter list (fb_modules.py:220): mlp(g, Hb, "ntanh", Hb, "relu", d)."""
    g, d, Hb = cfg.goal_dim, cfg.z_dim, cfg.backward_hidden_dim
    if cfg.debug:                 # IdentityMap holds nn.Identity(): an empty state_dict
        return []
    return [("B.0.weight", (Hb, g)), ("B.0.bias", (Hb,)),
            ("B.1.weight", (Hb,)), ("B.1.bias", (Hb,)),
            ("B.3.weight", (Hb, Hb)), ("B.3.bias", (Hb,)),
            ("B.5.weight", (d, Hb)), ("B.5.bias", (d,))]


NET_SHAPES = {"actor": actor_shapes, "forward_net": forward_map_shapes, "backward_net": backward_map_shapes}


def synthetic_params(rng: np.random.Generator, shapes, ln_jitter: float = 0.1) -> Params:
    """Platform-independent synthetic weights: Linear W ~ N(0, 1/fan_in), small random bias,
    LayerNorm gamma ~ 1 + jitter, beta ~ jitter.  (NOT the reference's orthogonal init --
    used for fixtures so both sides can regenerate identical weights from a seed.)"""
    out: Params = {}
    for name, shape in shapes:
        if len(shape) == 2:
            w = rng.standard_normal(shape) / math.sqrt(shape[1])
        elif ".1." in name and name.endswith("weight"):       # LayerNorm gamma
            w = 1.0 + ln_jitter * rng.standard_normal(shape)
        else:
            w = ln_jitter * rng.standard_normal(shape)
        out[name] = torch.from_numpy(w.astype(np.float32))
    return out


# --------------------------------------------------------------------------- #
# modules
# --------------------------------------------------------------------------- #
def _trunk(p: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """Linear -> LayerNorm(eps=1e-5, affine) -> Tanh -> Linear -> ReLU   (fb_modules.py:43-78)."""
    h = F.linear(x, p[f"{prefix}.0.weight"], p[f"{prefix}.0.bias"])
    h = F.layer_norm(h, (h.shape[-1],), p[f"{prefix}.1.weight"], p[f"{prefix}.1.bias"], LN_EPS)
    h = torch.tanh(h)
    return torch.relu(F.linear(h, p[f"{prefix}.3.weight"], p[f"{prefix}.3.bias"]))


def _single_trunk(p: Params, x: torch.Tensor) -> torch.Tensor:
    h = F.linear(x, p["trunk.0.weight"], p["trunk.0.bias"])
    h = torch.tanh(F.layer_norm(h, (h.shape[-1],), p["trunk.1.weight"], p["trunk.1.bias"], LN_EPS))
    h = torch.relu(F.linear(h, p["trunk.3.weight"], p["trunk.3.bias"]))
    return torch.relu(F.linear(h, p["trunk.5.weight"], p["trunk.5.bias"]))


def forward_map(p: Params, obs, z, action) -> tp.Tuple[torch.Tensor, torch.Tensor]:
    """ForwardMap.forward (fb_modules.py:186-199)."""
    if "trunk.5.weight" in p:                              # preprocess=False: cat([obs, z, action]) -> trunk
        h = _single_trunk(p, torch.cat([obs, z, action], dim=-1))
    else:
        obs_action = _trunk(p, "obs_action_net", torch.cat([obs, action], dim=-1))
        obs_z = _trunk(p, "obs_z_net", torch.cat([obs, z], dim=-1))
        h = torch.cat([obs_action, obs_z], dim=-1)
        if "trunk.0.weight" in p:                          # add_trunk (fb_modules.py:194-195)
            h = torch.relu(F.linear(h, p["trunk.0.weight"], p["trunk.0.bias"]))
    outs = []
    for head in ("F1", "F2"):
        t = torch.relu(F.linear(h, p[f"{head}.0.weight"], p[f"{head}.0.bias"]))
        outs.append(F.linear(t, p[f"{head}.2.weight"], p[f"{head}.2.bias"]))
    return outs[0], outs[1]


def backward_map_raw(p: Params, goal) -> torch.Tensor:
    """BackwardMap's inner mlp, before the L2 projection (fb_modules.py:220,227)."""
    h = F.linear(goal, p["B.0.weight"], p["B.0.bias"])
    h = torch.tanh(F.layer_norm(h, (h.shape[-1],), p["B.1.weight"], p["B.1.bias"], LN_EPS))
    h = torch.relu(F.linear(h, p["B.3.weight"], p["B.3.bias"]))
    return F.linear(h, p["B.5.weight"], p["B.5.bias"])


def backward_map(p: Params, goal, z_dim: int, norm_z: bool = True) -> torch.Tensor:
    """BackwardMap.forward (fb_modules.py:223-230); no parameters: IdentityMap.forward (fb_modules.py:207-208)."""
    if not p:
        return goal
    y = backward_map_raw(p, goal)
    return math.sqrt(z_dim) * F.normalize(y, dim=1) if norm_z else y


def diag_gaussian(p: Params, obs, z, log_std_min: float = -5.0, log_std_max: float = 2.0):
    """DiagGaussianActor.forward (fb_modules.py:141-151): (raw policy output [B, 2a], loc, scale) of the SquashedNormal."""
    pol = _trunk(p, "policy", torch.cat([obs, z], dim=-1))           # Linear, LayerNorm, Tanh, Linear, ReLU
    pol = F.linear(pol, p["policy.5.weight"], p["policy.5.bias"])
    mu, log_std = pol.chunk(2, dim=-1)
    log_std = torch.tanh(log_std)
    log_std = log_std_min + 0.5 * (log_std_max - log_std_min) * (log_std + 1)
    return pol, mu, log_std.exp()


def squashed_log_prob(mu, std, u) -> torch.Tensor:
    """SquashedNormal(mu, std).log_prob(tanh(u)) with the TanhTransform's cached pre-image u (utils.py:188-232):
    Normal.log_prob(u) - 2 (log 2 - u - softplus(-2u)), per element."""
    base = -((u - mu) ** 2) / (2 * std ** 2) - std.log() - math.log(math.sqrt(2 * math.pi))
    return base - 2.0 * (math.log(2.0) - u - F.softplus(-2.0 * u))


def actor_mu(p: Params, obs, z) -> torch.Tensor:
    """Actor.forward up to mu = tanh(policy(h)) (fb_modules.py:107-122); DiagGaussianActor: ``dist.mean`` = tanh(loc)."""
    if "policy.5.weight" in p:
        return torch.tanh(diag_gaussian(p, obs, z)[1])
    if "trunk.5.weight" in p:                              # preprocess=False: cat([obs, z]) -> trunk
        h = _single_trunk(p, torch.cat([obs, z], dim=-1))
    else:
        obs_z = _trunk(p, "obs_z_net", torch.cat([obs, z], dim=-1))
        ob = _trunk(p, "obs_net", obs)
        h = torch.cat([ob, obs_z], dim=-1)
        if "trunk.0.weight" in p:                          # add_trunk (fb_modules.py:116-117)
            h = torch.relu(F.linear(h, p["trunk.0.weight"], p["trunk.0.bias"]))
    t = torch.relu(F.linear(h, p["policy.0.weight"], p["policy.0.bias"]))
    return torch.tanh(F.linear(t, p["policy.2.weight"], p["policy.2.bias"]))


def truncated_normal_sample(mu, std: float, clip: tp.Optional[float], noise) -> torch.Tensor:
    """TruncatedNormal.sample with the straight-through clamp (utils.py:171-185)."""
    eps = noise * std
    if clip is not None:
        eps = torch.clamp(eps, -clip, clip)
    x = mu + eps
    clamped = torch.clamp(x, -1.0 + 1e-6, 1.0 - 1e-6)
    return x - x.detach() + clamped.detach()


def normal_log_prob(mu, std: float, x) -> torch.Tensor:
    """Normal(mu, std).log_prob(x) (torch.distributions.Normal; fb_ddpg.py:399)."""
    var = std * std
    return -((x - mu) ** 2) / (2 * var) - math.log(std) - math.log(math.sqrt(2 * math.pi))


def sample_z_from_gauss(gauss: torch.Tensor, z_dim: int, uniform: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
    """FBDDPGAgent.sample_z (fb_ddpg.py:224-232): ``uniform`` is the torch.rand draw of the norm_z=False branch."""
    g = F.normalize(gauss, dim=1)
    return math.sqrt(z_dim) * g if uniform is None else np.sqrt(z_dim) * uniform * g


# --------------------------------------------------------------------------- #
# replay sampling (in_memory_replay_buffer.py:139-190)
# --------------------------------------------------------------------------- #
def gather_batch(storage: tp.Dict[str, np.ndarray], ep_idx, step_idx, discount: float,
                 future_idx=None) -> tp.Dict[str, np.ndarray]:
    """Index arithmetic of ReplayBuffer.sample: obs = observation[ep, step-1],
    action/next_obs/reward/discount at [ep, step]; goal pair likewise (:163-180)."""
    out = {
        "obs": storage["observation"][ep_idx, step_idx - 1],
        "action": storage["action"][ep_idx, step_idx],
        "next_obs": storage["observation"][ep_idx, step_idx],
        "discount": discount * storage["discount"][ep_idx, step_idx],   # python float * f32 array -> f32 (:171)
    }
    if "reward" in storage:
        out["reward"] = storage["reward"][ep_idx, step_idx]
    if "goal" in storage:
        out["goal"] = storage["goal"][ep_idx, step_idx - 1]
        out["next_goal"] = storage["goal"][ep_idx, step_idx]
    if future_idx is not None:                                   # :176-183
        out["future_obs"] = storage["observation"][ep_idx, future_idx - 1]
        if "goal" in storage:
            out["future_goal"] = storage["goal"][ep_idx, future_idx - 1]
    return out


# --------------------------------------------------------------------------- #
# losses
# --------------------------------------------------------------------------- #
def fb_loss_terms(F1, F2, Bm, tF1, tF2, tB, discount, ortho_coef: float) -> tp.Dict[str, torch.Tensor]:
    """FB + orthonormality loss exactly as written in fb_ddpg.py:313-348 (boolean-mask indexing kept)."""
    target_M = torch.min(torch.einsum('sd, td -> st', tF1, tB), torch.einsum('sd, td -> st', tF2, tB))
    M1 = torch.einsum('sd, td -> st', F1, Bm)
    M2 = torch.einsum('sd, td -> st', F2, Bm)
    I = torch.eye(*M1.size(), dtype=M1.dtype)
    off_diag = ~I.bool()
    fb_offdiag = 0.5 * sum((M - discount * target_M)[off_diag].pow(2).mean() for M in [M1, M2])
    fb_diag = -sum(M.diag().mean() for M in [M1, M2])
    Cov = torch.matmul(Bm, Bm.T)
    orth_diag = -2 * Cov.diag().mean()
    orth_offdiag = Cov[off_diag].pow(2).mean()
    orth = orth_offdiag + orth_diag
    return dict(fb_offdiag=fb_offdiag, fb_diag=fb_diag, orth_loss=orth, orth_loss_diag=orth_diag,
                orth_loss_offdiag=orth_offdiag, fb_loss=fb_offdiag + fb_diag + ortho_coef * orth,
                target_M=target_M, M1=M1, M2=M2)


def fb_loss_closed_form(F1, F2, Bm, tF1, tF2, tB, discount, ortho_coef: float):
    """Mask-free statement of the same loss plus its analytic gradients wrt F1, F2, Bm
     -- the formula the HIP pairwise kernel implements.  float64 inside so it
    can serve as the accuracy yardstick for both the torch-fp32 statement above and the kernel."""
    F1, F2, Bm, tF1, tF2, tB = (x.double() for x in (F1, F2, Bm, tF1, tF2, tB))
    g = discount.double().reshape(-1, 1)
    Bn = F1.shape[0]
    n_off = Bn * (Bn - 1)
    eye = torch.eye(Bn, dtype=torch.float64)
    off = 1.0 - eye
    tM = torch.minimum(tF1 @ tB.T, tF2 @ tB.T)
    out: tp.Dict[str, torch.Tensor] = {}
    dB = torch.zeros_like(Bm)
    fb_off = 0.0
    fb_diag = 0.0
    for name, Fi in (("dF1", F1), ("dF2", F2)):
        M = Fi @ Bm.T
        D = (M - g * tM) * off
        fb_off = fb_off + 0.5 * (D ** 2).sum() / n_off
        fb_diag = fb_diag - M.diagonal().mean()
        G = D / n_off - eye / Bn
        out[name] = G @ Bm
        dB = dB + G.T @ Fi
    C = Bm @ Bm.T
    orth_off = ((C * off) ** 2).sum() / n_off
    orth_diag = -2 * C.diagonal().mean()
    Hm = 2 * C * off / n_off - 2 * eye / Bn
    dB = dB + ortho_coef * 2 * Hm @ Bm
    out.update(dB=dB, fb_offdiag=fb_off, fb_diag=fb_diag, orth_loss_offdiag=orth_off, orth_loss_diag=orth_diag,
               orth_loss=orth_off + orth_diag, fb_loss=fb_off + fb_diag + ortho_coef * (orth_off + orth_diag),
               target_M_mean=tM.mean(), M1_mean=(F1 @ Bm.T).mean())
    return out


# --------------------------------------------------------------------------- #
# optimiser / target update
# --------------------------------------------------------------------------- #
def adam_step(params: Params, grads: Params, m: Params, v: Params, t: int, lr: float) -> None:
    """torch.optim.Adam single-tensor path, defaults betas=(0.9,0.999) eps=1e-8 (fb_ddpg.py:146-151;
    torch/optim/adam.py _single_tensor_adam).  ``t`` is the 1-based step count."""
    bc1 = 1 - ADAM_BETA1 ** t
    bc2 = 1 - ADAM_BETA2 ** t
    step_size = lr / bc1
    bc2_sqrt = bc2 ** 0.5
    with torch.no_grad():
        for k, p in params.items():
            g = grads[k]
            m[k].lerp_(g, 1 - ADAM_BETA1)
            v[k].mul_(ADAM_BETA2).addcmul_(g, g, value=1 - ADAM_BETA2)
            denom = (v[k].sqrt() / bc2_sqrt).add_(ADAM_EPS)
            p.addcdiv_(m[k], denom, value=-step_size)


def soft_update(net: Params, target: Params, tau: float) -> None:
    """utils.soft_update_params (utils.py:66-69)."""
    with torch.no_grad():
        for k in net:
            target[k].copy_(tau * net[k] + (1 - tau) * target[k])


# --------------------------------------------------------------------------- #
# whole agent state + one update
# --------------------------------------------------------------------------- #
class OracleAgent:
    """State of one FBDDPGAgent (fb_ddpg.py:92-159) and its ``update`` (:427-520)."""

    NETS = ("actor", "forward_net", "backward_net")

    def __init__(self, cfg: OracleConfig, nets: tp.Dict[str, Params], dtype: torch.dtype = torch.float32) -> None:
        # dtype = torch.float64: the same arithmetic in double precision -- NOT the reference's numbers, but the yardstick
        # for "how far from exact is an fp32 evaluation of this step" (tests bound the HIP path's error by the fp32 oracle's own)
        self.cfg = cfg
        self.dtype = dtype
        self.actor = {k: v.clone().to(dtype) for k, v in nets["actor"].items()}
        self.forward_net = {k: v.clone().to(dtype) for k, v in nets["forward_net"].items()}
        self.backward_net = {k: v.clone().to(dtype) for k, v in nets["backward_net"].items()}
        # fb_ddpg.py:140-141: targets start as copies
        self.forward_target_net = {k: v.clone() for k, v in self.forward_net.items()}
        self.backward_target_net = {k: v.clone() for k, v in self.backward_net.items()}
        z = lambda d: {k: torch.zeros_like(v) for k, v in d.items()}
        self.adam = {n: {"m": z(getattr(self, n)), "v": z(getattr(self, n))} for n in self.NETS}
        self.fb_steps = 0
        self.actor_steps = 0
        self.last: tp.Dict[str, tp.Any] = {}     # intermediates of the last update (for kernel tests)

    # -- helpers ---------------------------------------------------------- #
    @staticmethod
    def _req(params: Params) -> Params:
        return {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}

    def mix_z(self, z: torch.Tensor, backward_input: torch.Tensor, draws: Draws,
              future_goal: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
        """z-mixing of fb_ddpg.py:467-485 (rand_weight=False, norm_z=True) and hindsight replay (:487-491)."""
        cfg = self.cfg
        bi = backward_input[torch.from_numpy(draws.perm)]
        if cfg.mix_ratio > 0:
            mix_idxs = np.where(draws.mix_uniform < cfg.mix_ratio)[0]
            with torch.no_grad():
                if cfg.rand_weight:                                # fb_ddpg.py:475-482
                    dt = getattr(self, "dtype", torch.float32)
                    weight = F.normalize(torch.from_numpy(draws.rand_weight[mix_idxs]).to(dt), dim=1)
                    weight = torch.from_numpy(draws.rand_weight_u[mix_idxs]).to(dt).reshape(-1, 1) * weight
                    mz = torch.matmul(weight, backward_map(self.backward_net, bi, cfg.z_dim, cfg.norm_z))
                else:
                    mz = backward_map(self.backward_net, bi[mix_idxs], cfg.z_dim, cfg.norm_z)
            if cfg.norm_z:                                         # fb_ddpg.py:483-484
                mz = math.sqrt(cfg.z_dim) * F.normalize(mz, dim=1)
            z = z.clone()
            z[mix_idxs] = mz
        if cfg.future_ratio > 0:                                   # fb_ddpg.py:487-491 (after, hence over, the mix)
            assert future_goal is not None and draws.future_uniform is not None
            future_idxs = np.where(draws.future_uniform < cfg.future_ratio)[0]
            with torch.no_grad():
                fz = backward_map(self.backward_net, future_goal[future_idxs], cfg.z_dim, cfg.norm_z)
            z = z.clone()
            z[future_idxs] = fz
        return z

    # -- one update ------------------------------------------------------- #
    def update(self, batch: tp.Dict[str, np.ndarray], draws: Draws, keep: bool = False) -> tp.Dict[str, float]:
        cfg = self.cfg
        t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).to(self.dtype)
        obs, action, next_obs = t(batch["obs"]), t(batch["action"]), t(batch["next_obs"])
        discount = t(batch["discount"]).reshape(-1, 1)
        next_goal = next_obs
        backward_input = obs
        if cfg.use_goal:                                           # fb_ddpg.py:441-443, 460-465
            next_goal = t(batch["next_goal"])
            backward_input = t(batch["goal"])
        future_goal = None
        if cfg.future_ratio > 0:                                   # fb_ddpg.py:461-465
            future_goal = t(batch["future_goal"] if cfg.use_goal else batch["future_obs"])
        z = sample_z_from_gauss(t(draws.z_gauss), cfg.z_dim, None if cfg.norm_z else t(draws.z_uniform))   # fb_ddpg.py:451
        z = self.mix_z(z, backward_input, draws, future_goal)
        metrics: tp.Dict[str, float] = {}

        # ---------------- update_fb (fb_ddpg.py:291-387) ---------------- #
        with torch.no_grad():
            if cfg.boltzmann:                                      # fb_ddpg.py:304-306: dist.sample() = tanh(loc + scale eps)
                _, loc_n, std_n = diag_gaussian(self.actor, next_obs, z, cfg.log_std_min, cfg.log_std_max)
                mu_n = torch.tanh(loc_n)
                next_action = torch.tanh(loc_n + std_n * t(draws.eps_next))
            else:
                mu_n = actor_mu(self.actor, next_obs, z)
                next_action = truncated_normal_sample(mu_n, cfg.stddev, cfg.stddev_clip, t(draws.eps_next))
            tF1, tF2 = forward_map(self.forward_target_net, next_obs, z, next_action)
            tB = backward_map(self.backward_target_net, next_goal, cfg.z_dim, cfg.norm_z)
        fp, bp = self._req(self.forward_net), self._req(self.backward_net)
        F1, F2 = forward_map(fp, obs, z, action)
        if cfg.debug:                                              # IdentityMap: B = next_goal, a leaf like any input
            y = Bm = next_goal.clone().requires_grad_(keep)
        else:
            y = backward_map_raw(bp, next_goal)
            Bm = math.sqrt(cfg.z_dim) * F.normalize(y, dim=1) if cfg.norm_z else y * 1.0
        if keep:
            for x in (F1, F2) + (() if cfg.debug else (Bm, y)):
                x.retain_grad()
        L = fb_loss_terms(F1, F2, Bm, tF1, tF2, tB, discount, cfg.ortho_coef)
        fb_loss = L["fb_loss"]
        if cfg.q_loss:                                             # fb_ddpg.py:330-340
            with torch.no_grad():
                nq = torch.min(torch.einsum('sd, sd -> s', tF1, z), torch.einsum('sd, sd -> s', tF2, z))
                cov = torch.matmul(Bm.T, Bm) / Bm.shape[0]
                inv_cov = torch.inverse(cov)
                implicit_reward = (torch.matmul(Bm, inv_cov) * z).sum(dim=1)
                target_Q = implicit_reward.detach() + discount.squeeze(1) * nq
            Q1, Q2 = [torch.einsum('sd, sd -> s', Fi, z) for Fi in (F1, F2)]
            q_loss = F.mse_loss(Q1, target_Q) + F.mse_loss(Q2, target_Q)
            fb_loss = fb_loss + cfg.q_loss_coef * q_loss
            metrics["q_loss"] = q_loss.item()
        metrics.update({                                            # fb_ddpg.py:356-377
            "target_M": L["target_M"].mean().item(), "M1": L["M1"].mean().item(), "F1": F1.mean().item(),
            "B": Bm.mean().item(), "B_norm": torch.norm(Bm, dim=-1).mean().item(),
            "z_norm": torch.norm(z, dim=-1).mean().item(), "fb_loss": fb_loss.item(),
            "fb_diag": L["fb_diag"].item(), "fb_offdiag": L["fb_offdiag"].item(),
            "orth_loss": L["orth_loss"].item(), "orth_loss_diag": L["orth_loss_diag"].item(),
            "orth_loss_offdiag": L["orth_loss_offdiag"].item()})
        with torch.no_grad():
            eye_diff = torch.matmul(Bm.T, Bm) / Bm.shape[0] - torch.eye(Bm.shape[1], dtype=Bm.dtype)
            metrics["orth_linf"] = torch.max(torch.abs(eye_diff)).item()
            metrics["orth_l2"] = eye_diff.norm().item() / math.sqrt(Bm.shape[1])
        metrics["fb_opt_lr"] = cfg.lr
        fb_loss.backward()
        gF = {k: v.grad for k, v in fp.items()}
        gB = {k: v.grad for k, v in bp.items()}
        self.fb_steps += 1
        adam_step(self.forward_net, gF, self.adam["forward_net"]["m"], self.adam["forward_net"]["v"],
                  self.fb_steps, cfg.lr)
        adam_step(self.backward_net, gB, self.adam["backward_net"]["m"], self.adam["backward_net"]["v"],
                  self.fb_steps, cfg.lr_coef * cfg.lr)

        # ---------------- update_actor (fb_ddpg.py:389-421) ------------- #
        ap = self._req(self.actor)
        fnew = self._req(self.forward_net)              # reference also tracks (and discards) these grads
        if cfg.boltzmann:                                          # fb_ddpg.py:391-393: rsample of the SquashedNormal
            premu, loc, std = diag_gaussian(ap, obs, z, cfg.log_std_min, cfg.log_std_max)
            u = loc + std * t(draws.eps_actor)
            act = torch.tanh(u)
            mu = torch.tanh(loc)
            log_prob = squashed_log_prob(loc, std, u).sum(-1, keepdim=True)
        else:
            mu = actor_mu(ap, obs, z)
            premu = None
            act = truncated_normal_sample(mu, cfg.stddev, cfg.stddev_clip, t(draws.eps_actor))
            log_prob = normal_log_prob(mu, cfg.stddev, act).sum(-1, keepdim=True)
        aF1, aF2 = forward_map(fnew, obs, z, act)
        Q1 = torch.einsum('sd, sd -> s', aF1, z)
        Q2 = torch.einsum('sd, sd -> s', aF2, z)
        Q = torch.min(Q1, Q2)
        actor_loss = (cfg.temp * log_prob.squeeze(1) - Q).mean() if cfg.boltzmann else -Q.mean()    # fb_ddpg.py:406
        if keep:
            for x in (aF1, aF2, act, mu) + ((premu,) if cfg.boltzmann else ()):
                x.retain_grad()
        actor_loss.backward()
        gA = {k: v.grad for k, v in ap.items()}
        self.actor_steps += 1
        adam_step(self.actor, gA, self.adam["actor"]["m"], self.adam["actor"]["v"], self.actor_steps, cfg.lr)
        metrics.update(actor_loss=actor_loss.item(), q=Q.mean().item(), actor_logprob=log_prob.mean().item())

        # ---------------- target EMA (fb_ddpg.py:500-503) --------------- #
        soft_update(self.forward_net, self.forward_target_net, cfg.fb_target_tau)
        soft_update(self.backward_net, self.backward_target_net, cfg.fb_target_tau)

        if keep:
            d = lambda x: x.detach().clone()
            self.last = dict(
                obs=obs, action=action, next_obs=next_obs, next_goal=next_goal, discount=discount, z=d(z),
                backward_input=backward_input, mu_next=d(mu_n), next_action=d(next_action),
                tF1=d(tF1), tF2=d(tF2), tB=d(tB), F1=d(F1), F2=d(F2), Bm=d(Bm), y=d(y),
                dF1=d(F1.grad), dF2=d(F2.grad), dBm=d(Bm.grad), dy=d(y.grad),
                grads_forward={k: d(v) for k, v in gF.items()}, grads_backward={k: d(v) for k, v in gB.items()},
                mu=d(mu), pi_action=d(act), aF1=d(aF1), aF2=d(aF2), daF1=d(aF1.grad), daF2=d(aF2.grad),
                d_pi_action=d(act.grad), d_mu=None if cfg.boltzmann else d(mu.grad),
                # gradient at the policy head's raw output: [B, a] pre-tanh (Actor) / [B, 2a] (loc | raw log-std) (DiagGaussianActor)
                d_premu=d(premu.grad) if cfg.boltzmann else d(mu.grad) * (1 - d(mu) ** 2),
                grads_actor={k: d(v) for k, v in gA.items()})
        return metrics

    # -- the same update cut at the optimiser steps (for the data-parallel schedule test) ---------------- #
    def dp_begin(self, batch: tp.Dict[str, np.ndarray], draws: Draws) -> None:
        t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32)
        cfg = self.cfg
        self._dp = dict(obs=t(batch["obs"]), action=t(batch["action"]), next_obs=t(batch["next_obs"]),
                        discount=t(batch["discount"]).reshape(-1, 1), draws=draws)
        self._dp["next_goal"] = t(batch["next_goal"]) if cfg.use_goal else self._dp["next_obs"]
        bi = t(batch["goal"]) if cfg.use_goal else self._dp["obs"]
        fg = None
        if cfg.future_ratio > 0:
            fg = t(batch["future_goal"] if cfg.use_goal else batch["future_obs"])
        self._dp["z"] = self.mix_z(sample_z_from_gauss(t(draws.z_gauss), cfg.z_dim, None if cfg.norm_z else t(draws.z_uniform)),
                                   bi, draws, fg)

    def dp_fb_grads(self) -> tp.Tuple[Params, Params]:
        """gradients of update_fb's loss wrt (forward_net, backward_net)  (fb_ddpg.py:303-383)"""
        cfg, d = self.cfg, self._dp
        t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32)
        with torch.no_grad():
            mu_n = actor_mu(self.actor, d["next_obs"], d["z"])
            na = truncated_normal_sample(mu_n, cfg.stddev, cfg.stddev_clip, t(d["draws"].eps_next))
            tF1, tF2 = forward_map(self.forward_target_net, d["next_obs"], d["z"], na)
            tB = backward_map(self.backward_target_net, d["next_goal"], cfg.z_dim, cfg.norm_z)
        fp, bp = self._req(self.forward_net), self._req(self.backward_net)
        F1, F2 = forward_map(fp, d["obs"], d["z"], d["action"])
        Bm = backward_map(bp, d["next_goal"], cfg.z_dim, cfg.norm_z)
        fb_loss_terms(F1, F2, Bm, tF1, tF2, tB, d["discount"], cfg.ortho_coef)["fb_loss"].backward()
        return {k: v.grad for k, v in fp.items()}, {k: v.grad for k, v in bp.items()}

    def dp_fb_step(self, gF: Params, gB: Params) -> None:
        self.fb_steps += 1
        adam_step(self.forward_net, gF, self.adam["forward_net"]["m"], self.adam["forward_net"]["v"], self.fb_steps, self.cfg.lr)
        adam_step(self.backward_net, gB, self.adam["backward_net"]["m"], self.adam["backward_net"]["v"], self.fb_steps,
                  self.cfg.lr_coef * self.cfg.lr)
        # the HIP path fuses the EMA into this pass (legal: nothing in between reads a target net)
        soft_update(self.forward_net, self.forward_target_net, self.cfg.fb_target_tau)
        soft_update(self.backward_net, self.backward_target_net, self.cfg.fb_target_tau)

    def dp_actor_grads(self) -> Params:
        cfg, d = self.cfg, self._dp
        t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32)
        ap = self._req(self.actor)
        assert not cfg.boltzmann, "the data-parallel cut of the oracle covers the default actor only"
        mu = actor_mu(ap, d["obs"], d["z"])
        act = truncated_normal_sample(mu, cfg.stddev, cfg.stddev_clip, t(d["draws"].eps_actor))
        aF1, aF2 = forward_map(self.forward_net, d["obs"], d["z"], act)
        Q = torch.min(torch.einsum('sd, sd -> s', aF1, d["z"]), torch.einsum('sd, sd -> s', aF2, d["z"]))
        (-Q.mean()).backward()
        return {k: v.grad for k, v in ap.items()}

    def dp_actor_step(self, gA: Params) -> None:
        self.actor_steps += 1
        adam_step(self.actor, gA, self.adam["actor"]["m"], self.adam["actor"]["v"], self.actor_steps, self.cfg.lr)

    # -- inference helpers (fb_ddpg.py:258-289, 177-222) ------------------ #
    def act_mean(self, obs: np.ndarray, z: np.ndarray) -> np.ndarray:
        with torch.no_grad():
            return actor_mu(self.actor, torch.as_tensor(obs, dtype=torch.float32).reshape(1, -1),
                            torch.as_tensor(z, dtype=torch.float32).reshape(1, -1))[0].numpy()

    def infer_z(self, goal_obs: torch.Tensor, reward: torch.Tensor) -> np.ndarray:
        """infer_meta_from_obs_and_rewards (fb_ddpg.py:201-222)."""
        with torch.no_grad():
            Bm = backward_map(self.backward_net, goal_obs, self.cfg.z_dim, self.cfg.norm_z)
        z = torch.matmul(reward.T, Bm) / reward.shape[0]
        if self.cfg.norm_z:                                        # fb_ddpg.py:217-218
            z = math.sqrt(self.cfg.z_dim) * F.normalize(z, dim=1)
        return z.squeeze().numpy()

    def state_tensors(self) -> tp.Dict[str, np.ndarray]:
        out = {}
        for n in ("actor", "forward_net", "backward_net", "forward_target_net", "backward_target_net"):
            for k, v in getattr(self, n).items():
                out[f"{n}/{k}"] = v.detach().numpy().copy()
        for n in self.NETS:
            for mv in ("m", "v"):
                for k, v in self.adam[n][mv].items():
                    out[f"adam_{mv}/{n}/{k}"] = v.numpy().copy()
        return out


def synthetic_storage(rng: np.random.Generator, n_episodes: int, T: int, obs_dim: int, action_dim: int,
                      goal_dim: tp.Optional[int] = None, lengths: tp.Optional[np.ndarray] = None
                      ) -> tp.Tuple[tp.Dict[str, np.ndarray], np.ndarray]:
    """Episode-major storage float32[n_eps, T+1, dim] like ReplayBuffer._storage
    (in_memory_replay_buffer.py:119-125); row 0 of each episode is the dummy FIRST step.
    Value distributions follow SURVEY.md section 8d (obs ~ N(0,1), action ~ U(-1,1), discount 1)."""
    st = {
        "observation": rng.standard_normal((n_episodes, T + 1, obs_dim)).astype(np.float32),
        "action": rng.uniform(-1, 1, (n_episodes, T + 1, action_dim)).astype(np.float32),
        "reward": rng.uniform(0, 1, (n_episodes, T + 1, 1)).astype(np.float32),
        "discount": np.ones((n_episodes, T + 1, 1), np.float32),
    }
    if goal_dim is not None:
        st["goal"] = rng.standard_normal((n_episodes, T + 1, goal_dim)).astype(np.float32)
    if lengths is None:
        lengths = np.full(n_episodes, T, np.int32)
    return st, np.asarray(lengths, np.int32)
