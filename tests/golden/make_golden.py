"""Generate the golden fixtures in this directory by running the *real* reference.

DEV-CONTAINER ONLY: imports facebookresearch/controllable_agent from /root/reference
(with in-process stubs for hydra / omegaconf / dm_env / dm_control -- none of which carry
arithmetic of the hot path) and records inputs and expected
outputs of ``FBDDPGAgent.update()`` and ``ReplayBuffer.sample()``.  Only the *outputs*
(``*.npz`` / ``*.json``, data not code) are committed; nothing here runs on the GPU box.

    python tests/golden/make_golden.py            # regenerates every fixture

Random draws are injected (monkey-patched ``np.random.*``, ``torch.randperm``, ``torch.randn``,
``utils._standard_normal``) from a numpy ``Generator`` so the same draws can be replayed through
``oracle.fb_oracle`` and through the HIP path.
"""
from __future__ import annotations

import contextlib
import enum
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parents[1]))

from oracle import fb_oracle as fo  # noqa: E402  (helpers for synthetic weights / storage / draws)
from oracle import discrete_fb_oracle as do  # noqa: E402


# --------------------------------------------------------------------------- #
def import_reference():
    if not REF.exists():
        raise SystemExit("make_golden.py needs /root/reference (dev container only)")
    sys.dont_write_bytecode = True
    sys.path.insert(0, str(REF))

    def mod(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
        return m

    class _CS:
        _i = None

        @classmethod
        def instance(cls):
            cls._i = cls._i or cls()
            return cls._i

        def store(self, **kw):
            pass

    mod("hydra").core = mod("hydra.core")
    mod("hydra.core.config_store", ConfigStore=_CS)
    mod("omegaconf", MISSING="???", II=lambda s: "${%s}" % s, SI=lambda s: s, DictConfig=dict)

    class StepType(enum.IntEnum):
        FIRST = 0
        MID = 1
        LAST = 2

    class _Spec:
        def __init__(self, shape=None, dtype=None, *a, **k):
            self.shape, self.dtype = shape, dtype

    specs = mod("dm_env.specs", Array=_Spec, BoundedArray=_Spec, DiscreteArray=_Spec)
    mod("dm_env", StepType=StepType, specs=specs, TimeStep=object, Environment=object)
    dc = mod("dm_control")
    dc.suite = mod("dm_control.suite", ALL_TASKS=())
    w = mod("dm_control.suite.wrappers")
    w.action_scale = mod("dm_control.suite.wrappers.action_scale")
    w.pixels = mod("dm_control.suite.wrappers.pixels")
    mod("url_benchmark.custom_dmc_tasks")
    # (sf.py:401 reads the goal dimension as len(next(iter(goals.goals.funcs[goal_space].values()))()))
    _gdims = {"simplified_walker": 3, "simplified_quadruped": 2}
    mod("url_benchmark.goals", get_goal_space_dim=_gdims.__getitem__,
        goals=types.SimpleNamespace(funcs={k: {"stub": (lambda n=n: np.zeros(n, np.float32))} for k, n in _gdims.items()}))
    import url_benchmark  # noqa: F401
    mod("url_benchmark.agent").__path__ = [str(REF / "url_benchmark/agent")]
    from url_benchmark.agent import fb_ddpg, discrete_fb, sf
    from url_benchmark.in_memory_replay_buffer import ReplayBuffer
    from url_benchmark import dmc, utils
    return types.SimpleNamespace(fb_ddpg=fb_ddpg, discrete_fb=discrete_fb, sf=sf, ReplayBuffer=ReplayBuffer, dmc=dmc, utils=utils,
                                 StepType=StepType)


# --------------------------------------------------------------------------- #
def make_ref_agent(R, cfg: fo.OracleConfig, goal_space=None, discrete=False, **extra):
    cls = R.discrete_fb.DiscreteFBAgent if discrete else R.fb_ddpg.FBDDPGAgent
    return cls(
        obs_type="states", obs_shape=(cfg.obs_dim,), action_shape=(cfg.action_dim,), device="cpu",
        num_expl_steps=0, use_tb=True, use_wandb=False, use_hiplog=False, update_encoder=True,
        goal_space=goal_space, lr=cfg.lr, lr_coef=cfg.lr_coef, fb_target_tau=cfg.fb_target_tau,
        hidden_dim=cfg.hidden_dim, backward_hidden_dim=cfg.backward_hidden_dim, feature_dim=cfg.feature_dim,
        z_dim=cfg.z_dim, stddev_schedule=str(cfg.stddev), stddev_clip=cfg.stddev_clip, batch_size=cfg.batch_size,
        ortho_coef=cfg.ortho_coef, mix_ratio=cfg.mix_ratio, q_loss=cfg.q_loss, q_loss_coef=cfg.q_loss_coef,
        future_ratio=cfg.future_ratio, norm_z=cfg.norm_z, rand_weight=cfg.rand_weight, add_trunk=cfg.add_trunk, preprocess=cfg.preprocess,
        boltzmann=cfg.boltzmann, temp=cfg.temp, log_std_bounds=(cfg.log_std_min, cfg.log_std_max), debug=cfg.debug,
        update_every_steps=1, **extra)


def load_nets(agent, nets):
    for name in nets:
        getattr(agent, name).load_state_dict(nets[name])
    agent.forward_target_net.load_state_dict(agent.forward_net.state_dict())
    agent.backward_target_net.load_state_dict(agent.backward_net.state_dict())


def fill_ref_buffer(R, storage, lengths, discount, future=0.99, max_len=None, meta_z=None):
    """Fill a reference ReplayBuffer through its real ``add()`` (in_memory_replay_buffer.py:104-133)."""
    n_eps = storage["observation"].shape[0]
    has_goal = "goal" in storage
    TS = R.dmc.ExtendedGoalTimeStep if has_goal else R.dmc.ExtendedTimeStep
    rb = R.ReplayBuffer(max_episodes=n_eps, discount=discount, future=future, max_episode_length=max_len)
    for e in range(n_eps):
        L = int(lengths[e])
        for s in range(L + 1):
            st = R.StepType.FIRST if s == 0 else (R.StepType.LAST if s == L else R.StepType.MID)
            kw = dict(step_type=st, reward=float(storage["reward"][e, s, 0]), discount=float(storage["discount"][e, s, 0]),
                      observation=storage["observation"][e, s], action=storage["action"][e, s])
            if has_goal:
                kw["goal"] = storage["goal"][e, s]
            ts = TS(**kw)
            ts.physics = np.zeros(2, np.float32)
            rb.add(ts, {} if meta_z is None else {"z": meta_z[e, s]})
    return rb


@contextlib.contextmanager
def inject(R, d: fo.Draws, variable_len: bool, mix_ratio: float = 0.5, action_noise: bool = True):
    """Route every random draw of one ``update()`` to the prepared values."""
    calls = {"randint": 0, "uniform": 0}
    o_randint, o_choice, o_uniform, o_geo = np.random.randint, np.random.choice, np.random.uniform, np.random.geometric
    o_randperm, o_randn, o_sn, o_rand = torch.randperm, torch.randn, R.utils._standard_normal, torch.rand
    import torch.distributions.normal as tdn     # boltzmann: Normal.sample -> torch.normal, Normal.rsample -> _standard_normal
    o_normal, o_tdn_sn = torch.normal, tdn._standard_normal
    eps_queue = [d.eps_next, d.eps_actor]

    def randint(low, high=None, size=None, **kw):
        calls["randint"] += 1
        if variable_len:            # only the step draw uses randint (in_memory_replay_buffer.py:155)
            return (d.step_idx - 1).copy()
        return d.ep_idx.copy() if calls["randint"] == 1 else (d.step_idx - 1).copy()

    def choice(a, size=None, p=None, **kw):
        return d.ep_idx.copy()

    def uniform(*a, size=None, **kw):
        assert size == len(d.mix_uniform)
        calls["uniform"] += 1                  # 1st: mix (fb_ddpg.py:471), 2nd: hindsight replay (:490)
        return d.mix_uniform.copy() if calls["uniform"] == 1 else d.future_uniform.copy()

    def geometric(p, size=None):               # in_memory_replay_buffer.py:159 (clip at :160 is then a no-op)
        if d.future_idx is None:               # a buffer with future < 1 under an agent that ignores future_obs: any draw will do
            return o_geo(p, size=size)
        return (d.future_idx - d.step_idx).copy()

    def randperm(n, **kw):
        return torch.from_numpy(d.perm.copy())

    def randn(*a, **kw):
        return torch.from_numpy(d.z_gauss.copy())

    def rand(*a, **kw):
        shape = tuple(kw.get("size", a[0] if len(a) == 1 and isinstance(a[0], (tuple, list, torch.Size)) else a))
        B = len(d.mix_uniform)
        if d.z_uniform is not None and shape == d.z_uniform.shape:      # sample_z, norm_z False (fb_ddpg.py:230)
            return torch.from_numpy(d.z_uniform.copy())
        mix_idxs = np.where(d.mix_uniform < mix_ratio)[0]               # rand_weight (fb_ddpg.py:477, :479): mixed rows only
        if shape == (len(mix_idxs), B):
            return torch.from_numpy(d.rand_weight[mix_idxs].copy())
        assert shape == (len(mix_idxs), 1), shape
        return torch.from_numpy(d.rand_weight_u[mix_idxs].copy()).reshape(-1, 1)

    def standard_normal(shape, dtype, device):
        return torch.from_numpy(eps_queue.pop(0).copy())

    def normal(mean, std, *a, **kw):           # torch.distributions.Normal.sample (fb_ddpg.py:306)
        return mean + std * torch.from_numpy(eps_queue.pop(0).copy())

    np.random.randint, np.random.choice, np.random.uniform, np.random.geometric = randint, choice, uniform, geometric
    torch.randperm, torch.randn, R.utils._standard_normal = randperm, randn, standard_normal
    torch.normal, tdn._standard_normal = normal, standard_normal
    if d.z_uniform is not None or d.rand_weight is not None:
        torch.rand = rand
    try:
        yield
    finally:
        np.random.randint, np.random.choice, np.random.uniform, np.random.geometric = o_randint, o_choice, o_uniform, o_geo
        torch.randperm, torch.randn, R.utils._standard_normal, torch.rand = o_randperm, o_randn, o_sn, o_rand
        torch.normal, tdn._standard_normal = o_normal, o_tdn_sn
    assert not eps_queue or not action_noise, "update() did not consume both action-noise draws"


def ref_state(agent):
    out = {}
    has_actor = hasattr(agent, "actor")          # DiscreteFBAgent has none
    for n in (("actor",) if has_actor else ()) + ("forward_net", "backward_net", "forward_target_net", "backward_target_net"):
        for k, v in getattr(agent, n).state_dict().items():
            out[f"{n}/{k}"] = v.detach().numpy().copy()
    for opt, nets in (((agent.actor_opt, ("actor",)),) if has_actor else ()) + ((agent.fb_opt, ("forward_net", "backward_net")),):
        for n in nets:
            for (k, p) in getattr(agent, n).named_parameters():
                st = opt.state.get(p, None)
                if st:
                    out[f"adam_m/{n}/{k}"] = st["exp_avg"].numpy().copy()
                    out[f"adam_v/{n}/{k}"] = st["exp_avg_sq"].numpy().copy()
    return out


def checksums(state):
    return {k: [float(np.sum(v, dtype=np.float64)), float(np.sqrt(np.sum(v.astype(np.float64) ** 2)))]
            for k, v in state.items() if not k.startswith("adam_")}


# --------------------------------------------------------------------------- #
def trace_fixture(R, name, cfg: fo.OracleConfig, seed, n_eps, T, n_steps, goal_space=None, variable_len=False,
                  full_state=True, checksum_steps=(), discrete=False):
    rng = np.random.default_rng(seed)
    shapes = do.NET_SHAPES if discrete else fo.NET_SHAPES
    nets = {n: fo.synthetic_params(rng, shapes[n](cfg)) for n in shapes}
    lengths = None
    if variable_len:
        lengths = rng.integers(max(2, T // 2), T + 1, size=n_eps).astype(np.int32)
        lengths[0] = T
    storage, lengths = fo.synthetic_storage(rng, n_eps, T, cfg.obs_dim, cfg.action_dim,
                                            cfg.goal_dim if cfg.use_goal else None, lengths)
    if discrete:
        do.synthetic_actions(rng, storage, cfg.action_dim)
    agent = make_ref_agent(R, cfg, goal_space=goal_space, discrete=discrete)
    load_nets(agent, nets)
    rb = fill_ref_buffer(R, storage, lengths, cfg.discount, future=cfg.future, max_len=(T + 1) if variable_len else None)
    assert rb._is_fixed_episode_length == (not variable_len)
    arrays, meta = {}, {"name": name, "seed": seed, "n_eps": n_eps, "T": T, "n_steps": n_steps,
                        "goal_space": goal_space, "variable_len": variable_len, "discrete": discrete,
                        "cfg": {k: getattr(cfg, k) for k in cfg.__dataclass_fields__}, "metrics": [], "checksums": {}}
    if full_state:
        for n, p in nets.items():
            for k, v in p.items():
                arrays[f"init/{n}/{k}"] = v.numpy()
        for k, v in storage.items():
            arrays[f"storage/{k}"] = v
        arrays["lengths"] = lengths
    for s in range(n_steps):
        d = fo.make_draws(rng, cfg, n_eps, lengths)
        with inject(R, d, variable_len, cfg.mix_ratio, action_noise=not discrete):
            m = agent.update(rb, s)
        meta["metrics"].append({k: float(v) for k, v in m.items()})
        if full_state:
            for f in d.__dataclass_fields__:
                if getattr(d, f) is not None:
                    arrays[f"draws/{s}/{f}"] = getattr(d, f)
            for k, v in ref_state(agent).items():
                arrays[f"state/{s}/{k}"] = v
        if (s + 1) in checksum_steps:
            meta["checksums"][str(s + 1)] = checksums(ref_state(agent))
    (HERE / f"{name}.json").write_text(json.dumps(meta, indent=1))
    if full_state:
        np.savez_compressed(HERE / f"{name}.npz", **arrays)
    print(f"[{name}] steps={n_steps} fb_loss={[round(m['fb_loss'], 4) for m in meta['metrics'][:3]]} "
          f"actor_loss={[round(m.get('actor_loss', float('nan')), 4) for m in meta['metrics'][:3]]}")


def future_fixtures(R):
    """hindsight replay (fb_ddpg.py:487-491) on buffers with future < 1 (in_memory_replay_buffer.py:157-161)"""
    trace_fixture(R, "tiny_future_trace", tiny_cfg(future=0.8, future_ratio=0.4), seed=103, n_eps=6, T=12, n_steps=4)
    trace_fixture(R, "tiny_future_goal_trace", tiny_cfg(goal_dim=3, use_goal=True, future=0.7, future_ratio=0.5, mix_ratio=0.3),
                  seed=104, n_eps=7, T=11, n_steps=3, goal_space="simplified_walker", variable_len=True)


def nonorm_fixture(R):
    """norm_z=False: BackwardMap unprojected, z = sqrt(d) U g/|g| (fb_ddpg.py:227-231, :483; fb_modules.py:228-229)"""
    trace_fixture(R, "tiny_nonorm_trace", tiny_cfg(norm_z=False, future=0.8, future_ratio=0.3), seed=105, n_eps=6, T=12, n_steps=4)


def debug_fixture(R):
    """cfg.debug (fb_ddpg.py:128-130): backward_net = backward_target_net = IdentityMap() -- B(goal) = goal, no parameters, no
    projection; z_dim must equal the goal dimension.  Once with the defaults (z-mix of raw observations, projected once), once with a
    goal space, variable episode lengths, norm_z off and the q_loss branch (covariance of the raw goals)."""
    trace_fixture(R, "tiny_debug_trace", tiny_cfg(z_dim=5, debug=True), seed=161, n_eps=6, T=12, n_steps=4)
    trace_fixture(R, "tiny_debug_goal_trace", tiny_cfg(goal_dim=3, use_goal=True, z_dim=3, debug=True, norm_z=False, q_loss=True, batch_size=24),
                  seed=162, n_eps=7, T=11, n_steps=4, goal_space="simplified_walker", variable_len=True)
    # hindsight rows are then the raw future goal (fb_ddpg.py:491), rand_weight mixes raw goals (:475-482)
    trace_fixture(R, "tiny_debug_future_randw_trace", tiny_cfg(goal_dim=3, use_goal=True, z_dim=3, debug=True, future=0.7, future_ratio=0.5,
                                                               rand_weight=True, mix_ratio=0.6, batch_size=24),
                  seed=164, n_eps=7, T=11, n_steps=4, goal_space="simplified_walker", variable_len=True)
    # DiscreteFBAgent has the same switch (discrete_fb.py:134-136)
    trace_fixture(R, "tiny_discrete_debug_trace", tiny_cfg(action_dim=4, preprocess=False, z_dim=5, debug=True, q_loss=True, batch_size=24),
                  seed=163, n_eps=6, T=12, n_steps=4, discrete=True)


def randweight_fixture(R):
    """rand_weight=True (fb_ddpg.py:475-482), once with the default projection and once together with norm_z=False"""
    trace_fixture(R, "tiny_randw_trace", tiny_cfg(rand_weight=True, mix_ratio=0.6), seed=106, n_eps=6, T=12, n_steps=3)
    trace_fixture(R, "tiny_randw_nonorm_trace", tiny_cfg(rand_weight=True, norm_z=False, goal_dim=3, use_goal=True),
                  seed=107, n_eps=7, T=11, n_steps=3, goal_space="simplified_walker")


def trunk_fixture(R):
    """add_trunk=True (fb_modules.py:93-98, 168-173): Linear(2 Fd, H) + ReLU between the preprocess nets and the heads"""
    trace_fixture(R, "tiny_trunk_trace", tiny_cfg(add_trunk=True, goal_dim=3, use_goal=True), seed=108, n_eps=6, T=12,
                  n_steps=4, goal_space="simplified_walker")


def single_trunk_fixture(R):
    """preprocess=False (fb_modules.py:99-103, 174-178): one LayerNorm trunk on cat([obs, z, action]) / cat([obs, z])"""
    trace_fixture(R, "tiny_single_trunk_trace", tiny_cfg(preprocess=False), seed=109, n_eps=6, T=12, n_steps=4)
    trace_fixture(R, "tiny_single_trunk_goal_trace", tiny_cfg(preprocess=False, goal_dim=3, use_goal=True, z_dim=10, batch_size=24),
                  seed=110, n_eps=7, T=11, n_steps=3, goal_space="simplified_walker", variable_len=True)


def boltzmann_fixture(R):
    """boltzmann=True (fb_ddpg.py:118-120, 304-306, 391-393, 406): DiagGaussianActor + SquashedNormal, entropy-regularised
    actor loss with temp != 1 and non-default log_std_bounds in the second trace"""
    trace_fixture(R, "tiny_boltzmann_trace", tiny_cfg(boltzmann=True), seed=111, n_eps=6, T=12, n_steps=4)
    trace_fixture(R, "tiny_boltzmann_goal_trace", tiny_cfg(boltzmann=True, temp=0.3, log_std_min=-3.0, log_std_max=1.0, goal_dim=3,
                                                           use_goal=True, z_dim=10, batch_size=24),
                  seed=112, n_eps=7, T=11, n_steps=3, goal_space="simplified_walker", variable_len=True)


def discrete_fixture(R):
    """DiscreteFBAgent.update (discrete_fb.py:277-468): [B, d, A] heads, greedy target column +
    goal space + hindsight replay + variable lengths; softmax targets (boltzmann, temp) + q_loss (pinv) + norm_z=False"""
    trace_fixture(R, "tiny_discrete_trace", tiny_cfg(action_dim=4, preprocess=False, goal_dim=3, use_goal=True, future=0.8,
                                                     future_ratio=0.3, lr_coef=0.5),
                  seed=120, n_eps=7, T=11, n_steps=4, goal_space="simplified_walker", variable_len=True, discrete=True)
    trace_fixture(R, "tiny_discrete_boltz_trace", tiny_cfg(action_dim=3, preprocess=False, boltzmann=True, temp=0.7, q_loss=True,
                                                           norm_z=False, batch_size=32, z_dim=6, backward_hidden_dim=20),
                  seed=121, n_eps=6, T=12, n_steps=4, discrete=True)


def sf_state(agent):
    out = {}
    for n in ("actor", "successor_net", "successor_target_net", "feature_learner"):
        for k, v in getattr(agent, n).state_dict().items():
            out[f"{n}/{k}"] = v.detach().numpy().copy()
    for opt, n in ((agent.actor_opt, "actor"), (agent.sf_opt, "successor_net"), (agent.phi_opt, "feature_learner")):
        if opt is None:                                   # feature_learner="random": no phi_opt (sf.py:447-449)
            continue
        for (k, p) in getattr(agent, n).named_parameters():
            st = opt.state.get(p, None)
            if st:
                out[f"adam_m/{n}/{k}"] = st["exp_avg"].numpy().copy()
                out[f"adam_v/{n}/{k}"] = st["exp_avg_sq"].numpy().copy()
    return out


def sf_fixture(R):
    """SFAgent.update (sf.py:594-768; SURVEY section 8 row n4, second sibling): the TD regression on successor features with the
    two feature learners the round targets.  (1) the defaults: feature_learner="icm", q_loss=True (scalar Q regression);
    (2) feature_learner="lap" (Laplacian + orthonormality loss), q_loss=False (regression in feature space), goal space, variable
    episode lengths, lr_coef=5 like the reference default."""
    _sf_traces(R, (
        ("tiny_sf_icm_trace", "icm", True, dict(z_dim=10, backward_hidden_dim=20, lr_coef=5.0, mix_ratio=0.0), dict(seed=131, n_eps=6, T=12, n_steps=4)),
        ("tiny_sf_lap_trace", "lap", False, dict(goal_dim=3, use_goal=True, z_dim=8, backward_hidden_dim=22, batch_size=24, lr_coef=5.0, mix_ratio=0.0),
         dict(seed=132, n_eps=7, T=11, n_steps=4, goal_space="simplified_walker", variable_len=True))))


def sf_more_fixture(R):
    """Three more feature learners of sf.py on the same update: "random" (:84-92, 447: feature_net frozen, no phi_opt, no phi_loss
    metric), "autoencoder" (:249-262, decoder(phi(goal)) vs goal; here with a goal space, so the decoder emits goal_dim columns) and
    "transition" (:215-227, forward_dynamic_net(cat[phi(goal), action]) vs next_goal); plus their constructors under seed 1."""
    _sf_traces(R, (
        ("tiny_sf_random_trace", "random", True, dict(z_dim=10, backward_hidden_dim=20, lr_coef=5.0, mix_ratio=0.0), dict(seed=133, n_eps=6, T=12, n_steps=4)),
        ("tiny_sf_autoencoder_trace", "autoencoder", False,
         dict(goal_dim=3, use_goal=True, z_dim=8, backward_hidden_dim=22, batch_size=24, lr_coef=5.0, mix_ratio=0.0),
         dict(seed=134, n_eps=7, T=11, n_steps=4, goal_space="simplified_walker", variable_len=True)),
        ("tiny_sf_transition_trace", "transition", True, dict(z_dim=10, backward_hidden_dim=20, lr_coef=2.0, mix_ratio=0.0),
         dict(seed=135, n_eps=6, T=12, n_steps=4))))
    _sf_inits(R, ("random", "autoencoder", "transition"))


def sf_mix_fixture(R):
    """SFAgent with mix_ratio > 0 (sf.py:725-739): rows with uniform < mix_ratio take z = sqrt(d) normalize(phi(next_goal[perm]) @
    pinv(phi^T phi / B)) instead of the gaussian draw.  Three learners: icm (feature-space critic loss), lap with a goal space and
    variable episode lengths, and identity (phi = the raw goal: the covariance of the observations themselves)."""
    _sf_traces(R, (
        ("tiny_sf_mix_icm_trace", "icm", True, dict(z_dim=10, backward_hidden_dim=20, lr_coef=5.0, mix_ratio=0.5), dict(seed=151, n_eps=6, T=12, n_steps=4)),
        ("tiny_sf_mix_lap_trace", "lap", False,
         dict(goal_dim=3, use_goal=True, z_dim=8, backward_hidden_dim=22, batch_size=24, lr_coef=5.0, mix_ratio=0.4),
         dict(seed=152, n_eps=7, T=11, n_steps=4, goal_space="simplified_walker", variable_len=True)),
        ("tiny_sf_mix_identity_trace", "identity", True, dict(z_dim=5, backward_hidden_dim=20, lr_coef=5.0, mix_ratio=0.6),
         dict(seed=153, n_eps=6, T=12, n_steps=4))))


def sf_svdp_fixture(R):
    """feature_learner="svd_p" (SVDP, sf.py:337-362; the paper's LRA-P): a second net mu_net on cat[goal, action], the low-rank
    loss on P = mu . phi(next_goal)^T plus the orthonormality loss of phi(next_goal).  Once with the defaults, once with a goal
    space + variable episode lengths + the feature-space critic loss; and the constructor under seed 1."""
    _sf_traces(R, (
        ("tiny_sf_svdp_trace", "svd_p", True, dict(z_dim=10, backward_hidden_dim=20, lr_coef=5.0, mix_ratio=0.0), dict(seed=136, n_eps=6, T=12, n_steps=4)),
        ("tiny_sf_svdp_goal_trace", "svd_p", False,
         dict(goal_dim=3, use_goal=True, z_dim=8, backward_hidden_dim=22, batch_size=24, lr_coef=5.0, mix_ratio=0.0),
         dict(seed=137, n_eps=7, T=11, n_steps=4, goal_space="simplified_walker", variable_len=True))))
    _sf_inits(R, ("svd_p",))


def sf_latent_fixture(R):
    """feature_learner="latent" (TransitionLatentModel, sf.py:230-246): forward_dynamic_net(cat[phi(goal), action]) regressed on
    target_feature_net(next_goal); the target net has its OWN initial weights and follows feature_net at rate 0.01 inside the
    learner's forward(), i.e. before phi_opt.step().  sf_target_tau is set to 0.03 so that a mix-up of the two rates shows."""
    _sf_traces(R, (
        ("tiny_sf_latent_trace", "latent", True, dict(z_dim=10, backward_hidden_dim=20, lr_coef=5.0, mix_ratio=0.0, fb_target_tau=0.03),
         dict(seed=138, n_eps=6, T=12, n_steps=4)),))
    _sf_inits(R, ("latent",))


def sf_svdsr_fixture(R):
    """feature_learner="svd_sr" (SVDSR, sf.py:264-299; the paper's LRA-SR): SR = phi(goal) . mu_net(next_goal)^T against 0.99 x the
    product of two target nets (own weights, moved at 0.01 inside forward()), plus the orthonormality loss of phi(goal); with a goal
    space + variable lengths in the second trace."""
    _sf_traces(R, (
        ("tiny_sf_svdsr_trace", "svd_sr", True, dict(z_dim=10, backward_hidden_dim=20, lr_coef=5.0, mix_ratio=0.0, fb_target_tau=0.03),
         dict(seed=139, n_eps=6, T=12, n_steps=4)),
        ("tiny_sf_svdsr_goal_trace", "svd_sr", False,
         dict(goal_dim=3, use_goal=True, z_dim=8, backward_hidden_dim=22, batch_size=24, lr_coef=5.0, mix_ratio=0.0),
         dict(seed=140, n_eps=7, T=11, n_steps=4, goal_space="simplified_walker", variable_len=True))))
    _sf_inits(R, ("svd_sr",))


def sf_svdsrv2_fixture(R):
    """feature_learner="svd_srv2" (SVDSRv2, sf.py:303-335): svd_sr with mu on the goal and the features of next_goal (0.98)."""
    _sf_traces(R, (
        ("tiny_sf_svdsrv2_trace", "svd_srv2", True, dict(z_dim=10, backward_hidden_dim=20, lr_coef=5.0, mix_ratio=0.0, fb_target_tau=0.03),
         dict(seed=141, n_eps=6, T=12, n_steps=4)),))
    _sf_inits(R, ("svd_srv2",))


def sf_contrastive_fixture(R):
    """feature_learner="contrastive" (ContrastiveFeature, sf.py:118-143; the paper's CL): the cosine of phi(goal) and mu_net(future_goal)
    over the batch, InfoNCE without the diagonal in the denominator; the buffer samples hindsight goals (future = 0.8).  Second trace
    with a goal space and variable episode lengths."""
    _sf_traces(R, (
        ("tiny_sf_contrastive_trace", "contrastive", True, dict(z_dim=10, backward_hidden_dim=20, lr_coef=5.0, mix_ratio=0.0, future=0.8),
         dict(seed=142, n_eps=6, T=12, n_steps=4)),
        ("tiny_sf_contrastive_goal_trace", "contrastive", False,
         dict(goal_dim=3, use_goal=True, z_dim=8, backward_hidden_dim=22, batch_size=24, lr_coef=5.0, mix_ratio=0.0, future=0.7),
         dict(seed=143, n_eps=7, T=11, n_steps=4, goal_space="simplified_walker", variable_len=True))))
    _sf_inits(R, ("contrastive",))


def sf_contrastivev2_fixture(R):
    """feature_learner="contrastivev2" (sf.py:159-186): contrastive with mu_net on the goal and feature_net on the hindsight goal."""
    _sf_traces(R, (
        ("tiny_sf_contrastivev2_trace", "contrastivev2", True,
         dict(goal_dim=3, use_goal=True, z_dim=8, backward_hidden_dim=22, batch_size=24, lr_coef=5.0, mix_ratio=0.0, future=0.75),
         dict(seed=144, n_eps=7, T=11, n_steps=4, goal_space="simplified_walker", variable_len=True)),))
    _sf_inits(R, ("contrastivev2",))


def sf_identity_fixture(R):
    """feature_learner="identity" (sf.py:94-98): feature_net = nn.Identity(), z_dim = the goal dimension, nothing trained (no phi_opt)."""
    _sf_traces(R, (
        ("tiny_sf_identity_trace", "identity", True, dict(z_dim=5, backward_hidden_dim=20, lr_coef=5.0, mix_ratio=0.0),
         dict(seed=145, n_eps=6, T=12, n_steps=4)),))


def _sf_traces(R, table):
    from oracle import sf_oracle as so
    for name, learner, q_loss, kw, extra in table:
        cfg = tiny_cfg(**kw)
        seed, n_eps, T, n_steps = extra["seed"], extra["n_eps"], extra["T"], extra["n_steps"]
        goal_space, variable_len = extra.get("goal_space"), extra.get("variable_len", False)
        rng = np.random.default_rng(seed)
        shapes = so.net_shapes(cfg, learner)
        nets = {n: fo.synthetic_params(rng, shapes[n]) for n in shapes}
        lengths = None
        if variable_len:
            lengths = rng.integers(max(2, T // 2), T + 1, size=n_eps).astype(np.int32)
            lengths[0] = T
        storage, lengths = fo.synthetic_storage(rng, n_eps, T, cfg.obs_dim, cfg.action_dim, cfg.goal_dim if cfg.use_goal else None, lengths)
        agent = R.sf.SFAgent(
            obs_type="states", obs_shape=(cfg.obs_dim,), action_shape=(cfg.action_dim,), device="cpu", num_expl_steps=0,
            use_tb=True, use_wandb=False, use_hiplog=False, update_encoder=True, goal_space=goal_space, lr=cfg.lr,
            lr_coef=cfg.lr_coef, sf_target_tau=cfg.fb_target_tau, hidden_dim=cfg.hidden_dim,
            backward_hidden_dim=cfg.backward_hidden_dim, feature_dim=cfg.feature_dim, z_dim=cfg.z_dim,
            stddev_schedule=str(cfg.stddev), stddev_clip=cfg.stddev_clip, batch_size=cfg.batch_size, q_loss=q_loss,
            feature_learner=learner, mix_ratio=cfg.mix_ratio, update_every_steps=1)
        for n in nets:
            getattr(agent, n).load_state_dict(nets[n])
        agent.successor_target_net.load_state_dict(agent.successor_net.state_dict())
        rb = fill_ref_buffer(R, storage, lengths, cfg.discount, future=cfg.future, max_len=(T + 1) if variable_len else None)
        arrays = {f"init/{n}/{k}": v.numpy() for n, p in nets.items() for k, v in p.items()}
        arrays.update({f"storage/{k}": v for k, v in storage.items()})
        arrays["lengths"] = lengths
        meta = {"name": name, "seed": seed, "n_eps": n_eps, "T": T, "n_steps": n_steps, "goal_space": goal_space,
                "variable_len": variable_len, "feature_learner": learner, "sf_q_loss": q_loss,
                "cfg": {k: getattr(cfg, k) for k in cfg.__dataclass_fields__}, "metrics": []}
        for s_ in range(n_steps):
            d = fo.make_draws(rng, cfg, n_eps, lengths)
            with inject(R, d, variable_len, 0.0):
                m = agent.update(rb, s_)
            meta["metrics"].append({k: float(v) for k, v in m.items()})
            for f in d.__dataclass_fields__:
                if getattr(d, f) is not None:
                    arrays[f"draws/{s_}/{f}"] = getattr(d, f)
            for k, v in sf_state(agent).items():
                arrays[f"state/{s_}/{k}"] = v
        (HERE / f"{name}.json").write_text(json.dumps(meta, indent=1))
        np.savez_compressed(HERE / f"{name}.npz", **arrays)
        print(f"[{name}] sf_loss={[round(m['sf_loss'], 4) for m in meta['metrics']]} phi_loss={[round(m.get('phi_loss', float('nan')), 4) for m in meta['metrics']]}")


def sf_init_fixture(R):
    """SFAgent's constructor under torch.manual_seed(1) (sf.py:419-463: actor, successor_net, successor_target_net, then the
    feature learner, whose ``self.apply(weight_init)`` runs once in FeatureLearner.__init__ and, for icm, AGAIN over every Linear
    in ICM.__init__) for both feature learners; plus a checkpoint-style pickle is not needed here (reference_io covers the format)."""
    _sf_inits(R, ("icm", "lap"))
    print("[sf init] ok")
    # a checkpoint written by the reference holding a live SFAgent (icm) after two updates (pretrain.py:437-449)
    cfg = tiny_cfg(z_dim=10, backward_hidden_dim=20, lr_coef=5.0, mix_ratio=0.0)
    from oracle import sf_oracle as so
    rng = np.random.default_rng(34)
    shapes = so.net_shapes(cfg, "icm")
    nets = {n: fo.synthetic_params(rng, shapes[n]) for n in shapes}
    storage, lengths = fo.synthetic_storage(rng, 6, 12, cfg.obs_dim, cfg.action_dim, None, None)
    agent = R.sf.SFAgent(obs_type="states", obs_shape=(cfg.obs_dim,), action_shape=(cfg.action_dim,), device="cpu", num_expl_steps=0,
                         use_tb=True, use_wandb=False, use_hiplog=False, update_encoder=True, goal_space=None, lr=cfg.lr,
                         lr_coef=cfg.lr_coef, hidden_dim=cfg.hidden_dim, backward_hidden_dim=cfg.backward_hidden_dim,
                         feature_dim=cfg.feature_dim, z_dim=cfg.z_dim, batch_size=cfg.batch_size, feature_learner="icm",
                         update_every_steps=1)
    for n in nets:
        getattr(agent, n).load_state_dict(nets[n])
    agent.successor_target_net.load_state_dict(agent.successor_net.state_dict())
    rb = fill_ref_buffer(R, storage, lengths, cfg.discount)
    for s_ in range(2):
        with inject(R, fo.make_draws(rng, cfg, 6, lengths), False, 0.0):
            agent.update(rb, s_)
    with (HERE / "ref_checkpoint_tiny_sf.pt").open("wb") as f:
        torch.save({"agent": agent, "global_step": 2, "global_episode": 1, "replay_loader": rb}, f, pickle_protocol=4)
    obs = rng.standard_normal((5, cfg.obs_dim)).astype(np.float32)
    zs = fo.sample_z_from_gauss(torch.from_numpy(rng.standard_normal((5, cfg.z_dim)).astype(np.float32)), cfg.z_dim).numpy()
    with torch.no_grad():
        acts = np.stack([agent.act(obs[i], {"z": zs[i]}, 0, eval_mode=True) for i in range(5)])
    arrays = {f"state/{k}": v for k, v in sf_state(agent).items()}
    arrays.update(obs=obs, z=zs, act_eval=acts)
    np.savez_compressed(HERE / "ref_checkpoint_sf_expect.npz", **arrays)
    print("[sf checkpoint] ok")


def _sf_inits(R, learners):
    for learner in learners:
        cfg = tiny_cfg(z_dim=10, backward_hidden_dim=20, lr_coef=5.0, mix_ratio=0.0)
        torch.manual_seed(1)
        agent = R.sf.SFAgent(obs_type="states", obs_shape=(cfg.obs_dim,), action_shape=(cfg.action_dim,), device="cpu",
                             num_expl_steps=0, use_tb=True, use_wandb=False, use_hiplog=False, update_encoder=True, goal_space=None,
                             hidden_dim=cfg.hidden_dim, backward_hidden_dim=cfg.backward_hidden_dim, feature_dim=cfg.feature_dim,
                             z_dim=cfg.z_dim, batch_size=cfg.batch_size, feature_learner=learner)
        arrays = {k: v for k, v in sf_state(agent).items()}
        arrays["torch_version"] = np.array(torch.__version__)
        np.savez_compressed(HERE / f"init_seed1_tiny_sf_{learner}.npz", **arrays)


def long_curve_fixtures(R):
    """50 free-running steps at full network dims, metric dict per step + parameter checksums at
    steps {1, 10, 50}: walker B=256; quadruped with goal_space B=256."""
    trace_fixture(R, "walker_b256_50", fo.OracleConfig(obs_dim=24, action_dim=6, goal_dim=24, z_dim=50, batch_size=256),
                  seed=211, n_eps=20, T=100, n_steps=50, full_state=False, checksum_steps=(1, 10, 50))
    trace_fixture(R, "quadruped_goal_b256_50",
                  fo.OracleConfig(obs_dim=78, action_dim=12, goal_dim=2, z_dim=100, batch_size=256, use_goal=True),
                  seed=212, n_eps=12, T=60, n_steps=50, goal_space="simplified_quadruped", full_state=False,
                  checksum_steps=(1, 10, 50))


def bench_config_curve_fixture(R):
    """configs[1] exactly as bench.py runs it -- walker dims, batch 1024 -- for 50 free-running reference steps (metric dict
    per step, parameter checksums at steps 10 / 32 / 50): the GPU test feeds the same draws through the 32-step PIPELINED
    graph (fbhip_update_many_injected), i.e. the code path the headline number times.  And configs[2] at its own batch size:
    quadruped + goal space, batch 2048, z_dim 100, 3 steps."""
    trace_fixture(R, "walker_b1024_50", fo.OracleConfig(obs_dim=24, action_dim=6, goal_dim=24, z_dim=50, batch_size=1024),
                  seed=221, n_eps=20, T=100, n_steps=50, full_state=False, checksum_steps=(10, 32, 50))
    trace_fixture(R, "quadruped_goal_b2048",
                  fo.OracleConfig(obs_dim=78, action_dim=12, goal_dim=2, z_dim=100, batch_size=2048, use_goal=True),
                  seed=222, n_eps=12, T=60, n_steps=3, goal_space="simplified_quadruped", full_state=False,
                  checksum_steps=(1, 3))


def sampler_fixture(R):
    """ReplayBuffer.sample KAT: variable lengths + goal + stored meta z; real numpy RNG, fixed seed."""
    rng = np.random.default_rng(7)
    n_eps, T, o, a, g, d = 6, 9, 4, 2, 3, 5
    lengths = np.array([9, 5, 7, 9, 3, 8], np.int32)
    storage, lengths = fo.synthetic_storage(rng, n_eps, T, o, a, g, lengths)
    meta_z = rng.standard_normal((n_eps, T + 1, d)).astype(np.float32)
    rb = fill_ref_buffer(R, storage, lengths, 0.98, future=1.0, max_len=T + 1, meta_z=meta_z)
    B = 64
    rec = {}
    o_choice, o_randint = np.random.choice, np.random.randint

    def choice(*a_, **k):
        rec["ep"] = o_choice(*a_, **k)
        return rec["ep"]

    def randint(*a_, **k):
        rec["step0"] = o_randint(*a_, **k)
        return rec["step0"]
    np.random.seed(123)
    np.random.choice, np.random.randint = choice, randint
    try:
        b = rb.sample(B)
    finally:
        np.random.choice, np.random.randint = o_choice, o_randint
    arrays = {f"storage/{k}": v for k, v in storage.items()}
    arrays.update(lengths=lengths, meta_z=meta_z, ep_idx=rec["ep"], step_idx=rec["step0"] + 1,
                  obs=b.obs, action=b.action, next_obs=b.next_obs, reward=b.reward, discount=b.discount,
                  goal=b.goal, next_goal=b.next_goal, meta_z_out=b.meta["z"])
    # the buffer's own bookkeeping (in_memory_replay_buffer.py:127-133, 135-137)
    arrays.update(rb_len=np.int64(len(rb)), rb_full=np.bool_(rb._full), rb_idx=np.int64(rb._idx),
                  avg_episode_length=np.int64(rb.avg_episode_length), rb_episodes_length=rb._episodes_length)
    np.savez_compressed(HERE / "sampler_kat.npz", **arrays)
    print("[sampler_kat] ok", b.obs.shape, "fixed_len:", rb._is_fixed_episode_length)


def init_fixture(R):
    """Reference constructor under torch.manual_seed: pins net construction order + orthogonal init
    (fb_ddpg.py:119-141, utils.py:81-93).  Valid for the torch build it was generated with."""
    cfg = tiny_cfg()
    torch.manual_seed(1)
    agent = make_ref_agent(R, cfg)
    arrays = {k: v for k, v in ref_state(agent).items()}
    arrays["torch_version"] = np.array(torch.__version__)
    np.savez_compressed(HERE / "init_seed1_tiny.npz", **arrays)
    big = fo.OracleConfig(obs_dim=24, action_dim=6, goal_dim=24, batch_size=256)
    torch.manual_seed(1)
    agent = make_ref_agent(R, big)
    (HERE / "init_seed1_walker.json").write_text(json.dumps(
        {"torch_version": torch.__version__, "checksums": checksums(ref_state(agent))}, indent=1))
    torch.manual_seed(1)       # the sibling's constructor (discrete_fb.py:131-150): forward_net, backward_net, the two targets
    agent = make_ref_agent(R, tiny_cfg(action_dim=4, preprocess=False), discrete=True)
    arrays = {k: v for k, v in ref_state(agent).items()}
    arrays["torch_version"] = np.array(torch.__version__)
    np.savez_compressed(HERE / "init_seed1_tiny_discrete.npz", **arrays)
    print("[init] ok")


def inference_fixture(R):
    """act (eval) / infer_meta_from_obs_and_rewards / get_goal_meta / compute_z_correl on fixed weights
    (fb_ddpg.py:177-222, 258-289)."""
    cfg = tiny_cfg()
    rng = np.random.default_rng(11)
    nets = {n: fo.synthetic_params(rng, fo.NET_SHAPES[n](cfg)) for n in ("actor", "forward_net", "backward_net")}
    agent = make_ref_agent(R, cfg)
    load_nets(agent, nets)
    obs = rng.standard_normal((7, cfg.obs_dim)).astype(np.float32)
    zs = fo.sample_z_from_gauss(torch.from_numpy(rng.standard_normal((7, cfg.z_dim)).astype(np.float32)), cfg.z_dim).numpy()
    with torch.no_grad():       # callers wrap act() in no_grad + eval_mode (pretrain.py:628-632)
        acts = np.stack([agent.act(obs[i], {"z": zs[i]}, 0, eval_mode=True) for i in range(7)])
    rew = rng.uniform(0, 1, (40, 1)).astype(np.float32)
    gobs = rng.standard_normal((40, cfg.obs_dim)).astype(np.float32)
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        zinf = agent.infer_meta_from_obs_and_rewards(torch.from_numpy(gobs), torch.from_numpy(rew))["z"]
    zgoal = agent.get_goal_meta(gobs[0])["z"]
    ts = types.SimpleNamespace(observation=obs[0], goal=None)
    correl = agent.compute_z_correl(ts, {"z": zs[0]})
    with torch.no_grad():
        Bout = agent.backward_net(torch.from_numpy(gobs)).numpy()
    arrays = {f"init/{n}/{k}": v.numpy() for n, p in nets.items() for k, v in p.items()}
    arrays.update(obs=obs, z=zs, act_eval=acts, reward=rew, goal_obs=gobs, z_inferred=zinf, z_goal=zgoal,
                  z_correl=np.float64(correl), backward_out=Bout)
    np.savez_compressed(HERE / "inference_kat.npz", **arrays)
    print("[inference_kat] ok")


def checkpoint_fixture(R):
    """A checkpoint exactly as ``Workspace.save_checkpoint`` writes it (pretrain.py:437-449): ``torch.save`` of
    ``{'agent', 'global_step', 'global_episode', 'replay_loader'}`` holding LIVE reference objects (an ``FBDDPGAgent``
    after two updates, a filled ``ReplayBuffer``), plus a bare ``torch.save(replay_loader)`` like
    ``train_offline.py:88-90``.  The ``.pt`` files are data (pickled tensors / arrays + class PATHS, no code);
    ``ref_checkpoint_expect.npz`` holds what a reader must recover from them."""
    cfg = tiny_cfg()
    rng = np.random.default_rng(31)
    nets = {n: fo.synthetic_params(rng, fo.NET_SHAPES[n](cfg)) for n in ("actor", "forward_net", "backward_net")}
    storage, lengths = fo.synthetic_storage(rng, 6, 12, cfg.obs_dim, cfg.action_dim, None, None)
    agent = make_ref_agent(R, cfg)
    load_nets(agent, nets)
    rb = fill_ref_buffer(R, storage, lengths, cfg.discount)
    for s in range(2):
        with inject(R, fo.make_draws(rng, cfg, 6, lengths), False):
            agent.update(rb, s)
    payload = {"agent": agent, "global_step": 7, "global_episode": 3, "replay_loader": rb}
    with (HERE / "ref_checkpoint_tiny.pt").open("wb") as f:
        torch.save(payload, f, pickle_protocol=4)
    with (HERE / "ref_replay_tiny.pt").open("wb") as f:
        torch.save(rb, f)
    obs = rng.standard_normal((5, cfg.obs_dim)).astype(np.float32)
    zs = fo.sample_z_from_gauss(torch.from_numpy(rng.standard_normal((5, cfg.z_dim)).astype(np.float32)), cfg.z_dim).numpy()
    with torch.no_grad():
        acts = np.stack([agent.act(obs[i], {"z": zs[i]}, 0, eval_mode=True) for i in range(5)])
    arrays = {f"state/{k}": v for k, v in ref_state(agent).items()}
    arrays.update({f"storage/{k}": np.asarray(v) for k, v in rb._storage.items()})
    arrays.update(lengths=np.asarray(rb._episodes_length), obs=obs, z=zs, act_eval=acts,
                  fb_steps=np.int64(2), actor_steps=np.int64(2))
    np.savez_compressed(HERE / "ref_checkpoint_expect.npz", **arrays)
    (HERE / "ref_checkpoint_expect.json").write_text(json.dumps(
        {"cfg": {k: getattr(cfg, k) for k in cfg.__dataclass_fields__}, "global_step": 7, "global_episode": 3,
         "agent_cfg": {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(agent.cfg).items()
                       if isinstance(v, (int, float, str, bool, tuple, type(None)))}}, indent=1))
    # the sibling agent's checkpoint (a pickled DiscreteFBAgent after two updates)
    cfg = tiny_cfg(action_dim=4, preprocess=False)
    rng = np.random.default_rng(32)
    nets = {n: fo.synthetic_params(rng, do.NET_SHAPES[n](cfg)) for n in do.NET_SHAPES}
    storage, lengths = fo.synthetic_storage(rng, 6, 12, cfg.obs_dim, cfg.action_dim, None, None)
    do.synthetic_actions(rng, storage, cfg.action_dim)
    agent = make_ref_agent(R, cfg, discrete=True)
    load_nets(agent, nets)
    rb = fill_ref_buffer(R, storage, lengths, cfg.discount)
    for s in range(2):
        with inject(R, fo.make_draws(rng, cfg, 6, lengths), False, action_noise=False):
            agent.update(rb, s)
    with (HERE / "ref_checkpoint_tiny_discrete.pt").open("wb") as f:
        torch.save({"agent": agent, "global_step": 5, "global_episode": 2, "replay_loader": rb}, f, pickle_protocol=4)
    obs = rng.standard_normal((9, cfg.obs_dim)).astype(np.float32)
    zs = fo.sample_z_from_gauss(torch.from_numpy(rng.standard_normal((9, cfg.z_dim)).astype(np.float32)), cfg.z_dim).numpy()
    with torch.no_grad():
        acts = np.array([agent.act(obs[i], {"z": zs[i]}, 0, eval_mode=True) for i in range(9)], np.int64)
    arrays = {f"state/{k}": v for k, v in ref_state(agent).items()}
    arrays.update(obs=obs, z=zs, act_eval=acts, fb_steps=np.int64(2))
    np.savez_compressed(HERE / "ref_checkpoint_discrete_expect.npz", **arrays)
    print("[ref_checkpoint] ok")


def hydra_checkpoint_fixture(R):
    """The checkpoint a HYDRA-launched reference run writes.  ``make_agent`` (pretrain.py:112-120) calls
    ``hydra.utils.instantiate(cfg.agent)`` (``_convert_`` = "none"), so the list-valued kwargs reach ``FBDDPGAgentConfig`` as
    ``omegaconf.ListConfig`` objects and are pickled as such inside ``agent.cfg`` (``obs_shape``, ``action_shape``,
    ``log_std_bounds``).  omegaconf is not in this image, so the containers here are stand-ins registered under omegaconf's
    class paths (``omegaconf.listconfig.ListConfig``, ``omegaconf.nodes.AnyNode``, ``omegaconf.base.ContainerMetadata`` /
    ``Metadata``) whose pickled STATE has the shape omegaconf 2.1-2.3 produces (``BaseContainer.__getstate__``: the
    instance dict without ``_flags_cache`` -> ``_metadata``, ``_parent``, ``_content`` = list of value nodes;
    ``Node.__getstate__``: ``_metadata``, ``_parent``, ``_val``) -- restated from omegaconf's source, not executed from it.
    Everything else in the file (agent, nets, Adam state) is the live reference, as in ``checkpoint_fixture``."""
    def mod(name):
        m = sys.modules.get(name) or types.ModuleType(name)
        sys.modules[name] = m
        return m

    base, nodes, listconfig = mod("omegaconf.base"), mod("omegaconf.nodes"), mod("omegaconf.listconfig")

    def cls(m, name, **ns):
        c = type(name, (), dict(ns, __module__=m.__name__, __qualname__=name))
        setattr(m, name, c)
        return c

    Metadata = cls(base, "Metadata")
    ContainerMetadata = cls(base, "ContainerMetadata")

    def meta(kind, **kw):
        m = kind()
        m.__dict__.update(dict(ref_type=None, object_type=None, optional=True, key=None, flags=None, flags_root=False,
                               resolver_cache={}), **kw)
        return m

    AnyNode = cls(nodes, "AnyNode")
    ListConfig = cls(listconfig, "ListConfig",
                     __len__=lambda self: len(self._content), __iter__=lambda self: (n._val for n in self._content),
                     __getitem__=lambda self, i: self._content[i]._val,
                     __getstate__=lambda self: {k: v for k, v in self.__dict__.items() if k != "_flags_cache"})

    def lc(values, key):
        out = ListConfig()
        out.__dict__.update(_metadata=meta(ContainerMetadata, key=key, element_type=None), _parent=None, _flags_cache=None,
                            _content=[])
        for i, v in enumerate(values):
            n = AnyNode()
            n.__dict__.update(_metadata=meta(Metadata, key=i), _parent=out, _val=v)
            out._content.append(n)
        return out

    cfg = tiny_cfg()
    rng = np.random.default_rng(33)
    nets = {n: fo.synthetic_params(rng, fo.NET_SHAPES[n](cfg)) for n in ("actor", "forward_net", "backward_net")}
    storage, lengths = fo.synthetic_storage(rng, 6, 12, cfg.obs_dim, cfg.action_dim, None, None)
    agent = make_ref_agent(R, cfg)
    load_nets(agent, nets)
    rb = fill_ref_buffer(R, storage, lengths, cfg.discount)
    with inject(R, fo.make_draws(rng, cfg, 6, lengths), False):
        agent.update(rb, 0)
    agent.cfg.obs_shape = lc([cfg.obs_dim], "obs_shape")
    agent.cfg.action_shape = lc([cfg.action_dim], "action_shape")
    agent.cfg.log_std_bounds = lc([-5, 2], "log_std_bounds")
    with (HERE / "ref_checkpoint_hydra_tiny.pt").open("wb") as f:
        torch.save({"agent": agent, "global_step": 1, "global_episode": 1}, f, pickle_protocol=4)
    arrays = {f"state/{k}": v for k, v in ref_state(agent).items()}
    np.savez_compressed(HERE / "ref_checkpoint_hydra_expect.npz", **arrays)
    print("[ref_checkpoint_hydra] ok")


def tiny_cfg(**kw):
    base = dict(obs_dim=5, action_dim=3, goal_dim=5, z_dim=8, hidden_dim=32, feature_dim=16,
                backward_hidden_dim=18, batch_size=16, lr=1e-3)
    base.update(kw)
    return fo.OracleConfig(**base)


def main():
    R = import_reference()
    torch.manual_seed(0)
    np.random.seed(0)
    trace_fixture(R, "tiny_trace", tiny_cfg(), seed=101, n_eps=6, T=12, n_steps=5)
    trace_fixture(R, "tiny_goal_trace", tiny_cfg(goal_dim=3, use_goal=True, q_loss=True, lr_coef=0.5, z_dim=10,
                                                 backward_hidden_dim=22, batch_size=24),
                  seed=102, n_eps=7, T=11, n_steps=4, goal_space="simplified_walker", variable_len=True)
    future_fixtures(R)
    nonorm_fixture(R)
    randweight_fixture(R)
    trunk_fixture(R)
    single_trunk_fixture(R)
    boltzmann_fixture(R)
    discrete_fixture(R)
    sf_fixture(R)
    sf_init_fixture(R)
    sf_more_fixture(R)
    sf_svdp_fixture(R)
    sf_latent_fixture(R)
    sf_svdsr_fixture(R)
    sf_svdsrv2_fixture(R)
    sf_contrastive_fixture(R)
    sf_contrastivev2_fixture(R)
    sf_identity_fixture(R)
    sf_mix_fixture(R)
    debug_fixture(R)
    walker = dict(obs_dim=24, action_dim=6, goal_dim=24, z_dim=50)
    trace_fixture(R, "walker_b256", fo.OracleConfig(batch_size=256, **walker), seed=201, n_eps=20, T=100,
                  n_steps=10, full_state=False, checksum_steps=(1, 5, 10))
    trace_fixture(R, "walker_b1024", fo.OracleConfig(batch_size=1024, **walker), seed=202, n_eps=20, T=100,
                  n_steps=3, full_state=False, checksum_steps=(1, 3))
    trace_fixture(R, "quadruped_goal_b512",
                  fo.OracleConfig(obs_dim=78, action_dim=12, goal_dim=2, z_dim=100, batch_size=512, use_goal=True),
                  seed=203, n_eps=12, T=60, n_steps=3, goal_space="simplified_quadruped", full_state=False,
                  checksum_steps=(1, 3))
    long_curve_fixtures(R)
    bench_config_curve_fixture(R)
    sampler_fixture(R)
    init_fixture(R)
    inference_fixture(R)
    checkpoint_fixture(R)
    hydra_checkpoint_fixture(R)


if __name__ == "__main__":
    if len(sys.argv) > 1:                      # e.g. ``make_golden.py checkpoint_fixture``: regenerate one fixture only
        R_ = import_reference()
        torch.manual_seed(0)
        np.random.seed(0)
        for fn in sys.argv[1:]:
            globals()[fn](R_)
    else:
        main()
