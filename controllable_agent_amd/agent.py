"""FBHipAgent: the reference's FB-DDPG agent surface on top of libfbhip.so (MI355X / gfx950).

Mirrors ``url_benchmark/agent/fb_ddpg.py`` -- ``FBDDPGAgentConfig`` (:37-82) and ``FBDDPGAgent`` (:89-520):
same constructor kwargs (so a hydra ``_target_`` pointing here works unchanged), same methods
(``train, init_from, init_meta, update_meta, act, update, sample_z, infer_meta,
infer_meta_from_obs_and_rewards, get_goal_meta, compute_z_correl``), same metric keys and gating.

All state lives in flat fp32 device buffers (params / grads / Adam m,v / targets) owned by torch; the nets
are exposed as ``NetView`` objects (``state_dict()``, ``parameters()``, ``load_state_dict()``) whose tensors
are strided views into those buffers, so ``init_from``, pickling and checkpoint code keep working.  Every
arithmetic step of ``update`` / ``act`` / ``infer_meta`` runs in hand-written HIP kernels behind the C ABI
(include/fbhip.h); there is no eager-PyTorch or CPU fallback.
"""
from __future__ import annotations

import collections
import copy
import ctypes as C
import dataclasses
import logging
import math
import os
import re
import operator
import typing as tp
import weakref

import numpy as np
import torch
import torch.distributed as _dist

from . import _lib
from ._lib import Dims, HParams, Inject, TensorDesc, check, ptr, stream_ptr
from .replay import DeviceReplayBuffer, EpisodeBatch

logger = logging.getLogger(__name__)
MetaDict = tp.Mapping[str, np.ndarray]
MISSING: tp.Any = "???"

# goals.get_goal_space_dim (goals.py:218-221) instantiates a MuJoCo env just to read a vector length; the
# lengths are fixed by the goal-space functions (goals.py:54-112), restated here.
GOAL_SPACE_DIMS: tp.Dict[str, int] = {
    "simplified_jaco": 3, "simplified_point_mass_maze": 2, "simplified_walker": 3, "walker_pos_speed": 4,
    "walker_pos_speed_z": 6, "simplified_quadruped": 2, "quad_pos_speed": 7,
}


def register_goal_space(name: str, dim: int) -> None:
    GOAL_SPACE_DIMS[name] = int(dim)


def get_goal_space_dim(name: str) -> int:
    if name not in GOAL_SPACE_DIMS:
        raise KeyError(f"unknown goal_space {name!r}: call controllable_agent_amd.agent.register_goal_space(name, dim)")
    return GOAL_SPACE_DIMS[name]


def schedule(schdl: tp.Union[str, float], step: int) -> float:
    """``utils.schedule`` (utils.py:235-255): a constant, ``linear(init,final,T)`` or
    ``step_linear(init,final1,T1,final2,T2)``.  All three are piecewise-linear in ``step`` and constant outside their
    knots, so one clamped interpolation evaluates them."""
    try:
        return float(schdl)
    except (TypeError, ValueError):
        pass
    m = re.fullmatch(r"\s*(linear|step_linear)\((.*)\)\s*", str(schdl))
    if m is not None:
        v = [float(x) for x in m.group(2).split(",")]
        if m.group(1) == "linear" and len(v) == 3:
            xs, ys = [0.0, v[2]], [v[0], v[1]]
        elif m.group(1) == "step_linear" and len(v) == 5:
            xs, ys = [0.0, v[2], v[2] + v[4]], [v[0], v[1], v[3]]
        else:
            raise NotImplementedError(schdl)
        return float(np.interp(float(step), xs, ys))
    raise NotImplementedError(schdl)


@dataclasses.dataclass
class FBDDPGAgentConfig:
    """Field-for-field mirror of fb_ddpg.py:37-82 (omegaconf interpolations become plain required fields)."""
    _target_: str = "controllable_agent_amd.agent.FBHipAgent"
    name: str = "fb_ddpg"
    obs_type: str = MISSING
    obs_shape: tp.Tuple[int, ...] = MISSING
    action_shape: tp.Tuple[int, ...] = MISSING
    device: str = "cuda"
    lr: float = 1e-4
    lr_coef: float = 1
    fb_target_tau: float = 0.01
    update_every_steps: int = 2
    use_tb: bool = False
    use_wandb: bool = False
    use_hiplog: bool = False
    num_expl_steps: int = MISSING
    num_inference_steps: int = 5120
    hidden_dim: int = 1024
    backward_hidden_dim: int = 526
    feature_dim: int = 512
    z_dim: int = 50
    stddev_schedule: str = "0.2"
    stddev_clip: float = 0.3
    update_z_every_step: int = 300
    update_z_proba: float = 1.0
    nstep: int = 1
    batch_size: int = 1024
    init_fb: bool = True
    update_encoder: bool = True
    goal_space: tp.Optional[str] = None
    ortho_coef: float = 1.0
    log_std_bounds: tp.Tuple[float, float] = (-5, 2)
    temp: float = 1
    boltzmann: bool = False
    debug: bool = False
    future_ratio: float = 0.0
    mix_ratio: float = 0.5
    rand_weight: bool = False
    preprocess: bool = True
    norm_z: bool = True
    q_loss: bool = False
    q_loss_coef: float = 0.01
    additional_metric: bool = False
    add_trunk: bool = False
    # --- not a reference field: data-parallel loss semantics (distributed.py).  False: every rank's own B x B block with
    # gradient averaging (mode A).  True: the exact loss of the concatenated world*B batch (mode B, one all-gather more)
    dp_global_batch: bool = False


# the reference's Linear-layer construction order per net (fb_modules.py:91-105, 165-182, 220); each entry is
# (state_dict prefix, in_features, out_features) -- drives an RNG-stream-identical orthogonal init
def _linear_order(net: str, o: int, a: int, g: int, d: int, H: int, Fd: int, Hb: int, add_trunk: bool = False,
                  preprocess: bool = True, boltzmann: bool = False, discrete: bool = False):
    """(name, in, out) of every nn.Linear in module-construction order (fb_modules.py:91-105, 138, 165-182, 220)."""
    if discrete and net == "forward_net":              # discrete_fb.ForwardMap, preprocess=False (discrete_fb.py:74-83)
        return [("trunk.0", o + d, H), ("trunk.3", H, H), ("trunk.5", H, H),
                ("F1.0", H, H), ("F1.2", H, d * a), ("F2.0", H, H), ("F2.2", H, d * a)]
    if boltzmann and net == "actor":                   # DiagGaussianActor: mlp(o + d, H, "ntanh", H, "relu", 2a)
        return [("policy.0", o + d, H), ("policy.3", H, H), ("policy.5", H, 2 * a)]
    if not preprocess and net != "backward_net":       # one trunk on the concatenated input (fb_modules.py:99-103, 174-178)
        if net == "actor":
            return [("trunk.0", o + d, H), ("trunk.3", H, H), ("trunk.5", H, H), ("policy.0", H, H), ("policy.2", H, a)]
        return [("trunk.0", o + d + a, H), ("trunk.3", H, H), ("trunk.5", H, H),
                ("F1.0", H, H), ("F1.2", H, d), ("F2.0", H, H), ("F2.2", H, d)]
    feat = H if add_trunk else 2 * Fd
    trunk = [("trunk.0", 2 * Fd, H)] if add_trunk else []
    if net == "actor":
        return [("obs_net.0", o, H), ("obs_net.3", H, Fd), ("obs_z_net.0", o + d, H), ("obs_z_net.3", H, Fd)] + trunk + \
               [("policy.0", feat, H), ("policy.2", H, a)]
    if net == "forward_net":
        return [("obs_action_net.0", o + a, H), ("obs_action_net.3", H, Fd), ("obs_z_net.0", o + d, H),
                ("obs_z_net.3", H, Fd)] + trunk + [("F1.0", feat, H), ("F1.2", H, d), ("F2.0", feat, H), ("F2.2", H, d)]
    return [("B.0", g, Hb), ("B.3", Hb, Hb), ("B.5", Hb, d)]


class NetView:
    """A network stored inside a flat device buffer; quacks like the ``nn.Module`` the reference code expects
    for parameter access (``state_dict / load_state_dict / parameters / named_parameters / train``)."""

    def __init__(self, name: str, flat: torch.Tensor, layout: tp.List[TensorDesc],
                 forward: tp.Optional[tp.Callable[[torch.Tensor], torch.Tensor]] = None,
                 sync: tp.Optional[tp.Callable[[], None]] = None) -> None:
        self._name = name
        self._flat = flat
        self._forward = forward
        # called before any access to the tensors: the agent launches the update() calls it still holds back (FBHipAgent.flush)
        self._sync = sync if sync is not None else (lambda: None)
        self.training = True
        self._views: "collections.OrderedDict[str, torch.Tensor]" = collections.OrderedDict()
        self._pads: tp.List[torch.Tensor] = []        # the alignment columns right of every matrix (must stay zero)
        for t in layout:
            n = t.name.decode()
            full = flat[t.offset:t.offset + t.rows * t.ld].view(t.rows, t.ld)
            seg = full[:, :t.cols]
            if t.ld > t.cols:
                self._pads.append(full[:, t.cols:])
            self._views[n] = seg[0] if (n.endswith("bias") or ".1." in n) else seg      # vectors: 1-D views

    def state_dict(self) -> "collections.OrderedDict[str, torch.Tensor]":
        self._sync()
        return collections.OrderedDict(self._views)

    def pad_abs_max(self) -> float:
        """max |x| over the alignment columns.  The kernels run GEMMs over padded widths and rely on these being zero:
        anything else means a panel leaked foreign columns into a weight gradient."""
        self._sync()
        return max([float(p.abs().max()) for p in self._pads], default=0.0)

    def named_parameters(self) -> tp.Iterator[tp.Tuple[str, torch.Tensor]]:
        self._sync()
        return iter(self._views.items())

    def parameters(self) -> tp.Iterator[torch.Tensor]:
        self._sync()
        return iter(self._views.values())

    def load_state_dict(self, sd: tp.Mapping[str, tp.Any], strict: bool = True) -> None:
        missing = set(self._views) - set(sd)
        extra = set(sd) - set(self._views)
        if strict and (missing or extra):
            raise KeyError(f"{self._name}: missing {sorted(missing)}, unexpected {sorted(extra)}")
        self._sync()
        with torch.no_grad():
            for k, v in self._views.items():
                if k in sd:
                    v.copy_(torch.as_tensor(sd[k], dtype=torch.float32))

    def train(self, mode: bool = True) -> "NetView":
        self.training = mode
        return self

    def to(self, *a: tp.Any, **k: tp.Any) -> "NetView":
        return self

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if self._forward is None:
            raise TypeError(f"{self._name} is not directly callable; use the agent's methods")
        return self._forward(x)


class AdamView:
    """torch.optim.Adam-shaped handle (``state_dict / load_state_dict / param_groups``) on the flat m / v buffers."""

    def __init__(self, agent: "FBHipAgent", which: str, nets: tp.List[str], lrs: tp.List[float]) -> None:
        # a WEAK reference: agent -> optimiser view -> agent was a reference cycle, so a dropped agent (and its native context:
        # graph execs with their runtime-internal streams, events, pinned staging) lived on until some later cyclic-GC pass --
        # dozens of dead contexts at a time in a process that builds many agents (round-3 lifecycle stress)
        self._agent_ref, self._which, self._nets = weakref.ref(agent), which, nets
        self.param_groups = [dict(lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False) for lr in lrs]

    @property
    def _agent(self) -> "FBHipAgent":
        a = self._agent_ref()
        if a is None:
            raise ReferenceError("the agent this optimiser view belongs to no longer exists")
        return a

    def _mv(self) -> tp.List[tp.Tuple[torch.Tensor, torch.Tensor]]:
        out = []
        for n in self._nets:
            for (_, m), (_, v) in zip(self._agent._adam_views[n]["m"].items(), self._agent._adam_views[n]["v"].items()):
                out.append((m, v))
        return out

    def state_dict(self) -> tp.Dict[str, tp.Any]:
        fb, ac = self._agent.step_counts()
        t = fb if self._which == "fb" else ac
        state = {}
        if t > 0:
            for i, (m, v) in enumerate(self._mv()):
                state[i] = {"step": torch.tensor(float(t)), "exp_avg": m.detach().cpu().clone(),
                            "exp_avg_sq": v.detach().cpu().clone()}
        groups, i0 = [], 0
        for n, g in zip(self._nets, self.param_groups):
            k = len(self._agent._adam_views[n]["m"])
            groups.append(dict(g, params=list(range(i0, i0 + k))))
            i0 += k
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd: tp.Mapping[str, tp.Any]) -> None:
        state = sd["state"]
        t = 0
        with torch.no_grad():
            for i, (m, v) in enumerate(self._mv()):
                if i in state:
                    m.copy_(torch.as_tensor(state[i]["exp_avg"], dtype=torch.float32))
                    v.copy_(torch.as_tensor(state[i]["exp_avg_sq"], dtype=torch.float32))
                    t = int(float(state[i]["step"]))
                else:
                    m.zero_(), v.zero_()
        fb, ac = self._agent.step_counts()
        self._agent.set_step_counts(*( (t, ac) if self._which == "fb" else (fb, t) ))
        for g, src in zip(self.param_groups, sd.get("param_groups", [])):
            g["lr"] = src.get("lr", g["lr"])


class FBHipAgent:
    _config_cls: tp.Any = FBDDPGAgentConfig
    _discrete = False               # DiscreteFBHipAgent: no actor, [B, d, A] ForwardMap heads (discrete_fb.py)

    # pylint: disable=unused-argument
    def __init__(self, **kwargs: tp.Any) -> None:
        cfg = self._config_cls(**kwargs)
        self.cfg = cfg
        for f in ("obs_type", "obs_shape", "action_shape", "num_expl_steps"):
            if getattr(cfg, f) is MISSING or getattr(cfg, f) == "???":
                raise ValueError(f"FBHipAgent: missing required config field {f!r}")
        # (``nstep`` is accepted and ignored like in the reference: neither FBDDPGAgent nor the in-memory ReplayBuffer reads
        # it -- only the file-based loader of url_benchmark/replay_buffer.py:182-259 did)
        # cfg.debug (fb_ddpg.py:128-130, discrete_fb.py:134-136): IdentityMap backward nets.  SFAgent declares the field and never
        # reads it (sf.py:70, 436): ignored there as in the reference
        unsupported = {"obs_type": cfg.obs_type == "pixels"}
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(f"FBHipAgent: non-default options not implemented in the HIP path yet: {bad}")
        assert len(cfg.action_shape) == 1
        self.action_dim = int(cfg.action_shape[0])
        self.obs_dim = int(cfg.obs_shape[0])
        self.solved_meta: tp.Any = None
        self.actor_success: tp.List[float] = []
        goal_dim = self.obs_dim
        if cfg.goal_space is not None:
            goal_dim = get_goal_space_dim(cfg.goal_space)
        self.goal_dim = goal_dim
        self._identity_b = bool(cfg.debug) and not self._sf_mode
        if self._identity_b:
            if goal_dim != cfg.z_dim:               # (the reference fails in update_fb: F [B, z_dim] x B(goal)^T [goal_dim, B])
                raise ValueError(f"debug=True makes the backward map the identity: z_dim ({cfg.z_dim}) must equal the goal dimension ({goal_dim})")
        if cfg.feature_dim < self.obs_dim:
            logger.warning(f"feature_dim {cfg.feature_dim} should not be smaller that obs_dim {self.obs_dim}")
        if cfg.z_dim < goal_dim:
            logger.warning(f"z_dim {cfg.z_dim} should not be smaller that goal_dim {goal_dim}")
        self.training = True
        self._device = self._resolve_device(cfg.device)
        self._dims = self._make_dims()
        self._ctx: tp.Optional[C.c_void_p] = None
        self._replay_token: tp.Optional[tp.Tuple[int, int]] = None
        self._ext_replay: tp.Optional[DeviceReplayBuffer] = None
        self._use_graph = True
        self._seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        self._allocate(self._reference_init())
        self.train()

    # ------------------------------------------------------------------ construction
    _sf_mode = 0                    # SFHipAgent: 1 icm / 2 lap (fbhip_dims.sf)

    def _make_dims(self) -> Dims:
        cfg = self.cfg
        return Dims(cfg.batch_size, self.obs_dim, self.action_dim, self.goal_dim, cfg.z_dim, cfg.hidden_dim, cfg.feature_dim,
                    cfg.backward_hidden_dim, int(cfg.goal_space is not None), int(bool(cfg.add_trunk)), int(bool(cfg.preprocess)),
                    int(bool(getattr(cfg, "norm_z", True))), int(bool(cfg.boltzmann)), int(self._discrete), int(self._sf_mode),
                    int(getattr(self, "_identity_b", False)))

    @staticmethod
    def _resolve_device(device: tp.Any) -> torch.device:
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError(f"FBHipAgent runs on an MI355X only (device={device!r}); there is no CPU fallback")
        _lib.require_device()
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        return dev

    def _reference_init(self) -> tp.Dict[str, tp.Dict[str, torch.Tensor]]:
        """Initial weights with the SAME torch RNG consumption as the reference constructor (fb_ddpg.py:119-141:
        actor, forward_net, backward_net, backward_target_net, forward_target_net; each = nn.Linear default init
        followed by utils.weight_init: orthogonal weights, zero bias, LayerNorm (1, 0); utils.py:81-93)."""
        c = self.cfg
        dims = (self.obs_dim, self.action_dim, self.goal_dim, c.z_dim, c.hidden_dim, c.feature_dim, c.backward_hidden_dim)

        def build(net: str) -> tp.Dict[str, torch.Tensor]:
            lins = [(p, torch.nn.Linear(i, o)) for p, i, o in _linear_order(net, *dims, add_trunk=bool(c.add_trunk),
                                                                               preprocess=bool(c.preprocess),
                                                                               boltzmann=bool(c.boltzmann),
                                                                               discrete=self._discrete)]
            sd: tp.Dict[str, torch.Tensor] = {}
            for p, lin in lins:
                torch.nn.init.orthogonal_(lin.weight.data)
                sd[f"{p}.weight"] = lin.weight.data
                sd[f"{p}.bias"] = torch.zeros_like(lin.bias.data)
                single = (not c.preprocess or self._discrete) and net != "backward_net"
                no_ln = ("F1", "F2") + (() if single else ("trunk",)) + (() if (c.boltzmann and net == "actor") else ("policy",))
                if p.endswith(".0") and not p.startswith(no_ln):                              # LayerNorm next
                    pre = p[:-2]
                    sd[f"{pre}.1.weight"] = torch.ones(lin.out_features)
                    sd[f"{pre}.1.bias"] = torch.zeros(lin.out_features)
            return sd

        if self._discrete:        # discrete_fb.py:131-147: forward_net, backward_net, backward_target_net, forward_target_net
            nets = {"forward_net": build("forward_net"), "backward_net": {} if self._identity_b else build("backward_net")}
        elif self._identity_b:    # cfg.debug: two IdentityMap()s, no parameters, no RNG draws (fb_ddpg.py:128-130)
            nets = {"actor": build("actor"), "forward_net": build("forward_net"), "backward_net": {}}
        else:
            nets = {"actor": build("actor"), "forward_net": build("forward_net"), "backward_net": build("backward_net")}
        if not self._identity_b:
            build("backward_net") # backward_target_net: constructed (RNG consumed), then overwritten by a copy
        build("forward_net")      # forward_target_net
        return nets

    def _allocate(self, nets: tp.Optional[tp.Dict[str, tp.Dict[str, torch.Tensor]]]) -> None:
        _lib.require_device()
        lib = _lib.load()
        d, dev = self._dims, self._device
        torch.cuda.set_device(dev)
        self._stream = torch.cuda.Stream(device=dev)      # hipGraph capture is illegal on the legacy default stream
        self._numel = [lib.fbhip_net_numel(C.byref(d), n) for n in range(3)]
        if min(self._numel) < 0:
            raise ValueError(_lib.last_error())
        nfb = self._numel[0] + self._numel[1]
        z = lambda n: torch.zeros(n, device=dev, dtype=torch.float32)
        self._fb_params, self._fb_grads, self._fb_m, self._fb_v, self._fb_targets = z(nfb), z(nfb), z(nfb), z(nfb), z(nfb)
        self._actor_params, self._actor_grads, self._actor_m, self._actor_v = (z(self._numel[2]) for _ in range(4))
        self._workspace = torch.zeros(lib.fbhip_workspace_bytes(C.byref(d)), device=dev, dtype=torch.uint8)
        ctx = C.c_void_p()
        check(lib.fbhip_create(C.byref(d), C.byref(ctx)))
        self._ctx = ctx
        check(lib.fbhip_bind_buffers(ctx, ptr(self._fb_params), ptr(self._fb_grads), ptr(self._fb_m), ptr(self._fb_v),
                                     ptr(self._fb_targets), ptr(self._actor_params), ptr(self._actor_grads),
                                     ptr(self._actor_m), ptr(self._actor_v), ptr(self._workspace),
                                     self._workspace.numel()), ctx)
        check(lib.fbhip_set_seed(ctx, self._seed, self._rank()), ctx)
        if self.cfg.boltzmann:                         # fb_ddpg.py:70-71, 118-120
            lo, hi = self.cfg.log_std_bounds
            check(lib.fbhip_set_policy_squash(ctx, float(self.cfg.temp), float(lo), float(hi)), ctx)

        def layout(net: int) -> tp.List[TensorDesc]:
            out = []
            for i in range(lib.fbhip_layout_count(C.byref(d), net)):
                t = TensorDesc()
                check(lib.fbhip_layout_entry(C.byref(d), net, i, C.byref(t)))
                out.append(t)
            return out

        nf = self._numel[0]
        lay = {n: layout(i) for i, n in enumerate(("forward_net", "backward_net", "actor"))}
        seg = {"forward_net": slice(0, nf), "backward_net": slice(nf, nfb)}
        me = weakref.ref(self)                   # (the views' callables must not keep the agent alive: no reference cycles, see AdamView)

        def sync() -> None:                      # every access to the tensors first launches the update() calls still held back
            a = me()
            if a is not None:
                a.flush()
        self.forward_net = NetView("forward_net", self._fb_params[seg["forward_net"]], lay["forward_net"], sync=sync)
        self.backward_net = NetView("backward_net", self._fb_params[seg["backward_net"]], lay["backward_net"],
                                    forward=lambda x: me()._backward_map(x, target=False), sync=sync)
        self.forward_target_net = NetView("forward_target_net", self._fb_targets[seg["forward_net"]], lay["forward_net"], sync=sync)
        self.backward_target_net = NetView("backward_target_net", self._fb_targets[seg["backward_net"]],
                                           lay["backward_net"], forward=lambda x: me()._backward_map(x, target=True), sync=sync)
        self.actor = NetView("actor", self._actor_params, lay["actor"], sync=sync)
        self.encoder = torch.nn.Identity()      # states only (fb_ddpg.py:108-110)
        self.aug = torch.nn.Identity()
        self._grad_views = {"forward_net": NetView("g", self._fb_grads[seg["forward_net"]], lay["forward_net"], sync=sync),
                            "backward_net": NetView("g", self._fb_grads[seg["backward_net"]], lay["backward_net"], sync=sync),
                            "actor": NetView("g", self._actor_grads, lay["actor"], sync=sync)}
        self._adam_views = {
            "forward_net": {"m": NetView("m", self._fb_m[seg["forward_net"]], lay["forward_net"]).state_dict(),
                            "v": NetView("v", self._fb_v[seg["forward_net"]], lay["forward_net"]).state_dict()},
            "backward_net": {"m": NetView("m", self._fb_m[seg["backward_net"]], lay["backward_net"]).state_dict(),
                             "v": NetView("v", self._fb_v[seg["backward_net"]], lay["backward_net"]).state_dict()},
            "actor": {"m": NetView("m", self._actor_m, lay["actor"]).state_dict(),
                      "v": NetView("v", self._actor_v, lay["actor"]).state_dict()}}
        if getattr(self, "_identity_b", False):
            # cfg.debug: IdentityMap holds no parameters.  The context keeps the block (unused, zero gradients); these views show none of it
            for v in (self.backward_net, self.backward_target_net, self._grad_views["backward_net"]):
                v._views.clear()
            self._adam_views["backward_net"] = {"m": collections.OrderedDict(), "v": collections.OrderedDict()}
        c = self.cfg
        self.encoder_opt = None
        if self._discrete:                       # discrete_fb.py:103-165 has neither
            del self.actor
        else:
            self.actor_opt = AdamView(self, "actor", ["actor"], [c.lr])
        self.fb_opt = AdamView(self, "fb", ["forward_net", "backward_net"], [c.lr, c.lr_coef * c.lr])   # fb_ddpg.py:149-151
        if nets is not None:
            self.load_nets(nets)

    def load_nets(self, nets: tp.Mapping[str, tp.Mapping[str, tp.Any]], copy_targets: bool = True) -> None:
        """Load {net: state_dict} (reference key names); targets start as copies (fb_ddpg.py:140-141)."""
        self.flush()
        self._replicas_verified = False
        for n in (() if self._discrete else ("actor",)) + ("forward_net", "backward_net"):
            if n in nets:
                getattr(self, n).load_state_dict(nets[n])
        if copy_targets:
            self._fb_targets.copy_(self._fb_params)
        for n in ("forward_target_net", "backward_target_net"):
            if n in nets:
                getattr(self, n).load_state_dict(nets[n])

    def _rank(self) -> int:
        import torch.distributed as dist
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    def _world(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def __del__(self) -> None:
        ctx = self.__dict__.get("_ctx")          # (not the flushing attribute: update() calls nobody can observe any more are dropped)
        if ctx:
            try:
                _lib.load().fbhip_destroy(ctx)
            except Exception:       # interpreter shutdown
                pass
            self.__dict__["_ctx"] = None

    # ------------------------------------------------------------------ pickling (pretrain.py:437-449 pickles the agent object)
    def __getstate__(self) -> tp.Dict[str, tp.Any]:
        self.flush()
        fb, ac = self.step_counts()
        flat = {k: getattr(self, k).detach().cpu() for k in ("_fb_params", "_fb_m", "_fb_v", "_fb_targets",
                                                              "_actor_params", "_actor_m", "_actor_v")}
        # (the device RNG counters travel too: a resumed run continues its Philox streams instead of replaying the batches,
        # z draws and exploration noise of the first steps of training)
        # (the flat buffers are the library's PHYSICAL layout -- padded leading dimensions, 32-float matrix alignment --, which has
        # changed between rounds: the pickle says which one it holds)
        return dict(layout_abi=int(_lib.load().fbhip_abi_version()), layout_digest=self._layout_digest(), flat_numel={k: int(v.numel()) for k, v in flat.items()},
                    cfg=dataclasses.asdict(self.cfg), flat=flat, fb_steps=fb, actor_steps=ac, seed=self._seed,
                    rng_counts=self.rng_counts(), solved_meta=self.solved_meta, training=self.training, goal_dim=self.goal_dim)

    def __setstate__(self, st: tp.Dict[str, tp.Any]) -> None:
        cfg = self._config_cls(**st["cfg"])
        self.cfg = cfg
        self.action_dim, self.obs_dim, self.goal_dim = int(cfg.action_shape[0]), int(cfg.obs_shape[0]), st["goal_dim"]
        self.solved_meta, self.actor_success, self.training = st["solved_meta"], [], st["training"]
        self._device = self._resolve_device(cfg.device)
        self._identity_b = bool(cfg.debug) and not self._sf_mode
        self._dims = self._make_dims()
        self._ctx, self._replay_token, self._ext_replay, self._use_graph, self._seed = None, None, None, True, st["seed"]
        self._allocate(None)
        # the flat buffers are the library's PHYSICAL layout: a pickle is only loadable into the layout that wrote it.  The digest covers
        # every tensor's (name, offset, rows, cols, ld); the element counts below stay as the guard for pickles older than the digest
        if "layout_digest" in st and st["layout_digest"] != self._layout_digest():
            raise RuntimeError(
                f"FBHipAgent pickle was written with the flat parameter layout {st['layout_digest']} (library ABI {st.get('layout_abi', '?')}); "
                f"this library (ABI {_lib.load().fbhip_abi_version()}) lays the same nets out as {self._layout_digest()}.  The flat buffers "
                "are the physical, padded layout and are not portable across layouts: re-export the agent with the version that wrote "
                "it through state_dict() of its nets / optimisers (name-based) and load_nets().")
        for k, v in st["flat"].items():
            if getattr(self, k).numel() != v.numel():
                raise RuntimeError(
                    f"FBHipAgent pickle holds {k} with {v.numel()} floats in the flat layout of library ABI {st.get('layout_abi', '< 17')}; "
                    f"this library (ABI {_lib.load().fbhip_abi_version()}) lays the same nets out in {getattr(self, k).numel()}.  The flat "
                    "buffers are the physical, padded layout and are not portable across library versions: re-export the agent "
                    "with the version that wrote it through state_dict() of its nets / optimisers (name-based) and load_nets().")
            getattr(self, k).copy_(v)
        self.set_step_counts(st["fb_steps"], st["actor_steps"])
        if "rng_counts" in st:                   # (pickles of round 1 do not have it: those restart their streams)
            self.set_rng_counts(*st["rng_counts"])

    def _layout_digest(self) -> str:
        """sha1 over every tensor's (name, offset, rows, cols, ld) of the three flat segments: equal digests == same physical layout"""
        import hashlib
        lib, h = _lib.load(), hashlib.sha1()
        for net in range(3):
            for i in range(lib.fbhip_layout_count(C.byref(self._dims), net)):
                t = TensorDesc()
                check(lib.fbhip_layout_entry(C.byref(self._dims), net, i, C.byref(t)))
                h.update(repr((net, t.name, int(t.offset), int(t.rows), int(t.cols), int(t.ld))).encode())
        return h.hexdigest()[:16]

    # ------------------------------------------------------------------ small surface methods
    def train(self, training: bool = True) -> None:                      # fb_ddpg.py:161-164
        self.flush()
        self.training = training
        for net in (() if self._discrete else (self.actor,)) + (self.forward_net, self.backward_net):
            net.train(training)

    def step_counts(self) -> tp.Tuple[int, int]:
        self.flush()
        fb, ac = C.c_int32(), C.c_int32()
        check(_lib.load().fbhip_get_step_counts(self._ctx, C.byref(fb), C.byref(ac), stream_ptr()), self._ctx)
        return fb.value, ac.value

    def set_step_counts(self, fb_steps: int, actor_steps: int) -> None:
        self.flush()
        check(_lib.load().fbhip_set_step_counts(self._ctx, int(fb_steps), int(actor_steps), stream_ptr()), self._ctx)

    def rng_counts(self) -> tp.Tuple[int, int]:
        """(update() calls drawn, fast-path act() calls drawn): the counters of the device Philox streams"""
        self.flush()
        u, a = C.c_uint32(), C.c_uint32()
        check(_lib.load().fbhip_get_rng_counts(self._ctx, C.byref(u), C.byref(a), stream_ptr()), self._ctx)
        return int(u.value), int(a.value)

    def set_rng_counts(self, update_count: int, act_count: int) -> None:
        self.flush()
        check(_lib.load().fbhip_set_rng_counts(self._ctx, int(update_count), int(act_count), stream_ptr()), self._ctx)

    def _join_fast_path_stream(self) -> None:
        """The batch-1 entry points launch on the agent's own stream.  Whatever last wrote the weights -- an update enqueued
        on the caller's (non-default) stream, ``load_state_dict`` / ``init_from`` copies -- was issued on torch's current
        stream: order the fast path behind it (an event, no host synchronisation)."""
        self.flush()
        cur = torch.cuda.current_stream(self._device)
        if cur != self._stream:
            self._stream.wait_stream(cur)

    def init_from(self, other: tp.Any) -> None:                          # fb_ddpg.py:166-175
        self.flush()
        self._replicas_verified = False
        names = [] if self._discrete else ["actor"]          # (discrete_fb.py:170-178 copies "encoder" only, + the FB nets)
        if self.cfg.init_fb:
            names += ["forward_net", "backward_net", "backward_target_net", "forward_target_net"]
        for name in names:
            src = getattr(other, name)
            getattr(self, name).load_state_dict({k: v.detach() for k, v in src.state_dict().items()})
        for key in ("actor_opt", "fb_opt"):
            if getattr(other, key, None) is not None and getattr(self, key, None) is not None:
                getattr(self, key).load_state_dict(copy.deepcopy(getattr(other, key).state_dict()))

    @classmethod
    def from_reference_checkpoint(cls, path: tp.Any, device: tp.Any = "cuda", **overrides: tp.Any) -> "FBHipAgent":
        """Rebuild an agent from a checkpoint the REFERENCE wrote (``latest.pt`` / ``snapshot_*.pt``: ``torch.save`` of
        ``{'agent': FBDDPGAgent, ...}``, pretrain.py:437-449) without the reference installed: the pickled config gives
        the constructor arguments, then ``init_from`` copies the five nets and both Adam states exactly like
        ``load_checkpoint`` does (pretrain.py:476-478)."""
        from . import reference_io
        parts = reference_io.payload_parts(reference_io.load_reference_payload(path))
        if "agent" not in parts:
            raise KeyError(f"{path}: no 'agent' in the payload (keys: {sorted(parts)})")
        ref = parts["agent"]
        fields = reference_io.reference_agent_config(ref)
        fields.update(overrides)
        fields["device"] = device
        agent = cls(**fields)
        agent.init_from(ref)
        return agent

    def sample_z(self, size: int, device: str = "cpu") -> torch.Tensor:  # fb_ddpg.py:224-232
        gaussian_rdv = torch.nn.functional.normalize(
            torch.randn((size, self.cfg.z_dim), dtype=torch.float32, device=device), dim=1)
        if self.cfg.norm_z:
            return math.sqrt(self.cfg.z_dim) * gaussian_rdv
        return np.sqrt(self.cfg.z_dim) * torch.rand((size, self.cfg.z_dim), dtype=torch.float32, device=device) * gaussian_rdv

    def init_meta(self) -> MetaDict:                                      # fb_ddpg.py:234-243
        if self.solved_meta is not None:
            return self.solved_meta
        z = self.sample_z(1).squeeze().numpy()
        meta: tp.Dict[str, np.ndarray] = collections.OrderedDict()
        meta["z"] = z
        return meta

    def update_meta(self, meta: MetaDict, global_step: int, time_step: tp.Any, finetune: bool = False,
                    replay_loader: tp.Any = None) -> MetaDict:            # fb_ddpg.py:246-256
        if global_step % self.cfg.update_z_every_step == 0 and np.random.rand() < self.cfg.update_z_proba:
            return self.init_meta()
        return meta

    # ------------------------------------------------------------------ inference paths (HIP)
    def _dev(self, x: tp.Any) -> torch.Tensor:
        t = torch.as_tensor(x, dtype=torch.float32, device=self._device)
        if t.dim() == 1:
            t = t.unsqueeze(0)
        return t.contiguous()

    def _backward_map(self, goal: tp.Any, target: bool = False) -> torch.Tensor:
        """B(goal) with norm_z (fb_modules.py:223-230)"""
        self.flush()
        g = self._dev(goal)
        assert g.shape[1] == self.goal_dim, (g.shape, self.goal_dim)
        out = torch.empty((g.shape[0], self.cfg.z_dim), device=self._device)
        check(_lib.load().fbhip_backward_map(self._ctx, int(target), ptr(g), g.stride(0), g.shape[0], ptr(out),
                                             out.stride(0), stream_ptr()), self._ctx)
        return out

    def _forward_map(self, obs: torch.Tensor, z: torch.Tensor, action: torch.Tensor, target: bool = False):
        self.flush()
        f1 = torch.empty((obs.shape[0], self.cfg.z_dim), device=self._device)
        f2 = torch.empty_like(f1)
        check(_lib.load().fbhip_forward_map(self._ctx, int(target), ptr(obs), obs.stride(0), ptr(z), z.stride(0),
                                            ptr(action), action.stride(0), obs.shape[0], ptr(f1), ptr(f2), f1.stride(0),
                                            stream_ptr()), self._ctx)
        return f1, f2

    def _actor(self, obs: torch.Tensor, z: torch.Tensor, noise: tp.Optional[torch.Tensor], std: float,
               clip: tp.Optional[float]) -> torch.Tensor:
        self.flush()
        out = torch.empty((obs.shape[0], self.action_dim), device=self._device)
        check(_lib.load().fbhip_actor_forward(self._ctx, ptr(obs), obs.stride(0), ptr(z), z.stride(0), obs.shape[0],
                                              ptr(noise), float(std), -1.0 if clip is None else float(clip), ptr(out),
                                              out.stride(0), stream_ptr()), self._ctx)
        return out

    def act(self, obs: tp.Any, meta: MetaDict, step: int, eval_mode: bool) -> np.ndarray:   # fb_ddpg.py:258-281
        stddev = schedule(self.cfg.stddev_schedule, step)
        if not (eval_mode and self.cfg.additional_metric):
            # batch-1 fast path: host arrays in, host action out, one graph launch (fbhip_act)
            if not eval_mode and step < self.cfg.num_expl_steps:          # fb_ddpg.py:277-279: uniform exploration
                return torch.empty(self.action_dim).uniform_(-1.0, 1.0).numpy()
            return self._act_fast(np.asarray(obs, np.float32).reshape(-1), np.asarray(meta["z"], np.float32).reshape(-1),
                                  None, stddev, eval_mode)
        o = self._dev(obs)
        z = self._dev(meta["z"])
        action = self._actor(o, z, None, stddev, None)
        f_mean = self._forward_map(o, z, action)
        f_rand = self._forward_map(o, z, torch.zeros_like(action).uniform_(-1.0, 1.0))
        qs = [torch.min(*( (f * z).sum(1) for f in fs)) for fs in (f_mean, f_rand)]
        self.actor_success = (qs[0] > qs[1]).cpu().numpy().tolist()
        return action.cpu().numpy()[0]

    def _act_fast(self, obs: np.ndarray, z: np.ndarray, noise: tp.Optional[np.ndarray], stddev: float,
                  eval_mode: bool) -> np.ndarray:
        if obs.shape[0] != self.obs_dim or z.shape[0] != self.cfg.z_dim:
            raise ValueError(f"act: expected obs[{self.obs_dim}] and z[{self.cfg.z_dim}], got {obs.shape} / {z.shape}")
        obs, z = np.ascontiguousarray(obs), np.ascontiguousarray(z)
        out = np.empty(self.action_dim, np.float32)
        nz = None if noise is None else np.ascontiguousarray(noise, np.float32)
        self._join_fast_path_stream()
        ctx = self.__dict__["_ctx"]                  # (the join above has flushed; the stream travels as an argument: no context switch
        call = lambda: check(_lib.load().fbhip_act(ctx, obs.ctypes.data, z.ctypes.data, None if nz is None else nz.ctypes.data,  # noqa: E731
                                                   float(stddev), int(bool(eval_mode)), out.ctypes.data, self._stream.cuda_stream), ctx)
        if torch.cuda.current_device() == self._device.index:      #  -- unless the caller sits on another device)
            call()
        else:
            with torch.cuda.device(self._device):
                call()
        return out

    def _normalize_z(self, z: torch.Tensor) -> torch.Tensor:            # ``if self.cfg.norm_z:`` of fb_ddpg.py:181, :217
        if not getattr(self.cfg, "norm_z", True):
            return z
        from . import kernels
        return kernels.l2norm_fwd(z.contiguous())[0]

    def get_goal_meta(self, goal_array: np.ndarray) -> MetaDict:          # fb_ddpg.py:177-186
        z = self._backward_map(np.asarray(goal_array, np.float32))
        z = self._normalize_z(z)
        meta: tp.Dict[str, np.ndarray] = collections.OrderedDict()
        meta["z"] = z.squeeze(0).cpu().numpy()
        return meta

    def infer_meta(self, replay_loader: tp.Any) -> MetaDict:              # fb_ddpg.py:188-199
        obs_list, reward_list = [], []
        batch_size = 0
        while batch_size < self.cfg.num_inference_steps:
            batch = replay_loader.sample(self.cfg.batch_size).to(self.cfg.device)
            obs_list.append(batch.next_goal if self.cfg.goal_space is not None else batch.next_obs)
            reward_list.append(batch.reward)
            batch_size += batch.next_obs.size(0)
        obs, reward = torch.cat(obs_list, 0), torch.cat(reward_list, 0)
        obs, reward = obs[:self.cfg.num_inference_steps], reward[:self.cfg.num_inference_steps]
        return self.infer_meta_from_obs_and_rewards(obs, reward)

    def infer_meta_from_obs_and_rewards(self, obs: torch.Tensor, reward: torch.Tensor) -> MetaDict:   # fb_ddpg.py:201-222
        from . import kernels
        Bm = self._backward_map(obs)
        r = self._dev(reward).reshape(-1, 1)
        z = kernels.gemm(r, Bm, a_kcontig=False, b_kcontig=False)        # reward^T . B   [1, d]
        z = z / r.shape[0]
        z = self._normalize_z(z)
        meta: tp.Dict[str, np.ndarray] = collections.OrderedDict()
        meta["z"] = z.squeeze().cpu().numpy()
        return meta

    def compute_z_correl(self, time_step: tp.Any, meta: MetaDict) -> float:   # fb_ddpg.py:283-289
        """NB: the reference writes ``F.normalize(z, 1)`` -- the positional 1 is ``p``, so both vectors are scaled by
        their L1 norm (dim defaults to 1).  Kept as is for result parity (fbhip_z_correl)."""
        goal = time_step.goal if self.cfg.goal_space is not None else time_step.observation
        g = np.ascontiguousarray(np.asarray(goal, np.float32).reshape(-1))
        z = np.ascontiguousarray(np.asarray(meta["z"], np.float32).reshape(-1))
        if g.shape[0] != self.goal_dim or z.shape[0] != self.cfg.z_dim:
            raise ValueError(f"compute_z_correl: expected goal[{self.goal_dim}] and z[{self.cfg.z_dim}]")
        out = np.empty(1, np.float32)
        self._join_fast_path_stream()
        ctx = self.__dict__["_ctx"]
        if torch.cuda.current_device() == self._device.index:
            check(_lib.load().fbhip_z_correl(ctx, g.ctypes.data, z.ctypes.data, out.ctypes.data, self._stream.cuda_stream), ctx)
        else:
            with torch.cuda.device(self._device):
                check(_lib.load().fbhip_z_correl(ctx, g.ctypes.data, z.ctypes.data, out.ctypes.data, self._stream.cuda_stream), ctx)
        return float(out[0])

    # ------------------------------------------------------------------ the hot path
    def _stddev_is_constant(self) -> bool:
        try:
            float(self.cfg.stddev_schedule)
            return True
        except (TypeError, ValueError):
            return False

    def _hparams(self, step: int, want_metrics: bool, grad_scale: float, discount: float, future: float = 1.0) -> HParams:
        c = self.cfg
        if c.future_ratio > 0 and not future < 1:
            # the reference asserts ``future_goal is not None`` (fb_ddpg.py:489): only buffers with future < 1 sample it
            raise ValueError("future_ratio > 0 needs a replay buffer built with future < 1 (hindsight replay)")
        return HParams(future_ratio=float(c.future_ratio), future=float(future), rand_weight=int(bool(c.rand_weight)), lr=c.lr, lr_coef=c.lr_coef, fb_target_tau=c.fb_target_tau,
                       stddev=schedule(c.stddev_schedule, step), stddev_clip=c.stddev_clip, ortho_coef=c.ortho_coef,
                       mix_ratio=c.mix_ratio, q_loss_coef=c.q_loss_coef, discount=discount, grad_scale=grad_scale,
                       q_loss=int(c.q_loss), want_metrics=int(want_metrics))

    def _bind_replay(self, rb: DeviceReplayBuffer) -> None:
        token = (id(rb), rb._version)
        if token == self._replay_token:
            return
        if rb.device != self._device:
            raise RuntimeError(f"replay buffer lives on {rb.device}, agent on {self._device}")
        v = rb.device_view()
        if self._dims.use_goal and v["goal"] is None:
            raise RuntimeError("goal_space is set but the replay buffer stores no 'goal'")
        for name, dim in (("observation", self.obs_dim), ("action", 1 if self._discrete else self.action_dim)):
            if v[name].shape[2] != dim:
                raise RuntimeError(f"replay '{name}' has dim {v[name].shape[2]}, agent expects {dim}")
        check(_lib.load().fbhip_replay_bind(self._ctx, ptr(v["observation"]), ptr(v["action"]), ptr(v["discount"]),
                                            ptr(v["goal"]), ptr(v["episode_len"]), ptr(v["cum_len"]), v["n_episodes"],
                                            v["t1"], int(v["fixed_length"])), self._ctx)
        self._replay_view = v           # keeps the tensors alive while bound
        self._replay_token = token
        self._replay_binds = self.__dict__.get("_replay_binds", 0) + 1

    def _run_update(self, hp: HParams, inject: tp.Optional[Inject], use_graph: bool) -> None:
        self._on_update_stream(lambda: self._launch_update(hp, inject, use_graph))

    def _on_update_stream(self, fn: tp.Callable[[], None]) -> None:
        cur = torch.cuda.current_stream(self._device)
        if cur.cuda_stream != 0:
            fn()
            return
        # the caller sits on the legacy default stream, where stream capture is not allowed: run on the agent's own stream, ordered
        # after the caller's earlier work (fbhip_order_stream_after_legacy) and before its later work
        # (fbhip_order_legacy_stream_after) -- NOT by events recorded on / waited for by the legacy stream: a command pending
        # there while the n-step graph is enqueued and runs slowed that graph down 1.5x (include/fbhip.h; DESIGN.md section 6
        # "The legacy default stream").  No host synchronisation either way.
        gate = os.environ.get("FBHIP_LEGACY_STREAM_ORDER", "gate") != "event"
        if not gate:
            self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            if gate:
                check(_lib.load().fbhip_order_stream_after_legacy(self._ctx, stream_ptr()), self._ctx)
            fn()
            if gate:
                check(_lib.load().fbhip_order_legacy_stream_after(self._ctx, stream_ptr()), self._ctx)
            else:
                cur.wait_stream(self._stream)

    def _c10d_collectives_in_flight(self) -> bool:
        """The torch.distributed schedule (distributed.py) under backend nccl: c10d's watchdog thread polls the end events of the
        collectives it has in flight, and a poll that lands while a stream capture is open in the process makes the runtime answer
        hipErrorCapturedEvent -- the watchdog throws and the process aborts (round 2: 8 of 14 stress processes).  So that
        schedule issues its phases as EAGER launches there: no capture, no hazard, no sleep.  It is the fallback transport; the
        default one (the library's own communicator inside the n-step graph, rccl.py) has no c10d collective in flight at all."""
        import torch.distributed as dist
        return self._world() > 1 and dist.get_backend() == "nccl"

    def _launch_update(self, hp: HParams, inject: tp.Optional[Inject], use_graph: bool) -> None:
        from .distributed import dp_update
        lib = _lib.load()
        s = stream_ptr()

        self._verify_replicas()
        global_batch = bool(getattr(self.cfg, "dp_global_batch", False))
        hp_fb = hp
        if global_batch:                # the FB loss is normalised by the GLOBAL pair counts: its gradients are summed
            hp_fb = HParams.from_buffer_copy(hp)
            hp_fb.grad_scale = 1.0

        def run_phases(mask: int) -> None:
            if self._discrete:          # no actor: a schedule's actor-only calls have nothing to enqueue
                mask &= ~(_lib.PHASE_ACTOR_GRAD | _lib.PHASE_ACTOR_STEP | _lib.PHASE_ACTOR_FWD)
                if mask == 0:
                    return
            # injected draws only matter to the SAMPLE phase
            inj = C.byref(inject) if (inject is not None and mask & _lib.PHASE_SAMPLE) else None
            h = hp_fb if (mask & _lib.PHASE_FB_STEP) else hp
            check(lib.fbhip_update(self._ctx, C.byref(h), inj, mask, int(use_graph and not self._c10d_collectives_in_flight()), s), self._ctx)

        dp_update(run_phases, self._fb_grads, self._actor_grads, self._exchange_embeddings if global_batch else None,
                  early=self._early_grad_range())

    # ---- data-parallel replica consistency (distributed.py relies on bit-identical replicas: no parameter broadcast per step)
    def _replica_buffers(self) -> tp.List[torch.Tensor]:
        bufs = [self._fb_params, self._fb_targets, self._fb_m, self._fb_v]
        if not self._discrete:
            bufs += [self._actor_params, self._actor_m, self._actor_v]
        return bufs

    def sync_from_rank0(self) -> None:
        """Make every rank's replica rank 0's: parameters, targets, both Adam moment sets and the Adam step counts are
        broadcast.  Call it after constructing the agents under different seeds (the common ``seed + rank`` convention),
        after ``init_from`` / ``load_nets`` / unpickling on some ranks only -- anything that may leave replicas different.
        (The RNG seed stays per rank: ranks must draw different batches.)"""
        import torch.distributed as dist
        self.flush()
        if self._world() < 2:
            return
        on_dev = dist.get_backend() == "nccl"
        for t in self._replica_buffers():
            if on_dev:
                dist.broadcast(t, src=0)
            else:                                    # gloo (tests: several ranks sharing one GPU)
                h = t.detach().cpu()
                dist.broadcast(h, src=0)
                t.copy_(h)
        counts = torch.tensor(self.step_counts(), dtype=torch.int64, device=self._device if on_dev else "cpu")
        dist.broadcast(counts, src=0)
        self.set_step_counts(int(counts[0]), int(counts[1]))
        self._replicas_verified = False

    def _verify_replicas(self) -> None:
        """First data-parallel update (and again after anything that rewrites the weights): all ranks must hold the same
        replica, else averaged gradients would be applied to different weights and the replicas would silently stay
        different.  One small all-reduce of checksums; raises on every rank."""
        if getattr(self, "_replicas_verified", False) or self._world() < 2:
            return
        import torch.distributed as dist
        sums = [t.double().sum() for t in self._replica_buffers()] + [t.double().abs().sum() for t in self._replica_buffers()]
        vec = torch.stack(sums + [torch.tensor(float(x), dtype=torch.float64, device=self._device) for x in self.step_counts()])
        both = torch.cat([vec, -vec])
        if dist.get_backend() != "nccl":
            both = both.cpu()
        dist.all_reduce(both, op=dist.ReduceOp.MAX)
        n = vec.numel()
        hi, lo = both[:n], -both[n:]
        if not torch.equal(hi, lo):
            bad = [i for i in range(n) if float(hi[i]) != float(lo[i])]
            raise RuntimeError(
                f"FBHipAgent: data-parallel replicas differ across ranks (checksum slots {bad}; rank {self._rank()}).  The "
                "schedule never broadcasts parameters: construct every rank's agent under the SAME torch seed (e.g. "
                "torch.manual_seed(seed) before FBHipAgent(...), as bench.py does) or call agent.sync_from_rank0() after "
                "construction / init_from / load_nets.")
        self._replicas_verified = True

    def _early_grad_range(self) -> tp.Tuple[int, int]:
        """(offset, count) of the FB gradient bucket that FB_BWD_A completes (both ForwardMap heads): reduced under FB_BWD_B."""
        off, cnt = C.c_int64(), C.c_int64()
        check(_lib.load().fbhip_fb_early_grad_range(C.byref(self._dims), C.byref(off), C.byref(cnt)))
        return int(off.value), int(cnt.value)

    def _exchange_embeddings(self) -> None:
        """Mode B exchange step (distributed.py): all-gather the six embedding panels + discounts of every rank and bind
        them as the global batch of the next FB_BWD phase.  Buffers are allocated once, so steady-state steps re-use the
        same pointers (captured graphs stay valid)."""
        import torch.distributed as dist
        lib = _lib.load()
        world, rank = self._world(), self._rank()
        pretend = int(os.environ.get("FBHIP_PRETEND_WORLD", "0")) if world == 1 else 0      # bench.py --pretend-world
        if pretend > 1:
            world = pretend
        B, Lz = self.cfg.batch_size, (self.cfg.z_dim + 3) // 4 * 4
        if world > 1 and B % 32:
            raise ValueError("dp_global_batch needs batch_size to be a multiple of 32")
        n = int(lib.fbhip_embeddings_floats(C.byref(self._dims)))
        assert n == 6 * B * Lz + B
        if getattr(self, "_gb_send", None) is None or self._gb_recv.shape[0] != world:
            self._gb_send = torch.empty(n, device=self._device)
            self._gb_recv = torch.empty((world, n), device=self._device)
            self._gb_panels = torch.empty((6, world * B, Lz), device=self._device)
            self._gb_disc = torch.empty(world * B, device=self._device)
        check(lib.fbhip_export_embeddings(self._ctx, ptr(self._gb_send), stream_ptr()), self._ctx)
        if pretend > 1 or world == 1:
            self._gb_recv.copy_(self._gb_send.view(1, -1).expand(world, -1))
        elif dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(self._gb_recv.view(-1), self._gb_send)
        else:                                             # gloo (tests: several ranks on one GPU)
            dist.all_gather(list(self._gb_recv.unbind(0)), self._gb_send)
        # [rank][matrix][row] -> [matrix][rank * B + row]
        self._gb_panels.view(6, world, B, Lz).copy_(self._gb_recv[:, :6 * B * Lz].view(world, 6, B, Lz).permute(1, 0, 2, 3))
        self._gb_disc.view(world, B).copy_(self._gb_recv[:, 6 * B * Lz:])
        check(lib.fbhip_bind_global_batch(self._ctx, ptr(self._gb_panels), ptr(self._gb_disc), world * B, rank * B), self._ctx)

    def _wait_metrics(self) -> tp.Any:
        """the metric array of the last metrics-on update, as soon as the running step has published it (fbhip_wait_metrics: no
        copy command, no stream synchronise -- the update's tail is still running when this returns)"""
        d = self.__dict__
        buf = d.get("_metrics_buf")
        if buf is None:
            buf = d["_metrics_buf"] = (C.c_float * _lib.NUM_METRICS)()
        check(_lib.load().fbhip_wait_metrics(self._ctx, buf), d["_ctx"])
        return buf

    def _metrics(self) -> tp.Dict[str, float]:
        c = self.cfg
        out: tp.Dict[str, float] = {}
        if not (c.use_tb or c.use_wandb or c.use_hiplog):
            return out
        buf = self._wait_metrics()
        if getattr(c, "dp_global_batch", False) and self._world() > 1:
            # every rank holds its SHARE of the pairwise terms (global normalisers) and local means of the row-wise
            # ones: sum the former, average the latter (orth_linf / orth_l2 come from the gathered B: equal everywhere)
            import torch.distributed as dist
            t = torch.tensor(list(buf), dtype=torch.float64)
            share = [_lib.METRIC_INDEX[k] for k in ("target_M", "M1", "fb_loss", "fb_diag", "fb_offdiag", "orth_loss",
                                                     "orth_loss_diag", "orth_loss_offdiag", "q_loss")]
            scale = torch.full_like(t, 1.0 / self._world())
            scale[share] = 1.0
            t = (t * scale).to(self._device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t)
            for i, v in enumerate(t.tolist()):
                buf[i] = v
        g = lambda k: float(buf[_lib.METRIC_INDEX[k]])
        for k in ("target_M", "M1", "F1", "B", "B_norm", "z_norm", "fb_loss", "fb_diag", "fb_offdiag"):
            out[k] = g(k)
        if c.q_loss:
            out["q_loss"] = g("q_loss")
        for k in ("orth_loss", "orth_loss_diag", "orth_loss_offdiag", "orth_linf", "orth_l2"):
            out[k] = g(k)
        out["fb_opt_lr"] = self.fb_opt.param_groups[0]["lr"]
        if (c.use_tb or c.use_wandb) and not self._discrete:               # fb_ddpg.py:413-418
            out["actor_loss"], out["q"], out["actor_logprob"] = g("actor_loss"), g("q"), g("actor_logprob")
            if c.additional_metric:                                        # fb_ddpg.py:403-404, 416-417
                out["q1_success"] = g("q1_success")
        return out

    def update(self, replay_loader: tp.Any, step: int) -> tp.Dict[str, float]:      # fb_ddpg.py:427-520
        if step % self.cfg.update_every_steps != 0:
            return {}
        key = self._defer_key(replay_loader)
        if key is not None:                      # one process, graphs allowed, the buffer on the device: the short ways
            d = self.__dict__
            f = key[4]
            want = bool(f[-1] or f[-2] or f[-3])
            if not want:                         # metrics off: a candidate for the queue (see "deferred batching")
                p = d.get("_pending")
                if p is not None and p[2] == key:
                    p[3] += 1
                    if p[3] >= self.DEFER_MAX:
                        self.flush()
                        d["_run_key"] = key      # (a full queue going out does not end the run of update() calls)
                    return {}
                # Run-length rule: a call is QUEUED only when the previous call into the agent was itself an update() with the same
                # key.  The first update() after anything else (act, compute_z_correl, a state read, a buffer mutation ...) is
                # launched at once: the online loop of pretrain.py:627-652 (act -> update -> env.step -> add -> compute_z_correl)
                # puts its update on the device BEFORE the host steps the environment, and never queues.
                if d.get("_run_key") == key and self._defer_update(replay_loader, key, step):
                    return {}
            if self._update_now(replay_loader, key, step, want):
                return self._metrics() if want else {}
        self.flush()
        c = self.cfg
        want = bool(c.use_tb or c.use_wandb or c.use_hiplog)
        if isinstance(replay_loader, DeviceReplayBuffer):
            from . import peer, rccl
            split_ = self._world() > 1 or os.environ.get("FBHIP_FORCE_PHASE_SPLIT", "0") == "1"
            if (split_ and self._rccl_ready() and not getattr(c, "dp_global_batch", False) and self._use_graph and self._stddev_is_constant()):
                self._bind_replay(replay_loader)
                self._verify_replicas()
                hp = self._hparams(step, want, 1.0 / self._world(), float(replay_loader._discount), float(replay_loader._future))
                if self._rccl_run(hp, 1):
                    return self._metrics()
                return self.update(replay_loader, step)
            if self._world() > 1 and peer.enabled() and not getattr(c, "dp_global_batch", False) and self._use_graph and self._stddev_is_constant():
                self._bind_replay(replay_loader)
                self._verify_replicas()
                peer.bind(self)
                hp = self._hparams(step, want, 1.0 / self._world(), float(replay_loader._discount), float(replay_loader._future))
                self._on_update_stream(lambda: check(_lib.load().fbhip_update_many_dp(self._ctx, C.byref(hp), 1, stream_ptr()), self._ctx))
                self._check_peer_status(every=64)
                return self._metrics()
            self._bind_replay(replay_loader)
            hp = self._hparams(step, want, 1.0 / self._world(), float(replay_loader._discount), float(replay_loader._future))
            # a captured graph bakes stddev in: with a time-varying stddev_schedule (utils.py:235-255) every step would be a
            # fresh capture + instantiation (milliseconds), so those configurations run as eager launches instead
            graph_ok = self._use_graph and self._stddev_is_constant()
            self._run_update(hp, None, graph_ok)
        else:
            # any other loader with the reference's .sample(batch_size) -> EpisodeBatch contract (host sampling)
            return self.update_from_batch(replay_loader.sample(c.batch_size), step)
        return self._metrics()

    # ---- deferred batching: the drop-in call of train_offline.py:118 at the rate of the n-step graph
    # With metrics off (the default, pretrain.py:58-60) ``update()`` returns {} and the reference's offline loop does nothing
    # else between two calls (train_offline.py:116-119).  Such a call is QUEUED -- validated, counted, nothing launched -- and
    # the queue goes out as ONE ``fbhip_update_many(k <= 32)`` when it is full or when anything is about to observe or change
    # what the queued updates read or write: act / infer_meta / compute_z_correl / the batched inference entry points, any
    # parameter or optimiser access through the net / optimiser views (state_dict, parameters, load_state_dict, init_from,
    # pickling), the step / RNG counters, another update entry point, a call with other hyper-parameters, another replay
    # buffer or another stream, and every mutation of the replay buffer (the buffer asks its observers to flush BEFORE it
    # writes: queued updates sample the contents they were called on).  Same kernels, operands and draws as eager calls:
    # the queue is exactly ``update_many`` over the same steps.  What cannot be intercepted is a bare
    # ``torch.cuda.synchronize()`` (a caller timing the loop, say): ``agent.flush()`` is the explicit form.
    DEFER_MAX = 32
    DEFER_MENU = (32, 16, 8, 4, 2, 1)            # the only n-step graph sizes a queue is ever launched as (flush)
    # every cfg field the hyper-parameter struct, the metrics switch and the update cadence are made of (FBDDPGAgentConfig)
    _hp_fields = operator.attrgetter("lr", "lr_coef", "fb_target_tau", "stddev_schedule", "stddev_clip", "ortho_coef", "mix_ratio",
                                     "q_loss_coef", "q_loss", "future_ratio", "rand_weight", "use_tb", "use_wandb", "use_hiplog")

    def _defer_key(self, rb: tp.Any) -> tp.Optional[tp.Tuple]:
        """None when this call is not the plain single-process graph update (it then takes the general path of ``update``); else
        everything a queued update depends on besides the library's own state: the
        replay contents (mutation counter) and its discount / future, the cfg fields behind the hyper-parameters, the caller's
        stream, and torch's version counters of the flat parameter / optimiser tensors -- every host-side in-place write
        through a state_dict view bumps them, the library's own kernels do not.  Equal keys == the call joins the queue (a few
        microseconds per call: the host's share of a queued update)."""
        d = self.__dict__
        if (rb.__class__ is not DeviceReplayBuffer or not d.get("defer_updates", True) or not d["_use_graph"] or (_dist.is_available() and _dist.is_initialized() and _dist.get_world_size() > 1) or
                os.environ.get("FBHIP_UPDATE_DEFER", "1") == "0" or os.environ.get("FBHIP_FORCE_PHASE_SPLIT", "0") == "1" or
                torch.cuda.is_current_stream_capturing() or      # (inside a caller's capture the launches must land IN it)
                getattr(self.cfg, "dp_global_batch", False)):    # (mode B: an embedding exchange between the phases of every update)
            return None
        f = self._hp_fields(self.cfg)            # (the last three fields are the metrics switches: such a call is never QUEUED)
        return (id(rb), rb._version, rb._discount, rb._future, f, d["_fb_params"]._version, d["_fb_targets"]._version, d["_fb_m"]._version,
                d["_fb_v"]._version, d["_actor_params"]._version, d["_actor_m"]._version, d["_actor_v"]._version,
                torch._C._cuda_getCurrentRawStream(self._device.index))

    def _update_now(self, rb: DeviceReplayBuffer, key: tp.Tuple, step: int, want: bool) -> bool:
        """ONE update, launched at once through the cached single-update graph -- the short way for a call ``_defer_key`` accepted:
        no phase machinery, no data-parallel checks, the hyper-parameter struct built once per key.  What a metrics-on caller
        (README.md:50, ``use_tb=1 use_hiplog=1``) pays per call on the host is GPU idle time: the step publishes its metrics
        from inside (``fbhip_wait_metrics``), this call returns, and the next one must be enqueued before the running step's tail
        (actor backward + optimiser step) has drained.  False: not eligible (time-varying stddev: eager launches)."""
        self.flush()
        if not self._stddev_is_constant():       # (a captured graph bakes stddev in)
            return False
        d = self.__dict__
        fh = d.get("_now_hp")
        if fh is None or fh[0] != key:
            hp = self._hparams(step, want, 1.0, float(rb._discount), float(rb._future))
            fh = d["_now_hp"] = (key, hp, C.byref(hp))
        self._bind_replay(rb)
        lib, ctx, raw = _lib.load(), d["_ctx"], key[-1]
        if raw != 0:
            check(lib.fbhip_update(ctx, fh[2], None, _lib.PHASE_ALL, 1, raw), ctx)
        else:                                    # (the legacy default stream: on the agent's own stream, ordered both ways)
            self._on_update_stream(lambda: check(lib.fbhip_update(ctx, fh[2], None, _lib.PHASE_ALL, 1, stream_ptr()), ctx))
        if not want:
            d["_run_key"] = key                  # an update() that COULD have queued: the next one with this key starts a queue
        return True

    def _defer_update(self, rb: DeviceReplayBuffer, key: tp.Tuple, step: int) -> bool:
        """the slow path of a queued call: a new queue (the old one, if any, goes out first)"""
        self.flush()
        if not self._stddev_is_constant():       # (a captured graph bakes stddev in: time-varying schedules run as eager launches)
            return False
        hp = self._hparams(step, False, 1.0, float(rb._discount), float(rb._future))
        self._bind_replay(rb)                     # (dimension / device errors surface at the call, not at the flush)
        self._pending = [rb, hp, key, 1, torch.cuda.current_stream(self._device)]
        rb._observe(self)
        return True

    def flush(self) -> None:
        """Launch every queued ``update()`` call (see above) on the stream it was called on.  Asynchronous like the updates
        themselves: follow with a stream / device synchronise to wait for the results."""
        d = self.__dict__
        d["_run_key"] = None                     # whoever asks for a flush is not a queued update(): the run of calls ends here
        p = d.get("_pending")
        if p is None:
            return
        d["_pending"] = None
        rb, hp, _, n, stream = p
        rb._unobserve(self)
        now = torch.cuda.current_stream(self._device)
        if now != stream:
            # the queue goes out later than its calls were made, on THEIR stream: whatever the caller has enqueued on the stream it
            # is on now comes first, and what it enqueues next sees the updates -- as if they had run when they were called and the
            # caller had ordered its streams then
            stream.wait_stream(now)
        # A queue of k goes out as graphs from a FIXED MENU of sizes, largest first (k = 23 -> 16 + 4 + 2 + 1): whatever the flush
        # points of a caller are, at most len(DEFER_MENU) n-step graphs ever exist per (hyper-parameters, replay binding), and none
        # is captured inside a steady-state loop once each size has been seen (``graph_captures()`` counts them).
        done = 0
        try:
            with torch.cuda.stream(stream):
                self._bind_replay(rb)
                for size in self.DEFER_MENU:
                    while n - done >= size:
                        if size == 1:
                            self._run_update(hp, None, True)
                        else:
                            self._launch_many(hp, size)
                        done += size
        except BaseException:
            # nothing is dropped silently: the calls that were not launched stay queued (the error repeats at the next flush if its
            # cause persists), and the step / RNG counters never fall behind what the caller issued without an exception saying so
            if done < n:
                p[3] = n - done
                d["_pending"] = p
                rb._observe(self)
            raise
        finally:
            if now != stream:
                now.wait_stream(stream)

    def graph_captures(self) -> int:
        """update graphs the library has captured for this agent so far (``fbhip_graph_captures``); does not launch the queue"""
        return int(_lib.load().fbhip_graph_captures(self.__dict__["_ctx"]))

    def _launch_many(self, hp: HParams, n_steps: int) -> None:
        done = 0
        while done < n_steps:
            n = min(64, n_steps - done)
            self._on_update_stream(lambda n=n: check(_lib.load().fbhip_update_many(self._ctx, C.byref(hp), n, stream_ptr()), self._ctx))
            done += n

    def _all_ranks_ok(self, ok: bool) -> bool:
        """A transport decision must be the SAME on every rank (a rank that falls back alone leaves the others waiting inside a
        collective it never joins): MAX-reduce the local failure flag over the default process group."""
        import torch.distributed as dist
        if self._world() < 2:
            return ok
        flag = torch.tensor([0.0 if ok else 1.0], device=self._device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return float(flag.item()) == 0.0

    def _rccl_ready(self) -> bool:
        """Bind the library-owned RCCL transport on first use (rccl.py).  A refusal on ANY rank (no librccl, communicator set-up
        failed) demotes EVERY rank to the torch.distributed schedule of distributed.py -- agreed through the default process
        group, loudly, and visible in ``_dp_transport``."""
        from . import rccl
        if getattr(self, "_rccl_failed", False) or not rccl.usable():
            return False
        if getattr(self, "_rccl_bound", False):
            return True
        err = None
        try:
            rccl.bind(self)
        except Exception as e:                               # noqa: BLE001
            err = e
            self._rccl_bound = False
        if self._all_ranks_ok(err is None):
            return True
        import warnings
        why = f"{type(err).__name__}: {err}" if err is not None else "refused on another rank"
        self._rccl_failed, self._rccl_bound = True, False
        self._dp_transport = f"c10d (library RCCL transport refused: {why})"
        warnings.warn(f"fbhip: library-owned RCCL transport unavailable ({why}); using torch.distributed collectives")
        return False

    def _rccl_run(self, hp, n_total: int) -> bool:
        """``n_total`` updates through ``fbhip_update_many_dp`` on the library's communicator, 64 per graph.  Every graph is first
        PREPARED (captured and instantiated, not launched: ``fbhip_update_many_dp_prepare``) and the ranks agree that all of them
        could build theirs before anyone launches -- a capture of the collectives refused on one rank only would otherwise leave
        the others inside an all-reduce it never joins.  A refusal demotes every rank to the torch.distributed schedule, recorded
        in ``_dp_transport``, and returns False with nothing applied."""
        lib = _lib.load()
        # mirrors the library's graph cache: 16 entries, oldest evicted first (a graph the library has dropped must be PREPARED again,
        # under the ranks' agreement, not re-captured inside a launch); keyed on the replay BIND COUNTER -- a rebind drops every graph
        # in the library, and an id() can be reused by a later token
        prepared: tp.List[tp.Tuple] = self.__dict__.setdefault("_rccl_prepared", [])
        sizes = sorted({min(64, n_total - d) for d in range(0, n_total, 64)})
        key = lambda n: (n, bytes(hp), self.__dict__.get("_replay_binds", 0))
        todo = [n for n in sizes if key(n) not in prepared]
        if todo:
            err = None
            try:
                for n in todo:
                    self._on_update_stream(lambda n=n: check(lib.fbhip_update_many_dp_prepare(self._ctx, C.byref(hp), n, stream_ptr()), self._ctx))
            except RuntimeError as e:
                err = e
            if not self._all_ranks_ok(err is None):
                import warnings
                self._rccl_failed = True
                self._dp_transport = ("c10d (library RCCL transport could not build its graph: "
                                      f"{err if err is not None else 'refused on another rank'})")
                warnings.warn(f"fbhip: {self._dp_transport}")
                return False
            for n in todo:
                prepared.append(key(n))
            del prepared[:max(0, len(prepared) - 16)]
        done = 0
        while done < n_total:
            n = min(64, n_total - done)
            self._on_update_stream(lambda n=n: check(lib.fbhip_update_many_dp(self._ctx, C.byref(hp), n, stream_ptr()), self._ctx))
            done += n
        return True

    def _check_peer_status(self, every: int = 1) -> None:
        """The peer-access all-reduce kernels (csrc/peer.hip) give up on a cross-rank barrier after a bounded spin and set a status
        word instead of hanging the GPU -- the optimiser step that follows then runs on unreduced gradients and the replicas
        diverge.  Read that word (one 16-byte D2H + a stream synchronise) after every multi-step launch, every ``every``-th single
        step, and raise: a lost peer must stop the run, not corrupt it."""
        n = self.__dict__.get("_peer_calls", 0) + 1
        self._peer_calls = n
        if n % max(1, every):
            return
        st = C.c_int32(0)
        with torch.cuda.stream(self._stream if torch.cuda.current_stream(self._device).cuda_stream == 0 else torch.cuda.current_stream(self._device)):
            check(_lib.load().fbhip_dp_status(self._ctx, C.byref(st), stream_ptr()), self._ctx)
        if st.value != 0:
            self._replicas_verified = False
            raise RuntimeError(f"fbhip: peer all-reduce reported status {st.value} (a rank did not reach a gradient barrier in time): "
                               "the replicas can no longer be trusted; restart from a checkpoint")

    def update_many(self, replay_loader: DeviceReplayBuffer, step: int, n_steps: int) -> tp.Dict[str, float]:
        """``n_steps`` consecutive ``update(replay_loader, step + i)`` calls as ONE graph launch (``fbhip_update_many``):
        same kernels, same order, same results (up to the fp32 summation order of a few split-K GEMMs at large dims, see
        DESIGN.md section 3 "pipelined steps") -- for loops that do nothing between updates (train_offline.py:101-134
        between two log lines).  With world > 1 the steps are pipelined around the gradient all-reduces instead
        (``distributed.dp_update_many``).  Falls back to single updates whenever that would not be equivalent: the
        global-batch schedule, a host-sampling loader, ``update_every_steps != 1``, a time-varying ``stddev_schedule``.
        Returns the metrics of the LAST step (if metrics are on).  NOTE: step t+1's batch is sampled while step t is still
        running -- from the same buffer contents; do not use it when transitions are added between updates."""
        c = self.cfg
        self.flush()
        stds = {schedule(c.stddev_schedule, step + i) for i in range(n_steps)}
        split = self._world() > 1 or os.environ.get("FBHIP_FORCE_PHASE_SPLIT", "0") == "1"
        if (n_steps < 2 or getattr(c, "dp_global_batch", False) or not isinstance(replay_loader, DeviceReplayBuffer) or
                c.update_every_steps != 1 or len(stds) != 1 or not self._use_graph):
            out: tp.Dict[str, float] = {}
            for i in range(n_steps):
                out = self.update(replay_loader, step + i)
            return out
        from . import peer
        if split and self._rccl_ready():
            # data parallel with the library's own RCCL communicator: ncclAllReduce of the two gradient buckets INSIDE the n-step
            # graph (csrc/rccl.hip): one graph launch per rank per call, no torch.distributed object on the hot path
            want = bool(c.use_tb or c.use_wandb or c.use_hiplog)
            self._bind_replay(replay_loader)
            self._verify_replicas()
            hp = self._hparams(step, want, 1.0 / self._world(), float(replay_loader._discount), float(replay_loader._future))
            if self._rccl_run(hp, n_steps):
                return self._metrics()
            return self.update_many(replay_loader, step, n_steps)
        if split and self._world() > 1 and peer.enabled():
            # data parallel without host-issued collectives: the peers' gradient buckets are mapped into this process and the
            # all-reduces are kernels INSIDE the n-step graph (csrc/peer.hip): one graph launch per rank per call
            want = bool(c.use_tb or c.use_wandb or c.use_hiplog)
            self._bind_replay(replay_loader)
            self._verify_replicas()
            peer.bind(self)
            hp = self._hparams(step, want, 1.0 / self._world(), float(replay_loader._discount), float(replay_loader._future))
            done = 0
            while done < n_steps:
                n = min(64, n_steps - done)
                self._on_update_stream(lambda n=n: check(_lib.load().fbhip_update_many_dp(self._ctx, C.byref(hp), n, stream_ptr()), self._ctx))
                done += n
            self._check_peer_status(every=1)
            return self._metrics()
        if split:
            # data parallel: the steps are pipelined around the gradient all-reduces (distributed.dp_update_many)
            from .distributed import dp_update_many
            want = bool(c.use_tb or c.use_wandb or c.use_hiplog)
            self._bind_replay(replay_loader)
            self._verify_replicas()
            hp = self._hparams(step, want, 1.0 / self._world(), float(replay_loader._discount), float(replay_loader._future))
            lib = _lib.load()

            def launch() -> None:
                actor_bits = (_lib.PHASE_ACTOR_GRAD | _lib.PHASE_ACTOR_STEP | _lib.PHASE_ACTOR_FWD) if self._discrete else 0
                phase_graphs = 0 if self._c10d_collectives_in_flight() else 1

                def phases(mask: int) -> None:           # (on torch's CURRENT stream: dp_update_many switches to its side stream)
                    if mask & ~actor_bits:
                        check(lib.fbhip_update(self._ctx, C.byref(hp), None, mask & ~actor_bits, phase_graphs, stream_ptr()), self._ctx)
                side = None
                if not self._discrete and os.environ.get("FBHIP_DP_SIDE_STREAM", "1") != "0":
                    if getattr(self, "_side_stream", None) is None:
                        self._side_stream = torch.cuda.Stream(device=self._device)
                    side = self._side_stream
                dp_update_many(phases,
                               lambda which: (self.__dict__.__setitem__("_cur_set", which),
                                              check(lib.fbhip_select_workspace_set(self._ctx, which), self._ctx))[1],
                               self._fb_grads, self._actor_grads, n_steps, early=self._early_grad_range(), side=side)
            self._on_update_stream(launch)
            return self._metrics()
        want = bool(c.use_tb or c.use_wandb or c.use_hiplog)
        self._bind_replay(replay_loader)
        hp = self._hparams(step, want, 1.0, float(replay_loader._discount), float(replay_loader._future))
        done = 0
        while done < n_steps:
            n = min(64, n_steps - done)


            def launch(n: int = n) -> None:
                check(_lib.load().fbhip_update_many(self._ctx, C.byref(hp), n, stream_ptr()), self._ctx)
            self._on_update_stream(launch)
            done += n
        return self._metrics()

    def update_from_batch(self, batch: tp.Any, step: int, draws: tp.Optional[tp.Mapping[str, tp.Any]] = None,
                          use_graph: bool = False) -> tp.Dict[str, float]:
        """One update on an externally sampled batch (``EpisodeBatch``-like: obs, action, next_obs, discount
        [, goal, next_goal]).  ``draws`` optionally injects the remaining random draws (parity tests):
        z_gauss [B,d], perm [B], mix_uniform [B], eps_next [B,a], eps_actor [B,a]."""
        self.flush()
        c, dev, Bn = self.cfg, self._device, self.cfg.batch_size
        f = lambda x: torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x, dtype=torch.float32,
                                      device=dev).reshape(Bn, -1)
        obs, nobs, act, disc = f(batch.obs), f(batch.next_obs), f(batch.action), f(batch.discount)
        # (hindsight rows: FB's future_ratio > 0, fb_ddpg.py:487-491; SFAgent's contrastive learners read batch.future_goal, sf.py:125, 167)
        hindsight = getattr(c, "future_ratio", 0.0) > 0 or self._sf_mode in (10, 11)
        # a one-transition-per-episode storage [B, 2 (+1), dim]: row 0 = (obs, goal), row 1 = (next_obs, action, ...),
        # row 2 = (future_obs, future_goal) when hindsight replay is on
        rows = lambda *xs: torch.stack(list(xs), 1).contiguous()
        if hindsight:
            if batch.future_obs is None or (c.goal_space is not None and batch.future_goal is None):
                raise ValueError("future_ratio > 0 (and SFAgent's contrastive learners) need batch.future_obs / future_goal (sample from a buffer with future < 1)")
            storage = {"observation": rows(obs, nobs, f(batch.future_obs)), "action": rows(torch.zeros_like(act), act, torch.zeros_like(act)),
                       "discount": rows(torch.ones_like(disc), disc, torch.ones_like(disc))}
            if c.goal_space is not None:
                storage["goal"] = rows(f(batch.goal), f(batch.next_goal), f(batch.future_goal))
        else:
            storage = {"observation": rows(obs, nobs), "action": rows(torch.zeros_like(act), act),
                       "discount": rows(torch.ones_like(disc), disc)}
            if c.goal_space is not None:
                storage["goal"] = rows(f(batch.goal), f(batch.next_goal))
        rb = DeviceReplayBuffer(Bn, 1.0, 0.5 if hindsight else 1.0, device=dev)
        rb._storage = storage
        rb._episodes_length = np.full(Bn, 2 if hindsight else 1, np.int32)
        rb._idx, rb._full = 0, True
        rb._touch()
        self._ext_replay = rb
        self._bind_replay(rb)
        i32 = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.int32, device=dev).contiguous()
        keep: tp.List[torch.Tensor] = [i32(np.arange(Bn)), i32(np.ones(Bn))]
        inj = Inject(ep_idx=ptr(keep[0]), step_idx=ptr(keep[1]))
        if hindsight:
            keep.append(i32(np.full(Bn, 3)))                               # future row = storage[:, 2] = [ep, 3 - 1]
            inj.future_idx = ptr(keep[-1])
        if draws is not None:
            for name in ("z_gauss", "mix_uniform", "eps_next", "eps_actor", "future_uniform", "z_uniform", "rand_weight",
                         "rand_weight_u"):
                if name in draws and draws[name] is not None:
                    t = torch.as_tensor(np.asarray(draws[name], dtype=np.float32), device=dev).contiguous()
                    keep.append(t)
                    setattr(inj, name, ptr(t))
            if draws.get("perm") is not None:
                keep.append(i32(draws["perm"]))
                inj.perm = ptr(keep[-1])
        want = bool(c.use_tb or c.use_wandb or c.use_hiplog)
        hp = self._hparams(step, want, 1.0 / self._world(), 1.0, rb._future)   # batch.discount is already gamma-scaled
        self._run_update(hp, inj, use_graph)
        torch.cuda.current_stream().synchronize()                          # ``keep`` must outlive the launches
        return self._metrics()

    def update_injected(self, replay_loader: DeviceReplayBuffer, step: int, draws: tp.Mapping[str, tp.Any],
                        use_graph: bool = False) -> tp.Dict[str, float]:
        """Parity mode: every random draw of the step is supplied (``ep_idx, step_idx, z_gauss, perm,
        mix_uniform, eps_next, eps_actor``), exactly as recorded from the reference run."""
        self.flush()
        dev = self._device
        self._bind_replay(replay_loader)
        keep = {}
        inj = Inject()
        for name in ("ep_idx", "step_idx", "perm"):
            keep[name] = torch.as_tensor(np.asarray(draws[name]), dtype=torch.int32, device=dev).contiguous()
            setattr(inj, name, ptr(keep[name]))
        for name in ("z_gauss", "mix_uniform", "eps_next", "eps_actor"):
            keep[name] = torch.as_tensor(np.asarray(draws[name], dtype=np.float32), device=dev).contiguous()
            setattr(inj, name, ptr(keep[name]))
        if getattr(self.cfg, "rand_weight", False):                    # raw weights [B,B] + row scales [B] (fb_ddpg.py:477-479)
            keep["rand_weight"] = torch.as_tensor(np.asarray(draws["rand_weight"], dtype=np.float32), device=dev).contiguous()
            keep["rand_weight_u"] = torch.as_tensor(np.asarray(draws["rand_weight_u"], dtype=np.float32), device=dev).contiguous()
            inj.rand_weight, inj.rand_weight_u = ptr(keep["rand_weight"]), ptr(keep["rand_weight_u"])
        if not getattr(self.cfg, "norm_z", True):                      # sample_z's uniform factor (fb_ddpg.py:230)
            keep["z_uniform"] = torch.as_tensor(np.asarray(draws["z_uniform"], dtype=np.float32), device=dev).contiguous()
            inj.z_uniform = ptr(keep["z_uniform"])
        if getattr(self.cfg, "future_ratio", 0.0) > 0:                 # hindsight replay draws (fb_ddpg.py:487-491)
            keep["future_idx"] = torch.as_tensor(np.asarray(draws["future_idx"]), dtype=torch.int32, device=dev).contiguous()
            keep["future_uniform"] = torch.as_tensor(np.asarray(draws["future_uniform"], dtype=np.float32), device=dev).contiguous()
            inj.future_idx, inj.future_uniform = ptr(keep["future_idx"]), ptr(keep["future_uniform"])
        elif self._sf_mode in (10, 11):                                # SFAgent's contrastive learners read batch.future_goal (sf.py:125, 167)
            keep["future_idx"] = torch.as_tensor(np.asarray(draws["future_idx"]), dtype=torch.int32, device=dev).contiguous()
            inj.future_idx = ptr(keep["future_idx"])
        self._inject_keep = keep
        c = self.cfg
        want = bool(c.use_tb or c.use_wandb or c.use_hiplog)
        hp = self._hparams(step, want, 1.0 / self._world(), float(replay_loader._discount), float(replay_loader._future))
        self._run_update(hp, inj, use_graph)
        return self._metrics()

    def update_many_injected(self, replay_loader: DeviceReplayBuffer, step: int,
                             draws_per_step: tp.Sequence[tp.Mapping[str, tp.Any]]) -> tp.Dict[str, float]:
        """Parity mode of ``update_many``: ``len(draws_per_step)`` consecutive updates as ONE pipelined multi-step graph
        (``fbhip_update_many_injected``), every random draw of every step supplied like ``update_injected`` does for one.
        Returns the metrics of the last step.  Single rank, constant stddev only."""
        n = len(draws_per_step)
        if not 1 <= n <= 64:
            raise ValueError("update_many_injected: 1..64 steps per call")
        if self._world() > 1 or len({schedule(self.cfg.stddev_schedule, step + i) for i in range(n)}) != 1:
            raise ValueError("update_many_injected: single rank and a constant stddev over the steps only")
        self.flush()
        dev = self._device
        self._bind_replay(replay_loader)
        # (SFAgentConfig has no future_ratio / rand_weight / norm_z; its contrastive learners read the hindsight goal, sf.py:125, 167)
        hindsight = getattr(self.cfg, "future_ratio", 0.0) > 0
        ints = ["ep_idx", "step_idx", "perm"] + (["future_idx"] if hindsight or self._sf_mode in (10, 11) else [])
        flts = ["z_gauss", "mix_uniform", "eps_next", "eps_actor"] + (["rand_weight", "rand_weight_u"] if getattr(self.cfg, "rand_weight", False) else []) + \
               ([] if getattr(self.cfg, "norm_z", True) else ["z_uniform"]) + (["future_uniform"] if hindsight else [])
        keep: tp.List[torch.Tensor] = []
        injs = (Inject * n)()
        for i, draws in enumerate(draws_per_step):
            for name in ints:
                keep.append(torch.as_tensor(np.asarray(draws[name]), dtype=torch.int32, device=dev).contiguous())
                setattr(injs[i], name, ptr(keep[-1]))
            for name in flts:
                keep.append(torch.as_tensor(np.asarray(draws[name], dtype=np.float32), device=dev).contiguous())
                setattr(injs[i], name, ptr(keep[-1]))
        c = self.cfg
        want = bool(c.use_tb or c.use_wandb or c.use_hiplog)
        hp = self._hparams(step, want, 1.0, float(replay_loader._discount), float(replay_loader._future))
        self._on_update_stream(lambda: check(_lib.load().fbhip_update_many_injected(self._ctx, C.byref(hp), n, injs, stream_ptr()),
                                             self._ctx))
        torch.cuda.current_stream(dev).synchronize()                  # ``keep`` must outlive the launches
        return self._metrics()

    def workspace_view(self, name: str) -> torch.Tensor:
        """A named intermediate of the last update as a tensor view into the workspace (tests / debugging)."""
        self.flush()
        p, rows, cols, ld = C.c_void_p(), C.c_int32(), C.c_int32(), C.c_int32()
        check(_lib.load().fbhip_workspace_view(self._ctx, name.encode(), C.byref(p), C.byref(rows), C.byref(cols),
                                               C.byref(ld)), self._ctx)
        base = self._workspace.data_ptr()
        off = p.value - base
        is_int = name in ("ep_idx", "step_idx", "perm", "future_idx")
        flat = self._workspace[off:off + 4 * rows.value * ld.value].view(torch.int32 if is_int else torch.float32)
        return flat.view(rows.value, ld.value)[:, :cols.value]


def _flushing_attribute(name: str) -> property:
    """The flat device buffers (and the views built on them) as attributes that first launch the agent's queued updates:
    whoever reads ``agent._fb_params`` / ``agent._adam_views`` ... sees every update() call made so far (FBHipAgent.flush)."""
    def get(self: "FBHipAgent") -> tp.Any:
        self.flush()
        try:
            return self.__dict__[name]
        except KeyError:
            raise AttributeError(name) from None

    def put(self: "FBHipAgent", value: tp.Any) -> None:
        self.__dict__[name] = value
    return property(get, put)


# (``_ctx`` too: EVERY call into the library on behalf of this agent first launches what the agent still holds back)
for _n in ("_fb_params", "_fb_grads", "_fb_m", "_fb_v", "_fb_targets", "_actor_params", "_actor_grads", "_actor_m", "_actor_v",
           "_workspace", "_adam_views", "_grad_views", "_ctx"):
    setattr(FBHipAgent, _n, _flushing_attribute(_n))
del _n


# ================================================================================================== sibling agent
@dataclasses.dataclass
class DiscreteFBAgentConfig(FBDDPGAgentConfig):
    """discrete_fb.py:37-46.  (The reference's un-annotated ``boltzmann = True`` / ``temp = 100`` class attributes are
    shadowed by the inherited dataclass fields' defaults in ``__init__``: the effective defaults stay False / 1.)"""
    _target_: str = "controllable_agent_amd.agent.DiscreteFBHipAgent"
    name: str = "discrete_fb"
    preprocess: bool = False
    expl_eps: float = 0.2


class DiscreteFBHipAgent(FBHipAgent):
    """``url_benchmark/agent/discrete_fb.py:103-468`` (DiscreteFBAgent) on the same kernels: the FB step of FBHipAgent with
    a ForwardMap that has no action input and emits one embedding per action ([B, z_dim, A]), a greedy / softmax selection
    of the target embedding, a gather of the taken action's column (scatter in the backward), and no actor.
    ``action_shape[0]`` is the NUMBER of actions; the replay buffer stores the action index as one float per transition."""
    _config_cls = DiscreteFBAgentConfig
    _discrete = True

    def __init__(self, **kwargs: tp.Any) -> None:
        cfg = DiscreteFBAgentConfig(**kwargs)
        if cfg.preprocess:
            # discrete_fb.ForwardMap.forward (:91-94) reads self.obs_action_net, which its constructor never builds
            raise NotImplementedError("DiscreteFBHipAgent: preprocess=True cannot run in the reference either (discrete_fb.py:91-94)")
        super().__init__(**kwargs)

    def greedy_action(self, obs: tp.Any, z: tp.Any, target: bool = False) -> torch.Tensor:
        """argmax_a min_i F_i(obs, z)[:, :, a] . z for a batch of rows (device int32 tensor), discrete_fb.py:263-268."""
        o, zz = self._dev(obs), self._dev(z)
        out = torch.empty(o.shape[0], dtype=torch.int32, device=self._device)
        check(_lib.load().fbhip_discrete_act(self._ctx, int(target), ptr(o), o.stride(0), ptr(zz), zz.stride(0), o.shape[0],
                                             ptr(out), None, None, None, 0, stream_ptr()), self._ctx)
        return out

    def target_embedding(self, obs: tp.Any, z: tp.Any, target: bool = True):
        """(F1, F2, next_Q) that update_fb's target side selects for these rows (greedy column / softmax mix, :289-303)."""
        o, zz = self._dev(obs), self._dev(z)
        n, d = o.shape[0], self.cfg.z_dim
        f1, f2 = (torch.empty(n, d, device=self._device) for _ in range(2))
        nq = torch.empty(n, device=self._device)
        check(_lib.load().fbhip_discrete_act(self._ctx, int(target), ptr(o), o.stride(0), ptr(zz), zz.stride(0), n, None,
                                             ptr(nq), ptr(f1), ptr(f2), d, stream_ptr()), self._ctx)
        return f1, f2, nq

    def act(self, obs: tp.Any, meta: MetaDict, step: int, eval_mode: bool) -> tp.Any:      # discrete_fb.py:258-275
        # batch-1 fast path: host arrays in, host index out, one graph launch (fbhip_discrete_act_host)
        o = np.ascontiguousarray(np.asarray(obs, np.float32).reshape(-1))
        z = np.ascontiguousarray(np.asarray(meta["z"], np.float32).reshape(-1))
        if o.shape[0] != self.obs_dim or z.shape[0] != self.cfg.z_dim:
            raise ValueError(f"act: expected obs[{self.obs_dim}] and z[{self.cfg.z_dim}], got {o.shape} / {z.shape}")
        out = C.c_int32()
        self._join_fast_path_stream()
        with torch.cuda.stream(self._stream):
            check(_lib.load().fbhip_discrete_act_host(self._ctx, o.ctypes.data, z.ctypes.data, C.byref(out),
                                                      self._stream.cuda_stream), self._ctx)
        action = int(out.value)
        if not eval_mode:
            if step < self.cfg.num_expl_steps:
                action = int(torch.randint(0, self.action_dim, (1,))[0])
            elif np.random.rand() < self.cfg.expl_eps:
                action = int(torch.randint(0, self.action_dim, (1,))[0])
        return action

    @property
    def compute_z_correl(self) -> tp.Any:            # DiscreteFBAgent has none (discrete_fb.py): hasattr() must say so
        raise AttributeError("DiscreteFBAgent has no compute_z_correl (discrete_fb.py)")


# ================================================================================================== second sibling
@dataclasses.dataclass
class SFAgentConfig:
    """Field-for-field mirror of sf.py:37-82 (omegaconf interpolations become plain required fields)."""
    _target_: str = "controllable_agent_amd.agent.SFHipAgent"
    name: str = "sf"
    obs_type: str = MISSING
    obs_shape: tp.Tuple[int, ...] = MISSING
    action_shape: tp.Tuple[int, ...] = MISSING
    device: str = "cuda"
    lr: float = 1e-4
    lr_coef: float = 5
    sf_target_tau: float = 0.01
    update_every_steps: int = 2
    use_tb: bool = False
    use_wandb: bool = False
    use_hiplog: bool = False
    num_expl_steps: int = MISSING
    num_inference_steps: int = 5120
    hidden_dim: int = 1024
    backward_hidden_dim: int = 512
    feature_dim: int = 512
    z_dim: int = 100
    stddev_schedule: str = "0.2"
    stddev_clip: float = 0.3
    update_z_every_step: int = 100
    nstep: int = 1
    batch_size: int = 1024
    init_sf: bool = True
    update_encoder: bool = True
    goal_space: tp.Optional[str] = None
    log_std_bounds: tp.Tuple[float, float] = (-5, 2)
    temp: float = 1
    boltzmann: bool = False
    debug: bool = False
    preprocess: bool = True
    num_sf_updates: int = 1
    feature_learner: str = "icm"
    mix_ratio: float = 0.0
    q_loss: bool = True
    update_cov_every_step: int = 1000
    add_trunk: bool = False


class FeatureLearnerView(NetView):
    """``agent.feature_learner``: parameters of ``feature_net`` (and ``inverse_dynamic_net`` for icm) under the reference's
    state_dict names, plus the callable ``feature_net`` the reference's inference code uses (sf.py:521, 532, 564)."""

    def __init__(self, *a: tp.Any, **k: tp.Any) -> None:
        super().__init__(*a, **k)
        self.feature_net = self._forward


class SFHipAgent(FBHipAgent):
    """``url_benchmark/agent/sf.py:383-768`` (SFAgent) on the kernels of the FB step: the same Actor and ForwardMap modules
    (the latter as ``successor_net`` / ``successor_target_net``), the same actor phase, Adam passes and target EMA; the critic
    loss is the TD regression on successor features (``sf_loss_kernel``) and ``feature_learner`` -- ``feature_net`` = the
    BackwardMap architecture, trained by its own loss at ``lr_coef * lr`` -- is one of

        "icm"          inverse dynamics  mean((action - tanh-mlp(cat[phi(goal), phi(next_goal)]))^2)      sf.py:194-213
        "lap"          Laplacian         mean((phi - next_phi)^2) + orthonormality loss of phi phi^T        sf.py:100-116
        "random"       feature_net keeps its initial weights: no loss, no phi_opt                         sf.py:84-92, 447
        "autoencoder"  mean((decoder(phi(goal)) - goal)^2)                                                sf.py:249-262
        "transition"   mean((forward_dynamic_net(cat[phi(goal), action]) - next_goal)^2)                  sf.py:215-227
        "FB"           feature_net = the backward_net of a trained FB agent (``fb_features=``), frozen       sf.py:368-380
        "latent"       mean((forward_dynamic_net(cat[phi(goal), action]) - target_feature_net(next_goal))^2), the
                       target net following feature_net at rate 0.01                                          sf.py:230-246
        "contrastive"  logits = cos(phi(goal), mu_net(future_goal)):  mean(-diag + logsumexp over the off-diagonal of each row);
                       the buffer must sample hindsight goals (future < 1)                                    sf.py:118-143
        "contrastivev2" the same with the roles swapped: cos(mu_net(goal), phi(future_goal))                    sf.py:159-186
        "identity"     feature_net = nn.Identity(): phi(goal) = goal (z_dim must equal the goal dimension), nothing trained   sf.py:94-98
        "svd_sr"       SR = phi(goal) . mu_net(next_goal)^T against 0.99 x the same product of two target nets:
                       -2 mean diag SR + mean offdiag (SR - 0.99 target_SR)^2 + orthonormality loss of phi (LRA-SR)   sf.py:264-299
        "svd_srv2"     the same with the roles swapped: SR = mu_net(goal) . phi(next_goal)^T, 0.98, orthonormality of phi(next_goal)   sf.py:303-335
        "svd_p"        P = mu_net(cat[goal, action]) . phi(next_goal)^T:  -2 mean diag P + mean offdiag P^2
                       + orthonormality loss of phi(next_goal) (the paper's LRA-P)                         sf.py:337-362

    Those are all thirteen feature learners of the reference.  ``boltzmann`` raises: the reference's own constructor does (sf.py:415 passes
    ``cfg.obs_type`` to DiagGaussianActor as an extra positional argument: TypeError), so there is nothing to pin; pixels raise
    NotImplementedError at construction.  ``mix_ratio > 0`` (sf.py:725-739): the rows drawn by the mix uniform take
    ``z = sqrt(d) normalize(phi(next_goal[perm]) @ inverse(phi^T phi / B))`` -- the reference's ``pinv`` equals this inverse while the
    feature covariance has full rank, which needs ``batch_size >= z_dim`` (checked here); a rank-deficient covariance is not supported.  ``num_sf_updates = k`` (sf.py:706): every ``update`` call runs k complete
    updates, each on a fresh batch."""
    _config_cls = SFAgentConfig
    _LEARNERS = {"icm": 1, "lap": 2, "random": 3, "autoencoder": 4, "transition": 5, "FB": 3, "svd_p": 6, "latent": 7, "svd_sr": 8,
                 "svd_srv2": 9, "contrastive": 10, "contrastivev2": 11,
                 "identity": 12}                 # -> fbhip_dims.sf
    # the head mlp(in, Hb, 'irelu', Hb, 'irelu', out) next to feature_net: (module name, in, out) from (z, a, g)
    _HEADS = {1: ("inverse_dynamic_net", lambda z, a, g: (2 * z, a)), 4: ("decoder", lambda z, a, g: (z, g)),
              5: ("forward_dynamic_net", lambda z, a, g: (z + a, g)), 7: ("forward_dynamic_net", lambda z, a, g: (z + a, z))}

    def __init__(self, fb_features: tp.Any = None, **kwargs: tp.Any) -> None:
        cfg = SFAgentConfig(**kwargs)
        bad = [k for k, v in dict(feature_learner=cfg.feature_learner not in self._LEARNERS, boltzmann=cfg.boltzmann,
                                  mix_ratio=cfg.mix_ratio > 0 and cfg.batch_size < cfg.z_dim,
                                  num_sf_updates=cfg.num_sf_updates < 1).items() if v]
        if bad:
            raise NotImplementedError(f"SFHipAgent: not implemented in the HIP path: {bad} (feature_learner in {sorted(self._LEARNERS)})")
        if cfg.feature_learner == "FB" and fb_features is None:
            # FBFeatures (sf.py:368-380) reads a trained FB agent from a path hard-coded in the reference's source
            raise ValueError('feature_learner="FB" needs fb_features=<checkpoint the reference wrote | an FB agent>: its backward_net '
                             "becomes the (frozen) feature_net")
        self._sf_mode = self._LEARNERS[cfg.feature_learner]
        self.inv_cov: tp.Optional[torch.Tensor] = None
        super().__init__(**kwargs)
        self.inv_cov = torch.eye(self.cfg.z_dim, dtype=torch.float32, device=self._device)      # sf.py:469
        if cfg.feature_learner == "FB":
            self.load_fb_features(fb_features)

    def load_fb_features(self, source: tp.Any) -> None:
        """``feature_net`` <- ``backward_net`` of a trained FB agent (sf.py:368-380, FBFeatures): ``source`` is a checkpoint file
        the reference (or ``torch.save`` of a reference agent) wrote, or an agent object (reference placeholder, FBHipAgent).
        The features stay frozen (sf.py:447: no phi_opt for "FB")."""
        ref = source
        if isinstance(source, (str, os.PathLike)):
            from . import reference_io
            parts = reference_io.payload_parts(reference_io.load_reference_payload(source))
            if "agent" not in parts:
                raise KeyError(f"{source}: no 'agent' in the payload (keys: {sorted(parts)})")
            ref = parts["agent"]
        sd = {k: torch.as_tensor(np.asarray(v.detach().cpu() if isinstance(v, torch.Tensor) else v)) for k, v in ref.backward_net.state_dict().items()}
        if not all(k.startswith("B.") for k in sd):
            raise KeyError(f"backward_net has unexpected entries {sorted(sd)[:3]}...: not a BackwardMap (fb_modules.py:211-230)")
        self.feature_learner.load_state_dict({"feature_net." + k[2:]: v for k, v in sd.items()})

    def __getstate__(self) -> tp.Dict[str, tp.Any]:
        st = super().__getstate__()
        st["inv_cov"] = self.inv_cov.detach().cpu()
        return st

    def __setstate__(self, st: tp.Dict[str, tp.Any]) -> None:
        self._sf_mode = self._LEARNERS[st["cfg"]["feature_learner"]]
        super().__setstate__(st)
        self.inv_cov = st["inv_cov"].to(self._device)

    # ---- construction: sf.py:419-463 builds actor, successor_net, successor_target_net, feature_learner in this order
    def _reference_init(self) -> tp.Dict[str, tp.Dict[str, torch.Tensor]]:
        c = self.cfg
        dims = (self.obs_dim, self.action_dim, self.goal_dim, c.z_dim, c.hidden_dim, c.feature_dim, c.backward_hidden_dim)

        def ortho(lins: tp.List[tp.Tuple[str, torch.nn.Linear]]) -> None:      # utils.weight_init via Module.apply, in module order
            for _, lin in lins:
                torch.nn.init.orthogonal_(lin.weight.data)

        def build_fb(net: str) -> tp.Dict[str, torch.Tensor]:
            lins = [(p_, torch.nn.Linear(i, o)) for p_, i, o in _linear_order(net, *dims, add_trunk=bool(c.add_trunk),
                                                                               preprocess=bool(c.preprocess))]
            ortho(lins)
            sd: tp.Dict[str, torch.Tensor] = {}
            for p_, lin in lins:
                sd[f"{p_}.weight"], sd[f"{p_}.bias"] = lin.weight.data, torch.zeros_like(lin.bias.data)
                no_ln = ("F1", "F2", "policy") + (() if not c.preprocess else ("trunk",))
                if p_.endswith(".0") and not p_.startswith(no_ln):
                    sd[f"{p_[:-2]}.1.weight"], sd[f"{p_[:-2]}.1.bias"] = torch.ones(lin.out_features), torch.zeros(lin.out_features)
            return sd

        nets = {"actor": build_fb("actor"), "successor_net": build_fb("forward_net")}
        build_fb("forward_net")                                                    # successor_target_net (overwritten by a copy)
        g, d, Hb, a = self.goal_dim, c.z_dim, c.backward_hidden_dim, self.action_dim
        feat = [("feature_net.0", torch.nn.Linear(g, Hb)), ("feature_net.3", torch.nn.Linear(Hb, Hb)), ("feature_net.5", torch.nn.Linear(Hb, d))]
        ortho(feat)                                                                # FeatureLearner.__init__: self.apply(weight_init), sf.py:88
        if self._sf_mode in self._HEADS:        # e.g. ICM.__init__ (sf.py:195-200): builds its head, then applies weight_init AGAIN to everything
            name, io = self._HEADS[self._sf_mode]
            fin, fout = io(d, a, g)
            feat = feat + [(f"{name}.0", torch.nn.Linear(fin, Hb)), (f"{name}.2", torch.nn.Linear(Hb, Hb)), (f"{name}.4", torch.nn.Linear(Hb, fout))]
            if self._sf_mode != 7:               # (latent builds one more net before its second weight_init, below)
                ortho(feat)
        if self._sf_mode == 6:                  # SVDP.__init__ (sf.py:338-342): mu_net = mlp(g + a, Hb, "ntanh", Hb, "relu", z), then weight_init again
            feat = feat + [("mu_net.0", torch.nn.Linear(g + a, Hb)), ("mu_net.3", torch.nn.Linear(Hb, Hb)), ("mu_net.5", torch.nn.Linear(Hb, d))]
            ortho(feat)
        if self._sf_mode in (10, 11):           # ContrastiveFeature(v2).__init__ (sf.py:119-123, 160-164): mu_net = feature_net's architecture (with "L2"), weight_init again
            feat = feat + [("mu_net.0", torch.nn.Linear(g, Hb)), ("mu_net.3", torch.nn.Linear(Hb, Hb)), ("mu_net.5", torch.nn.Linear(Hb, d))]
            ortho(feat)
        if self._sf_mode in (8, 9):             # SVDSR.__init__ / SVDSRv2.__init__ (sf.py:265-270, 304-309): mu_net on the goal alone, then BOTH target nets (own weights), one weight_init
            feat = feat + [(f"{n}.{i}", torch.nn.Linear(*io)) for n in ("mu_net", "target_feature_net", "target_mu_net")
                           for i, io in ((0, (g, Hb)), (3, (Hb, Hb)), (5, (Hb, d)))]
            ortho(feat)
        if self._sf_mode == 7:                  # TransitionLatentModel.__init__ (sf.py:231-236): forward_dynamic_net (above), then target_feature_net --
            # its OWN random weights, never a copy of feature_net -- and weight_init over all three
            tgt = [("target_feature_net.0", torch.nn.Linear(g, Hb)), ("target_feature_net.3", torch.nn.Linear(Hb, Hb)),
                   ("target_feature_net.5", torch.nn.Linear(Hb, d))]
            feat = feat + tgt
            ortho(feat)
        sd = {}
        for p_, lin in feat:
            sd[f"{p_}.weight"], sd[f"{p_}.bias"] = lin.weight.data, torch.zeros_like(lin.bias.data)
        sd["feature_net.1.weight"], sd["feature_net.1.bias"] = torch.ones(Hb), torch.zeros(Hb)
        if self._sf_mode in (6, 10, 11):
            sd["mu_net.1.weight"], sd["mu_net.1.bias"] = torch.ones(Hb), torch.zeros(Hb)
        if self._sf_mode == 7:
            sd["target_feature_net.1.weight"], sd["target_feature_net.1.bias"] = torch.ones(Hb), torch.zeros(Hb)
        if self._sf_mode in (8, 9):
            for n in ("mu_net", "target_feature_net", "target_mu_net"):
                sd[f"{n}.1.weight"], sd[f"{n}.1.bias"] = torch.ones(Hb), torch.zeros(Hb)
        nets["feature_learner"] = sd if self._sf_mode != 12 else {}     # (identity: the net above only consumed the RNG, sf.py:94-98)
        return nets

    def _allocate(self, nets: tp.Optional[tp.Dict[str, tp.Dict[str, torch.Tensor]]]) -> None:
        super()._allocate(None)
        # the FB context's forward / backward segments under the names of sf.py
        fwd, bwd, tgt = self.forward_net, self.backward_net, self.forward_target_net
        self.successor_net, self.successor_target_net = fwd, tgt
        fwd._name, tgt._name = "successor_net", "successor_target_net"
        me = weakref.ref(self)
        self.feature_learner = FeatureLearnerView("feature_learner", bwd._flat, self._layout_of(1),
                                                  forward=lambda x: me()._backward_map(x, target=False), sync=bwd._sync)
        if self._sf_mode in (7, 8, 9):
            # latent (sf.py:234) / svd_sr, svd_srv2 (:266-267, :306-307): ``feature_learner.target_feature_net`` [and ``target_mu_net``] are blocks of the
            # TARGET buffer -- modules without gradients to the optimiser, part of ``feature_learner.state_dict()`` for checkpoints
            tb = self.backward_target_net
            for k, v in tb._views.items():
                if k.startswith(("feature_net.", "mu_net.")):
                    self.feature_learner._views["target_" + k] = v
            self.feature_learner._pads += tb._pads
        self._adam_views["successor_net"] = self._adam_views["forward_net"]
        self._adam_views["feature_learner"] = self._adam_views["backward_net"]
        self._grad_views["successor_net"], self._grad_views["feature_learner"] = self._grad_views["forward_net"], self._grad_views["backward_net"]
        for stale in ("forward_net", "backward_net", "forward_target_net", "backward_target_net", "fb_opt"):
            delattr(self, stale)
        c = self.cfg
        self.sf_opt = AdamView(self, "fb", ["successor_net"], [c.lr])                          # sf.py:459
        # sf.py:461-463 ("random" trains nothing: no optimiser, its block of the fused pass sees zero gradients)
        self.phi_opt = AdamView(self, "fb", ["feature_learner"], [c.lr_coef * c.lr]) if self._sf_mode not in (3, 12) else None
        if self._sf_mode == 12:
            # identity (sf.py:94-98): FeatureLearner.__init__ builds (and initialises) a feature_net, then replaces it with nn.Identity():
            # no parameters, phi(goal) = goal.  The context keeps the unused block; this view shows none of it
            self.feature_learner._views.clear()
            self.feature_learner.feature_net = lambda x: torch.as_tensor(np.asarray(x, np.float32) if not isinstance(x, torch.Tensor) else x,
                                                                         dtype=torch.float32, device=self._device)
        if nets is not None:
            self.load_nets(nets)

    def _layout_of(self, net: int) -> tp.List[TensorDesc]:
        lib, out = _lib.load(), []
        for i in range(lib.fbhip_layout_count(C.byref(self._dims), net)):
            t = TensorDesc()
            check(lib.fbhip_layout_entry(C.byref(self._dims), net, i, C.byref(t)))
            out.append(t)
        return out

    def load_nets(self, nets: tp.Mapping[str, tp.Mapping[str, tp.Any]], copy_targets: bool = True) -> None:
        self._replicas_verified = False
        for n in ("actor", "successor_net"):
            if n in nets:
                getattr(self, n).load_state_dict(nets[n])
        if copy_targets:
            self._fb_targets.copy_(self._fb_params)                      # successor_target_net := successor_net (sf.py:451)
        if "feature_learner" in nets:                                    # (after the copy: latent's target_feature_net lives in the target buffer)
            self.feature_learner.load_state_dict(nets["feature_learner"])
        if "successor_target_net" in nets:
            self.successor_target_net.load_state_dict(nets["successor_target_net"])

    def train(self, training: bool = True) -> None:                      # sf.py:473-478
        self.training = training
        for net in (self.actor, self.successor_net):
            net.train(training)

    def init_from(self, other: tp.Any) -> None:                          # sf.py:480-489
        self._replicas_verified = False
        names = ["actor"] + (["successor_net", "feature_learner", "successor_target_net"] if self.cfg.init_sf else [])
        for name in names:
            getattr(self, name).load_state_dict({k: v.detach() for k, v in getattr(other, name).state_dict().items()})
        for key in ("actor_opt", "sf_opt", "phi_opt"):
            if getattr(other, key, None) is not None:
                getattr(self, key).load_state_dict(copy.deepcopy(getattr(other, key).state_dict()))

    # ---- surface (sf.py:491-592)
    def sample_z(self, size: int, device: str = "cpu") -> torch.Tensor:   # sf.py:570-573
        return math.sqrt(self.cfg.z_dim) * torch.nn.functional.normalize(torch.randn((size, self.cfg.z_dim), dtype=torch.float32, device=device), dim=1)

    def update_meta(self, meta: MetaDict, global_step: int, time_step: tp.Any, finetune: bool = False,
                    replay_loader: tp.Any = None) -> MetaDict:            # sf.py:587-592
        return self.init_meta() if global_step % self.cfg.update_z_every_step == 0 else meta

    def _backward_map(self, goal: tp.Any, target: bool = False) -> torch.Tensor:
        if self._sf_mode == 12:                  # identity features
            return self.feature_learner.feature_net(goal).reshape(-1, self.cfg.z_dim)
        return super()._backward_map(goal, target)

    def _compute_cov(self, goal: tp.Any) -> torch.Tensor:                # sf.py:509-515 (phi on the device; the d x d pinv in torch)
        phi = self._backward_map(goal)
        cov = (phi.double().T @ phi.double() / phi.shape[0]).cpu()
        return torch.linalg.pinv(cov).to(torch.float32).to(self._device)

    def precompute_cov(self, replay_loader: tp.Any) -> None:             # sf.py:491-507
        obs_list, n = [], 0
        while n < self.cfg.num_inference_steps:
            batch = replay_loader.sample(self.cfg.batch_size).to(self.cfg.device)
            obs = batch.next_goal if self.cfg.goal_space is not None else batch.next_obs
            if obs is None:
                raise ValueError("Obs should never be None")
            obs_list.append(obs)
            n += batch.next_obs.size(0)
        self.inv_cov = self._compute_cov(torch.cat(obs_list, 0))

    def get_goal_meta(self, goal_array: np.ndarray) -> MetaDict:         # sf.py:517-529
        z = self._backward_map(np.asarray(goal_array, np.float32)) @ self.inv_cov
        z = self._normalize_z(z)
        meta: tp.Dict[str, np.ndarray] = collections.OrderedDict()
        meta["z"] = z.squeeze(0).cpu().numpy()
        return meta

    def infer_meta_from_obs_and_rewards(self, obs: torch.Tensor, reward: torch.Tensor) -> MetaDict:     # sf.py:545-568
        phi = self._backward_map(obs).cpu().double()
        sol = torch.linalg.lstsq(phi, torch.as_tensor(reward).reshape(-1, 1).cpu().double()).solution     # z_dim x 1 (host: a d x d solve)
        z = math.sqrt(self.cfg.z_dim) * torch.nn.functional.normalize(sol.float(), dim=0)
        meta: tp.Dict[str, np.ndarray] = collections.OrderedDict()
        meta["z"] = z.squeeze().numpy()
        return meta

    @property
    def compute_z_correl(self) -> tp.Any:            # SFAgent has none (sf.py): hasattr() must say so (run_online asks)
        raise AttributeError("SFAgent has no compute_z_correl (sf.py)")

    def act(self, obs: tp.Any, meta: MetaDict, step: int, eval_mode: bool) -> np.ndarray:      # sf.py:594-609... the Actor path of FBDDPGAgent
        stddev = schedule(self.cfg.stddev_schedule, step)
        if not eval_mode and step < self.cfg.num_expl_steps:
            return torch.empty(self.action_dim).uniform_(-1.0, 1.0).numpy()
        return self._act_fast(np.asarray(obs, np.float32).reshape(-1), np.asarray(meta["z"], np.float32).reshape(-1), None, stddev, eval_mode)

    # ---- the hot path
    _hp_fields = operator.attrgetter("lr", "lr_coef", "sf_target_tau", "stddev_schedule", "stddev_clip", "mix_ratio", "q_loss",
                                     "num_sf_updates", "use_tb", "use_wandb", "use_hiplog")        # (SFAgentConfig, sf.py:57-96)

    def _hparams(self, step: int, want_metrics: bool, grad_scale: float, discount: float, future: float = 1.0) -> HParams:
        c = self.cfg
        if self._world() > 1 and getattr(c, "dp_global_batch", False):
            raise NotImplementedError("SFHipAgent: data parallel = gradient averaging (mode A) only; there is no global-batch loss to make exact (sf.py:594-698 has no batch x batch term)")
        # (grad_scale = 1 / world: the data-parallel schedule sums the ranks' gradient buckets, the optimiser passes average them)
        return HParams(lr=c.lr, lr_coef=c.lr_coef, fb_target_tau=c.sf_target_tau, stddev=schedule(c.stddev_schedule, step),
                       stddev_clip=c.stddev_clip, ortho_coef=1.0, mix_ratio=max(float(c.mix_ratio), 0.0), q_loss_coef=0.0, discount=discount, grad_scale=float(grad_scale),
                       q_loss=int(bool(c.q_loss)), want_metrics=int(want_metrics), future_ratio=0.0, future=float(future), rand_weight=0)

    def _metrics(self) -> tp.Dict[str, float]:                           # sf.py:627-639, 688-692
        c = self.cfg
        out: tp.Dict[str, float] = {}
        if not (c.use_tb or c.use_wandb or c.use_hiplog):
            return out
        buf = self._wait_metrics()
        g = lambda k: float(buf[_lib.METRIC_INDEX[k]])
        for k in ("target_F", "F1", "phi", "phi_norm", "z_norm", "sf_loss") + (("phi_loss",) if self._sf_mode not in (3, 12) else ()):
            out[k] = g(k)                        # (sf.py:634-635: "random" has no phi_loss)
        out["sf_opt_lr"] = self.sf_opt.param_groups[0]["lr"]
        if c.use_tb or c.use_wandb:
            out["actor_loss"], out["actor_logprob"] = g("actor_loss"), g("actor_logprob")
        return out

    def _early_grad_range(self) -> tp.Optional[tp.Tuple[int, int]]:
        return None                              # (one all-reduce per bucket: the SF backward is not cut for an early share)

    def update(self, replay_loader: tp.Any, step: int) -> tp.Dict[str, float]:                   # sf.py:700-754
        out: tp.Dict[str, float] = {}
        for _ in range(int(self.cfg.num_sf_updates)):             # :706: sample, update_sf, update_actor, target EMA -- num_sf_updates times
            out = super().update(replay_loader, step)
            if step % self.cfg.update_every_steps != 0:
                break
        return out

    def update_many(self, replay_loader: DeviceReplayBuffer, step: int, n_steps: int) -> tp.Dict[str, float]:
        c = self.cfg
        self.flush()
        total = n_steps * int(c.num_sf_updates)                  # (every update() call is num_sf_updates complete updates)
        split = self._world() > 1 or os.environ.get("FBHIP_FORCE_PHASE_SPLIT", "0") == "1"
        const_std = len({schedule(c.stddev_schedule, step + i) for i in range(n_steps)}) == 1
        from . import peer
        use_peer = self._world() > 1 and peer.enabled()
        if (split and total >= 2 and isinstance(replay_loader, DeviceReplayBuffer) and c.update_every_steps == 1 and self._use_graph and
                const_std and (use_peer or self._rccl_ready())):
            # data parallel, pipelined like the FB agent: the [sf | phi] and actor buckets are reduced INSIDE the n-step graph
            # (fbhip_update_many_dp) -- by the library's RCCL communicator (csrc/rccl.hip) or, FBHIP_DP_ALLREDUCE=peer, by the
            # peer-access kernels (csrc/peer.hip)
            want = bool(c.use_tb or c.use_wandb or c.use_hiplog)
            self._bind_replay(replay_loader)
            self._verify_replicas()
            hp = self._hparams(step, want, 1.0 / self._world(), float(replay_loader._discount), float(replay_loader._future))
            if use_peer:
                peer.bind(self)
                done = 0
                while done < total:
                    n = min(64, total - done)
                    self._on_update_stream(lambda n=n: check(_lib.load().fbhip_update_many_dp(self._ctx, C.byref(hp), n, stream_ptr()), self._ctx))
                    done += n
                self._check_peer_status(every=1)
                return self._metrics()
            if self._rccl_run(hp, total):
                return self._metrics()
            return self.update_many(replay_loader, step, n_steps)
        # (data parallel without that transport: single updates, each through the phase-split schedule with its two gradient all-reduces)
        if (total < 2 or not isinstance(replay_loader, DeviceReplayBuffer) or c.update_every_steps != 1 or not self._use_graph or
                split or not const_std):
            out: tp.Dict[str, float] = {}
            for i in range(n_steps):
                out = self.update(replay_loader, step + i)
            return out
        want = bool(c.use_tb or c.use_wandb or c.use_hiplog)
        self._bind_replay(replay_loader)
        hp = self._hparams(step, want, 1.0, float(replay_loader._discount), float(replay_loader._future))
        done = 0
        while done < total:
            n = min(64, total - done)
            self._on_update_stream(lambda n=n: check(_lib.load().fbhip_update_many(self._ctx, C.byref(hp), n, stream_ptr()), self._ctx))
            done += n
        return self._metrics()
