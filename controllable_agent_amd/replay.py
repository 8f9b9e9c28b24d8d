"""Device-resident replay buffer behind the reference's ``ReplayBuffer`` surface.

What has to match ``url_benchmark/in_memory_replay_buffer.py:65-216`` is the INTERFACE -- constructor, ``add``,
``sample``, ``load``, ``relabel``, ``__len__``, ``avg_episode_length``, and the private attributes the workspaces
poke (``_storage, _future, _discount, _max_episodes, _current_episode, _idx, _full, _episodes_length``;
pretrain.py:485-489, train_offline.py:92-95) -- and the RESULTS: the episode-major layout
``float32[max_episodes, T+1, dim]`` with a dummy first row per episode, and, for ``sample``, the same numpy
global-RNG draws in the same order, so a seeded run returns the reference's rows bit for bit
(tests/golden/sampler_kat.npz).  ``EpisodeBatch`` is the hand-off type of ``url_benchmark/replay_buffer.py:27-103``.

The implementation is this package's own:

* ``_storage[name]`` is a torch tensor in HBM; ``FBHipAgent.update`` samples it with a HIP kernel (``sampler.hip``), so
  the per-step host gathers and the pageable H2D copies of ``EpisodeBatch.to`` disappear from the hot path.
* The episode being collected is staged in ONE growable host block per field (``_EpisodeStage``; pinned when the
  storage is on a GPU); a finished episode goes up as a single ``[steps, dim]`` copy per field.
* ``sample`` draws indices on the host like the reference (``_draw``) and gathers every field through one helper
  (``_take``: a flat ``index_select`` on the ``[episodes * (T+1), dim]`` view, on the storage device).
* ``load`` / ``from_arrays`` / ``from_reference`` share one episode-ingestion path (``_put_episode``).
"""
from __future__ import annotations

import collections.abc
import dataclasses
import typing as tp
import weakref
from pathlib import Path

import numpy as np
import torch

Array = tp.Union[np.ndarray, torch.Tensor]

# fields of the reference's ExtendedGoalTimeStep (url_benchmark/dmc.py:35-73); anything else stored per step is "meta"
TIMESTEP_FIELDS = frozenset({"step_type", "reward", "discount", "observation", "action", "physics", "goal"})


def _to_tensor(value: tp.Any, device: tp.Union[str, torch.device]) -> tp.Any:
    if value is None:
        return None
    if isinstance(value, dict):
        return {k: _to_tensor(v, device) for k, v in value.items()}
    if isinstance(value, (np.ndarray, torch.Tensor)):
        return torch.as_tensor(value, device=device)
    raise TypeError(f"EpisodeBatch.to: cannot move a {type(value).__name__} to {device}")


@dataclasses.dataclass
class EpisodeBatch:
    """Batch of replayed transitions: same field names, defaults and methods as replay_buffer.py:27-103
    (``obs, action, reward, next_obs, discount, meta, _physics, goal, next_goal, future_obs, future_goal``)."""
    obs: Array
    action: Array
    reward: Array
    next_obs: Array
    discount: Array
    meta: tp.Dict[str, Array] = dataclasses.field(default_factory=dict)
    _physics: tp.Optional[Array] = None
    goal: tp.Optional[Array] = None
    next_goal: tp.Optional[Array] = None
    future_obs: tp.Optional[Array] = None
    future_goal: tp.Optional[Array] = None

    def __post_init__(self) -> None:
        for name in ("reward", "discount"):
            if not isinstance(getattr(self, name), (np.ndarray, torch.Tensor)):
                raise TypeError(f"EpisodeBatch.{name} must be an array or a tensor")
        if not isinstance(self.meta, dict):
            raise TypeError("EpisodeBatch.meta must be a dict")

    def to(self, device: tp.Union[str, torch.device]) -> "EpisodeBatch":
        """every array field as a tensor on ``device`` (no copy for tensors already there)"""
        return EpisodeBatch(**{f.name: _to_tensor(getattr(self, f.name), device) for f in dataclasses.fields(self)})

    def unpack(self) -> tp.Tuple[Array, Array, Array, Array, Array]:
        return self.obs, self.action, self.reward, self.discount, self.next_obs

    def with_no_reward(self) -> "EpisodeBatch":
        zero = torch.zeros_like(self.reward) if isinstance(self.reward, torch.Tensor) else np.zeros_like(self.reward)
        return dataclasses.replace(self, reward=zero)


@dataclasses.dataclass
class TimeStep:
    """Minimal stand-in for the reference's ExtendedGoalTimeStep (dmc.py:35-73) for callers without dm_env:
    ``step_type`` 0 = FIRST, 1 = MID, 2 = LAST."""
    step_type: int
    reward: float
    discount: float
    observation: np.ndarray
    action: np.ndarray
    physics: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(0, np.float32))
    goal: tp.Optional[np.ndarray] = None

    def first(self) -> bool:
        return int(self.step_type) == 0

    def last(self) -> bool:
        return int(self.step_type) == 2

    def __getitem__(self, attr: str) -> tp.Any:
        return getattr(self, attr)


def _resolve(device: tp.Union[str, torch.device]) -> torch.device:
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None and torch.cuda.is_available():
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def _step_items(time_step: tp.Any) -> tp.Iterator[tp.Tuple[str, np.ndarray]]:
    """(name, float32 row) of every storable field of a time step: scalars become 1-vectors, arrays are kept, everything
    else (``None`` goals, enums that are not numpy scalars ...) is skipped -- the filter of in_memory_replay_buffer.py:108-113"""
    if dataclasses.is_dataclass(time_step):
        pairs: tp.Iterable[tp.Tuple[str, tp.Any]] = ((f.name, getattr(time_step, f.name)) for f in dataclasses.fields(time_step))
    elif isinstance(time_step, collections.abc.Mapping):
        pairs = time_step.items()
    else:
        raise TypeError(f"unsupported time_step type {type(time_step)}")
    for name, value in pairs:
        if np.isscalar(value):
            yield name, np.full(1, value, np.float32)
        elif isinstance(value, np.ndarray):
            yield name, value.astype(np.float32, copy=False)


def _is_last(time_step: tp.Any) -> bool:
    return bool(time_step.last()) if hasattr(time_step, "last") else int(time_step["step_type"]) == 2


class _EpisodeStage:
    """The episode under collection: one growable float32 host block per field, rows appended in place.  Supports what
    callers do with the reference's ``_current_episode`` dict: ``.clear()`` (pretrain.py:485), ``name in stage``."""

    def __init__(self, pinned: bool, capacity: int = 64) -> None:
        self._pinned = pinned
        self._capacity = max(int(capacity), 2)
        self._blocks: tp.Dict[str, torch.Tensor] = {}
        self._rows: tp.Dict[str, int] = {}

    def _block(self, name: str, width: int, rows: int) -> torch.Tensor:
        blk = self._blocks.get(name)
        if blk is None or blk.shape[1] != width or blk.shape[0] < rows:
            cap = max(self._capacity, 2 * rows)
            new = torch.empty((cap, width), dtype=torch.float32, pin_memory=self._pinned)
            if blk is not None and blk.shape[1] == width:
                new[:self._rows.get(name, 0)] = blk[:self._rows.get(name, 0)]
            self._blocks[name] = blk = new
        return blk

    def append(self, name: str, row: np.ndarray) -> None:
        flat = np.asarray(row, np.float32).reshape(-1)
        n = self._rows.get(name, 0)
        self._block(name, flat.shape[0], n + 1)[n] = torch.from_numpy(np.ascontiguousarray(flat))
        self._rows[name] = n + 1

    def rows(self, name: str) -> torch.Tensor:
        return self._blocks[name][:self._rows[name]]

    def steps(self, name: str) -> int:
        return self._rows.get(name, 0)

    def names(self) -> tp.List[str]:
        return [n for n, k in self._rows.items() if k > 0]

    def clear(self) -> None:
        self._rows = {}

    def __contains__(self, name: object) -> bool:
        return self._rows.get(name, 0) > 0          # type: ignore[arg-type]

    def __len__(self) -> int:
        return max(self._rows.values(), default=0)


class DeviceReplayBuffer:
    def __init__(self, max_episodes: int, discount: float, future: float,
                 max_episode_length: tp.Optional[int] = None, device: tp.Union[str, torch.device] = "cuda") -> None:
        if not 0 <= future <= 1:
            raise ValueError(f"future must lie in [0, 1], got {future}")
        self._max_episodes = max_episodes
        self._discount = discount
        self._future = future
        self._max_episode_length = max_episode_length
        self._device = _resolve(device)
        self._storage: tp.Dict[str, torch.Tensor] = {}
        self._episodes_length = np.zeros(max_episodes, dtype=np.int32)
        self._idx = 0                       # ring position of the next finished episode
        self._full = False
        self._is_fixed_episode_length = True
        self._collected_episodes = 0
        self._batch_names = set(TIMESTEP_FIELDS)
        self._episodes_selection_probability: tp.Optional[np.ndarray] = None
        self._current_episode = _EpisodeStage(pinned=self._device.type == "cuda" and torch.cuda.is_available())
        self._version = 0                   # bumped on every mutation; FBHipAgent re-binds device pointers when it changes
        self._dev_cache: tp.Optional[tp.Dict[str, tp.Any]] = None

    # ------------------------------------------------------------------ observers
    # An FBHipAgent holds consecutive ``update()`` calls back and launches them as one n-step graph (agent.py "deferred
    # batching"); the batches of those updates are drawn from THIS buffer when they run.  Every mutation therefore first asks
    # the agents with calls in the queue to launch them: queued updates sample the contents they were called on.
    def _observe(self, agent: tp.Any) -> None:
        obs = self.__dict__.get("_observers")
        if obs is None:
            obs = self.__dict__["_observers"] = weakref.WeakSet()
        obs.add(agent)

    def _unobserve(self, agent: tp.Any) -> None:
        obs = self.__dict__.get("_observers")
        if obs is not None:
            obs.discard(agent)

    def _before_mutation(self) -> None:
        obs = self.__dict__.get("_observers")
        if obs:
            for agent in list(obs):
                agent.flush()

    # ------------------------------------------------------------------ bookkeeping
    def __len__(self) -> int:
        return self._max_episodes if self._full else self._idx

    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def avg_episode_length(self) -> int:
        stored = self._episodes_length[:len(self)]
        return round(float(stored.sum(dtype=np.int64)) / len(stored))

    def _touch(self) -> None:
        self._before_mutation()
        self._version += 1
        self._dev_cache = None
        self._episodes_selection_probability = None

    def _advance_ring(self, steps: int) -> None:
        """book a finished episode of ``steps`` transitions into ring slot ``_idx``"""
        before = int(self._episodes_length[self._idx - 1])            # slot -1 wraps to the newest slot of a full ring
        if before and before != steps:
            self._is_fixed_episode_length = False
        self._episodes_length[self._idx] = steps
        self._idx = (self._idx + 1) % self._max_episodes
        self._full = self._full or self._idx == 0

    def _slot(self, name: str, row_shape: tp.Tuple[int, ...], rows: int) -> torch.Tensor:
        """storage tensor of a field, allocated (zeroed) on first use: ``[max_episodes, rows or max_episode_length, *row_shape]``"""
        if name not in self._storage:
            depth = self._max_episode_length if self._max_episode_length is not None else rows
            self._storage[name] = torch.zeros((self._max_episodes, depth) + tuple(row_shape), dtype=torch.float32,
                                              device=self._device)
        return self._storage[name]

    def _put_episode(self, episode: tp.Mapping[str, tp.Any], steps: tp.Optional[int] = None) -> None:
        """one finished episode (``{name: [rows, dim]}``, host arrays or tensors) -> ring slot ``_idx``"""
        self._before_mutation()
        n_rows = 0
        for name, values in episode.items():
            block = values if isinstance(values, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(values, dtype=np.float32))
            n_rows = max(n_rows, block.shape[0])
            dst = self._slot(name, tuple(block.shape[1:]), block.shape[0])
            dst[self._idx, :block.shape[0]].copy_(block, non_blocking=block.is_pinned())
        self._advance_ring(n_rows - 1 if steps is None else steps)

    # ------------------------------------------------------------------ pickling (workspaces torch.save the buffer object)
    def __getstate__(self) -> tp.Dict[str, tp.Any]:
        state = {k: v for k, v in self.__dict__.items() if k not in ("_dev_cache", "_current_episode", "_observers")}
        state["_storage"] = {k: v.cpu().numpy() for k, v in self._storage.items()}
        state["_device"] = str(self._device)
        return state

    def __setstate__(self, state: tp.Dict[str, tp.Any]) -> None:
        self._before_mutation()
        dev = torch.device(state.get("_device", "cuda"))
        if dev.type == "cuda" and not torch.cuda.is_available():
            dev = torch.device("cpu")
        self.__dict__.update(state)
        self._device = _resolve(dev)
        self._storage = {k: torch.as_tensor(np.asarray(v, dtype=np.float32), device=self._device)
                         for k, v in state["_storage"].items()}
        self._dev_cache = None
        self._current_episode = _EpisodeStage(pinned=self._device.type == "cuda")
        if self._storage and "_episodes_length" not in state:
            # buffers pickled before episode lengths existed (in_memory_replay_buffer.py:95-102): every stored episode is full
            stored = self._storage["discount"]
            self._episodes_length = np.zeros(stored.shape[0], np.int32)
            self._episodes_length[:len(self)] = stored.shape[1] - 1
            self._episodes_selection_probability = None
            self._is_fixed_episode_length = True
            self._max_episode_length = None

    # ------------------------------------------------------------------ add (in_memory_replay_buffer.py:104-133)
    def add(self, time_step: tp.Any, meta: tp.Mapping[str, np.ndarray]) -> None:
        stage = self._current_episode
        if _is_last(time_step):
            # the transition that completes an episode changes the ring: queued updates of an observing agent go out FIRST, before
            # anything of this call is staged -- if their launch raises, the buffer is exactly as it was before the call
            self._before_mutation()
        for name, value in meta.items():
            stage.append(name, np.asarray(value, np.float32))
        for name, row in _step_items(time_step):
            stage.append(name, row)
        if not _is_last(time_step):
            return
        # the episode is complete: one [steps + 1, dim] host->device block per field
        steps = stage.steps("discount") - 1                                 # row 0 is the dummy FIRST step
        self._put_episode({name: stage.rows(name) for name in stage.names()}, steps=steps)
        if self._device.type == "cuda":
            torch.cuda.current_stream(self._device).synchronize()          # the stage is reused by the next episode
        stage.clear()
        self._collected_episodes += 1
        self._touch()

    # ------------------------------------------------------------------ sample (in_memory_replay_buffer.py:139-190)
    def _draw(self, batch_size: int) -> tp.Tuple[np.ndarray, np.ndarray, tp.Optional[np.ndarray]]:
        """(episode, step, future step) indices from numpy's GLOBAL generator, call for call like the reference (:146-161):
        ``randint`` over episodes (``choice`` weighted by length when lengths vary), ``randint(0, length) + 1``, and with
        hindsight replay ``step + geometric(1 - future)`` clipped to the episode."""
        if isinstance(self._future, bool):
            self._future = float(self._future)
        if self._is_fixed_episode_length:
            episode = np.random.randint(0, len(self), size=batch_size)
        else:
            if self._episodes_selection_probability is None:
                self._episodes_selection_probability = self._episodes_length / self._episodes_length.sum()
            episode = np.random.choice(np.arange(len(self._episodes_length)), size=batch_size,
                                       p=self._episodes_selection_probability)
        lengths = self._episodes_length[episode]
        step = np.random.randint(0, lengths) + 1
        future = None
        if self._future < 1:
            future = np.clip(step + np.random.geometric(p=(1 - self._future), size=batch_size), 0, lengths)
        return episode, step, future

    def _take(self, name: str, flat_index: torch.Tensor) -> torch.Tensor:
        """rows ``flat_index`` (= episode * depth + step) of a field, gathered on the storage device"""
        data = self._storage[name]
        return data.reshape(data.shape[0] * data.shape[1], *data.shape[2:]).index_select(0, flat_index)

    def sample(self, batch_size: int, custom_reward: tp.Optional[tp.Any] = None,
               with_physics: bool = False) -> EpisodeBatch:
        """A batch with the reference's field semantics: ``obs`` / ``goal`` / stored meta from row ``step - 1``; ``action``,
        ``reward``, ``next_obs``, ``next_goal``, physics and ``discount`` (times ``_discount``) from row ``step``;
        ``future_*`` from row ``future - 1``.  (``FBHipAgent.update`` does not come through here: it samples inside the
        fused HIP step.)"""
        episode, step, future = self._draw(batch_size)
        depth = self._storage["observation"].shape[1]
        base = torch.as_tensor(episode, dtype=torch.long, device=self._device) * depth
        at_step = base + torch.as_tensor(step, dtype=torch.long, device=self._device)
        before = at_step - 1
        has_goal = "goal" in self._storage
        fields: tp.Dict[str, tp.Any] = dict(
            obs=self._take("observation", before), next_obs=self._take("observation", at_step),
            action=self._take("action", at_step), discount=self._discount * self._take("discount", at_step),
            meta={name: self._take(name, before) for name in self._storage if name not in self._batch_names})
        if has_goal:
            fields.update(goal=self._take("goal", before), next_goal=self._take("goal", at_step))
        if future is not None:
            at_future = base + torch.as_tensor(future, dtype=torch.long, device=self._device) - 1
            fields["future_obs"] = self._take("observation", at_future)
            if has_goal:
                fields["future_goal"] = self._take("goal", at_future)
        physics = self._take("physics", at_step) if "physics" in self._storage else None
        if custom_reward is None:
            fields["reward"] = self._take("reward", at_step)
        else:
            if physics is None:
                raise ValueError("custom_reward needs stored physics")
            fields["reward"] = self._rewards_of(custom_reward, physics)
        if with_physics:
            fields["_physics"] = physics
        return EpisodeBatch(**fields)

    def _rewards_of(self, custom_reward: tp.Any, physics: torch.Tensor) -> torch.Tensor:
        """``custom_reward.from_physics`` (goals.py:233-237, host Python) per row -> ``[rows, 1]`` on the storage device"""
        host = physics.detach().cpu().numpy()
        out = np.fromiter((custom_reward.from_physics(p) for p in host), dtype=np.float32, count=host.shape[0])
        return torch.as_tensor(out.reshape(-1, 1), device=self._device)

    # ------------------------------------------------------------------ load / relabel (in_memory_replay_buffer.py:192-216)
    def load(self, env: tp.Any, replay_dir: Path, relabel: bool = True, goal_func: tp.Any = None) -> None:
        """Fill the ring from ExORL-style per-episode ``*.npz`` files (keys observation / action / reward / discount /
        physics, each ``[T+1, dim]``) in sorted order until it is full.  ``relabel=True`` recomputes rewards (and goals via
        ``goal_func``) from the stored physics and therefore needs the MuJoCo ``env``, as in the reference."""
        if relabel and env is None:
            raise ValueError("relabel=True needs an environment (MuJoCo physics), as in the reference")
        for path in sorted(Path(replay_dir).glob("*.npz")):
            if self._full:
                break
            with np.load(path) as data:
                episode = {k: np.asarray(data[k]) for k in data.files}
            if relabel:
                episode = _relabel_episode(env, episode, goal_func)
            self._put_episode(episode)
        self._touch()

    def relabel(self, custom_reward: tp.Any) -> None:
        self._before_mutation()
        physics = self._storage["physics"]
        n, depth = physics.shape[:2]
        self._storage["reward"] = self._rewards_of(custom_reward, physics.reshape(n * depth, -1)).reshape(n, depth, 1)
        self._max_episodes = n
        self._full = True
        self._touch()

    # ------------------------------------------------------------------ ingestion / sharding / device view
    def _adopt(self, storage: tp.Mapping[str, tp.Any], lengths: tp.Optional[np.ndarray]) -> None:
        """take complete episode-major arrays as the storage"""
        self._before_mutation()
        self._storage = {name: torch.as_tensor(np.asarray(arr, dtype=np.float32), device=self._device)
                         for name, arr in storage.items()}
        if lengths is None:                                    # no record: every stored episode is full length
            any_field = next(iter(self._storage.values()))
            lengths = np.zeros(any_field.shape[0], np.int32)
            lengths[:len(self)] = any_field.shape[1] - 1
        self._episodes_length = np.asarray(lengths, dtype=np.int32).copy()

    @classmethod
    def from_reference(cls, other: tp.Any, device: tp.Union[str, torch.device] = "cuda") -> "DeviceReplayBuffer":
        """Build from a reference ``ReplayBuffer`` (or anything exposing the same private attributes)."""
        rb = cls(other._max_episodes, other._discount, float(other._future),
                 getattr(other, "_max_episode_length", None), device=device)
        rb._idx, rb._full = other._idx, other._full
        rb._adopt(other._storage, getattr(other, "_episodes_length", None))
        rb._is_fixed_episode_length = bool(getattr(other, "_is_fixed_episode_length", True))
        rb._touch()
        return rb

    @classmethod
    def from_reference_file(cls, path: tp.Union[str, Path], device: tp.Union[str, torch.device] = "cuda",
                            discount: tp.Optional[float] = None, future: tp.Optional[float] = None) -> "DeviceReplayBuffer":
        """Ingest a file the reference wrote with ``torch.save``: a pickled ``ReplayBuffer`` (``replay.pt`` /
        ``relabeled_replay_*.pt``, train_offline.py:88-90) or a checkpoint dict holding one under ``'replay_loader'``
        (pretrain.py:437-449) -- without the reference installed (``reference_io``).  Mirrors what
        ``Workspace.load_checkpoint`` does to the buffer afterwards (pretrain.py:480-489): pending episode dropped,
        ``_discount`` / ``_future`` taken from the caller's config when given, ``_max_episodes`` = stored episodes."""
        from . import reference_io
        parts = reference_io.payload_parts(reference_io.load_reference_payload(path))
        if "replay_loader" not in parts:
            raise KeyError(f"{path}: no 'replay_loader' in the payload (keys: {sorted(parts)})")
        other = parts["replay_loader"]
        rb = cls.from_reference(other, device=device)
        if discount is not None:
            rb._discount = float(discount)
        if future is not None:
            rb._future = float(future)
        rb._max_episodes = int(next(iter(rb._storage.values())).shape[0]) if rb._storage else rb._max_episodes
        return rb

    @classmethod
    def from_arrays(cls, storage: tp.Mapping[str, np.ndarray], episode_lengths: np.ndarray, discount: float,
                    future: float = 1.0, device: tp.Union[str, torch.device] = "cuda") -> "DeviceReplayBuffer":
        """A full buffer from episode-major arrays ``[n_episodes, T+1, dim]``."""
        n = next(iter(storage.values())).shape[0]
        rb = cls(n, discount, future, device=device)
        rb._idx, rb._full = 0, True
        rb._adopt(storage, episode_lengths)
        rb._is_fixed_episode_length = bool((rb._episodes_length == rb._episodes_length[0]).all())
        rb._touch()
        return rb

    def shard(self, rank: int, world_size: int) -> "DeviceReplayBuffer":
        """Episodes ``ep % world_size == rank``: every rank samples its own shard."""
        n = len(self)
        keep = np.arange(rank, n, world_size)
        rb = DeviceReplayBuffer(len(keep), self._discount, float(self._future), self._max_episode_length, self._device)
        k = torch.as_tensor(keep, device=self._device, dtype=torch.long)
        rb._storage = {name: data[k].contiguous() for name, data in self._storage.items()}
        rb._episodes_length = self._episodes_length[keep].copy()
        rb._is_fixed_episode_length = bool((rb._episodes_length == rb._episodes_length[0]).all()) if len(keep) else True
        rb._idx, rb._full = 0, True
        rb._touch()
        return rb

    def device_view(self) -> tp.Dict[str, tp.Any]:
        """Pointers-to-be for ``fbhip_replay_bind``: contiguous storage tensors, episode lengths and their
        exclusive prefix sum on the device.  Cached until the buffer is mutated."""
        if self._dev_cache is not None:
            return self._dev_cache
        if "observation" not in self._storage:
            raise RuntimeError("replay buffer is empty")
        n = len(self)
        if n == 0:
            raise RuntimeError("replay buffer holds no finished episode")
        lens = self._episodes_length[:n].astype(np.int32)
        if (lens < 1).any():
            raise RuntimeError("replay buffer holds an empty episode")
        cum = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=cum[1:])
        view = {name: self._storage[name] for name in ("observation", "action", "discount")}
        view["goal"] = self._storage.get("goal")
        for k, v in list(view.items()):
            if v is not None:
                assert v.is_contiguous() and v.dtype == torch.float32
        view.update(episode_len=torch.as_tensor(lens, device=self._device),
                    cum_len=torch.as_tensor(cum, device=self._device),
                    n_episodes=n, t1=self._storage["observation"].shape[1],
                    fixed_length=bool((lens == lens[0]).all()))
        self._dev_cache = view
        return view


def _relabel_episode(env: tp.Any, episode: tp.Dict[str, np.ndarray], goal_func: tp.Any) -> tp.Dict[str, np.ndarray]:
    """Recompute ``reward`` (and ``goal`` when ``goal_func`` is given) of an episode by replaying its stored physics states
    through the MuJoCo task (what in_memory_replay_buffer.py:40-55 does; host CPU, needs dm_control)."""
    states = episode["physics"]
    rewards = np.empty((states.shape[0], 1), np.float32)
    goals: tp.List[np.ndarray] = []
    for i, state in enumerate(states):
        with env.physics.reset_context():
            env.physics.set_state(state)
        rewards[i, 0] = env.task.get_reward(env.physics)
        if goal_func is not None:
            goals.append(np.asarray(goal_func(env), np.float32))
    out = dict(episode, reward=rewards)
    if goals:
        out["goal"] = np.stack(goals)
    return out
