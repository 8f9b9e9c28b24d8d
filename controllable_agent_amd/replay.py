"""Device-resident replay buffer with the reference's ``ReplayBuffer`` surface.

Mirrors ``url_benchmark/in_memory_replay_buffer.py:65-216`` (constructor, ``add``, ``sample``, ``load``,
``relabel``, ``__len__``, ``avg_episode_length`` and the private attributes workspaces poke:
``_storage, _future, _discount, _max_episodes, _current_episode, _idx, _full, _episodes_length``) and the
``EpisodeBatch`` hand-off type of ``url_benchmark/replay_buffer.py:27-103``.

Difference by design: ``_storage[name]`` is a torch tensor living in HBM (episode-major
``float32[max_episodes, T+1, dim]``, same layout as the reference's numpy arrays), so ``FBHipAgent.update``
samples it with a HIP kernel and the per-step pageable H2D copies of ``EpisodeBatch.to`` disappear.  Finished
episodes are appended host->device one ``[T+1, dim]`` block at a time (online training, pretrain.py:649).
"""
from __future__ import annotations

import collections
import dataclasses
import typing as tp
from pathlib import Path

import numpy as np
import torch

T = tp.TypeVar("T", np.ndarray, torch.Tensor)
B = tp.TypeVar("B", bound="EpisodeBatch")

# fields of the reference's ExtendedGoalTimeStep (url_benchmark/dmc.py:35-73): everything else in an episode is "meta"
TIMESTEP_FIELDS = frozenset({"step_type", "reward", "discount", "observation", "action", "physics", "goal"})


@dataclasses.dataclass
class EpisodeBatch(tp.Generic[T]):
    """replay_buffer.py:27-48: a container for batchable replayed transitions"""
    obs: T
    action: T
    reward: T
    next_obs: T
    discount: T
    meta: tp.Dict[str, T] = dataclasses.field(default_factory=dict)
    _physics: tp.Optional[T] = None
    goal: tp.Optional[T] = None
    next_goal: tp.Optional[T] = None
    future_obs: tp.Optional[T] = None
    future_goal: tp.Optional[T] = None

    def __post_init__(self) -> None:
        assert isinstance(self.reward, (np.ndarray, torch.Tensor))
        assert isinstance(self.discount, (np.ndarray, torch.Tensor))
        assert isinstance(self.meta, dict)

    def to(self, device: tp.Union[str, torch.device]) -> "EpisodeBatch[torch.Tensor]":
        """replay_buffer.py:50-63 (a no-op copy when the fields already live on ``device``)"""
        out: tp.Dict[str, tp.Any] = {}
        for field in dataclasses.fields(self):
            data = getattr(self, field.name)
            if field.name == "meta":
                out[field.name] = {x: torch.as_tensor(y, device=device) for x, y in data.items()}
            elif isinstance(data, (torch.Tensor, np.ndarray)):
                out[field.name] = torch.as_tensor(data, device=device)
            elif data is None:
                out[field.name] = data
            else:
                raise RuntimeError(f"Not sure what to do with {field.name}: {data}")
        return EpisodeBatch(**out)

    def unpack(self) -> tp.Tuple[T, T, T, T, T]:
        return (self.obs, self.action, self.reward, self.discount, self.next_obs)

    def with_no_reward(self: B) -> B:
        reward = self.reward
        reward = torch.zeros_like(reward) if isinstance(reward, torch.Tensor) else 0 * reward
        return dataclasses.replace(self, reward=reward)


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


def _fields_of(time_step: tp.Any) -> tp.Iterable[tp.Tuple[str, tp.Any]]:
    if dataclasses.is_dataclass(time_step):
        return [(f.name, getattr(time_step, f.name)) for f in dataclasses.fields(time_step)]
    if isinstance(time_step, collections.abc.Mapping):
        return list(time_step.items())
    raise TypeError(f"unsupported time_step type {type(time_step)}")


def _is_last(time_step: tp.Any) -> bool:
    if hasattr(time_step, "last"):
        return bool(time_step.last())
    return int(time_step["step_type"]) == 2


class DeviceReplayBuffer:
    def __init__(self, max_episodes: int, discount: float, future: float,
                 max_episode_length: tp.Optional[int] = None, device: tp.Union[str, torch.device] = "cuda") -> None:
        self._max_episodes = max_episodes
        self._discount = discount
        assert 0 <= future <= 1
        self._future = future
        self._current_episode: tp.Dict[str, tp.List[np.ndarray]] = collections.defaultdict(list)
        self._idx = 0
        self._full = False
        self._num_transitions = 0
        self._storage: tp.Dict[str, torch.Tensor] = {}
        self._collected_episodes = 0
        self._batch_names = set(TIMESTEP_FIELDS)
        self._episodes_length = np.zeros(max_episodes, dtype=np.int32)
        self._episodes_selection_probability = None
        self._is_fixed_episode_length = True
        self._max_episode_length = max_episode_length
        self._device = _resolve(device)
        self._version = 0            # bumped on every mutation; FBHipAgent re-binds device pointers when it changes
        self._dev_cache: tp.Optional[tp.Dict[str, tp.Any]] = None

    # ------------------------------------------------------------------ bookkeeping (:88-102, 135-137)
    def __len__(self) -> int:
        return self._max_episodes if self._full else self._idx

    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def avg_episode_length(self) -> int:
        return round(self._episodes_length[:len(self)].mean())

    def _touch(self) -> None:
        self._version += 1
        self._dev_cache = None
        self._episodes_selection_probability = None

    def __getstate__(self) -> tp.Dict[str, tp.Any]:
        state = dict(self.__dict__)
        state["_storage"] = {k: v.cpu().numpy() for k, v in self._storage.items()}
        state["_dev_cache"] = None
        state["_device"] = str(self._device)
        return state

    def __setstate__(self, state: tp.Dict[str, tp.Any]) -> None:
        dev = torch.device(state.get("_device", "cuda"))
        if dev.type == "cuda" and not torch.cuda.is_available():
            dev = torch.device("cpu")
        dev = _resolve(dev)
        self.__dict__.update(state)
        self._device = dev
        self._storage = {k: torch.as_tensor(np.asarray(v, dtype=np.float32), device=dev) for k, v in state["_storage"].items()}
        self._dev_cache = None
        self._backward_compatibility()

    def _backward_compatibility(self) -> None:          # in_memory_replay_buffer.py:95-102
        if self._storage and not hasattr(self, "_episodes_length"):
            n = self._storage["discount"].shape[1] - 1
            self._episodes_length = np.full(self._storage["discount"].shape[0], n, dtype=np.int32)
            self._episodes_length[len(self):] = 0
            self._episodes_selection_probability = None
            self._is_fixed_episode_length = True
            self._max_episode_length = None

    # ------------------------------------------------------------------ add (:104-133)
    def add(self, time_step: tp.Any, meta: tp.Mapping[str, np.ndarray]) -> None:
        dtype = np.float32
        for key, value in meta.items():
            self._current_episode[key].append(value)
        for name, value in _fields_of(time_step):
            if np.isscalar(value):
                value = np.full((1,), value, dtype=dtype)
            if isinstance(value, np.ndarray):
                self._current_episode[name].append(np.array(value, dtype=dtype))
        if _is_last(time_step):
            for name, value_list in self._current_episode.items():
                values = np.array(value_list, dtype)
                if name not in self._storage:
                    _shape = values.shape
                    if self._max_episode_length is not None:
                        _shape = (self._max_episode_length,) + _shape[1:]
                    self._storage[name] = torch.zeros((self._max_episodes,) + _shape, dtype=torch.float32,
                                                      device=self._device)
                # one [T+1, dim] host->device block per finished episode
                self._storage[name][self._idx, :len(values)] = torch.from_numpy(values).to(self._device)
            self._episodes_length[self._idx] = len(self._current_episode["discount"]) - 1   # dummy first transition
            if self._episodes_length[self._idx] != self._episodes_length[self._idx - 1] \
                    and self._episodes_length[self._idx - 1] != 0:
                self._is_fixed_episode_length = False
            self._current_episode = collections.defaultdict(list)
            self._collected_episodes += 1
            self._idx = (self._idx + 1) % self._max_episodes
            self._full = self._full or self._idx == 0
            self._touch()

    # ------------------------------------------------------------------ sample (:139-190)
    def sample(self, batch_size: int, custom_reward: tp.Optional[tp.Any] = None,
               with_physics: bool = False) -> EpisodeBatch:
        """Same numpy-global-RNG index draws as the reference; rows are gathered on the storage device.
        (``FBHipAgent.update`` does not call this: it samples inside the fused HIP step.)"""
        if not isinstance(self._future, float):
            assert isinstance(self._future, bool)
            self._future = float(self._future)
        if self._is_fixed_episode_length:
            ep_idx = np.random.randint(0, len(self), size=batch_size)
        else:
            if self._episodes_selection_probability is None:
                self._episodes_selection_probability = self._episodes_length / self._episodes_length.sum()
            ep_idx = np.random.choice(np.arange(len(self._episodes_length)), size=batch_size,
                                      p=self._episodes_selection_probability)
        eps_lengths = self._episodes_length[ep_idx]
        step_idx = np.random.randint(0, eps_lengths) + 1            # +1 for the first dummy transition
        assert (step_idx <= eps_lengths).all()
        if self._future < 1:
            future_idx = step_idx + np.random.geometric(p=(1 - self._future), size=batch_size)
            future_idx = np.clip(future_idx, 0, eps_lengths)
            assert (future_idx <= eps_lengths).all()
        e = torch.as_tensor(ep_idx, device=self._device, dtype=torch.long)
        s = torch.as_tensor(step_idx, device=self._device, dtype=torch.long)
        st = self._storage
        meta = {name: data[e, s - 1] for name, data in st.items() if name not in self._batch_names}
        obs = st["observation"][e, s - 1]
        action = st["action"][e, s]
        next_obs = st["observation"][e, s]
        phy = st["physics"][e, s] if "physics" in st else None
        if custom_reward is not None:
            assert phy is not None, "custom_reward needs stored physics"
            reward = torch.as_tensor(np.array([[custom_reward.from_physics(p)] for p in phy.cpu().numpy()],
                                              dtype=np.float32), device=self._device)
        else:
            reward = st["reward"][e, s]
        discount = self._discount * st["discount"][e, s]
        goal = next_goal = future_obs = future_goal = None
        if "goal" in st:
            goal = st["goal"][e, s - 1]
            next_goal = st["goal"][e, s]
        if self._future < 1:
            f = torch.as_tensor(future_idx, device=self._device, dtype=torch.long)
            future_obs = st["observation"][e, f - 1]
            if "goal" in st:
                future_goal = st["goal"][e, f - 1]
        additional = {}
        if with_physics:
            additional["_physics"] = phy
        return EpisodeBatch(obs=obs, goal=goal, action=action, reward=reward, discount=discount, next_obs=next_obs,
                            next_goal=next_goal, future_obs=future_obs, future_goal=future_goal, meta=meta, **additional)

    # ------------------------------------------------------------------ load / relabel (:192-216)
    def load(self, env: tp.Any, replay_dir: Path, relabel: bool = True, goal_func: tp.Any = None) -> None:
        """Ingest ExORL-style per-episode ``*.npz`` files (keys observation/action/reward/discount/physics,
        each [T+1, dim]).  ``relabel=True`` needs a MuJoCo ``env`` exactly like the reference."""
        eps_fns = sorted(Path(replay_dir).glob("*.npz"))
        for eps_fn in eps_fns:
            if self._full:
                break
            with eps_fn.open("rb") as f:
                ep = np.load(f)
                episode = {k: ep[k] for k in ep.keys()}
            if relabel:
                if env is None:
                    raise ValueError("relabel=True needs an environment (MuJoCo physics), as in the reference")
                episode = _relabel_episode(env, episode, goal_func)
            for name, values in episode.items():
                if name not in self._storage:
                    self._storage[name] = torch.zeros((self._max_episodes,) + values.shape, dtype=torch.float32,
                                                      device=self._device)
                self._storage[name][self._idx] = torch.as_tensor(np.array(values, dtype=np.float32), device=self._device)
            self._episodes_length[self._idx] = next(iter(episode.values())).shape[0] - 1
            self._idx = (self._idx + 1) % self._max_episodes
            self._full = self._full or self._idx == 0
        self._touch()

    def relabel(self, custom_reward: tp.Any) -> None:
        phys = self._storage["physics"].cpu().numpy()
        for ep_idx, phy in enumerate(phys):
            reward = np.array([[custom_reward.from_physics(p)] for p in phy], dtype=np.float32)
            self._storage["reward"][ep_idx] = torch.as_tensor(reward, device=self._device)
        self._max_episodes = len(phys)
        self._full = True
        self._touch()

    # ------------------------------------------------------------------ ingestion / sharding / device view
    @classmethod
    def from_reference(cls, other: tp.Any, device: tp.Union[str, torch.device] = "cuda") -> "DeviceReplayBuffer":
        """Build from a reference ``ReplayBuffer`` (or anything exposing the same private attributes)."""
        rb = cls(other._max_episodes, other._discount, float(other._future),
                 getattr(other, "_max_episode_length", None), device=device)
        for name, arr in other._storage.items():
            rb._storage[name] = torch.as_tensor(np.asarray(arr, dtype=np.float32), device=rb._device)
        rb._idx, rb._full = other._idx, other._full
        n_store = next(iter(rb._storage.values())).shape[0] if rb._storage else other._max_episodes
        lens = getattr(other, "_episodes_length", None)
        if lens is None:
            lens = np.full(n_store, next(iter(rb._storage.values())).shape[1] - 1, np.int32)
            lens[len(rb):] = 0
        rb._episodes_length = np.asarray(lens, dtype=np.int32).copy()
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
        for name, arr in storage.items():
            rb._storage[name] = torch.as_tensor(np.asarray(arr, dtype=np.float32), device=rb._device)
        rb._episodes_length = np.asarray(episode_lengths, dtype=np.int32).copy()
        rb._is_fixed_episode_length = bool((rb._episodes_length == rb._episodes_length[0]).all())
        rb._idx, rb._full = 0, True
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
    """in_memory_replay_buffer.py:40-55 (needs MuJoCo physics; host CPU)."""
    goals, rewards = [], []
    states = episode["physics"]
    for i in range(states.shape[0]):
        with env.physics.reset_context():
            env.physics.set_state(states[i])
        reward = env.task.get_reward(env.physics)
        rewards.append(np.full((1,), reward, dtype=np.float32))
        if goal_func is not None:
            goals.append(goal_func(env))
    episode["reward"] = np.array(rewards, dtype=np.float32)
    if goals:
        episode["goal"] = np.array(goals, dtype=np.float32)
    return episode
