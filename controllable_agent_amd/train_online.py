"""Counterpart of the reference's ONLINE pre-training loop (``url_benchmark/pretrain.py:559-659``), reduced to the
calls that touch the agent and the replay buffer -- exactly the per-environment-step sequence of the reference:

    meta   = agent.update_meta(meta, step, time_step, finetune=False, replay_loader=replay)     (:627)
    action = agent.act(time_step.observation, meta, step, eval_mode=False)                      (:629-632, no_grad + eval_mode)
    if step >= num_seed_frames / action_repeat:  metrics = agent.update(replay, step)           (:635-643)
    time_step = env.step(action);  replay.add(time_step, meta)                                  (:646-649)
    z_correl += agent.compute_z_correl(time_step, meta)                                         (:651-652)

and on episode end ``env.reset()``, ``agent.init_meta()``, ``replay.add(first_step, meta)`` (:580-608).  Evaluation,
video, logging sinks and checkpoints stay with the caller (out of scope); the environment is
whatever object offers ``reset() -> time_step`` / ``step(action) -> time_step`` with ``.observation``, ``.reward``,
``.last()`` (dm_env's contract, ``dmc.py``)."""
from __future__ import annotations

import dataclasses
import time
import typing as tp


@dataclasses.dataclass
class OnlineStats:
    env_steps: int = 0
    updates: int = 0
    episodes: int = 0
    seconds: float = 0.0
    last_episode_reward: float = 0.0
    last_z_correl: float = 0.0

    @property
    def env_steps_per_s(self) -> float:
        return self.env_steps / max(self.seconds, 1e-9)

    @property
    def updates_per_s(self) -> float:
        return self.updates / max(self.seconds, 1e-9)


class _eval_mode:                                   # utils.eval_mode (utils.py:34-47)
    def __init__(self, agent: tp.Any) -> None:
        self.agent = agent

    def __enter__(self) -> None:
        self.prev = getattr(self.agent, "training", True)
        self.agent.train(False)

    def __exit__(self, *exc: tp.Any) -> None:
        self.agent.train(self.prev)


def run_online(agent: tp.Any, replay_loader: tp.Any, env: tp.Any, num_train_frames: int, num_seed_frames: int = 4000,
               action_repeat: int = 1, start_step: int = 0,
               log_fn: tp.Optional[tp.Callable[[int, tp.Dict[str, float]], None]] = None) -> OnlineStats:
    """Runs the loop until ``num_train_frames`` (``utils.Until``: step < frames / action_repeat); returns throughput
    counters.  ``agent.update`` applies its own ``update_every_steps`` gate like the reference (fb_ddpg.py:430)."""
    until = lambda frames, step: step < frames // max(action_repeat, 1)          # utils.Until.__call__ (utils.py:207-213)
    stats = OnlineStats()
    step = start_step
    episode_reward, z_correl = 0.0, 0.0
    time_step = env.reset()
    meta = agent.init_meta()
    replay_loader.add(time_step, meta)
    t0 = time.time()
    while until(num_train_frames, step):
        if time_step.last():
            stats.episodes += 1
            stats.last_episode_reward, stats.last_z_correl = episode_reward, z_correl
            if log_fn is not None:
                log_fn(step, {"episode_reward": episode_reward, "z_correl": z_correl, "episode": stats.episodes,
                              "buffer_size": len(replay_loader)})
            time_step = env.reset()
            meta = agent.init_meta()
            replay_loader.add(time_step, meta)
            episode_reward, z_correl = 0.0, 0.0
        meta = agent.update_meta(meta, step, time_step, finetune=False, replay_loader=replay_loader)
        with _eval_mode(agent):
            action = agent.act(time_step.observation, meta, step, eval_mode=False)
        if not until(num_seed_frames, step):
            metrics = agent.update(replay_loader, step)
            if step % max(int(getattr(agent.cfg, "update_every_steps", 1)), 1) == 0:      # the agent's own gate
                stats.updates += 1
            if log_fn is not None and metrics:
                log_fn(step, metrics)
        time_step = env.step(action)
        episode_reward += float(time_step.reward)
        replay_loader.add(time_step, meta)
        if hasattr(agent, "compute_z_correl"):
            z_correl += agent.compute_z_correl(time_step, meta)
        step += 1
        stats.env_steps += 1
    stats.seconds = time.time() - t0
    return stats
