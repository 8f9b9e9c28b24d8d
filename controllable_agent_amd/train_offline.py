"""Counterpart of the reference's offline training loop (``url_benchmark/train_offline.py:101-134``):
``agent.update(replay, step)`` every step (``update_every_steps`` forced to 1, :59), a metrics sink every
``log_every_steps`` and the reference's own ``fps = log_every / elapsed`` meter (:120-125).  MuJoCo evaluation
and checkpoint plumbing stay with the caller (out of scope)."""
from __future__ import annotations

import time
import typing as tp

import torch


def _flush(agent: tp.Any) -> None:
    """A bare ``torch.cuda.synchronize()`` does not launch what ``FBHipAgent.update`` has queued: the loop's timing points do."""
    f = getattr(agent, "flush", None)
    if f is not None:
        f()


def run_offline(agent: tp.Any, replay_loader: tp.Any, num_grad_steps: int, log_every_steps: int = 1000,
                log_fn: tp.Optional[tp.Callable[[int, tp.Dict[str, float]], None]] = None, start_step: int = 0,
                steps_per_launch: int = 1) -> float:
    """Runs ``num_grad_steps`` updates; returns update-steps/sec over the whole call (device-synchronised).
    ``steps_per_launch > 1`` hands that many consecutive updates to ``agent.update_many`` (one hipGraph launch; the
    reference loop does nothing between two updates except the metric sink, which then sees every n-th step)."""
    agent.cfg.update_every_steps = 1                                  # train_offline.py:59
    torch.cuda.synchronize()
    t_all = t0 = time.time()
    n = max(1, int(steps_per_launch)) if hasattr(agent, "update_many") else 1
    i = 0
    while i < num_grad_steps:
        k = min(n, num_grad_steps - i)
        if log_every_steps:
            k = min(k, log_every_steps - i % log_every_steps)        # never run across a log boundary
        step = start_step + i + k - 1
        metrics = agent.update(replay_loader, step) if k == 1 else agent.update_many(replay_loader, start_step + i, k)
        i += k - 1
        if log_fn is not None and metrics:
            log_fn(step, metrics)
        if log_every_steps and (i + 1) % log_every_steps == 0:
            _flush(agent)                                             # update() calls the agent still holds back (deferred batching)
            torch.cuda.synchronize()
            now = time.time()
            if log_fn is not None:
                log_fn(step, {"fps": log_every_steps / (now - t0)})
            t0 = now
        i += 1
    _flush(agent)
    torch.cuda.synchronize()
    return num_grad_steps / (time.time() - t_all)
