"""Counterpart of the reference's offline training loop (``url_benchmark/train_offline.py:101-134``):
``agent.update(replay, step)`` every step (``update_every_steps`` forced to 1, :59), a metrics sink every
``log_every_steps`` and the reference's own ``fps = log_every / elapsed`` meter (:120-125).  MuJoCo evaluation
and checkpoint plumbing stay with the caller (out of scope)."""
from __future__ import annotations

import time
import typing as tp

import torch


def run_offline(agent: tp.Any, replay_loader: tp.Any, num_grad_steps: int, log_every_steps: int = 1000,
                log_fn: tp.Optional[tp.Callable[[int, tp.Dict[str, float]], None]] = None, start_step: int = 0) -> float:
    """Runs ``num_grad_steps`` updates; returns update-steps/sec over the whole call (device-synchronised)."""
    agent.cfg.update_every_steps = 1                                  # train_offline.py:59
    torch.cuda.synchronize()
    t_all = t0 = time.time()
    for i in range(num_grad_steps):
        step = start_step + i
        metrics = agent.update(replay_loader, step)
        if log_fn is not None and metrics:
            log_fn(step, metrics)
        if log_every_steps and (i + 1) % log_every_steps == 0:
            torch.cuda.synchronize()
            now = time.time()
            if log_fn is not None:
                log_fn(step, {"fps": log_every_steps / (now - t0)})
            t0 = now
    torch.cuda.synchronize()
    return num_grad_steps / (time.time() - t_all)
