"""Data-parallel schedules of one FB-DDPG update (modes A and B).

The reference has no distributed code at all; this is new design for MI355X nodes: one process per GPU,
``torch.distributed`` (backend "nccl" == RCCL over xGMI), replay episodes sharded ``ep % world == rank``,
parameters / Adam state / targets replicated.  Every rank computes the FB loss on its OWN batch x batch block and
the two flat gradient buckets are sum-all-reduced:

    phase SAMPLE | FB_GRAD | ACTOR_FWD -> all_reduce(fb_grads)  (forward_net ++ backward_net, 14.7 MB at walker dims)
    phase FB_STEP | ACTOR_GRAD  -> all_reduce(actor_grads)   ( 8.9 MB)
    phase ACTOR_STEP

``1 / world`` is folded into the Adam pass (``grad_scale``), so the optimiser steps are bit-identical on every
rank and no parameter broadcast is ever needed.  This equals ONE device fed the same ``world`` micro-batches with
gradient averaging -- not the single-device loss on the concatenated batch (the contrastive off-diagonal mean
runs over world * B(B-1) pairs instead of (world*B)(world*B - 1)).

Mode B (``exchange`` given; ``FBHipAgent(dp_global_batch=True)``) IS the single-device loss on the concatenated batch: the
FB / orthonormality losses couple every row with every other row (fb_ddpg.py:313-326, 344-346), so after the forward
passes the ranks all-gather their six ``[B, d]`` embedding panels and discounts (7 x B x d floats per rank, 1.4 MB at
walker dims) and every rank evaluates ITS rows of the (world*B) x (world*B) loss -- each rank's dF / dB rows come out
complete, no reduce-scatter.  Normalisers are global, so the FB bucket is summed with grad_scale 1:

    phase SAMPLE | FB_FWD -> all_gather(embeddings) -> phase FB_BWD | ACTOR_FWD -> all_reduce(fb_grads)
    phase FB_STEP | ACTOR_GRAD -> all_reduce(actor_grads) -> phase ACTOR_STEP

The pairwise work per rank grows with world (B x world*B tiles): that is the price of the exact global-batch loss.
"""
from __future__ import annotations

import typing as tp

import torch

PHASE_SAMPLE, PHASE_FB_FWD, PHASE_FB_STEP, PHASE_ACTOR_GRAD, PHASE_ACTOR_STEP, PHASE_ACTOR_FWD, PHASE_FB_BWD = 1, 2, 4, 8, 16, 32, 64
PHASE_FB_GRAD = PHASE_FB_FWD | PHASE_FB_BWD
PHASE_ALL = 127


def world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def dp_update(run_phases: tp.Callable[[int], None], fb_grads: torch.Tensor, actor_grads: torch.Tensor,
              exchange: tp.Optional[tp.Callable[[], None]] = None) -> None:
    """Run one update through ``run_phases(mask)``; with world_size > 1 the two gradient buckets are
    sum-all-reduced between the phases (``run_phases`` applies grad_scale = 1/world in its optimiser steps).
    ``exchange`` (mode B): called between FB_FWD and FB_BWD; it all-gathers the embeddings and binds the global batch
    (``run_phases`` must then apply grad_scale = 1 in the FB optimiser step)."""
    import os
    import torch.distributed as dist
    world = world_size()
    if exchange is not None:
        reduce = dist.all_reduce if (dist.is_available() and dist.is_initialized()) else (lambda t: None)
        run_phases(PHASE_SAMPLE | PHASE_FB_FWD)
        exchange()
        run_phases(PHASE_FB_BWD | PHASE_ACTOR_FWD)
        reduce(fb_grads)
        run_phases(PHASE_FB_STEP | PHASE_ACTOR_GRAD)
        reduce(actor_grads)
        run_phases(PHASE_ACTOR_STEP)
        return
    if world == 1 and os.environ.get("FBHIP_FORCE_PHASE_SPLIT", "0") != "1":
        run_phases(PHASE_ALL)
        return
    # (FBHIP_FORCE_PHASE_SPLIT=1 runs this schedule on a single rank too: tests / 1-GPU rehearsal of the 8-GPU path)
    reduce = dist.all_reduce if (dist.is_available() and dist.is_initialized()) else (lambda t: None)
    run_phases(PHASE_SAMPLE | PHASE_FB_GRAD | PHASE_ACTOR_FWD)     # the actor's forward rides along the FB backward
    reduce(fb_grads)
    run_phases(PHASE_FB_STEP | PHASE_ACTOR_GRAD)
    reduce(actor_grads)
    run_phases(PHASE_ACTOR_STEP)


def shard_episodes(n_episodes: int, rank_: int, world: int) -> range:
    """episode ids owned by ``rank_``: ep % world == rank_"""
    return range(rank_, n_episodes, world)
