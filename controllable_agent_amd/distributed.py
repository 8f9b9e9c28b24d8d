"""Data-parallel schedules of one FB-DDPG update (modes A and B).

The reference has no distributed code at all; this is new design for MI355X nodes: one process per GPU,
``torch.distributed`` (backend "nccl" == RCCL over xGMI), replay episodes sharded ``ep % world == rank``,
parameters / Adam state / targets replicated.  Every rank computes the FB loss on its OWN batch x batch block and
the two flat gradient buckets are sum-all-reduced:

    phase SAMPLE | FB_GRAD | ACTOR_FWD -> all_reduce(fb_grads)  (forward_net ++ backward_net, 14.7 MB at walker dims)
    phase FB_STEP | ACTOR_GRAD  -> all_reduce(actor_grads)   ( 8.9 MB)
    phase ACTOR_STEP

``1 / world`` is folded into the Adam pass (``grad_scale``), so the optimiser steps are bit-identical on every
rank and no parameter broadcast is ever needed -- PROVIDED the replicas start identical: build every rank's agent under
the same torch seed (bench.py: ``torch.manual_seed(1)`` before the constructor) or call ``agent.sync_from_rank0()``
(broadcast of parameters, targets, Adam moments and step counts).  ``FBHipAgent`` verifies it with a checksum all-reduce
on its first data-parallel update and after every ``init_from`` / ``load_nets`` (``_verify_replicas``) and raises on a
mismatch instead of training diverged replicas.  This equals ONE device fed the same ``world`` micro-batches with
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

PHASE_SAMPLE, PHASE_FB_FWD_ONLINE, PHASE_FB_STEP, PHASE_ACTOR_GRAD, PHASE_ACTOR_STEP, PHASE_ACTOR_FWD, PHASE_FB_BWD_A = 1, 2, 4, 8, 16, 32, 64
PHASE_FB_FWD_TARGET, PHASE_FB_BWD_B = 128, 256
PHASE_FB_FWD = PHASE_FB_FWD_ONLINE | PHASE_FB_FWD_TARGET
PHASE_FB_BWD = PHASE_FB_BWD_A | PHASE_FB_BWD_B
PHASE_FB_GRAD = PHASE_FB_FWD | PHASE_FB_BWD
PHASE_ALL = 511


_COALESCE_OK = True


def world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _reduce_fb(run_phases: tp.Callable[[int], None], fb_grads: torch.Tensor, early: tp.Optional[tp.Tuple[int, int]],
               first_mask: int, live: bool) -> None:
    """FB backward + the sum-all-reduce of the FB bucket.  With ``early = (offset, count)`` (the gradients that are final after
    FB_BWD_A: both ForwardMap heads, 57 % of the bucket at walker dims) that range is reduced WHILE the rest of the backward
    runs; the remainder follows.  ``first_mask``: the phases issued together with FB_BWD_A."""
    import torch.distributed as dist
    if not live or early is None:
        run_phases(first_mask | PHASE_FB_BWD_B)
        if live:
            dist.all_reduce(fb_grads)
        return
    off, cnt = early
    run_phases(first_mask)
    work = dist.all_reduce(fb_grads[off:off + cnt], async_op=True)
    # (the actor's own forward pass rides with the target chain when the first call carries FB_FWD_TARGET; only when it shares
    # the FB backward's launches does the second half need the bit too)
    second = PHASE_ACTOR_FWD if (first_mask & PHASE_ACTOR_FWD) and not (first_mask & PHASE_FB_FWD_TARGET) else 0
    run_phases(PHASE_FB_BWD_B | second)
    work.wait()
    rest = [t for t in (fb_grads[:off], fb_grads[off + cnt:]) if t.numel() > 0]
    global _COALESCE_OK
    if len(rest) == 2 and _COALESCE_OK and dist.get_backend() == "nccl":
        # one grouped RCCL launch for the two remaining slices (trunks | backward_net) instead of two latencies.  The context
        # manager is a private torch API: if it is missing or refuses, fall back to two plain calls for good -- nothing has
        # been reduced yet when it raises on entry / exit without having issued the group
        try:
            from torch.distributed.distributed_c10d import _coalescing_manager
            with _coalescing_manager(device=fb_grads.device):
                for t in rest:
                    dist.all_reduce(t)
            return
        except (ImportError, AttributeError, TypeError, ValueError, NotImplementedError):
            _COALESCE_OK = False
    for t in rest:
        dist.all_reduce(t)


def dp_update(run_phases: tp.Callable[[int], None], fb_grads: torch.Tensor, actor_grads: torch.Tensor,
              exchange: tp.Optional[tp.Callable[[], None]] = None,
              early: tp.Optional[tp.Tuple[int, int]] = None) -> None:
    """Run one update through ``run_phases(mask)``; with world_size > 1 the two gradient buckets are
    sum-all-reduced between the phases (``run_phases`` applies grad_scale = 1/world in its optimiser steps).
    ``exchange`` (mode B): called between FB_FWD and FB_BWD; it all-gathers the embeddings and binds the global batch
    (``run_phases`` must then apply grad_scale = 1 in the FB optimiser step)."""
    import os
    import torch.distributed as dist
    world = world_size()
    if exchange is not None:
        live_ = dist.is_available() and dist.is_initialized()
        reduce = (lambda t: dist.all_reduce(t) if t.numel() > 0 else None) if live_ else (lambda t: None)
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
    live = dist.is_available() and dist.is_initialized()
    reduce = (lambda t: dist.all_reduce(t) if t.numel() > 0 else None) if live else (lambda t: None)   # (DiscreteFBHipAgent: no actor bucket)
    # the actor's forward rides along the FB backward
    _reduce_fb(run_phases, fb_grads, early, PHASE_SAMPLE | PHASE_FB_FWD | PHASE_FB_BWD_A | PHASE_ACTOR_FWD, live)
    run_phases(PHASE_FB_STEP | PHASE_ACTOR_GRAD)
    reduce(actor_grads)
    run_phases(PHASE_ACTOR_STEP)


def dp_update_many(run_phases: tp.Callable[[int], None], select_set: tp.Callable[[int], None], fb_grads: torch.Tensor,
                   actor_grads: torch.Tensor, n_steps: int, early: tp.Optional[tp.Tuple[int, int]] = None,
                   side: tp.Optional["torch.cuda.Stream"] = None) -> None:
    """``n_steps`` consecutive mode-A updates with the steps software-pipelined: step t+1's sampling, z mixing, B passes and
    online ForwardMap pass (``SAMPLE | FB_FWD_ONLINE``: they depend on step t only through its FB optimiser step) are
    launched on the OTHER workspace set while step t's actor-gradient all-reduce is in flight, so that all-reduce is hidden
    behind useful work instead of idling the GPU (the FB all-reduce still gates the FB step).  Same kernels, operands and
    order inside each step as ``n_steps`` calls of ``dp_update`` (bit-identical at small dims; at walker dims regrouped
    launches change a few split-K factors, i.e. fp32 summation order); replicas stay bit-identical to each other.

        head(0);  for t:  FB_FWD_TARGET|FB_BWD_A|ACTOR_FWD -> all_reduce(fb heads, async) || FB_BWD_B|ACTOR_FWD -> wait
                          -> all_reduce(fb rest) -> FB_STEP|ACTOR_GRAD
                          -> all_reduce(actor, async) || head(t+1) -> wait -> ACTOR_STEP

    With ``side`` (a second CUDA stream; ``run_phases`` must launch on torch's CURRENT stream) the next head starts even
    earlier, right after the FB optimiser step, on that stream: it then runs beside the actor phase itself (its ~20 dependent
    small launches leave most of the GPU idle) AND beside the all-reduce, exactly like the single-GPU pipeline of
    ``fbhip_update_many``:   FB_STEP -> [ side: head(t+1) ]  ||  [ main: ACTOR_GRAD -> all_reduce(actor) -> ACTOR_STEP ] -> join
    """
    import torch.distributed as dist
    live = dist.is_available() and dist.is_initialized()
    head = PHASE_SAMPLE | PHASE_FB_FWD_ONLINE
    cur = 0
    select_set(cur)
    try:
        run_phases(head)
        for t in range(n_steps):
            _reduce_fb(run_phases, fb_grads, early, PHASE_FB_FWD_TARGET | PHASE_FB_BWD_A | PHASE_ACTOR_FWD, live)
            if side is not None and t + 1 < n_steps:
                main = torch.cuda.current_stream()
                run_phases(PHASE_FB_STEP)
                side.wait_stream(main)                   # fork: the head needs the new FB weights, nothing of the actor phase
                select_set(cur ^ 1)
                with torch.cuda.stream(side):
                    run_phases(head)
                select_set(cur)
                run_phases(PHASE_ACTOR_GRAD)
                work = dist.all_reduce(actor_grads, async_op=True) if (live and actor_grads.numel() > 0) else None
                if work is not None:
                    work.wait()
                run_phases(PHASE_ACTOR_STEP)
                main.wait_stream(side)                   # join before the next step's target chain
                cur ^= 1
                select_set(cur)
                continue
            run_phases(PHASE_FB_STEP | PHASE_ACTOR_GRAD)
            work = dist.all_reduce(actor_grads, async_op=True) if (live and actor_grads.numel() > 0) else None
            if t + 1 < n_steps:
                select_set(cur ^ 1)
                run_phases(head)                     # the next step's head, under the all-reduce
                select_set(cur)
            if work is not None:
                work.wait()                          # nccl: the compute stream waits; gloo: the host does
            run_phases(PHASE_ACTOR_STEP)
            if t + 1 < n_steps:
                cur ^= 1
                select_set(cur)
    finally:
        select_set(0)


def shard_episodes(n_episodes: int, rank_: int, world: int) -> range:
    """episode ids owned by ``rank_``: ep % world == rank_"""
    return range(rank_, n_episodes, world)
