"""Host side of the peer-access all-reduce (csrc/peer.hip, SURVEY.md section 8e fallback): map every rank's gradient buckets and
barrier flags into this process and hand the pointers to the library, so that a data-parallel step is ONE hipGraph launch per
rank (``fbhip_update_many_dp``) instead of three graph launches and three c10d calls.

Selected with ``FBHIP_DP_ALLREDUCE=peer`` (default: RCCL through torch.distributed, ``distributed.py``).  Single node only:
the mapping is CUDA/HIP IPC (``torch.multiprocessing.reductions.reduce_tensor`` produces a picklable handle of a device
tensor; the handles travel through the already-initialised process group with ``all_gather_object``).  The environment needs
``HSA_ENABLE_IPC_MODE_LEGACY=0`` on this driver (dmabuf IPC).
"""
from __future__ import annotations

import ctypes as C
import os
import typing as tp

import torch

from . import _lib
from ._lib import check, ptr

FLAG_INTS = 3 * 8          # int32[3][PEER_MAX_WORLD]: one row of slots per kernel of the all-reduce (csrc/common.h)
STATE_INTS = 4             # PeerState: epoch[3], status


def enabled() -> bool:
    return os.environ.get("FBHIP_DP_ALLREDUCE", "rccl").lower() == "peer"


def bind(agent: tp.Any) -> None:
    """Exchange IPC handles of (fb gradient bucket, actor gradient bucket, flag array) with every rank of the default process
    group and bind the mapped pointers to ``agent``'s context.  Collective: every rank must call it.  Idempotent per agent."""
    import torch.distributed as dist
    from torch.multiprocessing.reductions import reduce_tensor
    if getattr(agent, "_peer_bound", False):
        return
    world, rank = dist.get_world_size(), dist.get_rank()
    if world > 8:
        raise RuntimeError("peer all-reduce: at most 8 ranks (one node)")
    dev = agent._device
    flags = torch.zeros(FLAG_INTS, dtype=torch.int32, device=dev)
    state = torch.zeros(STATE_INTS, dtype=torch.int32, device=dev)
    has_actor = not agent._discrete
    mine = {"fb": agent._fb_grads, "flags": flags}
    if has_actor:
        mine["actor"] = agent._actor_grads
    torch.cuda.synchronize(dev)
    handles: tp.List[tp.Any] = [None] * world
    dist.all_gather_object(handles, {k: reduce_tensor(v) for k, v in mine.items()})
    peers: tp.List[tp.Dict[str, torch.Tensor]] = []
    for q in range(world):
        if q == rank:
            peers.append(mine)
        else:
            peers.append({k: fn(*args) for k, (fn, args) in handles[q].items()})       # rebuild_cuda_tensor: maps the peer's memory
    table = lambda key: (C.c_void_p * world)(*[ptr(p[key]) for p in peers])
    fb_t, fl_t = table("fb"), table("flags")
    ac_t = table("actor") if has_actor else None
    check(_lib.load().fbhip_dp_bind_peers(agent._ctx, world, rank, fb_t, ac_t, fl_t, ptr(state)), agent._ctx)
    agent._peer_keep = (peers, flags, state, handles)            # mapped tensors must outlive the context's use of the pointers
    agent._peer_bound = True
    dist.barrier()                                               # nobody launches before every rank has mapped everybody


def status(agent: tp.Any) -> int:
    out = C.c_int32()
    check(_lib.load().fbhip_dp_status(agent._ctx, C.byref(out), _lib.stream_ptr()), agent._ctx)
    return int(out.value)
