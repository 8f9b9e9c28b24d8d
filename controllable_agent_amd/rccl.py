"""Host side of the library-owned RCCL transport (csrc/rccl.hip): load librccl into libfbhip, create one communicator per agent
from a unique id that rank 0 draws and the default process group carries (its only use: 128 bytes, once), and from then on a
data-parallel ``update_many`` is ONE hipGraph launch per rank -- ``fbhip_update_many_dp`` with ``ncclAllReduce`` of the two flat
gradient buckets captured inside it.  No ``torch.distributed`` call, and therefore no c10d watchdog poll, on the hot path.

``FBHIP_DP_ALLREDUCE``: ``rccl`` (default) this transport | ``c10d`` the round-2 path (``dist.all_reduce`` between phase graphs /
inside a torch-level graph, distributed.py) | ``peer`` hand-written peer-access kernels (peer.py).
"""
from __future__ import annotations

import ctypes as C
import os
import typing as tp

import torch

from . import _lib
from ._lib import check


def selected() -> str:
    return os.environ.get("FBHIP_DP_ALLREDUCE", "rccl").lower()


def usable() -> bool:
    return selected() == "rccl"


def library_path() -> str:
    """The librccl that shares the process's HIP runtime: the copy next to torch's libamdhip64."""
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return cand if os.path.exists(cand) else ""


def bind(agent: tp.Any) -> None:
    """Collective over the default process group (or local when there is none: world 1).  Idempotent per agent."""
    import torch.distributed as dist
    if getattr(agent, "_rccl_bound", False):
        return
    lib = _lib.load()
    live = dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(), dist.get_rank()) if live else (1, 0)
    load_rc = lib.fbhip_rccl_load(library_path().encode())       # (local; its outcome travels with the collective below)
    if not (live and world > 1):
        check(load_rc)
    if live and world > 1:
        # RCCL wants one device per rank: refuse BEFORE any communicator call when two ranks sit on one device (the one-GPU
        # rehearsals over gloo) -- the caller falls back to the torch.distributed schedule
        import socket
        props = torch.cuda.get_device_properties(agent._device)
        me = (socket.gethostname(), int(torch.device(agent._device).index or 0), str(getattr(props, "uuid", "")), int(getattr(props, "pci_bus_id", -1)),
              int(load_rc), rank if load_rc else -1)
        everyone: tp.List[tp.Any] = [None] * world
        dist.all_gather_object(everyone, me)
        bad = [e[5] for e in everyone if e[4] != 0]
        if bad:                                  # every rank raises together (a rank that raised alone would leave the others in the gather)
            raise RuntimeError(f"librccl could not be loaded on rank(s) {bad}: {_lib.last_error() if load_rc else 'see those ranks'}")
        everyone = [e[:4] for e in everyone]
        if len(set(everyone)) != world:
            raise RuntimeError(f"{world} ranks on {len(set(everyone))} distinct devices: RCCL needs one device per rank")
    # From here on every step that can fail on ONE rank is followed by an agreement of ALL ranks before anyone enters the next
    # collective: a rank that raised alone would leave the others inside a broadcast / ncclCommInitRank it never joins.
    uid = C.create_string_buffer(128)
    uid_rc, uid_err = 0, ""
    if rank == 0:
        uid_rc = int(lib.fbhip_rccl_unique_id(uid))
        uid_err = _lib.last_error() if uid_rc else ""
    box = [(uid_rc, uid_err, bytes(uid.raw))]
    if live and world > 1:
        dist.broadcast_object_list(box, src=0)      # rank 0 broadcasts (ok, id): every rank raises together on !ok
    if box[0][0] != 0:
        raise RuntimeError(f"fbhip_rccl_unique_id failed on rank 0: {box[0][1]}")
    torch.cuda.synchronize(agent._device)
    with torch.cuda.device(agent._device), torch.cuda.stream(agent._stream):
        init_rc = int(lib.fbhip_rccl_init(agent._ctx, box[0][2], world, rank, _lib.stream_ptr()))
    init_err = _lib.last_error(agent._ctx) if init_rc else ""
    torch.cuda.synchronize(agent._device)
    if live and world > 1:
        # (ncclCommInitRank is itself a rendezvous: a rank that fails in it fails on its peers too, or they time out inside RCCL;
        # what this catches is a LOCAL failure after the rendezvous -- exec teardown, a refused stream -- on some ranks only)
        flags: tp.List[tp.Any] = [None] * world
        dist.all_gather_object(flags, (rank, init_rc, init_err))
        bad = [f for f in flags if f[1] != 0]
        if bad:
            if init_rc == 0:                      # this rank's communicator is live: give it back before raising with the others
                # (release with rank = 1: ncclCommAbort -- a peer failed AFTER the rendezvous, e.g. in its warm-up all-reduce, and
                #  ncclCommDestroy would wait for that collective instead of tearing down; the library dropped its graphs too)
                agent._rccl_bound = False
                agent._rccl_prepared = []
                lib.fbhip_rccl_init(agent._ctx, None, 0, 1, _lib.stream_ptr())
            raise RuntimeError("fbhip_rccl_init failed on rank(s) " + ", ".join(f"{r}: {e}" for r, _, e in bad))
    elif init_rc != 0:
        raise RuntimeError(f"fbhip_rccl_init failed: {init_err}")
    agent._rccl_bound = True
    agent._dp_transport = f"rccl-library (in-graph ncclAllReduce, librccl {lib.fbhip_rccl_version()})"
