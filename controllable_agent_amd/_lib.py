"""ctypes binding of libfbhip.so (include/fbhip.h).

The HIP library is the product; there is NO CPU fallback.  Importing this module never needs a GPU (the
symbol table can be checked on any box), but every compute entry point raises ``RuntimeError`` when the
shared object is missing or no gfx950 device is current.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "libfbhip.so"

NET_FORWARD, NET_BACKWARD, NET_ACTOR = 0, 1, 2
PHASE_SAMPLE, PHASE_FB_FWD_ONLINE, PHASE_FB_STEP, PHASE_ACTOR_GRAD, PHASE_ACTOR_STEP, PHASE_ACTOR_FWD, PHASE_FB_BWD_A = 1, 2, 4, 8, 16, 32, 64
PHASE_FB_FWD_TARGET, PHASE_FB_BWD_B = 128, 256
PHASE_FB_BWD = PHASE_FB_BWD_A | PHASE_FB_BWD_B
PHASE_FB_FWD = PHASE_FB_FWD_ONLINE | PHASE_FB_FWD_TARGET
PHASE_FB_GRAD, PHASE_ALL = PHASE_FB_FWD | PHASE_FB_BWD, 511
NUM_METRICS = 32
# metric name -> index in the device metrics array (fb_ddpg.py:356-377, 413-418)
METRIC_INDEX = {n: i for i, n in enumerate(
    ["target_M", "M1", "F1", "B", "B_norm", "z_norm", "fb_loss", "fb_diag", "fb_offdiag", "q_loss", "orth_loss",
     "orth_loss_diag", "orth_loss_offdiag", "orth_linf", "orth_l2", "actor_loss", "q", "actor_logprob", "q1_success",
     "sf_loss", "target_F", "phi", "phi_norm", "phi_loss"])}
EPI_NONE, EPI_BIAS, EPI_BIAS_RELU, EPI_MASK_RELU, EPI_TANH_BWD = range(5)


class Dims(C.Structure):
    """include/fbhip.h::fbhip_dims; ``struct_size`` (first field) is filled in here, positional arguments start at ``batch``"""
    _fields_ = [("struct_size", C.c_uint32)] + [(n, C.c_int32) for n in ("batch", "obs_dim", "action_dim", "goal_dim", "z_dim", "hidden_dim",
                                          "feature_dim", "backward_hidden_dim", "use_goal", "add_trunk", "preprocess", "norm_z", "boltzmann",
                                          "discrete", "sf", "backward_identity")]

    def __init__(self, *args, **kw):
        super().__init__(C.sizeof(Dims), *args, **kw)


class HParams(C.Structure):
    """include/fbhip.h::fbhip_hparams; ``struct_size`` is filled in here"""
    _fields_ = [("struct_size", C.c_uint32)] + [(n, C.c_float) for n in ("lr", "lr_coef", "fb_target_tau", "stddev", "stddev_clip", "ortho_coef",
                                          "mix_ratio", "q_loss_coef", "discount", "grad_scale")] + \
               [("q_loss", C.c_int32), ("want_metrics", C.c_int32), ("future_ratio", C.c_float), ("future", C.c_float),
                ("rand_weight", C.c_int32)]

    def __init__(self, *args, **kw):
        super().__init__(C.sizeof(HParams), *args, **kw)


class Inject(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ep_idx", "step_idx", "z_gauss", "perm", "mix_uniform", "eps_next",
                                           "eps_actor", "future_idx", "future_uniform", "z_uniform", "rand_weight",
                                           "rand_weight_u")]


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("offset", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32),
                ("ld", C.c_int32)]


_P, _I, _F, _L, _Z = C.c_void_p, C.c_int32, C.c_float, C.c_int64, C.c_size_t
# every symbol include/fbhip.h declares: name -> (restype, argtypes)
PROTOTYPES = {
    "fbhip_abi_version": (C.c_int, []),
    "fbhip_last_error": (C.c_char_p, [_P]),
    "fbhip_device_ok": (C.c_int, []),
    "fbhip_branched_graphs": (C.c_int, [C.POINTER(C.c_char_p)]),
    "fbhip_net_numel": (_L, [C.POINTER(Dims), C.c_int]),
    "fbhip_net_param_count": (_L, [C.POINTER(Dims), C.c_int]),
    "fbhip_layout_count": (C.c_int, [C.POINTER(Dims), C.c_int]),
    "fbhip_layout_entry": (C.c_int, [C.POINTER(Dims), C.c_int, C.c_int, C.POINTER(TensorDesc)]),
    "fbhip_workspace_bytes": (_Z, [C.POINTER(Dims)]),
    "fbhip_create": (C.c_int, [C.POINTER(Dims), C.POINTER(_P)]),
    "fbhip_destroy": (C.c_int, [_P]),
    "fbhip_bind_buffers": (C.c_int, [_P] + [_P] * 9 + [_P, _Z]),
    "fbhip_replay_bind": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I]),
    "fbhip_set_seed": (C.c_int, [_P, C.c_uint64, C.c_uint32]),
    "fbhip_set_policy_squash": (C.c_int, [_P, _F, _F, _F]),
    "fbhip_set_step_counts": (C.c_int, [_P, _I, _I, _P]),
    "fbhip_get_step_counts": (C.c_int, [_P, C.POINTER(_I), C.POINTER(_I), _P]),
    "fbhip_get_rng_counts": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _P]),
    "fbhip_set_rng_counts": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P]),
    "fbhip_update": (C.c_int, [_P, C.POINTER(HParams), C.POINTER(Inject), _I, _I, _P]),
    "fbhip_update_many": (C.c_int, [_P, _P, _I, _P]),
    "fbhip_graph_captures": (C.c_int64, [_P]),
    "fbhip_update_many_injected": (C.c_int, [_P, _P, _I, _P, _P]),
    "fbhip_dp_bind_peers": (C.c_int, [_P, _I, _I, _P, _P, _P, _P]),
    "fbhip_peer_allreduce": (C.c_int, [_P, _I, _P]),
    "fbhip_update_many_dp": (C.c_int, [_P, _P, _I, _P]),
    "fbhip_update_many_dp_prepare": (C.c_int, [_P, _P, _I, _P]),
    "fbhip_dp_status": (C.c_int, [_P, C.POINTER(_I), _P]),
    "fbhip_order_legacy_stream_after": (C.c_int, [_P, _P]),
    "fbhip_order_stream_after_legacy": (C.c_int, [_P, _P]),
    "fbhip_rccl_load": (C.c_int, [C.c_char_p]),
    "fbhip_rccl_version": (C.c_int, []),
    "fbhip_rccl_unique_id": (C.c_int, [_P]),
    "fbhip_rccl_init": (C.c_int, [_P, _P, _I, _I, _P]),
    "fbhip_select_workspace_set": (C.c_int, [_P, _I]),
    "fbhip_fb_early_grad_range": (C.c_int, [C.POINTER(Dims), C.POINTER(_L), C.POINTER(_L)]),
    "fbhip_embeddings_floats": (_Z, [C.POINTER(Dims)]),
    "fbhip_export_embeddings": (C.c_int, [_P, _P, _P]),
    "fbhip_bind_global_batch": (C.c_int, [_P, _P, _P, _I, _I]),
    "fbhip_read_metrics": (C.c_int, [_P, C.POINTER(C.c_float), _P]),
    "fbhip_wait_metrics": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "fbhip_workspace_view": (C.c_int, [_P, C.c_char_p, C.POINTER(_P), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "fbhip_actor_forward": (C.c_int, [_P, _P, _I, _P, _I, _I, _P, _F, _F, _P, _I, _P]),
    "fbhip_backward_map": (C.c_int, [_P, _I, _P, _I, _I, _P, _I, _P]),
    "fbhip_act": (C.c_int, [_P, _P, _P, _P, _F, _I, _P, _P]),
    "fbhip_z_correl": (C.c_int, [_P, _P, _P, _P, _P]),
    "fbhip_forward_map": (C.c_int, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P, _I, _P]),
    "fbhip_discrete_act": (C.c_int, [_P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _P, _I, _P]),
    "fbhip_discrete_act_host": (C.c_int, [_P, _P, _P, _P, _P]),
    "fbhip_gemm": (C.c_int, [_P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P]),
    "fbhip_gemm_cfg": (C.c_int, [_P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _P]),
    "fbhip_head": (C.c_int, [_P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _F, _I, _I, _I, _I, _P]),
    "fbhip_ln_tanh_fwd": (C.c_int, [_P, _I, _P, _P, _P, _I, _P, _I, _I, _P]),
    "fbhip_ln_tanh_bwd": (C.c_int, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P]),
    "fbhip_l2norm_fwd": (C.c_int, [_P, _I, _P, _I, _P, _I, _I, _P]),
    "fbhip_l2norm_bwd": (C.c_int, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _P]),
    "fbhip_actor_loss": (C.c_int, [_P, _P, _I, _P, _I, _P, _I, _P, _I, _F, _P, _P, _P, _P, _I, _I, _I, _P]),
    "fbhip_policy_head": (C.c_int, [_P, _I, _P, _I, _P, _P, _F, _F, _P, _P, _P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "fbhip_actor_head_bwd": (C.c_int, [_P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _P]),
    "fbhip_pairwise_scratch_floats": (_Z, [_I, _I]),
    "fbhip_pairwise_fb": (C.c_int, [_P] * 7 + [_I, _I, _I, _F] + [_P] * 5 + [_P]),
    "fbhip_pairwise_fb_block": (C.c_int, [_P] * 7 + [_I, _I, _I, _F, _I, _I] + [_P] * 5 + [_P]),
    "fbhip_adam_ema": (C.c_int, [_P, _P, _P, _P, _P, _L, _F, _I, _F, _F, _P]),
    "fbhip_inverse": (C.c_int, [_P, _I, _I, _F, _P, _I, _P]),
}

_lib = None


def load() -> C.CDLL:
    """Load libfbhip.so (once) and attach prototypes.  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"or `make -C {LIB_PATH.parent / 'csrc'}`.  controllable_agent_amd has no CPU fallback.")
    # torch ships its own HIP runtime (torch/lib/libamdhip64.so); it must be the ONE runtime of the process.  If
    # libfbhip.so is loaded first, the system copy under /opt/rocm/lib gets mapped, torch later maps its own, and
    # this library then talks to a second, device-less runtime ("no HIP device").
    import torch  # noqa: F401
    lib = C.CDLL(str(LIB_PATH), mode=getattr(os, "RTLD_NOW", 2))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError here == header / library mismatch
        fn.restype, fn.argtypes = res, args
    if lib.fbhip_abi_version() != 19:
        raise RuntimeError("libfbhip.so ABI version mismatch")
    _lib = lib
    return lib


def last_error(ctx=None) -> str:
    return load().fbhip_last_error(ctx).decode()


def check(rc: int, ctx=None) -> None:
    if rc != 0:
        raise RuntimeError(f"fbhip error {rc}: {last_error(ctx)}")


def require_device() -> None:
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("controllable_agent_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False "
                           "and there is no CPU fallback")
    check(load().fbhip_device_ok())


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int:
    """device pointer of a torch tensor (None -> NULL)"""
    return None if t is None else t.data_ptr()
