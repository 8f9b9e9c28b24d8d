"""Torch-tensor front-ends of the individually callable gfx950 kernels in libfbhip.so.

Each function takes CUDA(=HIP) float32 tensors, passes raw device pointers + leading dimensions over the
C ABI (include/fbhip.h) and launches on torch's current stream.  No arithmetic happens in Python.
"""
from __future__ import annotations

import typing as tp

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def _ld(t: torch.Tensor) -> int:
    assert t.dtype == torch.float32 and t.is_cuda, "float32 device tensor expected"
    if t.dim() == 1:
        return t.shape[0]
    assert t.dim() == 2 and t.stride(1) == 1, "row-major 2-D tensor expected"
    return t.stride(0)


def gemm(A: torch.Tensor, B: torch.Tensor, *, a_kcontig: bool = True, b_kcontig: bool = True,
         bias: tp.Optional[torch.Tensor] = None, aux: tp.Optional[torch.Tensor] = None, epi: int = _lib.EPI_NONE,
         want_colsum: bool = False, out: tp.Optional[torch.Tensor] = None, cfg: tp.Optional[int] = None):
    """C[M,N] = epi(sum_k A(m,k) B(n,k)).  a_kcontig: A is [M,K] (else [K,M]); b_kcontig: B is [N,K] (else [K,N])."""
    _lib.require_device()
    M, K = (A.shape if a_kcontig else A.shape[::-1])
    N, K2 = (B.shape if b_kcontig else B.shape[::-1])
    assert K == K2, (A.shape, B.shape)
    Cm = out if out is not None else torch.empty((M, N), device=A.device, dtype=torch.float32)
    colsum = torch.zeros(M, device=A.device) if want_colsum else None
    lib = _lib.load()
    if cfg is not None:
        check(lib.fbhip_gemm_cfg(ptr(A), _ld(A), int(a_kcontig), ptr(B), _ld(B), int(b_kcontig), ptr(Cm), _ld(Cm),
                                 M, N, K, cfg, stream_ptr()))
        return Cm
    check(lib.fbhip_gemm(ptr(A), _ld(A), int(a_kcontig), ptr(B), _ld(B), int(b_kcontig), ptr(Cm), _ld(Cm), M, N, K,
                         ptr(bias), ptr(aux), 0 if aux is None else _ld(aux), epi, ptr(colsum), stream_ptr()))
    return (Cm, colsum) if want_colsum else Cm


def head(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, *, normalize: bool = False, scale: float = 1.0, replicas: int = 1):
    """c = x @ w.T + bias for an embedding head (N <= 64) and, with ``normalize``, scale * c / |c|_row and the row norms
    (csrc/fused.hip).  bias must be readable up to pad4(N)."""
    _lib.require_device()
    rows, K = x.shape
    N = w.shape[0]
    ld = (N + 3) // 4 * 4
    c = torch.full((rows, ld), float("nan"), device=x.device)
    out2 = torch.full((rows, ld), float("nan"), device=x.device) if normalize else None
    norms = torch.empty(rows, device=x.device) if normalize else None
    check(_lib.load().fbhip_head(ptr(x), _ld(x), ptr(w), _ld(w), ptr(bias), ptr(c), ld, ptr(out2), ld, ptr(norms), scale, rows, N, K,
                                 replicas, stream_ptr()))
    return (c, out2, norms) if normalize else c


def ln_tanh_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor):
    """y = tanh(LayerNorm(x)), stats[rows,2] = (mean, rstd)   (fb_modules.py:49-50)."""
    _lib.require_device()
    rows, n = x.shape
    y = torch.empty((rows, n), device=x.device)
    stats = torch.empty((rows, 2), device=x.device)
    check(_lib.load().fbhip_ln_tanh_fwd(ptr(x), _ld(x), ptr(gamma), ptr(beta), ptr(y), _ld(y), ptr(stats), rows, n,
                                        stream_ptr()))
    return y, stats


def ln_tanh_bwd(dy, y, x, stats, gamma, want_param_grads: bool = True):
    _lib.require_device()
    rows, n = x.shape
    dx = torch.empty((rows, n), device=x.device)
    dg = torch.empty(n, device=x.device) if want_param_grads else None
    db = torch.empty(n, device=x.device) if want_param_grads else None
    part = torch.empty(((rows + 7) // 8) * 2 * n, device=x.device) if want_param_grads else None
    check(_lib.load().fbhip_ln_tanh_bwd(ptr(dy), _ld(dy), ptr(y), _ld(y), ptr(x), _ld(x), ptr(stats), ptr(gamma), ptr(dx),
                                        _ld(dx), ptr(dg), ptr(db), ptr(part), rows, n, stream_ptr()))
    return dx, dg, db


def l2norm_fwd(y: torch.Tensor):
    """sqrt(d) * F.normalize(y, dim=1) and the row norms."""
    _lib.require_device()
    rows, d = y.shape
    out = torch.empty((rows, d), device=y.device)
    norms = torch.empty(rows, device=y.device)
    check(_lib.load().fbhip_l2norm_fwd(ptr(y), _ld(y), ptr(out), _ld(out), ptr(norms), rows, d, stream_ptr()))
    return out, norms


def l2norm_bwd(dB, y, norms):
    _lib.require_device()
    rows, d = y.shape
    dy = torch.empty((rows, d), device=y.device)
    check(_lib.load().fbhip_l2norm_bwd(ptr(dB), _ld(dB), ptr(y), _ld(y), ptr(norms), ptr(dy), _ld(dy), rows, d,
                                       stream_ptr()))
    return dy


def pairwise_fb(F1, F2, Bm, tF1, tF2, tB, discount, ortho_coef: float):
    """FB + orthonormality loss and dF1, dF2, dB (fb_ddpg.py:313-348).  Returns (dF1, dF2, dB, metrics dict)."""
    _lib.require_device()
    Bn, d = F1.shape
    ld = _ld(F1)
    for t in (F2, Bm, tF1, tF2, tB):
        assert t.shape == F1.shape and _ld(t) == ld
    lib = _lib.load()
    # outputs share the inputs' leading dimension (one ``ld`` in the C ABI)
    dF1, dF2, dB = (torch.empty((Bn, ld), device=F1.device)[:, :d] for _ in range(3))
    metrics = torch.zeros(_lib.NUM_METRICS, device=F1.device)
    scratch = torch.empty(lib.fbhip_pairwise_scratch_floats(Bn, d), device=F1.device)
    disc = discount.reshape(-1).contiguous()
    check(lib.fbhip_pairwise_fb(ptr(F1), ptr(F2), ptr(Bm), ptr(tF1), ptr(tF2), ptr(tB), ptr(disc), Bn, d, ld,
                                float(ortho_coef), ptr(dF1), ptr(dF2), ptr(dB), ptr(metrics), ptr(scratch), stream_ptr()))
    m = metrics.cpu()
    names = ("fb_loss", "fb_offdiag", "fb_diag", "orth_loss", "orth_loss_diag", "orth_loss_offdiag", "target_M", "M1")
    return dF1, dF2, dB, {k: float(m[_lib.METRIC_INDEX[k]]) for k in names}


def pairwise_fb_block(F1, F2, Bm, tF1, tF2, tB, discount, ortho_coef: float, row_offset: int, rows: int):
    """Rows [row_offset, row_offset + rows) of ``pairwise_fb`` on the same panels: (dF1, dF2, dB) of those rows and this block's
    share of the scalars (fbhip_pairwise_fb_block; the global-batch data-parallel schedule)."""
    _lib.require_device()
    Bn, d = F1.shape
    ld = _ld(F1)
    for t in (F2, Bm, tF1, tF2, tB):
        assert t.shape == F1.shape and _ld(t) == ld
    lib = _lib.load()
    dF1, dF2, dB = (torch.empty((rows, ld), device=F1.device)[:, :d] for _ in range(3))
    metrics = torch.zeros(_lib.NUM_METRICS, device=F1.device)
    scratch = torch.empty(lib.fbhip_pairwise_scratch_floats(rows, d), device=F1.device)
    disc = discount.reshape(-1).contiguous()
    check(lib.fbhip_pairwise_fb_block(ptr(F1), ptr(F2), ptr(Bm), ptr(tF1), ptr(tF2), ptr(tB), ptr(disc), Bn, d, ld,
                                      float(ortho_coef), int(row_offset), int(rows), ptr(dF1), ptr(dF2), ptr(dB), ptr(metrics),
                                      ptr(scratch), stream_ptr()))
    m = metrics.cpu()
    names = ("fb_loss", "fb_offdiag", "fb_diag", "orth_loss", "orth_loss_diag", "orth_loss_offdiag", "target_M", "M1")
    return dF1, dF2, dB, {k: float(m[_lib.METRIC_INDEX[k]]) for k in names}


def adam_ema(params, grads, m, v, target, lr: float, t: int, grad_scale: float = 1.0, tau: float = 0.0) -> None:
    """In-place fused Adam (+ EMA into ``target`` when given) over flat tensors (numel % 4 == 0)."""
    _lib.require_device()
    check(_lib.load().fbhip_adam_ema(ptr(params), ptr(grads), ptr(m), ptr(v), ptr(target), params.numel(), float(lr),
                                     int(t), float(grad_scale), float(tau), stream_ptr()))


def inverse(A: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """inverse(scale * A) of a [d, d] fp32 matrix (row stride >= d), 1 <= d <= 128."""
    _lib.require_device()
    d = A.shape[0]
    out = torch.empty((d, d), device=A.device)
    check(_lib.load().fbhip_inverse(ptr(A), A.stride(0), d, float(scale), ptr(out), d, stream_ptr()))
    return out


def actor_loss(F1, F2, z, mu, action, stddev: float):
    _lib.require_device()
    rows, d = F1.shape
    a = mu.shape[1]
    dF1, dF2 = (torch.empty((rows, _ld(F1)), device=F1.device)[:, :d] for _ in range(2))
    metrics = torch.zeros(_lib.NUM_METRICS, device=F1.device)
    scratch = torch.empty(3 * ((rows + 3) // 4), device=F1.device)
    check(_lib.load().fbhip_actor_loss(ptr(F1), ptr(F2), _ld(F1), ptr(z), _ld(z), ptr(mu), _ld(mu), ptr(action),
                                       _ld(action), float(stddev), ptr(dF1), ptr(dF2), ptr(metrics), ptr(scratch),
                                       rows, d, a, stream_ptr()))
    m = metrics.cpu()
    return dF1, dF2, {k: float(m[_lib.METRIC_INDEX[k]]) for k in ("actor_loss", "q", "actor_logprob", "q1_success")}


def policy_head(P, W4, b4, noise, stddev: float, clip: float, base=None, W1a=None, gamma=None, beta=None, keep_pre: bool = False):
    """(premu, mu, action[, t1, pre, stats]) of the policy head launch (csrc/rowops.hip row kernel or csrc/headtiles.hip MFMA tiles)."""
    _lib.require_device()
    rows, H = P.shape
    a = W4.shape[0]
    La = (a + 3) // 4 * 4
    premu, mu, action = (torch.zeros((rows, La), device=P.device) for _ in range(3))
    t1 = stats = pre = None
    if base is not None:
        pre = base.clone()
        t1 = torch.empty((rows, H), device=P.device)
        stats = torch.empty(2 * rows, device=P.device) if keep_pre else None
    check(_lib.load().fbhip_policy_head(ptr(P), _ld(P), ptr(W4), _ld(W4), ptr(b4), ptr(noise), float(stddev), float(clip), ptr(premu),
                                        ptr(mu), ptr(action), La, ptr(pre), 0 if pre is None else _ld(pre), ptr(W1a),
                                        0 if W1a is None else _ld(W1a), ptr(gamma), ptr(beta), ptr(t1), 0 if t1 is None else _ld(t1),
                                        ptr(stats), rows, H, a, stream_ptr()))
    return premu[:, :a], mu[:, :a], action[:, :a], t1, pre, stats


def actor_head_bwd(dt1, lnY, lnX, lnStats, lnGamma, W1a, mu, W4, P):
    """(d premu [rows, a], d p [rows, H]) of the actor's backward seam behind the LayerNorm + tanh backward."""
    _lib.require_device()
    rows, H = dt1.shape
    a = W4.shape[0]
    La = (a + 3) // 4 * 4
    dpremu = torch.zeros((rows, La), device=dt1.device)
    dp = torch.empty((rows, H), device=dt1.device)
    assert _ld(dt1) == _ld(lnY) == _ld(lnX) == _ld(P) == _ld(dp)
    check(_lib.load().fbhip_actor_head_bwd(ptr(dt1), _ld(dt1), ptr(lnY), ptr(lnX), ptr(lnStats), ptr(lnGamma), ptr(W1a), _ld(W1a), ptr(mu),
                                           _ld(mu), ptr(W4), _ld(W4), ptr(P), ptr(dpremu), La, ptr(dp), rows, H, a, stream_ptr()))
    return dpremu[:, :a], dp
