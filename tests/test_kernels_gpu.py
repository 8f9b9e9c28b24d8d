"""Kernel-level parity (GPU): every hand-written gfx950 kernel against an fp64 torch statement of the same op /
the oracle's closed form.  Tolerances are fp32 round-off class and are stated per test."""
import math

import numpy as np
import pytest
import torch

from oracle import fb_oracle as fo
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from controllable_agent_amd import kernels
    return kernels


def _r(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


def _padded(t, ld):
    """device copy of a 2-D tensor with leading dimension ld (pad columns filled with NaN to catch over-reads)"""
    buf = torch.full((t.shape[0], ld), float("nan"), device="cuda")
    buf[:, :t.shape[1]] = t.cuda()
    return buf[:, :t.shape[1]]


# ------------------------------------------------------------------------------------------------ GEMM
GEMM_SHAPES = [
    # M, N, K  -- layer shapes of the walker / quadruped nets plus ragged edge cases
    (1024, 1024, 74), (1024, 512, 1024), (1024, 2048, 1024), (1024, 50, 1024), (1024, 6, 1024), (1024, 526, 526),
    (1024, 1024, 30), (50, 1024, 1024), (526, 24, 1024), (2048, 1024, 1024), (1, 1024, 74), (16, 32, 13), (33, 65, 97),
    (256, 100, 1024), (7, 3, 5),
]


@pytest.mark.parametrize("M,N,K_", GEMM_SHAPES)
@pytest.mark.parametrize("akc,bkc", [(True, True), (True, False), (False, False)])
def test_gemm_layouts(K, M, N, K_, akc, bkc):
    """forward (NT), dgrad (NN), wgrad (TN) operand layouts; asymmetric random operands; NaN padding."""
    A = _r(M, K_, seed=1) if akc else _r(K_, M, seed=1)
    B = _r(N, K_, seed=2) if bkc else _r(K_, N, seed=2)
    Ad = _padded(A, (A.shape[1] + 3) // 4 * 4 + 4)
    Bd = _padded(B, (B.shape[1] + 3) // 4 * 4)
    C = K.gemm(Ad, Bd, a_kcontig=akc, b_kcontig=bkc)
    Am = A.double() if akc else A.double().T
    Bm = B.double() if bkc else B.double().T
    ref = Am @ Bm.T
    assert C.shape == (M, N)
    assert rel_err(C.cpu(), ref) < 2e-6


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4])      # (cfg 5 needs K % 32 == 0: test_gemm_lds_dma_kernels)
def test_gemm_every_tile_config(K, cfg):
    A, B = _r(200, 300, seed=3), _r(150, 300, seed=4)
    C = K.gemm(A.cuda(), B.cuda(), cfg=cfg)
    assert rel_err(C.cpu(), A.double() @ B.double().T) < 2e-6


@pytest.mark.parametrize("cfg", [5])
@pytest.mark.parametrize("akc,bkc", [(True, True), (True, False), (False, False), (False, True)])
@pytest.mark.parametrize("M,N,Kd", [(256, 192, 128), (200, 100, 96), (50, 1024, 1024), (1024, 52, 576), (130, 70, 32)])
def test_gemm_lds_dma_kernels(K, cfg, akc, bkc, M, N, Kd):
    """cfg 5 = the LDS-DMA kernel (128x64 tiles; K % 32 == 0, aligned operands); ragged M and N tiles read
    clamped rows.  NaN padding columns must never reach the result."""
    A = _r(M, Kd, seed=21) if akc else _r(Kd, M, seed=21)
    B = _r(N, Kd, seed=22) if bkc else _r(Kd, N, seed=22)
    Ad = _padded(A, (A.shape[1] + 3) // 4 * 4 + 4)
    Bd = _padded(B, (B.shape[1] + 3) // 4 * 4)
    C = K.gemm(Ad, Bd, a_kcontig=akc, b_kcontig=bkc, cfg=cfg)
    Am = A.double() if akc else A.double().T
    Bm = B.double() if bkc else B.double().T
    assert rel_err(C.cpu(), Am @ Bm.T) < 2e-6


def test_gemm_unaligned_views(K):
    """column-sliced operands (the actor step reads W1[:, o:o+a]) and odd leading dimensions take the scalar path"""
    W = _r(64, 37, seed=5).cuda()
    X = _r(100, 64, seed=6).cuda()
    C = K.gemm(X, W[:, 5:11], a_kcontig=True, b_kcontig=False)          # [100,64] . [64,6]
    assert rel_err(C.cpu(), X.cpu().double() @ W[:, 5:11].cpu().double()) < 2e-6


def test_gemm_epilogues(K):
    M, N, Kd = 300, 200, 128
    A, B, bias, aux = _r(M, Kd, seed=7), _r(N, Kd, seed=8), _r(N, seed=9), _r(M, N, seed=10)
    ref = A.double() @ B.double().T
    Ad, Bd = A.cuda(), B.cuda()
    from controllable_agent_amd import _lib
    assert rel_err(K.gemm(Ad, Bd, bias=bias.cuda(), epi=_lib.EPI_BIAS).cpu(), ref + bias.double()) < 2e-6
    assert rel_err(K.gemm(Ad, Bd, bias=bias.cuda(), epi=_lib.EPI_BIAS_RELU).cpu(), torch.relu(ref + bias.double())) < 2e-6
    assert rel_err(K.gemm(Ad, Bd, aux=aux.cuda(), epi=_lib.EPI_MASK_RELU).cpu(), ref * (aux.double() > 0)) < 2e-6
    y = torch.tanh(aux)
    assert rel_err(K.gemm(Ad, Bd, aux=y.cuda(), epi=_lib.EPI_TANH_BWD).cpu(), ref * (1 - y.double() ** 2)) < 2e-6


@pytest.mark.parametrize("M,N,Kd", [(1024, 30, 1024), (50, 1024, 1024), (526, 526, 1024), (33, 7, 45)])
def test_gemm_wgrad_colsum(K, M, N, Kd):
    """weight-gradient GEMM dW = dY^T X with the fused bias gradient (column sums of dY)."""
    dY, X = _r(Kd, M, seed=11), _r(Kd, N, seed=12)
    C, cs = K.gemm(dY.cuda(), X.cuda(), a_kcontig=False, b_kcontig=False, want_colsum=True)
    assert rel_err(C.cpu(), dY.double().T @ X.double()) < 2e-6
    assert rel_err(cs.cpu(), dY.double().sum(0)) < 2e-6


# ------------------------------------------------------------------------------------------------ LayerNorm + tanh
@pytest.mark.parametrize("rows,n", [(1024, 1024), (1024, 526), (5, 32), (1, 18), (37, 2048), (16, 700)])
def test_ln_tanh_fwd_bwd(K, rows, n):
    x, g, b, dy = _r(rows, n, seed=1, scale=2.0), 1 + 0.1 * _r(n, seed=2), 0.1 * _r(n, seed=3), _r(rows, n, seed=4)
    xd = x.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    y_ref = torch.tanh(torch.nn.functional.layer_norm(xd, (n,), gd, bd, 1e-5))
    y_ref.backward(dy.double())
    xp = _padded(x, (n + 3) // 4 * 4)
    y, stats = K.ln_tanh_fwd(xp, g.cuda(), b.cuda())
    assert rel_err(y.cpu(), y_ref.detach()) < 2e-6
    assert rel_err(stats[:, 0].cpu(), x.double().mean(1)) < 1e-5 or x.double().mean(1).abs().max() < 1e-3
    dx, dg, db = K.ln_tanh_bwd(dy.cuda(), y, xp, stats, g.cuda())
    assert rel_err(dx.cpu(), xd.grad) < 1e-5
    assert rel_err(dg.cpu(), gd.grad) < 1e-5
    assert rel_err(db.cpu(), bd.grad) < 1e-5
    dx2, none_g, none_b = K.ln_tanh_bwd(dy.cuda(), y, xp, stats, g.cuda(), want_param_grads=False)
    assert none_g is None and torch.equal(dx2, dx)


# ------------------------------------------------------------------------------------------------ L2 projection
@pytest.mark.parametrize("rows,d", [(1024, 50), (1024, 100), (3, 8), (17, 128), (5, 1)])
def test_l2norm_fwd_bwd(K, rows, d):
    y, dB = _r(rows, d, seed=1), _r(rows, d, seed=2)
    yd = y.double().requires_grad_(True)
    ref = math.sqrt(d) * torch.nn.functional.normalize(yd, dim=1)
    ref.backward(dB.double())
    yp = _padded(y, (d + 3) // 4 * 4)
    out, norms = K.l2norm_fwd(yp)
    assert rel_err(out.cpu(), ref.detach()) < 1e-6
    assert rel_err(norms.cpu(), y.double().norm(dim=1)) < 1e-6
    dy = K.l2norm_bwd(dB.cuda(), yp, norms)
    if d == 1:          # the projection of a 1-vector is constant: the true gradient is exactly zero
        assert dy.abs().max().item() < 1e-6
    else:
        assert rel_err(dy.cpu(), yd.grad) < 1e-5


# ------------------------------------------------------------------------------------------------ pairwise FB
@pytest.mark.parametrize("Bn,d", [(16, 8), (64, 50), (256, 50), (1024, 50), (100, 100), (48, 10), (33, 3), (1024, 100),
                                  (2048, 100)])
def test_pairwise_fb_vs_closed_form(K, Bn, d):
    """K3 against the oracle's fp64 closed form.  Tolerance: scalars rel 2e-5, grads rel-L2 1e-5
    (fp32 products accumulated over d and over the batch)."""
    rng = np.random.default_rng(Bn * 1000 + d)
    t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    F1, F2, tF1, tF2 = t(Bn, d), t(Bn, d), t(Bn, d), t(Bn, d)
    Bm = math.sqrt(d) * torch.nn.functional.normalize(t(Bn, d), dim=1)
    tB = math.sqrt(d) * torch.nn.functional.normalize(t(Bn, d), dim=1)
    disc = torch.from_numpy(rng.uniform(0.9, 0.99, (Bn, 1)).astype(np.float32))
    cf = fo.fb_loss_closed_form(F1, F2, Bm, tF1, tF2, tB, disc, 0.7)
    ld = (d + 3) // 4 * 4
    dev = [_padded(x, ld) for x in (F1, F2, Bm, tF1, tF2, tB)]
    dF1, dF2, dB, m = K.pairwise_fb(*dev, disc.cuda(), 0.7)
    for k, got in (("dF1", dF1), ("dF2", dF2), ("dB", dB)):
        assert rel_err(got.cpu(), cf[k]) < 1e-5, k
    for k in ("fb_loss", "fb_offdiag", "fb_diag", "orth_loss", "orth_loss_diag", "orth_loss_offdiag"):
        assert m[k] == pytest.approx(float(cf[k]), rel=2e-5, abs=1e-6), k
    assert m["target_M"] == pytest.approx(float(cf["target_M_mean"]), rel=1e-4, abs=1e-5)
    assert m["M1"] == pytest.approx(float(cf["M1_mean"]), rel=1e-4, abs=1e-5)


def test_pairwise_fb_vs_masked_autograd(K):
    """... and against autograd of the reference's masked formulation (fb_ddpg.py:313-348) in fp32."""
    rng = np.random.default_rng(3)
    Bn, d = 96, 50
    t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    F1, F2, tF1, tF2 = t(Bn, d).requires_grad_(True), t(Bn, d).requires_grad_(True), t(Bn, d), t(Bn, d)
    Bm = (math.sqrt(d) * torch.nn.functional.normalize(t(Bn, d), dim=1)).requires_grad_(True)
    tB = math.sqrt(d) * torch.nn.functional.normalize(t(Bn, d), dim=1)
    disc = torch.full((Bn, 1), 0.98)
    L = fo.fb_loss_terms(F1, F2, Bm, tF1, tF2, tB, disc, 1.0)
    L["fb_loss"].backward()
    ld = 52
    dev = [_padded(x.detach(), ld) for x in (F1, F2, Bm, tF1, tF2, tB)]
    dF1, dF2, dB, m = K.pairwise_fb(*dev, disc.cuda(), 1.0)
    assert rel_err(dF1.cpu(), F1.grad) < 1e-5 and rel_err(dF2.cpu(), F2.grad) < 1e-5 and rel_err(dB.cpu(), Bm.grad) < 1e-5
    assert m["fb_loss"] == pytest.approx(float(L["fb_loss"]), rel=2e-5)


@pytest.mark.parametrize("Bn,d,blocks", [(256, 50, 4), (2048, 50, 8), (1024, 100, 2)])
def test_pairwise_row_blocks_add_up_to_the_square_loss(K, Bn, d, blocks):
    """Block mode (the global-batch data-parallel schedule): every row block's dF / dB rows equal the square kernel's rows bit for
    bit where the J-chunking coincides, within fp32 otherwise, and the blocks' scalar shares add up to the square loss."""
    rng = np.random.default_rng(17)
    ld = (d + 3) // 4 * 4
    t = lambda: torch.from_numpy(rng.standard_normal((Bn, ld)).astype(np.float32)).cuda()[:, :d]
    F1, F2, tF1, tF2 = t(), t(), t(), t()
    Bm = t(); Bm.copy_(math.sqrt(d) * torch.nn.functional.normalize(Bm, dim=1))
    tB = t(); tB.copy_(math.sqrt(d) * torch.nn.functional.normalize(tB, dim=1))
    disc = torch.from_numpy(rng.uniform(0.9, 1.0, Bn).astype(np.float32)).cuda()
    dF1, dF2, dB, m = K.pairwise_fb(F1, F2, Bm, tF1, tF2, tB, disc, 1.0)
    rows = Bn // blocks
    tot = {k: 0.0 for k in m}
    for b in range(blocks):
        g1, g2, gb, mb = K.pairwise_fb_block(F1, F2, Bm, tF1, tF2, tB, disc, 1.0, b * rows, rows)
        sl = slice(b * rows, (b + 1) * rows)
        assert rel_err(g1.cpu(), dF1[sl].cpu()) < 2e-6 and rel_err(g2.cpu(), dF2[sl].cpu()) < 2e-6
        assert rel_err(gb.cpu(), dB[sl].cpu()) < 2e-6
        for k in tot:
            tot[k] += mb[k]
    for k in tot:
        assert tot[k] == pytest.approx(m[k], rel=2e-5, abs=1e-6), k


def test_pairwise_is_deterministic(K):
    """fixed-order reductions: two launches give bit-identical gradients and scalars"""
    rng = np.random.default_rng(9)
    t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).cuda()
    args = [t(512, 52)[:, :50] for _ in range(6)]
    disc = torch.full((512,), 0.99, device="cuda")
    a = K.pairwise_fb(*args, disc, 1.0)
    b = K.pairwise_fb(*args, disc, 1.0)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    assert a[3] == b[3]


# ------------------------------------------------------------------------------------------------ Adam + EMA
def test_adam_ema_matches_torch_optim(K):
    n = 4 * 1000 + 4 * 37
    p0, tgt0 = _r(n, seed=1), _r(n, seed=2)
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=3e-4)
    ref_t = tgt0.clone()
    p, m, v, tg = p0.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), tgt0.cuda()
    for step in range(1, 6):
        g = _r(n, seed=10 + step)
        ref_p.grad = g.clone()
        opt.step()
        with torch.no_grad():
            ref_t.copy_(0.01 * ref_p + (1 - 0.01) * ref_t)
        K.adam_ema(p, g.cuda(), m, v, tg, lr=3e-4, t=step, tau=0.01)
        assert (p.cpu() - ref_p.detach()).abs().max() < 3e-7 * step        # <= 2 ulp at |p| ~ 2 per step
        assert (tg.cpu() - ref_t).abs().max() < 3e-7 * step
    st = opt.state[ref_p]
    assert rel_err(m.cpu(), st["exp_avg"]) < 1e-6 and rel_err(v.cpu(), st["exp_avg_sq"]) < 1e-6


@pytest.mark.parametrize("d", [1, 2, 3, 31, 32, 33, 50, 64, 100, 127, 128])
def test_inverse_general_and_covariance_matrices(K, d):
    """inverse_kernel against torch.linalg.inv in fp64: (a) a general matrix whose largest column entries sit off the diagonal, so
    that nearly every pivot step picks another row (the kernel never moves rows: it scatters the result through the pivot order);
    (b) a covariance phi^T phi / n as the q_loss / z-mix paths produce, through a padded row stride and the scale argument."""
    g = torch.Generator().manual_seed(100 + d)
    A = torch.randn(d, d, generator=g) + 3.0 * torch.eye(d)[torch.randperm(d, generator=g)]
    want = torch.linalg.inv(A.double())
    got = K.inverse(A.cuda()).cpu().double()
    cond = float(torch.linalg.cond(A.double()))
    assert float((got - want).abs().max() / want.abs().max()) < 4e-7 * max(cond, 1.0), cond
    assert float((got @ A.double() - torch.eye(d, dtype=torch.float64)).abs().max()) < 2e-6 * max(cond, 1.0)
    n = 4 * d + 8
    phi = torch.randn(n, d, generator=g)
    C = torch.zeros(d, d + 5)
    C[:, :d] = phi.T @ phi
    want = torch.linalg.inv((C[:, :d].double() / n))
    got = K.inverse(C.cuda()[:, :d], scale=1.0 / n).cpu().double()
    assert float((got - want).abs().max() / want.abs().max()) < 4e-7 * float(torch.linalg.cond(C[:, :d].double()))


def test_adam_grad_scale(K):
    n = 64
    p, g = _r(n, seed=1).cuda(), _r(n, seed=2).cuda()
    p2 = p.clone()
    m, v, m2, v2 = (torch.zeros(n, device="cuda") for _ in range(4))
    K.adam_ema(p, g * 4, m, v, None, lr=1e-3, t=1, grad_scale=0.25)
    K.adam_ema(p2, g, m2, v2, None, lr=1e-3, t=1)
    assert torch.allclose(p, p2, atol=1e-8)


# ------------------------------------------------------------------------------------------------ actor loss
def test_actor_loss(K):
    rng = np.random.default_rng(2)
    Bn, d, a = 300, 50, 6
    t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    F1, F2, z = t(Bn, d).requires_grad_(True), t(Bn, d).requires_grad_(True), t(Bn, d)
    mu, act = torch.tanh(t(Bn, a)), torch.tanh(t(Bn, a))
    Q = torch.min(torch.einsum('sd, sd -> s', F1, z), torch.einsum('sd, sd -> s', F2, z))
    loss = -Q.mean()
    loss.backward()
    lp = fo.normal_log_prob(mu, 0.2, act).sum(-1).mean()
    dF1, dF2, m = K.actor_loss(_padded(F1.detach(), 52), _padded(F2.detach(), 52), _padded(z, 52), _padded(mu, 8),
                               _padded(act, 8), 0.2)
    assert rel_err(dF1.cpu(), F1.grad) < 1e-6 and rel_err(dF2.cpu(), F2.grad) < 1e-6
    assert m["actor_loss"] == pytest.approx(float(loss), rel=1e-5)
    assert m["q"] == pytest.approx(float(Q.mean()), rel=1e-5)
    assert m["actor_logprob"] == pytest.approx(float(lp), rel=1e-5)
    Q1, Q2 = torch.einsum('sd, sd -> s', F1, z), torch.einsum('sd, sd -> s', F2, z)
    assert m["q1_success"] == pytest.approx(float((Q1 > Q2).float().mean()), abs=1e-6)      # additional_metric (fb_ddpg.py:417)


def test_gemm_randomised_shapes_layouts_epilogues(K):
    """120 random problems: M, N, K in 1..300 (+ a few large), all operand layouts, aligned and unaligned leading
    dimensions (NaN in every pad element), every epilogue, optional bias-gradient column sums -- against fp64."""
    from controllable_agent_amd import _lib
    rng = np.random.default_rng(1234)
    for it in range(120):
        big = it % 10 == 0
        M, N, Kd = (int(rng.integers(1, 1200 if big else 300)) for _ in range(3))
        akc, bkc = bool(rng.integers(2)), bool(rng.integers(2))
        epi = int(rng.integers(0, 5))
        A = _r(M, Kd, seed=1000 + it) if akc else _r(Kd, M, seed=1000 + it)
        B = _r(N, Kd, seed=2000 + it) if bkc else _r(Kd, N, seed=2000 + it)
        pad = lambda t: _padded(t, t.shape[1] + int(rng.integers(0, 6)) if rng.integers(2) else (t.shape[1] + 3) // 4 * 4)
        Ad, Bd = pad(A), pad(B)
        Am = A.double() if akc else A.double().T
        Bm = B.double() if bkc else B.double().T
        ref = Am @ Bm.T
        kw = {}
        if epi in (_lib.EPI_BIAS, _lib.EPI_BIAS_RELU):
            bias = _r(N, seed=3000 + it)
            kw["bias"] = bias.cuda()
            ref = ref + bias.double()
            if epi == _lib.EPI_BIAS_RELU:
                ref = torch.relu(ref)
        elif epi in (_lib.EPI_MASK_RELU, _lib.EPI_TANH_BWD):
            aux = torch.tanh(_r(M, N, seed=4000 + it))
            kw["aux"] = _padded(aux, N + int(rng.integers(0, 3)))
            ref = ref * (aux.double() > 0) if epi == _lib.EPI_MASK_RELU else ref * (1 - aux.double() ** 2)
        want_cs = epi == 0 and not akc and bool(rng.integers(2))
        out = K.gemm(Ad, Bd, a_kcontig=akc, b_kcontig=bkc, epi=epi, want_colsum=want_cs, **kw)
        C, cs = out if want_cs else (out, None)
        tag = f"#{it} M{M} N{N} K{Kd} akc{akc} bkc{bkc} epi{epi} ldA{Ad.stride(0)} ldB{Bd.stride(0)}"
        assert torch.isfinite(C).all(), tag
        assert rel_err(C.cpu(), ref) < 3e-6, tag
        if cs is not None:
            assert rel_err(cs.cpu(), Am.sum(1)) < 3e-6, tag


def test_pairwise_randomised_sizes_and_alignment(K):
    """random batch sizes (incl. non-multiples of 32), every z_dim class of the kernel and unaligned panels (the
    predicated staging instantiation) against the fp64 closed form"""
    rng = np.random.default_rng(77)
    for it in range(24):
        Bn = int(rng.integers(2, 200)) if it % 6 else int(rng.integers(500, 1100))
        d = int(rng.choice([1, 3, 8, 16, 17, 31, 32, 50, 64, 65, 100, 128]))
        t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
        F1, F2, tF1, tF2 = t(Bn, d), t(Bn, d), t(Bn, d), t(Bn, d)
        Bm = math.sqrt(d) * torch.nn.functional.normalize(t(Bn, d), dim=1)
        tB = math.sqrt(d) * torch.nn.functional.normalize(t(Bn, d), dim=1)
        disc = torch.from_numpy(rng.uniform(0.9, 0.99, (Bn, 1)).astype(np.float32))
        cf = fo.fb_loss_closed_form(F1, F2, Bm, tF1, tF2, tB, disc, 1.3)
        ld = (d + 3) // 4 * 4 if it % 2 else d + 1 + int(rng.integers(0, 3))          # aligned / unaligned leading dimension
        dev = [_padded(x, ld) for x in (F1, F2, Bm, tF1, tF2, tB)]
        dF1, dF2, dB, m = K.pairwise_fb(*dev, disc.cuda(), 1.3)
        tag = f"#{it} B{Bn} d{d} ld{ld}"
        for k, got in (("dF1", dF1), ("dF2", dF2), ("dB", dB)):
            assert torch.isfinite(got).all() and rel_err(got.cpu(), cf[k]) < 2e-5, (tag, k)
        for k in ("fb_loss", "fb_offdiag", "fb_diag", "orth_loss"):
            assert m[k] == pytest.approx(float(cf[k]), rel=5e-5, abs=1e-5), (tag, k)


# ------------------------------------------------------------------------------------------------ embedding heads in one launch
@pytest.mark.parametrize("rows,N,Kd", [(1024, 50, 1024), (1024, 50, 576), (2048, 64, 1024), (37, 6, 40), (5, 1, 4), (16, 48, 2048),
                                       (3, 17, 36), (200, 64, 512), (64, 33, 1000)])
@pytest.mark.parametrize("normalize", [False, True])
def test_head_last_layer_in_one_launch(K, rows, N, Kd, normalize):
    """Linear(H, z) of an embedding head (+ sqrt(d) F.normalize, fb_modules.py:229) as ONE kernel (csrc/fused.hip): K split over
    the workgroup's waves and folded in wave order -- against the fp64 statement; pad columns zero; bit-reproducible."""
    x, w = _r(rows, Kd, seed=1), _r(N, Kd, seed=2, scale=0.2)
    Np = (N + 3) // 4 * 4
    b = torch.zeros(Np)
    b[:N] = _r(N, seed=3)
    ref = x.double() @ w.double().T + b[:N].double()
    scale = math.sqrt(N)
    out = K.head(x.cuda(), w.cuda(), b.cuda(), normalize=normalize, scale=scale)
    c = out[0] if normalize else out
    assert rel_err(c[:, :N].cpu(), ref) < 2e-6
    assert torch.count_nonzero(c[:, N:]) == 0
    if normalize:
        nrm = ref.norm(dim=1)
        assert rel_err(out[2].cpu(), nrm) < 2e-6
        assert rel_err(out[1][:, :N].cpu(), scale * ref / nrm.clamp_min(1e-12)[:, None]) < 3e-6
        assert torch.count_nonzero(out[1][:, N:]) == 0
    again = K.head(x.cuda(), w.cuda(), b.cuda(), normalize=normalize, scale=scale, replicas=4)
    assert torch.equal(again[0] if normalize else again, c)


def test_head_refuses_what_it_cannot_run(K):
    x, w, b = torch.zeros(8, 64, device="cuda"), torch.zeros(100, 64, device="cuda"), torch.zeros(100, device="cuda")
    with pytest.raises(RuntimeError, match="N <= 64"):
        K.head(x, w, b)


@pytest.mark.parametrize("M,N,Kd,lay", [(1024, 2048, 1024, "NT"), (2048, 1024, 1024, "TN"), (1024, 1024, 2048, "NN"), (1024, 512, 1024, "NT"),
                                        (512, 1024, 1024, "TN"), (576, 576, 1024, "TN"), (1000, 520, 96, "NT"), (128, 256, 64, "NT")])
def test_gemm_tile_to_xcd_maps_give_the_same_bits(K, M, N, Kd, lay, monkeypatch):
    """Which XCD's L2 a tile's operands go through is a permutation of the launch's workgroups (gemm.hip: bands of tile rows -- the
    default -- or, FBHIP_GEMM_XCD2D=1, an xgm x xgn grid of tile blocks that minimises what one L2 has to hold: FETCH_SIZE 753 -> 545
    MB raw per update, the step 0.6 % slower).  The results must not depend on it, bit for bit, incl. shapes whose tile counts do not
    divide (those keep the bands)."""
    akc, bkc = {"NT": (True, True), "NN": (True, False), "TN": (False, False)}[lay]
    A = _r(*((M, Kd) if akc else (Kd, M)), seed=1).cuda()
    B = _r(*((N, Kd) if bkc else (Kd, N)), seed=2).cuda()
    c0 = K.gemm(A, B, a_kcontig=akc, b_kcontig=bkc)
    monkeypatch.setenv("FBHIP_GEMM_XCD2D", "1")
    c1 = K.gemm(A, B, a_kcontig=akc, b_kcontig=bkc)
    monkeypatch.delenv("FBHIP_GEMM_XCD2D")
    assert torch.equal(c0, c1)
    ref = (A.double() if akc else A.double().T) @ (B.double().T if bkc else B.double())
    assert rel_err(c1.cpu(), ref.cpu()) < 2e-6


# ------------------------------------------------------------------------------------------------ the actor's a-wide seams
def _ln_stats64(x64):
    mean = x64.mean(1, keepdim=True)
    var = ((x64 - mean) ** 2).mean(1, keepdim=True)
    return mean, 1.0 / torch.sqrt(var + 1e-5)


@pytest.mark.parametrize("tiles", ["0", "3"])
@pytest.mark.parametrize("rows,H,a,aoff,ldw1", [(100, 1024, 12, 78, 96), (64, 512, 6, 24, 32), (37, 512, 3, 5, 8), (2048, 1024, 12, 78, 96)])
def test_policy_head_and_first_layer_vs_fp64(monkeypatch, tiles, rows, H, a, aoff, ldw1):
    """premu = p W4^T + b4, mu = tanh, TruncatedNormal sample with clip and straight-through clamp (fb_modules.py:117-126,
    utils.py:176-185), then the first layer of the trunk that consumes the action: tanh(LayerNorm(base + W1[:, action] action))
    (fb_modules.py:190) with the pre-activation and (mean, rstd) kept -- row kernel (FBHIP_HEAD_TILES=0, rowops.hip) and 16-row MFMA
    tiles (=3, headtiles.hip) against an fp64 statement; ragged row counts, action columns at 16-byte-aligned and unaligned offsets."""
    from controllable_agent_amd import kernels as K
    monkeypatch.setenv("FBHIP_HEAD_TILES", tiles)
    g = torch.Generator(device="cuda").manual_seed(rows + H + a)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    P = torch.relu(rn(rows, H))
    W4, b4 = rn(a, H) / H ** 0.5, rn(16)[:a] * 0.1
    noise = rn(rows, a)
    W1 = rn(H, ldw1) * 0.3
    W1a = W1[:, aoff:aoff + a]
    base, gamma, beta = rn(rows, H), 1.0 + 0.1 * rn(H), 0.1 * rn(H)
    stddev, clip = 0.2, 0.3
    premu, mu, action, t1, pre, stats = K.policy_head(P, W4, b4, noise, stddev, clip, base, W1a, gamma, beta, keep_pre=True)
    torch.cuda.synchronize()
    d = lambda t: t.double()
    premu64 = d(P) @ d(W4).T + d(b4)
    mu64 = torch.tanh(premu64)
    act64 = torch.clamp(mu64 + torch.clamp(d(noise) * stddev, -clip, clip), -1.0 + 1e-6, 1.0 - 1e-6)
    pre64 = d(base) + act64 @ d(W1a).T
    mean, rstd = _ln_stats64(pre64)
    t164 = torch.tanh((pre64 - mean) * rstd * d(gamma) + d(beta))
    for name, got, ref, tol in (("premu", premu, premu64, 2e-6), ("mu", mu, mu64, 2e-6), ("action", action, act64, 2e-6),
                                ("pre", pre, pre64, 2e-6), ("t1", t1, t164, 1e-5)):
        err = float((d(got) - ref).abs().max() / ref.abs().max())
        assert err < tol, (name, err)
    st = stats.view(rows, 2).double()
    assert float((st[:, 0:1] - mean).abs().max()) < 1e-5 and float(((st[:, 1:2] - rstd) / rstd).abs().max()) < 1e-5
    # the plain head (no first layer, no noise): action = mu
    premu2, mu2, action2, *_ = K.policy_head(P, W4, b4, None, stddev, clip)
    assert float((d(premu2) - premu64).abs().max() / premu64.abs().max()) < 2e-6 and torch.equal(mu2, action2)


@pytest.mark.parametrize("tiles", ["0", "3"])
@pytest.mark.parametrize("rows,H,a,aoff,ldw1", [(100, 1024, 12, 78, 96), (64, 512, 6, 24, 32), (37, 512, 3, 5, 8), (2048, 1024, 12, 78, 96)])
def test_actor_head_bwd_vs_fp64(monkeypatch, tiles, rows, H, a, aoff, ldw1):
    """d action = LayerNormTanhBackward(dt1) W1[:, action columns]; d premu = d action (1 - mu^2) (straight-through clamp + tanh,
    utils.py:171-174); d p = (d premu W4) relu'(p) (fb_modules.py:119-124 reversed) -- both kernel forms against an fp64 statement."""
    from controllable_agent_amd import kernels as K
    monkeypatch.setenv("FBHIP_HEAD_TILES", tiles)
    g = torch.Generator(device="cuda").manual_seed(7 * rows + H + a)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    d = lambda t: t.double()
    x, gamma, beta = rn(rows, H), 1.0 + 0.1 * rn(H), 0.1 * rn(H)
    mean, rstd = _ln_stats64(d(x))
    y = torch.tanh((d(x) - mean) * rstd * d(gamma) + d(beta)).float()
    stats = torch.cat([mean, rstd], 1).float().contiguous().view(-1)
    dt1 = rn(rows, H)
    W1 = rn(H, ldw1) * 0.3
    W1a = W1[:, aoff:aoff + a]
    W4 = rn(a, H) / H ** 0.5
    mu = torch.tanh(rn(rows, (a + 3) // 4 * 4))[:, :a]
    P = rn(rows, H)
    dpremu, dp = K.actor_head_bwd(dt1, y, x, stats, gamma, W1a, mu, W4, P)
    torch.cuda.synchronize()
    du = d(dt1) * (1.0 - d(y) ** 2)
    gg = du * d(gamma)
    xh = (d(x) - d(stats.view(rows, 2)[:, 0:1])) * d(stats.view(rows, 2)[:, 1:2])
    dx = d(stats.view(rows, 2)[:, 1:2]) * (gg - gg.mean(1, keepdim=True) - xh * (gg * xh).mean(1, keepdim=True))
    dpremu64 = (dx @ d(W1a)) * (1.0 - d(mu) ** 2)
    dp64 = (dpremu64 @ d(W4)) * (d(P) > 0)
    for name, got, ref in (("dpremu", dpremu, dpremu64), ("dp", dp, dp64)):
        err = float((d(got) - ref).abs().max() / ref.abs().max())
        assert err < 5e-6, (name, err)
    assert torch.equal(dp == 0, ~(P > 0) | (dp == 0))                                # masked entries are exact zeros
