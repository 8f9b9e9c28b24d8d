"""Pin the oracle (oracle/fb_oracle.py) to the reference: replay every golden trace that
tests/golden/make_golden.py recorded from the real FBDDPGAgent / ReplayBuffer."""
import numpy as np
import pytest
import torch

from oracle import fb_oracle as fo
from oracle import discrete_fb_oracle as do
from tests import helpers as H


def _replay(name, full_state):
    meta = H.load_meta(name)
    cfg, nets, storage, lengths, rng = H.regenerate_inputs(meta)
    z = np.load(H.GOLDEN / f"{name}.npz") if full_state else None
    if full_state:      # the explicit arrays must equal the regenerated ones
        for n, p in nets.items():
            for k, v in p.items():
                np.testing.assert_array_equal(z[f"init/{n}/{k}"], v.numpy())
        for k, v in storage.items():
            np.testing.assert_array_equal(z[f"storage/{k}"], v)
    agent = do.DiscreteOracleAgent(cfg, nets) if meta.get("discrete") else fo.OracleAgent(cfg, nets)
    out = []
    for s in range(meta["n_steps"]):
        d = fo.make_draws(rng, cfg, meta["n_eps"], lengths)
        if full_state:
            np.testing.assert_array_equal(z[f"draws/{s}/perm"], d.perm)
            np.testing.assert_array_equal(z[f"draws/{s}/z_gauss"], d.z_gauss)
        if full_state and d.future_idx is not None:
            np.testing.assert_array_equal(z[f"draws/{s}/future_idx"], d.future_idx)
        batch = fo.gather_batch(storage, d.ep_idx, d.step_idx, cfg.discount, d.future_idx)
        m = agent.update(batch, d)
        out.append((m, agent.state_tensors() if (full_state or str(s + 1) in meta["checksums"]) else None))
    return meta, z, out


@pytest.mark.parametrize("name", ["tiny_trace", "tiny_goal_trace", "tiny_future_trace", "tiny_future_goal_trace", "tiny_nonorm_trace",
                                  "tiny_randw_trace", "tiny_randw_nonorm_trace", "tiny_trunk_trace",
                                  "tiny_single_trunk_trace", "tiny_single_trunk_goal_trace",
                                  "tiny_boltzmann_trace", "tiny_boltzmann_goal_trace", "tiny_debug_trace", "tiny_debug_goal_trace", "tiny_debug_future_randw_trace",
                                  "tiny_discrete_trace", "tiny_discrete_boltz_trace", "tiny_discrete_debug_trace"])
def test_tiny_traces_full_state(name):
    """Every parameter / target / Adam tensor after every step, tiny dims (incl. goal_space, q_loss,
    variable episode lengths, lr_coef != 1, hindsight replay with future_ratio > 0 on future < 1 buffers, norm_z=False,
    rand_weight=True, add_trunk=True, preprocess=False, boltzmann=True; the last two: DiscreteFBAgent, oracle/discrete_fb_oracle.py).  Tolerance: abs 2e-6 on params (fp32, Adam lr 1e-3)."""
    meta, z, out = _replay(name, True)
    for s, (m, state) in enumerate(out):
        for k, v in meta["metrics"][s].items():
            assert m[k] == pytest.approx(v, rel=2e-5, abs=1e-6), (s, k)
        for k, v in state.items():
            ref = z[f"state/{s}/{k}"]
            np.testing.assert_allclose(v, ref, rtol=1e-4, atol=2e-6, err_msg=f"step {s} {k}")


@pytest.mark.parametrize("name,tol", [("walker_b256", 2e-4), ("walker_b1024", 1e-4), ("quadruped_goal_b512", 1e-4),
                                      ("walker_b256_50", 2e-4), ("quadruped_goal_b256_50", 2e-4),      # 50 steps, checksums at 1/10/50
                                      ("walker_b1024_50", 2e-4),                                      # configs[1] as benchmarked, 50 steps
                                      ("quadruped_goal_b2048", 1e-4)])                                # configs[2] at its batch size
def test_full_dim_metric_curves(name, tol):
    """Full network dims: loss curves + parameter checksums.  Free-running, so the tolerance is the
    reference's own 1-vs-8-thread envelope, not bit equality."""
    meta, _, out = _replay(name, False)
    for s, (m, state) in enumerate(out):
        scale = max(1.0, abs(meta["metrics"][s]["fb_offdiag"]))     # fb_diag / q / actor_loss are sums of O(1) terms that can
        for k in H.LOSS_KEYS:                                        # sit near zero: absolute bound at the loss scale
            assert m[k] == pytest.approx(meta["metrics"][s][k], rel=tol * (1 + s), abs=max(1e-5, tol * (1 + s) * scale)), (s, k)
        assert m["B_norm"] == pytest.approx(np.sqrt(meta["cfg"]["z_dim"]), rel=1e-5)
        assert m["orth_loss_diag"] == pytest.approx(-2 * meta["cfg"]["z_dim"], rel=1e-5)
        if state is not None:
            ref = meta["checksums"][str(s + 1)]
            for k, (ssum, l2) in H.checksums(state).items():
                # (bit-identical with the thread count the fixture was made with; other counts drift like BASELINE.md section 2)
                assert l2 == pytest.approx(ref[k][1], rel=1e-5 * (1 + s / 4)), (s, k)
                assert ssum == pytest.approx(ref[k][0], rel=1e-3 * (1 + s), abs=1e-3 * (1 + s)), (s, k)


def test_sampler_kat():
    """ReplayBuffer.sample index arithmetic incl. variable lengths, goal pair and stored meta."""
    z = np.load(H.GOLDEN / "sampler_kat.npz")
    storage = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith("storage/")}
    b = fo.gather_batch(storage, z["ep_idx"], z["step_idx"], 0.98)
    for k in ("obs", "action", "next_obs", "reward", "discount", "goal", "next_goal"):
        np.testing.assert_array_equal(b[k], z[k], err_msg=k)
    np.testing.assert_array_equal(z["meta_z"][z["ep_idx"], z["step_idx"] - 1], z["meta_z_out"])
    assert (z["step_idx"] >= 1).all() and (z["step_idx"] <= z["lengths"][z["ep_idx"]]).all()


def test_closed_form_gradients_match_autograd():
    """The mask-free closed form (what the HIP pairwise kernel implements) == autograd of the faithful
    masked statement."""
    rng = np.random.default_rng(5)
    Bn, d = 48, 10
    t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    F1, F2, y, tF1, tF2, ty = t(Bn, d), t(Bn, d), t(Bn, d), t(Bn, d), t(Bn, d), t(Bn, d)
    Bm = (np.sqrt(d) * torch.nn.functional.normalize(y, dim=1)).requires_grad_(True)
    tB = np.sqrt(d) * torch.nn.functional.normalize(ty, dim=1)
    F1.requires_grad_(True), F2.requires_grad_(True)
    disc = torch.from_numpy(rng.uniform(0.9, 1.0, (Bn, 1)).astype(np.float32))
    L = fo.fb_loss_terms(F1, F2, Bm, tF1, tF2, tB, disc, 0.7)
    L["fb_loss"].backward()
    cf = fo.fb_loss_closed_form(F1.detach(), F2.detach(), Bm.detach(), tF1, tF2, tB, disc, 0.7)
    for k, g in (("dF1", F1.grad), ("dF2", F2.grad), ("dB", Bm.grad)):
        assert H.rel_err(g.numpy(), cf[k].numpy()) < 2e-6, k
    for k in ("fb_loss", "fb_offdiag", "fb_diag", "orth_loss"):
        assert float(L[k]) == pytest.approx(float(cf[k]), rel=1e-5)


def test_inference_kat():
    z = np.load(H.GOLDEN / "inference_kat.npz")
    cfg = fo.OracleConfig(obs_dim=5, action_dim=3, goal_dim=5, z_dim=8, hidden_dim=32, feature_dim=16,
                          backward_hidden_dim=18, batch_size=16)
    nets = {n: {k.split("/", 2)[2]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"init/{n}/")}
            for n in ("actor", "forward_net", "backward_net")}
    ag = fo.OracleAgent(cfg, nets)
    acts = np.stack([ag.act_mean(z["obs"][i], z["z"][i]) for i in range(7)])
    np.testing.assert_allclose(acts, z["act_eval"], rtol=1e-5, atol=1e-6)
    zi = ag.infer_z(torch.from_numpy(z["goal_obs"]), torch.from_numpy(z["reward"]))
    np.testing.assert_allclose(zi, z["z_inferred"], rtol=1e-5, atol=1e-6)
    Bout = fo.backward_map(ag.backward_net, torch.from_numpy(z["goal_obs"]), cfg.z_dim).numpy()
    np.testing.assert_allclose(Bout, z["backward_out"], rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------------ SFAgent (row n4, second sibling)
def sf_trace_inputs(name):
    """(meta, npz, cfg, nets, storage, lengths) of a tiny_sf_* trace (tests/golden/make_golden.py::sf_fixture)"""
    meta = H.load_meta(name)
    z = np.load(H.GOLDEN / f"{name}.npz")
    cfg = H.cfg_from_meta(meta)
    nets = {n: {k.split("/", 2)[2]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"init/{n}/")}
            for n in ("actor", "successor_net", "feature_learner")}
    storage = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith("storage/")}
    return meta, z, cfg, nets, storage, z["lengths"]


@pytest.mark.parametrize("name", ["tiny_sf_icm_trace", "tiny_sf_lap_trace", "tiny_sf_random_trace", "tiny_sf_autoencoder_trace", "tiny_sf_transition_trace",
                                  "tiny_sf_svdp_trace", "tiny_sf_svdp_goal_trace", "tiny_sf_latent_trace",
                                  "tiny_sf_svdsr_trace", "tiny_sf_svdsr_goal_trace", "tiny_sf_svdsrv2_trace",
                                  "tiny_sf_contrastive_trace", "tiny_sf_contrastive_goal_trace", "tiny_sf_contrastivev2_trace",
                                  "tiny_sf_identity_trace", "tiny_sf_mix_icm_trace", "tiny_sf_mix_lap_trace", "tiny_sf_mix_identity_trace"])
def test_sf_oracle_full_state_against_the_reference(name):
    """oracle/sf_oracle.py against traces of the real url_benchmark.agent.sf.SFAgent: metrics and every parameter / target /
    Adam tensor after every step (icm + scalar Q regression; lap + feature-space regression + goal space + variable lengths)."""
    from oracle import sf_oracle as so
    meta, z, cfg, nets, storage, lengths = sf_trace_inputs(name)
    agent = so.SFOracleAgent(cfg, nets, meta["feature_learner"], meta["sf_q_loss"])
    for s in range(meta["n_steps"]):
        d = fo.Draws(**{f: z[f"draws/{s}/{f}"] for f in fo.Draws.__dataclass_fields__ if f"draws/{s}/{f}" in z.files})
        m = agent.update(fo.gather_batch(storage, d.ep_idx, d.step_idx, cfg.discount, d.future_idx), d)
        for k, v in meta["metrics"][s].items():
            # (phi_loss of the low-rank learners is a difference of O(1..10) terms: its last digits follow the thread count's summation order)
            assert m[k] == pytest.approx(v, rel=2e-5, abs=1e-5 if k == "phi_loss" else 1e-6), (s, k)
        for k, v in agent.state_tensors().items():
            if f"state/{s}/{k}" not in z.files:               # no Adam state in the reference: "random" has no phi_opt; latent's target net has no gradients
                assert k.startswith(("adam_m/feature_learner", "adam_v/feature_learner"))
                assert meta["feature_learner"] == "random" or "/target_" in k, k
                assert float(np.abs(v).max()) == 0.0, k
                continue
            np.testing.assert_allclose(v, z[f"state/{s}/{k}"], rtol=1e-4, atol=2e-6, err_msg=f"step {s} {k}")
