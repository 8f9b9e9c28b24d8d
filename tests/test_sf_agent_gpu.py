"""SFHipAgent (second sibling: url_benchmark/agent/sf.py) on the GPU against
  (1) the traces recorded from the real reference SFAgent (tests/golden/tiny_sf_*_trace, teacher-forced), and
  (2) oracle/sf_oracle.py (pinned to the same traces on CPU) at larger dims."""
import pickle

import numpy as np
import pytest
import torch

from oracle import fb_oracle as fo
from oracle import sf_oracle as so
from tests import helpers as H
from tests.test_oracle_golden import sf_trace_inputs
from tests.test_update_parity_gpu import GRAD_REL_L2, LOSS_RTOL, _buffer, _param_close, _v_from_trace

pytestmark = pytest.mark.gpu

SF_NETS = ("actor", "successor_net", "feature_learner")


def sf_kwargs(cfg: fo.OracleConfig, learner: str, q_loss: bool, goal_space=None, metrics=True, **extra):
    kw = dict(obs_type="states", obs_shape=(cfg.obs_dim,), action_shape=(cfg.action_dim,), device="cuda", num_expl_steps=0,
              use_tb=metrics, use_wandb=False, use_hiplog=False, goal_space=goal_space, lr=cfg.lr, lr_coef=cfg.lr_coef,
              sf_target_tau=cfg.fb_target_tau, hidden_dim=cfg.hidden_dim, backward_hidden_dim=cfg.backward_hidden_dim,
              feature_dim=cfg.feature_dim, z_dim=cfg.z_dim, stddev_schedule=str(cfg.stddev), stddev_clip=cfg.stddev_clip,
              batch_size=cfg.batch_size, q_loss=q_loss, feature_learner=learner, mix_ratio=cfg.mix_ratio, update_every_steps=1,
              add_trunk=cfg.add_trunk, preprocess=cfg.preprocess)
    kw.update(extra)
    return kw


def make_sf_agent(cfg, nets, learner, q_loss, goal_space=None, metrics=True):
    from controllable_agent_amd.agent import SFHipAgent
    ag = SFHipAgent(**sf_kwargs(cfg, learner, q_loss, goal_space, metrics))
    ag.load_nets({n: dict(p) for n, p in nets.items()})
    return ag


def get_sf_state(agent) -> dict:
    out = {}
    for n in SF_NETS + ("successor_target_net",):
        for k, v in getattr(agent, n).state_dict().items():
            out[f"{n}/{k}"] = v.detach().cpu().numpy().copy()
    for n in SF_NETS:
        for mv in ("m", "v"):
            for k, v in agent._adam_views[n][mv].items():
                out[f"adam_{mv}/{n}/{k}"] = v.detach().cpu().numpy().copy()
    return out


def set_sf_state(agent, state: dict, steps: int) -> None:
    for n in SF_NETS + ("successor_target_net",):
        getattr(agent, n).load_state_dict({k.split("/", 1)[1]: torch.from_numpy(np.asarray(v)) for k, v in state.items()
                                           if k.startswith(n + "/")})
    for n in SF_NETS:
        for mv in ("m", "v"):
            for k, view in agent._adam_views[n][mv].items():
                key = f"adam_{mv}/{n}/{k}"
                view.copy_(torch.from_numpy(np.asarray(state[key]))) if key in state else view.zero_()
    agent.set_step_counts(steps, steps)


@pytest.mark.parametrize("name", ["tiny_sf_icm_trace", "tiny_sf_lap_trace", "tiny_sf_random_trace", "tiny_sf_autoencoder_trace", "tiny_sf_transition_trace",
                                  "tiny_sf_svdp_trace", "tiny_sf_svdp_goal_trace", "tiny_sf_latent_trace",
                                  "tiny_sf_svdsr_trace", "tiny_sf_svdsr_goal_trace", "tiny_sf_svdsrv2_trace",
                                  "tiny_sf_contrastive_trace", "tiny_sf_contrastive_goal_trace", "tiny_sf_contrastivev2_trace",
                                  "tiny_sf_identity_trace", "tiny_sf_mix_icm_trace", "tiny_sf_mix_lap_trace", "tiny_sf_mix_identity_trace"])
def test_sf_teacher_forced_against_reference_trace(name):
    """Each step starts from the REFERENCE SFAgent's recorded state, runs one HIP update with the recorded draws and must land
    on the reference's next state and metrics; intermediates and gradients are compared with the oracle's autograd."""
    meta, z, cfg, nets, storage, lengths = sf_trace_inputs(name)
    learner, q_loss = meta["feature_learner"], meta["sf_q_loss"]
    agent = make_sf_agent(cfg, nets, learner, q_loss, meta["goal_space"])
    rb = _buffer(storage, lengths, cfg.discount, cfg.future)
    oracle = so.SFOracleAgent(cfg, nets, learner, q_loss)
    B = cfg.batch_size
    for s in range(meta["n_steps"]):
        draws = fo.Draws(**{f: z[f"draws/{s}/{f}"] for f in fo.Draws.__dataclass_fields__ if f"draws/{s}/{f}" in z.files})
        if s > 0:
            set_sf_state(agent, {k.split("/", 2)[2]: z[k] for k in z.files if k.startswith(f"state/{s - 1}/")}, s)
        oracle.update(fo.gather_batch(storage, draws.ep_idx, draws.step_idx, cfg.discount, draws.future_idx), draws, keep=True)
        m = agent.update_injected(rb, s, H.draws_dict(draws))
        for k, v in meta["metrics"][s].items():
            # (phi_loss of the low-rank learners is a difference of O(1..10) terms: absolute tolerance of that scale)
            assert m[k] == pytest.approx(v, rel=LOSS_RTOL if k not in ("target_F", "F1", "phi") else 2e-4, abs=2e-5 if k == "phi_loss" else 2e-6), (s, k)
        L = oracle.last
        phi2 = agent.workspace_view("phi2").cpu()
        assert H.rel_err(phi2[:B], L["phi"]) < 2e-5 and H.rel_err(phi2[B:2 * B], L["next_phi"]) < 2e-5, s   # (contrastivev2: a third block, phi(future_goal))
        for view, ref in (("z", L["z"]), ("next_action", L["next_action"]), ("F1", L["F1"]), ("F2", L["F2"]), ("tF1", L["nF1"]),
                          ("tF2", L["nF2"]), ("mu", L["mu"]), ("pi_action", L["pi_action"])):
            assert H.rel_err(agent.workspace_view(view).cpu(), ref) < 2e-5, (s, view)
        for view, ref in (("dF1", L["dF1"]), ("dF2", L["dF2"]), ("d_premu", L["d_premu"])):
            assert H.rel_err(agent.workspace_view(view).cpu(), ref) < GRAD_REL_L2, (s, view)
        dphi2 = agent.workspace_view("dphi2").cpu()
        # "random" has no feature loss (sf.py:447-449); autoencoder / transition never read next_phi, svd_p never reads phi
        for got, ref, at in ((dphi2[:B], L["dphi"], L["phi"]), (dphi2[B:2 * B], L["dnext_phi"], L["next_phi"])):
            if float(ref.abs().max()) == 0.0:
                assert float(got.abs().max()) == 0.0, s
            else:
                if learner.startswith("contrastive"):     # F.normalize(phi) drops the radial part of the gradient; here feature_net's own L2 stage does
                    got = got - at * (got * at).sum(1, keepdim=True) / (at * at).sum(1, keepdim=True)
                assert H.rel_err(got, ref) < GRAD_REL_L2, s
        for net, key in (("successor_net", "grads_successor"), ("feature_learner", "grads_feature"), ("actor", "grads_actor")):
            for k, g in agent._grad_views[net].state_dict().items():
                ref = L[key].get(k, torch.zeros(1)) if learner in ("random", "identity") else L[key][k]
                if float(ref.abs().max()) == 0.0:
                    assert float(g.abs().max()) == 0.0, (s, net, k)
                else:
                    assert H.rel_err(g.cpu(), ref) < GRAD_REL_L2, (s, net, k)
        for k, v in get_sf_state(agent).items():
            if f"state/{s}/{k}" not in z.files:               # "random": no phi_opt in the reference; ours must not have moved
                assert learner in ("random", "identity") and k.startswith("adam_") and "feature_learner" in k and float(np.abs(v).max()) == 0.0, k
                continue
            ref = z[f"state/{s}/{k}"]
            if k.startswith("adam_"):
                assert H.rel_err(v, ref) < 2e-4, (s, k)
            else:
                _param_close(v, ref, cfg.lr * max(1.0, cfg.lr_coef), f"step {s} {k}", _v_from_trace(z, s, k), s + 1)
        assert agent.step_counts() == (s + 1, s + 1)
        for nv in (agent.successor_net, agent.feature_learner, agent.actor, agent.successor_target_net, *agent._grad_views.values()):
            assert nv.pad_abs_max() == 0.0, (s, nv._name)


@pytest.mark.parametrize("learner,q_loss,goal", [("icm", True, False), ("lap", False, True), ("icm", False, True), ("lap", True, False),
                                                 ("random", True, False), ("autoencoder", False, True), ("autoencoder", True, False),
                                                 ("transition", True, False), ("transition", False, True), ("svd_p", True, False),
                                                 ("svd_p", False, True), ("latent", True, False), ("svd_sr", True, False),
                                                 ("svd_sr", False, True), ("svd_srv2", True, False), ("contrastive", True, False),
                                                 ("contrastive", False, True), ("contrastivev2", True, True)])
def test_sf_free_running_at_full_width_against_the_oracle(learner, q_loss, goal):
    """hidden 1024 / feature 512 / Hb 512 (the reference defaults), z 100, batch 256, walker-sized inputs: three free-running
    updates against the oracle (metrics + parameter checksums)."""
    _free_running_full_width(learner, q_loss, goal, 0.0)


@pytest.mark.parametrize("learner,q_loss,goal,z_dim", [("icm", True, False, 50), ("lap", False, True, 50), ("svd_sr", True, False, 50)])
def test_sf_mix_free_running_at_full_width_against_the_oracle(learner, q_loss, goal, z_dim):
    """mix_ratio = 0.5 at the full widths (sf.py:725-739): half of the tasks are whitened features of permuted next goals.  The d x d
    covariance of 256 unit-norm features is inverted (fp64 Gauss-Jordan here, SVD pinv in the oracle)."""
    _free_running_full_width(learner, q_loss, goal, 0.5, z_dim)


def _free_running_full_width(learner, q_loss, goal, mix_ratio, z_dim=100):
    cfg = fo.OracleConfig(obs_dim=24, action_dim=6, goal_dim=3 if goal else 24, z_dim=z_dim, backward_hidden_dim=512, batch_size=256,
                          lr_coef=5.0, mix_ratio=mix_ratio, use_goal=goal, future=0.8 if learner.startswith("contrastive") else 1.0)
    rng = np.random.default_rng(41)
    shapes = so.net_shapes(cfg, learner)
    nets = {n: fo.synthetic_params(rng, shapes[n]) for n in shapes}
    storage, lengths = fo.synthetic_storage(rng, 10, 40, cfg.obs_dim, cfg.action_dim, cfg.goal_dim if goal else None)
    agent = make_sf_agent(cfg, nets, learner, q_loss, "simplified_walker" if goal else None)
    rb = _buffer(storage, lengths, cfg.discount, cfg.future)
    torch.set_num_threads(8)
    oracle = so.SFOracleAgent(cfg, nets, learner, q_loss)
    for s in range(3):
        d = fo.make_draws(rng, cfg, 10, lengths)
        mo = oracle.update(fo.gather_batch(storage, d.ep_idx, d.step_idx, cfg.discount, d.future_idx), d)
        m = agent.update_injected(rb, s, H.draws_dict(d))
        for k in ("sf_loss", "phi_loss", "actor_loss", "target_F", "phi_norm", "z_norm"):
            if k == "phi_loss" and learner == "random":
                assert k not in m and k not in mo
                continue
            assert m[k] == pytest.approx(mo[k], rel=3e-4 * (1 + s), abs=3e-5 * (1 + s)), (s, k)
    got, want = get_sf_state(agent), oracle.state_tensors()
    for k, v in want.items():
        if not k.startswith("adam_"):
            # (bias / LayerNorm vectors of a few entries move by ~lr per Adam step whatever the gradient's size: one sign-level
            # difference on one entry is 1e-5 of such a vector's norm after three steps -- matrices average that out)
            assert float(np.linalg.norm(got[k])) == pytest.approx(float(np.linalg.norm(v)), rel=2e-5 if v.ndim == 2 else 2e-4), k


@pytest.mark.parametrize("learner", ["icm", "lap", "random", "autoencoder", "transition", "svd_p", "latent", "svd_sr", "svd_srv2", "contrastive", "contrastivev2"])
def test_sf_constructor_init_matches_reference_seed(learner):
    """same torch.manual_seed => SFAgent's orthogonal init tensor for tensor (sf.py:419-463; ICM re-applies weight_init)"""
    z = np.load(H.GOLDEN / f"init_seed1_tiny_sf_{learner}.npz")
    if str(z["torch_version"]) != torch.__version__:
        pytest.skip("fixture generated with another torch build")
    cfg = fo.OracleConfig(obs_dim=5, action_dim=3, goal_dim=5, z_dim=10, hidden_dim=32, feature_dim=16, backward_hidden_dim=20,
                          batch_size=16, lr=1e-3, lr_coef=5.0, mix_ratio=0.0)
    from controllable_agent_amd.agent import SFHipAgent
    torch.manual_seed(1)
    agent = SFHipAgent(**sf_kwargs(cfg, learner, True))
    for k, v in get_sf_state(agent).items():
        if not k.startswith("adam_"):
            np.testing.assert_allclose(v, z[k], rtol=0, atol=2e-6, err_msg=k)


def test_sf_pickle_init_from_update_many_and_inference():
    meta, z, cfg, nets, storage, lengths = sf_trace_inputs("tiny_sf_icm_trace")
    rb = _buffer(storage, lengths, cfg.discount)
    a1 = make_sf_agent(cfg, nets, "icm", True)
    for s in range(2):
        a1.update(rb, s)
    a2 = pickle.loads(pickle.dumps(a1))
    a3 = make_sf_agent(cfg, nets, "icm", True)
    a3.init_from(a1)
    s1 = get_sf_state(a1)
    for other in (a2, a3):
        for k, v in get_sf_state(other).items():
            np.testing.assert_array_equal(v, s1[k], err_msg=k)
        assert other.step_counts() == (2, 2)
    # update_many == the same number of single updates (same device-drawn batches: the RNG counters travelled with the pickle)
    a1.update_many(rb, 2, 3)
    for s in range(3):
        a2.update(rb, 2 + s)
    s1, s2 = get_sf_state(a1), get_sf_state(a2)
    for k in s1:
        np.testing.assert_array_equal(s1[k], s2[k], err_msg=k)
    # inference surface (sf.py:509-568) against torch on the same weights
    oracle_p = {k: v.detach().cpu() for k, v in a1.feature_learner.state_dict().items()}
    rng = np.random.default_rng(4)
    goal = rng.standard_normal((64, cfg.goal_dim)).astype(np.float32)
    phi_ref = so.feature_net(oracle_p, torch.from_numpy(goal), cfg.z_dim)
    np.testing.assert_allclose(a1.feature_learner.feature_net(goal).cpu().numpy(), phi_ref.numpy(), rtol=2e-5, atol=2e-6)
    a1.inv_cov = a1._compute_cov(goal)
    cov = phi_ref.double().T @ phi_ref.double() / 64
    np.testing.assert_allclose(a1.inv_cov.cpu().numpy(), torch.linalg.pinv(cov).float().numpy(), rtol=2e-3, atol=2e-4)
    zg = a1.get_goal_meta(goal[0])["z"]
    want = np.sqrt(cfg.z_dim) * torch.nn.functional.normalize(phi_ref[:1] @ a1.inv_cov.cpu(), dim=1)[0].numpy()
    np.testing.assert_allclose(zg, want, rtol=1e-4, atol=1e-5)
    reward = rng.standard_normal((64, 1)).astype(np.float32)
    zi = a1.infer_meta_from_obs_and_rewards(torch.from_numpy(goal), torch.from_numpy(reward))["z"]
    sol = torch.linalg.lstsq(phi_ref.double(), torch.from_numpy(reward).double()).solution
    np.testing.assert_allclose(zi, (np.sqrt(cfg.z_dim) * torch.nn.functional.normalize(sol.float(), dim=0)).squeeze().numpy(), rtol=1e-3, atol=1e-4)
    act = a1.act(goal[0, :cfg.obs_dim] if cfg.goal_dim >= cfg.obs_dim else rng.standard_normal(cfg.obs_dim).astype(np.float32), {"z": zg}, 0, eval_mode=True)
    assert act.shape == (cfg.action_dim,) and np.all(np.abs(act) <= 1)
    assert not hasattr(a1, "fb_opt") and not hasattr(a1, "forward_net") and a1.sf_opt.param_groups[0]["lr"] == cfg.lr
    assert a1.phi_opt.param_groups[0]["lr"] == pytest.approx(cfg.lr_coef * cfg.lr)


def test_sf_unsupported_options_fail_loudly():
    from controllable_agent_amd.agent import SFHipAgent
    base = dict(obs_type="states", obs_shape=(5,), action_shape=(3,), num_expl_steps=0)
    for bad in (dict(feature_learner="no_such_learner"), dict(mix_ratio=0.3, batch_size=8, z_dim=10), dict(boltzmann=True),
                dict(num_sf_updates=0)):
        with pytest.raises(NotImplementedError):
            SFHipAgent(**{**base, **bad})


def test_sf_agent_from_reference_checkpoint_file():
    """a checkpoint written by the REAL reference holding a live SFAgent (pretrain.py:437-449) -> SFHipAgent.from_reference_checkpoint
    without the reference installed: nets, target, Adam moments, step counts, and the same greedy actions"""
    from controllable_agent_amd.agent import SFHipAgent
    from controllable_agent_amd.replay import DeviceReplayBuffer
    exp = np.load(H.GOLDEN / "ref_checkpoint_sf_expect.npz")
    agent = SFHipAgent.from_reference_checkpoint(H.GOLDEN / "ref_checkpoint_tiny_sf.pt", device="cuda")
    assert agent.cfg.feature_learner == "icm" and agent.cfg.z_dim == 10 and agent.cfg.lr_coef == 5.0
    got = get_sf_state(agent)
    for k in exp.files:
        if k.startswith("state/"):
            np.testing.assert_array_equal(got[k[len("state/"):]], exp[k], err_msg=k)
    assert agent.step_counts() == (2, 2)
    acts = np.stack([agent.act(exp["obs"][i], {"z": exp["z"][i]}, 0, eval_mode=True) for i in range(5)])
    np.testing.assert_allclose(acts, exp["act_eval"], rtol=2e-5, atol=2e-6)
    rb = DeviceReplayBuffer.from_reference_file(H.GOLDEN / "ref_checkpoint_tiny_sf.pt", device="cuda")
    m = agent.update(rb, 2)
    assert np.isfinite(m["sf_loss"]) and np.isfinite(m["phi_loss"]) and agent.step_counts() == (3, 3)


def test_sf_fb_features_take_the_backward_net_of_a_trained_fb_agent():
    """feature_learner="FB" (sf.py:368-380, FBFeatures): feature_net is the backward_net of a trained FB agent -- here read from a
    checkpoint file the reference wrote -- and stays frozen (sf.py:447: no phi_opt); the SF update runs on top of it."""
    from controllable_agent_amd.agent import FBHipAgent, SFHipAgent
    from controllable_agent_amd.replay import DeviceReplayBuffer
    path = H.GOLDEN / "ref_checkpoint_tiny.pt"
    fb = FBHipAgent.from_reference_checkpoint(path, device="cuda")
    c = fb.cfg
    cfg = fo.OracleConfig(obs_dim=fb.obs_dim, action_dim=fb.action_dim, goal_dim=fb.goal_dim, z_dim=c.z_dim, hidden_dim=c.hidden_dim,
                          feature_dim=c.feature_dim, backward_hidden_dim=c.backward_hidden_dim, batch_size=16, lr=1e-3, lr_coef=5.0,
                          mix_ratio=0.0)
    with pytest.raises(ValueError, match="fb_features"):
        SFHipAgent(**sf_kwargs(cfg, "FB", True))
    agent = SFHipAgent(fb_features=path, **sf_kwargs(cfg, "FB", True))
    assert agent.phi_opt is None
    x = torch.randn(7, fb.goal_dim, generator=torch.Generator().manual_seed(5)).cuda()
    torch.testing.assert_close(agent.feature_learner.feature_net(x), fb.backward_net(x), rtol=0, atol=0)
    before = {k: v.clone() for k, v in agent.feature_learner.state_dict().items()}
    for k, v in fb.backward_net.state_dict().items():
        torch.testing.assert_close(before["feature_net." + k[2:]], v, rtol=0, atol=0)
    rng = np.random.default_rng(9)
    storage, lengths = fo.synthetic_storage(rng, 6, 12, cfg.obs_dim, cfg.action_dim)
    rb = DeviceReplayBuffer.from_arrays(storage, lengths, cfg.discount, device="cuda")
    for s in range(3):
        m = agent.update(rb, s)
    assert np.isfinite(m["sf_loss"]) and "phi_loss" not in m and agent.step_counts() == (3, 3)
    for k, v in agent.feature_learner.state_dict().items():
        torch.testing.assert_close(v, before[k], rtol=0, atol=0)
    # an agent object works as the source too
    again = SFHipAgent(fb_features=fb, **sf_kwargs(cfg, "FB", True))
    torch.testing.assert_close(again.feature_learner.feature_net(x), fb.backward_net(x), rtol=0, atol=0)


@pytest.mark.parametrize("name", ["tiny_sf_lap_trace", "tiny_sf_autoencoder_trace", "tiny_sf_transition_trace", "tiny_sf_svdp_goal_trace",
                                  "tiny_sf_random_trace", "tiny_sf_latent_trace", "tiny_sf_svdsr_goal_trace", "tiny_sf_contrastive_goal_trace",
                                  "tiny_sf_mix_icm_trace", "tiny_sf_mix_lap_trace", "tiny_sf_mix_identity_trace"])
def test_sf_pipelined_update_many_equals_single_updates(name):
    """fbhip_update_many cuts an SF update into head (sampling, online successor_net, feature_net [, mu_net]), middle and actor phase
    and runs the next step's head beside the actor phase: same kernels and operands, so the state after n pipelined steps equals n
    single updates bit for bit at these dimensions, for every feature learner's schedule."""
    meta, z, cfg, nets, storage, lengths = sf_trace_inputs(name)
    rb = _buffer(storage, lengths, cfg.discount, cfg.future)
    a1 = make_sf_agent(cfg, nets, meta["feature_learner"], meta["sf_q_loss"], meta["goal_space"], metrics=False)
    a2 = pickle.loads(pickle.dumps(a1))
    a2.defer_updates = False                               # single fbhip_update launches (not queued into an n-step graph)
    a1.update_many(rb, 0, 5)
    for s in range(5):
        a2.update(rb, s)
    s1, s2 = get_sf_state(a1), get_sf_state(a2)
    assert a1.step_counts() == a2.step_counts() == (5, 5)
    for k in s1:
        np.testing.assert_array_equal(s1[k], s2[k], err_msg=k)


@pytest.mark.parametrize("name", ["tiny_sf_mix_lap_trace", "tiny_sf_contrastive_goal_trace"])
def test_sf_update_many_injected_equals_injected_single_updates(name):
    """the parity entry point of the pipelined graph (fbhip_update_many_injected) with an SF agent: the recorded draws of four steps in
    one call against four update_injected calls -- bit for bit at these dimensions (mix_ratio > 0; the contrastive hindsight goal)."""
    meta, z, cfg, nets, storage, lengths = sf_trace_inputs(name)
    rb = _buffer(storage, lengths, cfg.discount, cfg.future)
    a1 = make_sf_agent(cfg, nets, meta["feature_learner"], meta["sf_q_loss"], meta["goal_space"])
    a2 = pickle.loads(pickle.dumps(a1))
    draws = [H.draws_dict(fo.Draws(**{f: z[f"draws/{s}/{f}"] for f in fo.Draws.__dataclass_fields__ if f"draws/{s}/{f}" in z.files}))
             for s in range(meta["n_steps"])]
    m1 = a1.update_many_injected(rb, 0, draws)
    for s, d in enumerate(draws):
        m2 = a2.update_injected(rb, s, d)
    for k in m2:
        assert m1[k] == pytest.approx(m2[k], rel=1e-6, abs=1e-7), k
    s1, s2 = get_sf_state(a1), get_sf_state(a2)
    for k in s1:
        np.testing.assert_array_equal(s1[k], s2[k], err_msg=k)


@pytest.mark.parametrize("name", ["tiny_sf_mix_icm_trace", "tiny_sf_contrastive_trace", "tiny_sf_contrastivev2_trace"])
def test_sf_external_batch_path_matches_device_path(name):
    """update_from_batch with an SF agent (a host-sampled EpisodeBatch, the reference ReplayBuffer contract; the contrastive learners
    also take batch.future_obs / future_goal, sf.py:713-719) == the fused sampler fed the same indices, bit for bit."""
    from controllable_agent_amd.replay import EpisodeBatch
    meta, z, cfg, nets, storage, lengths = sf_trace_inputs(name)
    rb = _buffer(storage, lengths, cfg.discount, cfg.future)
    a1, a2 = (make_sf_agent(cfg, nets, meta["feature_learner"], meta["sf_q_loss"], meta["goal_space"]) for _ in range(2))
    d = fo.Draws(**{f: z[f"draws/0/{f}"] for f in fo.Draws.__dataclass_fields__ if f"draws/0/{f}" in z.files})
    m1 = a1.update_injected(rb, 0, H.draws_dict(d))
    b = fo.gather_batch(storage, d.ep_idx, d.step_idx, cfg.discount, d.future_idx)
    batch = EpisodeBatch(**{k: b[k] for k in ("obs", "action", "reward", "next_obs", "discount", "goal", "next_goal", "future_obs", "future_goal") if k in b})
    m2 = a2.update_from_batch(batch, 0, draws=H.draws_dict(d))
    assert m1 == m2
    s1, s2 = get_sf_state(a1), get_sf_state(a2)
    for k in s1:
        np.testing.assert_array_equal(s1[k], s2[k], err_msg=k)


@pytest.mark.parametrize("learner,kw", [("lap", dict(mix_ratio=0.5)), ("contrastivev2", dict(future=0.8)), ("identity", dict(z_dim=5, mix_ratio=0.3))])
def test_sf_online_and_offline_loops(learner, kw):
    """run_online (pretrain.py:559-659 counterpart: the ring fills through add(), act on the batch-1 fast path, no compute_z_correl
    -- SFAgent has none and hasattr() must say so) and run_offline with an SF agent, end to end on tiny dims."""
    from controllable_agent_amd.replay import DeviceReplayBuffer, TimeStep
    from controllable_agent_amd.train_offline import run_offline
    from controllable_agent_amd.train_online import run_online
    cfg = fo.OracleConfig(**{**dict(obs_dim=5, action_dim=3, goal_dim=5, z_dim=8, hidden_dim=32, feature_dim=16, backward_hidden_dim=18,
                                    batch_size=16, lr=1e-3, lr_coef=5.0, mix_ratio=0.0), **kw})
    rng = np.random.default_rng(3)
    shapes = so.net_shapes(cfg, learner)
    agent = make_sf_agent(cfg, {n: fo.synthetic_params(rng, shapes[n]) for n in shapes}, learner, True)
    assert not hasattr(agent, "compute_z_correl")
    agent.cfg.update_every_steps = 2

    class Env:
        T, t = 6, 0

        def _ts(self, kind, action):
            return TimeStep(step_type=kind, reward=0.5, discount=1.0, observation=rng.standard_normal(5).astype(np.float32),
                            action=np.asarray(action, np.float32), physics=np.zeros(2, np.float32))

        def reset(self):
            self.t = 0
            return self._ts(0, np.zeros(3))

        def step(self, action):
            assert action.shape == (3,) and np.all(np.abs(action) <= 1.0)
            self.t += 1
            return self._ts(2 if self.t == self.T else 1, action)

    rb = DeviceReplayBuffer(max_episodes=4, discount=0.98, future=cfg.future, device="cuda")
    st = run_online(agent, rb, Env(), num_train_frames=40, num_seed_frames=18)
    assert (st.env_steps, st.episodes, st.updates) == (40, 6, 11) and agent.step_counts() == (11, 11)
    run_offline(agent, rb, num_grad_steps=37, log_every_steps=10, steps_per_launch=8)
    assert agent.step_counts() == (48, 48)
    assert all(np.isfinite(v).all() for v in get_sf_state(agent).values())


def test_sf_contrastive_needs_hindsight_goals():
    """the contrastive learner reads batch.future_goal (sf.py:125): a buffer with future = 1 has none -- loud error, no update"""
    meta, z, cfg, nets, storage, lengths = sf_trace_inputs("tiny_sf_contrastive_trace")
    agent = make_sf_agent(cfg, nets, "contrastive", True)
    with pytest.raises(RuntimeError, match="future < 1"):
        agent.update(_buffer(storage, lengths, cfg.discount, 1.0), 0)
    assert agent.step_counts() == (0, 0)
    agent.update(_buffer(storage, lengths, cfg.discount, 0.8), 0)        # device-drawn hindsight indices
    fut, step = (agent.workspace_view(n).cpu().numpy()[0] for n in ("future_idx", "step_idx"))
    assert np.all(fut >= step) and np.all(fut <= 12) and agent.step_counts() == (1, 1)


@pytest.mark.parametrize("name", ["tiny_sf_icm_trace", "tiny_sf_svdp_goal_trace", "tiny_sf_mix_icm_trace"])
def test_sf_phase_split_schedule_equals_single_call(name, monkeypatch):
    """The data-parallel cut of an SF update (gradients | sf_opt + phi_opt step + actor gradient | actor step, distributed.dp_update)
    on one rank against the single-graph update on the same draws: launches group differently, so fp32 tolerance."""
    meta, z, cfg, nets, storage, lengths = sf_trace_inputs(name)
    rb = _buffer(storage, lengths, cfg.discount, cfg.future)
    a1, a2 = (make_sf_agent(cfg, nets, meta["feature_learner"], meta["sf_q_loss"], meta["goal_space"]) for _ in range(2))
    for s in range(3):
        d = H.draws_dict(fo.Draws(**{f: z[f"draws/{s}/{f}"] for f in fo.Draws.__dataclass_fields__ if f"draws/{s}/{f}" in z.files}))
        monkeypatch.delenv("FBHIP_FORCE_PHASE_SPLIT", raising=False)
        m1 = a1.update_injected(rb, s, d)
        monkeypatch.setenv("FBHIP_FORCE_PHASE_SPLIT", "1")
        monkeypatch.setenv("FBHIP_DP_ALLREDUCE", "c10d")      # (the host-issued / torch-level schedule is what this test is about)
        m2 = a2.update_injected(rb, s, d)
        for k in m1:
            assert m2[k] == pytest.approx(m1[k], rel=2e-5, abs=1e-6), (s, k)
    monkeypatch.delenv("FBHIP_FORCE_PHASE_SPLIT", raising=False)
    s1, s2 = get_sf_state(a1), get_sf_state(a2)
    assert a1.step_counts() == a2.step_counts() == (3, 3)
    for k in s1:
        np.testing.assert_allclose(s2[k], s1[k], rtol=0, atol=3e-6, err_msg=k)


def test_sf_mix_draws_on_device():
    """mix_ratio > 0 with nothing injected: the permutation and the mix uniforms come from the device sampler, the rows below the
    ratio carry sqrt(d) normalize(phi(next_goal[perm]) @ inv_cov) recomputed here from the agent's own feature_net, the others the
    gaussian draw."""
    meta, z, cfg, nets, storage, lengths = sf_trace_inputs("tiny_sf_mix_lap_trace")
    agent = make_sf_agent(cfg, nets, "lap", False, meta["goal_space"])
    rb = _buffer(storage, lengths, cfg.discount, cfg.future)
    w0 = {k: v.detach().cpu().clone() for k, v in agent.feature_learner.state_dict().items()}    # weights BEFORE the update
    agent.update(rb, 0)
    v = lambda n: agent.workspace_view(n).cpu()
    perm, u, ep, st = v("perm")[0].long(), v("mix_uniform")[0], v("ep_idx")[0].long(), v("step_idx")[0].long()
    assert sorted(perm.tolist()) == list(range(cfg.batch_size)) and perm.tolist() != list(range(cfg.batch_size))
    mixed = u < cfg.mix_ratio
    assert 0 < int(mixed.sum()) < cfg.batch_size
    goal = torch.from_numpy(storage["goal"])
    next_goal = goal[ep, st]
    ref = so.feature_net({k: v_ for k, v_ in w0.items()}, next_goal[perm], cfg.z_dim)
    inv = torch.linalg.pinv(ref.T @ ref / ref.shape[0])
    want = np.sqrt(cfg.z_dim) * torch.nn.functional.normalize(ref @ inv, dim=1)
    gauss = fo.sample_z_from_gauss(v("z_gauss"), cfg.z_dim)
    zz = v("z")
    assert H.rel_err(zz[mixed], want[mixed]) < 2e-5 and H.rel_err(zz[~mixed], gauss[~mixed]) < 1e-6


def test_sf_identity_features_surface():
    """feature_learner="identity" (sf.py:94-98): no feature parameters are exposed, phi(goal) is the goal on the inference surface too,
    and z_dim != goal_dim is refused by the library"""
    from controllable_agent_amd.agent import SFHipAgent
    meta, z, cfg, nets, storage, lengths = sf_trace_inputs("tiny_sf_identity_trace")
    agent = make_sf_agent(cfg, nets, "identity", True)
    assert agent.phi_opt is None and len(agent.feature_learner.state_dict()) == 0 and list(agent.feature_learner.parameters()) == []
    goal = np.random.default_rng(2).standard_normal((9, cfg.goal_dim)).astype(np.float32)
    np.testing.assert_array_equal(agent.feature_learner.feature_net(goal).cpu().numpy(), goal)
    agent.inv_cov = agent._compute_cov(goal)
    want = np.sqrt(cfg.z_dim) * torch.nn.functional.normalize(torch.from_numpy(goal[:1]) @ agent.inv_cov.cpu(), dim=1)[0].numpy()
    np.testing.assert_allclose(agent.get_goal_meta(goal[0])["z"], want, rtol=1e-5, atol=1e-6)
    bad = fo.OracleConfig(**{**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__}, "z_dim": cfg.goal_dim + 1})
    with pytest.raises(ValueError, match="z_dim == goal_dim"):
        SFHipAgent(**sf_kwargs(bad, "identity", True))


def test_sf_num_sf_updates_runs_that_many_updates_per_call():
    """num_sf_updates = k (sf.py:706): one update() call = k complete updates on fresh batches; update_many(n) = n k of them, and the
    two agree bit for bit (same device RNG stream)"""
    from controllable_agent_amd.agent import SFHipAgent
    meta, z, cfg, nets, storage, lengths = sf_trace_inputs("tiny_sf_icm_trace")
    rb = _buffer(storage, lengths, cfg.discount)
    mk = lambda: SFHipAgent(**sf_kwargs(cfg, "icm", True, metrics=False, num_sf_updates=3))
    torch.manual_seed(7)
    a1 = mk()
    a1.load_nets({n: dict(p) for n, p in nets.items()})
    a2 = pickle.loads(pickle.dumps(a1))
    a1.defer_updates = False
    a1.update(rb, 0)
    assert a1.step_counts() == (3, 3)
    a1.update(rb, 1)
    a2.update_many(rb, 0, 2)
    assert a1.step_counts() == a2.step_counts() == (6, 6)
    s1, s2 = get_sf_state(a1), get_sf_state(a2)
    for k in s1:
        np.testing.assert_array_equal(s1[k], s2[k], err_msg=k)


def test_sf_deferred_update_calls_equal_eager_calls():
    """SFHipAgent.update() = num_sf_updates queued updates per call (agent.py "deferred batching"): the queue is launched as n-step
    graphs and equals eager single launches bit for bit; reading the state launches it."""
    from controllable_agent_amd.agent import SFHipAgent
    meta, z, cfg, nets, storage, lengths = sf_trace_inputs("tiny_sf_icm_trace")
    rb = _buffer(storage, lengths, cfg.discount)
    mk = lambda: SFHipAgent(**sf_kwargs(cfg, "icm", True, metrics=False, num_sf_updates=2))
    torch.manual_seed(7)
    a1 = mk()
    a1.load_nets({n: dict(p) for n, p in nets.items()})
    a2 = pickle.loads(pickle.dumps(a1))
    a2.defer_updates = False
    for s in range(9):
        assert a1.update(rb, s) == {} and a2.update(rb, s) == {}
    assert a1.__dict__["_pending"][3] == 17 and a2.__dict__.get("_pending") is None      # (the first call of the run went out at once)
    s1, s2 = get_sf_state(a1), get_sf_state(a2)
    assert a1.__dict__.get("_pending") is None and a1.step_counts() == a2.step_counts() == (18, 18)
    for k in s1:
        np.testing.assert_array_equal(s1[k], s2[k], err_msg=k)
