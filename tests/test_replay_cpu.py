"""Host logic of DeviceReplayBuffer on CPU tensors: the reference's bookkeeping tests
(url_benchmark/test_in_memory_replay_buffer.py:19-53) plus the sampler KAT recorded from the real ReplayBuffer."""
import pickle

import numpy as np
import pytest
import torch

from controllable_agent_amd.replay import DeviceReplayBuffer, EpisodeBatch, TimeStep
from tests import helpers as H


def _fill(rb, lengths, o=4, a=2, goal_dim=None, meta_z=None, storage=None):
    for e, L in enumerate(lengths):
        for s in range(L + 1):
            st = 0 if s == 0 else (2 if s == L else 1)
            kw = {}
            if storage is not None:
                kw = dict(reward=float(storage["reward"][e, s, 0]), discount=float(storage["discount"][e, s, 0]),
                          observation=storage["observation"][e, s], action=storage["action"][e, s])
                if "goal" in storage:
                    kw["goal"] = storage["goal"][e, s]
            else:
                kw = dict(reward=0.0, discount=1.0, observation=np.full(o, e, np.float32), action=np.zeros(a, np.float32))
            ts = TimeStep(step_type=st, physics=np.zeros(2, np.float32), **kw)
            rb.add(ts, {} if meta_z is None else {"z": meta_z[e, s]})


def test_fixed_episode_length_bookkeeping():
    rb = DeviceReplayBuffer(max_episodes=3, discount=1.0, future=1.0, device="cpu")
    _fill(rb, [10, 10])
    assert len(rb) == 2 and not rb._full and rb._is_fixed_episode_length and rb.avg_episode_length == 10
    _fill(rb, [10, 10])                      # wraps around
    assert len(rb) == 3 and rb._full and rb._idx == 1


def test_variable_episode_length_bookkeeping():
    rb = DeviceReplayBuffer(max_episodes=4, discount=1.0, future=1.0, max_episode_length=11, device="cpu")
    _fill(rb, [10, 4, 6])
    assert not rb._is_fixed_episode_length
    assert rb.avg_episode_length == round((10 + 4 + 6) / 3)
    np.testing.assert_array_equal(rb._episodes_length, [10, 4, 6, 0])
    b = rb.sample(64)
    assert b.obs.shape == (64, 4) and b.discount.shape == (64, 1)


def test_sample_matches_reference_kat(golden_dir):
    """Same numpy global-RNG seed => the real ReplayBuffer's indices and rows, bit for bit."""
    z = np.load(golden_dir / "sampler_kat.npz")
    storage = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith("storage/")}
    rb = DeviceReplayBuffer(max_episodes=6, discount=0.98, future=1.0, max_episode_length=10, device="cpu")
    _fill(rb, z["lengths"], storage=storage, meta_z=z["meta_z"])
    assert len(rb) == int(z["rb_len"]) and rb._full == bool(z["rb_full"]) and rb._idx == int(z["rb_idx"])
    assert rb.avg_episode_length == int(z["avg_episode_length"])
    np.testing.assert_array_equal(rb._episodes_length, z["rb_episodes_length"])
    np.random.seed(123)
    b = rb.sample(64)
    for k in ("obs", "action", "next_obs", "reward", "discount", "goal", "next_goal"):
        np.testing.assert_array_equal(getattr(b, k).numpy(), z[k], err_msg=k)
    np.testing.assert_array_equal(b.meta["z"].numpy(), z["meta_z_out"])
    assert b.future_obs is None


def test_pickle_round_trip_and_from_arrays():
    rng = np.random.default_rng(0)
    from oracle import fb_oracle as fo
    storage, lengths = fo.synthetic_storage(rng, 5, 8, 4, 2, goal_dim=3)
    rb = DeviceReplayBuffer.from_arrays(storage, lengths, 0.99, device="cpu")
    rb2 = pickle.loads(pickle.dumps(rb))
    assert len(rb2) == 5 and rb2._discount == 0.99
    for k in storage:
        np.testing.assert_array_equal(rb2._storage[k].numpy(), storage[k])
    v = rb.device_view()
    assert v["n_episodes"] == 5 and v["t1"] == 9 and v["fixed_length"]
    np.testing.assert_array_equal(v["cum_len"].numpy(), np.arange(6) * 8)
    sh = rb.shard(1, 2)
    assert len(sh) == 2
    np.testing.assert_array_equal(sh._storage["observation"].numpy(), storage["observation"][[1, 3]])


def test_episode_batch_surface():
    b = EpisodeBatch(obs=np.zeros((3, 2), np.float32), action=np.zeros((3, 1), np.float32),
                     reward=np.ones((3, 1), np.float32), next_obs=np.zeros((3, 2), np.float32),
                     discount=np.ones((3, 1), np.float32))
    t = b.to("cpu")
    assert isinstance(t.obs, torch.Tensor) and t.goal is None
    assert float(t.with_no_reward().reward.sum()) == 0.0
    assert len(t.unpack()) == 5


def test_agent_rejects_unsupported_flags_loudly():
    from controllable_agent_amd.agent import FBHipAgent
    base = dict(obs_type="states", obs_shape=(4,), action_shape=(2,), num_expl_steps=0)
    from controllable_agent_amd.agent import DiscreteFBHipAgent
    # (debug=True: the IdentityMap backward nets of fb_ddpg.py:128-130 need z_dim == goal dimension)
    for flag in (dict(obs_type="pixels"),):
        with pytest.raises(NotImplementedError):
            FBHipAgent(**{**base, **flag})
    with pytest.raises(ValueError, match="must equal the goal dimension"):
        DiscreteFBHipAgent(**{**base, "debug": True, "z_dim": 8, "preprocess": False})
    with pytest.raises(ValueError, match="must equal the goal dimension"):
        FBHipAgent(**{**base, "debug": True, "z_dim": 8})
    with pytest.raises(ValueError):
        FBHipAgent(obs_type="states", obs_shape=(4,), action_shape=(2,))            # num_expl_steps missing
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            FBHipAgent(**base)


# ------------------------------------------------------------------------------------------------ reference files (n2)
def test_reference_checkpoint_reads_without_the_reference_installed():
    """``latest.pt`` as the reference's Workspace.save_checkpoint writes it (pretrain.py:437-449; made by the real
    reference in tests/golden/make_golden.py::checkpoint_fixture) unpickles through reference_io's placeholder classes:
    every tensor of the five nets, the Adam moments and the replay storage come back bit-exactly."""
    import sys
    from controllable_agent_amd import reference_io as rio
    assert not any(m == "url_benchmark" or m.startswith("url_benchmark.") for m in sys.modules)
    import json
    GOLDEN = H.GOLDEN
    payload = rio.load_reference_payload(GOLDEN / "ref_checkpoint_tiny.pt")
    exp = np.load(GOLDEN / "ref_checkpoint_expect.npz")
    meta = json.loads((GOLDEN / "ref_checkpoint_expect.json").read_text())
    assert payload["global_step"] == meta["global_step"] and payload["global_episode"] == meta["global_episode"]
    agent = payload["agent"]
    assert isinstance(agent, rio.ReferenceObject) and agent._ref_name == "FBDDPGAgent"
    fields = rio.reference_agent_config(agent)
    for k, v in meta["agent_cfg"].items():
        got = fields[k]
        assert (list(got) if isinstance(got, tuple) else got) == v, k
    for net in ("actor", "forward_net", "backward_net", "forward_target_net", "backward_target_net"):
        sd = getattr(agent, net).state_dict()
        want = [k.split("/", 2)[2] for k in exp.files if k.startswith(f"state/{net}/")]
        assert list(sd) == want                                  # nn.Module.state_dict() order and names
        for k, v in sd.items():
            np.testing.assert_array_equal(v.numpy(), exp[f"state/{net}/{k}"], err_msg=f"{net}/{k}")
    # torch.optim.Adam unpickles as itself; its state follows parameters() order (fb_ddpg.py:146-151)
    osd = agent.fb_opt.state_dict()
    names = ([f"forward_net/{k}" for k in agent.forward_net.state_dict()] +
             [f"backward_net/{k}" for k in agent.backward_net.state_dict()])
    assert len(osd["state"]) == len(names)
    for i, n in enumerate(names):
        np.testing.assert_array_equal(osd["state"][i]["exp_avg"].numpy(), exp[f"state/adam_m/{n}"], err_msg=n)
        assert int(osd["state"][i]["step"]) == int(exp["fb_steps"])
    # the buffer: placeholder -> DeviceReplayBuffer (CPU device here), storage and lengths bit-exact, sampling works
    for rb in (DeviceReplayBuffer.from_reference(payload["replay_loader"], device="cpu"),
               DeviceReplayBuffer.from_reference_file(GOLDEN / "ref_replay_tiny.pt", device="cpu", discount=0.9, future=1.0),
               DeviceReplayBuffer.from_reference_file(GOLDEN / "ref_checkpoint_tiny.pt", device="cpu")):
        for k in ("observation", "action", "discount", "reward"):
            np.testing.assert_array_equal(rb._storage[k].numpy(), exp[f"storage/{k}"], err_msg=k)
        np.testing.assert_array_equal(rb._episodes_length, exp["lengths"])
        assert len(rb) == exp["storage/observation"].shape[0] and rb._full
        b = rb.sample(8)
        assert tuple(b.obs.shape) == (8, exp["storage/observation"].shape[2])
    assert rb._discount == pytest.approx(0.99)
    assert DeviceReplayBuffer.from_reference_file(GOLDEN / "ref_replay_tiny.pt", device="cpu", discount=0.9)._discount == 0.9


def test_reference_discrete_checkpoint_reads_without_the_reference():
    """The sibling agent's checkpoint (a pickled ``DiscreteFBAgent``, discrete_fb.py) through the same reader: config fields,
    the four nets in state_dict() order, the FB Adam moments."""
    from controllable_agent_amd import reference_io as rio
    payload = rio.load_reference_payload(H.GOLDEN / "ref_checkpoint_tiny_discrete.pt")
    exp = np.load(H.GOLDEN / "ref_checkpoint_discrete_expect.npz")
    agent = payload["agent"]
    assert isinstance(agent, rio.ReferenceObject) and agent._ref_name == "DiscreteFBAgent" and not hasattr(agent, "actor")
    fields = rio.reference_agent_config(agent)
    assert fields["name"] == "discrete_fb" and fields["preprocess"] is False and fields["action_shape"] == (4,) and fields["expl_eps"] == 0.2
    for net in ("forward_net", "backward_net", "forward_target_net", "backward_target_net"):
        sd = getattr(agent, net).state_dict()
        assert list(sd) == [k.split("/", 2)[2] for k in exp.files if k.startswith(f"state/{net}/")]
        for k, v in sd.items():
            np.testing.assert_array_equal(v.numpy(), exp[f"state/{net}/{k}"], err_msg=f"{net}/{k}")
    assert sd["F2.2.weight"].shape == (8 * 4, 32) if net.startswith("forward") else True
    osd = agent.fb_opt.state_dict()
    names = [f"forward_net/{k}" for k in agent.forward_net.state_dict()] + [f"backward_net/{k}" for k in agent.backward_net.state_dict()]
    for i, n in enumerate(names):
        np.testing.assert_array_equal(osd["state"][i]["exp_avg"].numpy(), exp[f"state/adam_m/{n}"], err_msg=n)


def test_hydra_written_checkpoint_config_unwraps_omegaconf_containers():
    """pretrain.py:112-120 builds the agent with hydra.utils.instantiate, so a real checkpoint's ``agent.cfg`` holds
    ``omegaconf.ListConfig`` objects for obs_shape / action_shape / log_std_bounds.  Without omegaconf they unpickle as
    placeholders; ``reference_agent_config`` must hand the constructor plain tuples (tests/golden/make_golden.py::
    hydra_checkpoint_fixture)."""
    from controllable_agent_amd import reference_io as rio
    payload = rio.load_reference_payload(H.GOLDEN / "ref_checkpoint_hydra_tiny.pt")
    agent = payload["agent"]
    raw = agent.cfg.obs_shape
    assert isinstance(raw, rio.ReferenceObject) and raw._ref_name == "ListConfig" and not hasattr(raw, "__len__")
    fields = rio.reference_agent_config(agent)
    assert fields["obs_shape"] == (5,) and fields["action_shape"] == (3,) and fields["log_std_bounds"] == (-5, 2)
    assert len(fields["action_shape"]) == 1 and int(fields["obs_shape"][0]) == 5
    exp = np.load(H.GOLDEN / "ref_checkpoint_hydra_expect.npz")
    for k, v in agent.forward_net.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), exp[f"state/forward_net/{k}"])


def test_reader_refuses_globals_outside_the_allow_list_and_never_imports(tmp_path, monkeypatch):
    """a pickle naming os.system (or any module outside the allow-list) is refused; a module on sys.path that shadows one of
    the reference's generic top-level names (``utils``) is NOT imported -- its classes become placeholders regardless"""
    import pickle, sys
    from controllable_agent_amd import reference_io as rio
    evil = tmp_path / "evil.pt"
    evil.write_bytes(pickle.dumps(__import__("os").getcwd, protocol=4))         # GLOBAL posix.getcwd / os.getcwd
    with pytest.raises(Exception, match="allow-list|refusing|Unsupported|Invalid"):
        rio.load_reference_payload(evil)
    (tmp_path / "utils.py").write_text("raise SystemExit('imported by the unpickler')\nclass Until: pass\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    sys.modules.pop("utils", None)
    cls = rio._Unpickler(__import__("io").BytesIO(b"")).find_class("utils", "Until")
    assert issubclass(cls, rio.ReferenceObject) and "utils" not in sys.modules


def test_nested_storage_payloads_stay_inside_the_allow_list(tmp_path):
    """``torch.storage._load_from_bytes`` (what ``pickle.dumps(tensor)`` reduces a storage through) used to be on the
    allow-list and runs a default-unpickler ``torch.load``: a refused global executed when wrapped in it.  It is now replaced by a
    stand-in that re-enters the allow-listed unpickler: the harmless nested payload is refused, a plain pickled tensor loads."""
    import pickle
    from controllable_agent_amd import reference_io as rio

    class Nested:
        def __reduce__(self):
            import torch.storage
            inner = pickle.dumps(__import__("os").getpid, protocol=2)           # GLOBAL posix.getpid inside the inner stream
            return (torch.storage._load_from_bytes, (inner,))

    evil = tmp_path / "nested.pt"
    evil.write_bytes(pickle.dumps(Nested(), protocol=4))
    with pytest.raises(Exception, match="allow-list|refusing|Unsupported|Invalid|pickle"):
        rio.load_reference_payload(evil)
    t = torch.arange(6, dtype=torch.float32).reshape(2, 3)
    ok = tmp_path / "tensor.pkl"
    ok.write_bytes(pickle.dumps({"x": t}, protocol=4))                          # storages travel through _load_from_bytes
    with ok.open("rb") as f:
        back = rio._Unpickler(f).load()
    assert torch.equal(back["x"], t)


# ------------------------------------------------------------------------------------------------ staging / load / relabel
def test_episode_stage_supports_what_callers_do_with_current_episode():
    """pretrain.py:485 calls ``_current_episode.clear()``; url_benchmark/test_dmc.py:25 asserts ``"physics" in
    rb._current_episode``; episodes longer than the stage's first capacity grow it without losing rows"""
    rb = DeviceReplayBuffer(max_episodes=3, discount=1.0, future=1.0, device="cpu")
    ts = lambda st, v: TimeStep(step_type=st, reward=v, discount=1.0, observation=np.full(4, v, np.float32),
                                action=np.zeros(2, np.float32), physics=np.full(3, -v, np.float32))
    rb.add(ts(0, 0.0), {"z": np.arange(5, dtype=np.float32)})
    assert "physics" in rb._current_episode and "z" in rb._current_episode and "goal" not in rb._current_episode
    assert len(rb._current_episode) == 1 and len(rb) == 0
    rb._current_episode.clear()                                    # load_checkpoint drops the pending episode
    assert "physics" not in rb._current_episode and len(rb._current_episode) == 0
    L = 150                                                        # > the stage's initial 64-row capacity
    for s in range(L + 1):
        rb.add(ts(0 if s == 0 else (2 if s == L else 1), float(s)), {"z": np.full(5, s, np.float32)})
    assert len(rb) == 1 and rb._episodes_length[0] == L and len(rb._current_episode) == 0
    np.testing.assert_array_equal(rb._storage["observation"][0, :, 0].numpy(), np.arange(L + 1, dtype=np.float32))
    np.testing.assert_array_equal(rb._storage["z"][0, :, 4].numpy(), np.arange(L + 1, dtype=np.float32))
    np.testing.assert_array_equal(rb._storage["physics"][0, 7].numpy(), np.full(3, -7, np.float32))
    assert rb._storage["reward"].shape == (3, L + 1, 1)
    # a second, shorter episode reuses the stage: rows beyond its length stay zero in the storage
    for s in range(11):
        rb.add(ts(0 if s == 0 else (2 if s == 10 else 1), 100.0 + s), {"z": np.zeros(5, np.float32)})
    assert rb._episodes_length[1] == 10 and not rb._is_fixed_episode_length
    assert float(rb._storage["observation"][1, 10, 0]) == 110.0 and float(rb._storage["observation"][1, 11:].abs().sum()) == 0.0


class _RewardFromPhysics:
    def from_physics(self, p):
        return float(p[0] * 2 + 1)


def test_load_npz_directory_and_relabel(tmp_path):
    """ExORL layout (in_memory_replay_buffer.py:33-37, 192-216): one ``*.npz`` per episode, keys observation / action /
    reward / discount / physics, each [T+1, dim]; ``load`` fills the ring in sorted order and stops when it is full;
    ``relabel`` / ``sample(custom_reward=...)`` recompute rewards from the stored physics."""
    rng = np.random.default_rng(3)
    T1, eps = 9, []
    for e in range(4):
        ep = dict(observation=rng.standard_normal((T1, 4)).astype(np.float32), action=rng.standard_normal((T1, 2)).astype(np.float32),
                  reward=rng.standard_normal((T1, 1)).astype(np.float32), discount=np.ones((T1, 1), np.float32),
                  physics=rng.standard_normal((T1, 3)).astype(np.float32))
        np.savez(tmp_path / f"episode_{e:03d}.npz", **ep)
        eps.append(ep)
    rb = DeviceReplayBuffer(max_episodes=3, discount=0.9, future=1.0, device="cpu")
    with pytest.raises(ValueError, match="relabel=True needs"):
        rb.load(None, tmp_path)
    rb.load(None, tmp_path, relabel=False)
    assert rb._full and len(rb) == 3 and rb._idx == 0 and rb._is_fixed_episode_length       # the 4th file did not fit
    for e in range(3):
        for k in eps[e]:
            np.testing.assert_array_equal(rb._storage[k][e].numpy(), eps[e][k], err_msg=f"{k}[{e}]")
    np.testing.assert_array_equal(rb._episodes_length, [T1 - 1] * 3)
    np.random.seed(5)
    b = rb.sample(32, custom_reward=_RewardFromPhysics(), with_physics=True)
    np.testing.assert_allclose(b.reward.numpy()[:, 0], b._physics.numpy()[:, 0] * 2 + 1, rtol=0, atol=0)
    assert b.discount.numpy().max() == pytest.approx(0.9)
    rb.relabel(_RewardFromPhysics())
    want = np.stack([eps[e]["physics"][:, :1] * 2 + 1 for e in range(3)]).astype(np.float32)
    np.testing.assert_array_equal(rb._storage["reward"].numpy(), want)
    assert rb._max_episodes == 3 and rb._full


def test_reference_sf_checkpoint_reads_without_the_reference():
    """the second sibling's checkpoint (a pickled ``sf.SFAgent``) through the same reader"""
    from controllable_agent_amd import reference_io as rio
    payload = rio.load_reference_payload(H.GOLDEN / "ref_checkpoint_tiny_sf.pt")
    exp = np.load(H.GOLDEN / "ref_checkpoint_sf_expect.npz")
    agent = payload["agent"]
    assert isinstance(agent, rio.ReferenceObject) and agent._ref_name == "SFAgent"
    fields = rio.reference_agent_config(agent)
    assert fields["name"] == "sf" and fields["feature_learner"] == "icm" and fields["q_loss"] is True and fields["obs_shape"] == (5,)
    for net in ("actor", "successor_net", "successor_target_net", "feature_learner"):
        sd = getattr(agent, net).state_dict()
        assert list(sd) == [k.split("/", 2)[2] for k in exp.files if k.startswith(f"state/{net}/")]
        for k, v in sd.items():
            np.testing.assert_array_equal(v.numpy(), exp[f"state/{net}/{k}"], err_msg=f"{net}/{k}")
    assert "inverse_dynamic_net.4.bias" in agent.feature_learner.state_dict()
    assert len(agent.phi_opt.state_dict()["state"]) == 14 and len(agent.sf_opt.state_dict()["state"]) == 20


def _episode(steps, seed, o=4, a=2):
    rng = np.random.default_rng(seed)
    return [TimeStep(step_type=0 if t == 0 else (2 if t == steps else 1), reward=0.0, discount=1.0, physics=np.zeros(2, np.float32),
                     observation=rng.standard_normal(o).astype(np.float32), action=np.zeros(a, np.float32)) for t in range(steps + 1)]


def test_observers_are_asked_to_flush_before_every_mutation(tmp_path):
    """FBHipAgent queues metrics-off update() calls and launches them as n-step graphs (agent.py "deferred batching"); the batches of
    queued updates are drawn from the buffer when they RUN, so the buffer asks every agent with calls in its queue to launch them
    BEFORE it writes -- a finished episode (add), load, relabel, adopting new arrays, unpickling over it -- and never on the steps
    that only stage a transition.  Observers are weak references and do not travel with a pickle."""
    import gc

    class Observer:
        def __init__(self, rb):
            self.rb, self.seen = rb, []

        def flush(self):
            # what a queued update would sample from at this moment
            self.seen.append((len(self.rb), int(self.rb._version),
                              None if "observation" not in self.rb._storage else float(self.rb._storage["observation"].sum())))

    rb = DeviceReplayBuffer(max_episodes=3, discount=0.98, future=1.0, device="cpu")
    ob = Observer(rb)
    rb._observe(ob)
    for ep in range(2):
        for ts in _episode(4, seed=ep):
            before = len(ob.seen)
            rb.add(ts, {})
            if not ts.last():
                assert len(ob.seen) == before                      # staging a transition touches nothing a sampler can see
    # each finished episode: asked once before the block is written (and once more by the bookkeeping that follows it)
    assert [s[0] for s in ob.seen][:1] == [0] and ob.seen[0][2] is None   # first flush saw the EMPTY buffer
    assert any(s[0] == 1 for s in ob.seen)                                  # the second episode's flush saw exactly one stored episode
    first_two = float(rb._storage["observation"].sum())
    n = len(ob.seen)
    rb._adopt({k: v.numpy() for k, v in rb._storage.items()}, None)         # new arrays: asked first
    assert len(ob.seen) == n + 1 and ob.seen[-1][2] == pytest.approx(first_two)
    rb2 = pickle.loads(pickle.dumps(rb))
    assert "_observers" not in rb2.__dict__ or len(rb2.__dict__["_observers"]) == 0
    rb._unobserve(ob)
    n = len(ob.seen)
    for ts in _episode(4, seed=9):
        rb.add(ts, {})
    assert len(ob.seen) == n                                                 # no longer an observer
    rb._observe(ob)
    del ob
    gc.collect()
    for ts in _episode(4, seed=10):                                          # a dead observer is dropped, not called
        rb.add(ts, {})
    assert len(rb) == 3
