"""The online loop (controllable_agent_amd.train_online.run_online) issues the reference's per-step call sequence
(pretrain.py:559-659) -- checked on CPU with a recording stub agent / buffer / environment."""
import types

import numpy as np

from controllable_agent_amd.replay import TimeStep
from controllable_agent_amd.train_online import run_online


class _Env:
    def __init__(self, episode_len):
        self.T, self.t = episode_len, 0

    def _ts(self, kind, action):
        return TimeStep(step_type=kind, reward=1.0, discount=1.0, observation=np.full(3, self.t, np.float32),
                        action=np.asarray(action, np.float32))

    def reset(self):
        self.t = 0
        return self._ts(0, np.zeros(2))

    def step(self, action):
        self.t += 1
        return self._ts(2 if self.t == self.T else 1, action)


class _Agent:
    def __init__(self, log):
        self.log, self.training = log, True
        self.cfg = types.SimpleNamespace(update_every_steps=2)
        self.n_meta = 0

    def train(self, training=True):
        self.training = training

    def init_meta(self):
        self.n_meta += 1
        self.log.append(("init_meta",))
        return {"z": np.full(4, self.n_meta, np.float32)}

    def update_meta(self, meta, step, time_step, finetune=False, replay_loader=None):
        self.log.append(("update_meta", step))
        assert finetune is False and replay_loader is not None
        return meta

    def act(self, obs, meta, step, eval_mode):
        assert eval_mode is False and self.training is False       # wrapped in utils.eval_mode (pretrain.py:629)
        self.log.append(("act", step, float(obs[0]), float(meta["z"][0])))
        return np.full(2, step, np.float32)

    def update(self, replay, step):
        self.log.append(("update", step))
        return {"fb_loss": 1.0} if step % self.cfg.update_every_steps == 0 else {}

    def compute_z_correl(self, time_step, meta):
        self.log.append(("z_correl", float(time_step.observation[0])))
        return 0.5


class _Replay:
    def __init__(self, log):
        self.log, self.n = log, 0

    def add(self, time_step, meta):
        self.n += 1
        self.log.append(("add", int(time_step.step_type), float(meta["z"][0])))

    def __len__(self):
        return self.n


def test_online_loop_call_sequence():
    log = []
    agent, rb, env = _Agent(log), _Replay(log), _Env(episode_len=3)
    seen = []
    st = run_online(agent, rb, env, num_train_frames=8, num_seed_frames=4, action_repeat=1,
                    log_fn=lambda step, m: seen.append((step, sorted(m))))
    assert (st.env_steps, st.episodes) == (8, 2) and agent.training is True
    assert st.updates == 2                                   # steps 4 and 6 (seed frames 0..3, update_every_steps = 2)
    # reset: init_meta + add(FIRST); then per step update_meta, act, [update], add(step), z_correl
    assert log[:3] == [("init_meta",), ("add", 0, 1.0), ("update_meta", 0)]
    per_step = [e for e in log if e[0] in ("update_meta", "act", "update", "add", "z_correl", "init_meta")]
    i = per_step.index(("update_meta", 4))
    assert [e[0] for e in per_step[i:i + 5]] == ["update_meta", "act", "update", "add", "z_correl"]
    assert [e[1] for e in log if e[0] == "update"] == [4, 5, 6, 7]        # the agent gates update_every_steps itself
    # episode boundary after 3 steps: LAST added with the OLD meta, then a new meta and a FIRST step
    j = log.index(("add", 2, 1.0))
    assert log[j + 1][0] == "z_correl" and log[j + 2] == ("init_meta",) and log[j + 3] == ("add", 0, 2.0)
    acts = [e for e in log if e[0] == "act"]
    assert [a[1] for a in acts] == list(range(8)) and acts[3][2] == 0.0 and acts[3][3] == 2.0   # step 3 = first of episode 2
    assert any(k == ["buffer_size", "episode", "episode_reward", "z_correl"] for _, k in seen)


def test_schedule_matches_utils_schedule_semantics():
    """utils.schedule (utils.py:235-255): constants, linear(init,final,T), step_linear(init,f1,T1,f2,T2)."""
    from controllable_agent_amd.agent import schedule
    import pytest
    assert schedule("0.2", 123) == 0.2 and schedule(0.3, 0) == 0.3
    lin = "linear(1.0,0.1,500)"
    assert schedule(lin, 0) == 1.0 and schedule(lin, 250) == pytest.approx(0.55) and schedule(lin, 10_000) == pytest.approx(0.1)
    sl = "step_linear(1.0,0.5,100,0.1,400)"
    assert schedule(sl, 50) == pytest.approx(0.75) and schedule(sl, 100) == pytest.approx(0.5)
    assert schedule(sl, 300) == pytest.approx(0.3) and schedule(sl, 9999) == pytest.approx(0.1)
    with pytest.raises(NotImplementedError):
        schedule("cosine(1,0,10)", 3)
