"""configs[4] rehearsal: the online loop -- act every env step, add to a 2000-episode ring buffer,
update every 2nd step after the seed frames, compute_z_correl every step -- on quadruped dims with a SYNTHETIC environment
(MuJoCo is not in the image): obs ~ N(0,1), fixed-length episodes, optional busy-wait per step to emulate physics cost.

    python tools/online_bench.py [--frames 12000] [--env-us 0]
"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from controllable_agent_amd.agent import FBHipAgent
from controllable_agent_amd.replay import DeviceReplayBuffer, TimeStep
from controllable_agent_amd.train_online import run_online


class SyntheticEnv:
    def __init__(self, obs_dim, action_dim, episode_len, env_us, seed=0):
        self.o, self.a, self.T, self.env_us = obs_dim, action_dim, episode_len, env_us
        self.rng = np.random.default_rng(seed)
        self.bank = self.rng.standard_normal((4096, obs_dim)).astype(np.float32)
        self.t = 0

    def _ts(self, kind, action):
        return TimeStep(step_type=kind, reward=float(self.t % 7) * 0.1, discount=1.0,
                        observation=self.bank[self.rng.integers(4096)], action=np.asarray(action, np.float32),
                        physics=np.zeros(2, np.float32))

    def reset(self):
        self.t = 0
        return self._ts(0, np.zeros(self.a, np.float32))

    def step(self, action):
        if self.env_us > 0:
            t_end = time.perf_counter() + self.env_us * 1e-6
            while time.perf_counter() < t_end:
                pass
        self.t += 1
        return self._ts(2 if self.t == self.T else 1, action)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=12000)
    ap.add_argument("--seed-frames", type=int, default=4000)
    ap.add_argument("--env-us", type=float, default=0.0, help="busy-wait per env.step (emulated physics)")
    ap.add_argument("--episode-len", type=int, default=1000)
    a = ap.parse_args()
    agent = FBHipAgent(obs_type="states", obs_shape=(78,), action_shape=(12,), device="cuda", num_expl_steps=2000,
                       use_tb=False, use_wandb=False, use_hiplog=False, goal_space=None, z_dim=50, batch_size=1024,
                       update_every_steps=2)
    rb = DeviceReplayBuffer(max_episodes=2000, discount=0.99, future=0.99, device="cuda")
    env = SyntheticEnv(78, 12, a.episode_len, a.env_us)
    st = run_online(agent, rb, env, num_train_frames=a.frames, num_seed_frames=a.seed_frames)
    torch.cuda.synchronize()
    print(f"online loop, quadruped dims (o=78, a=12, d=50, B=1024), synthetic env ({a.env_us} us/step): "
          f"{st.env_steps} env steps, {st.updates} updates, {st.episodes} episodes in {st.seconds:.2f} s -> "
          f"{st.env_steps_per_s:.0f} env-steps/s, {st.updates_per_s:.0f} update-steps/s "
          f"(buffer {len(rb)} episodes)")


if __name__ == "__main__":
    main()
