"""A tiny deterministic env with the reference's env duck-type, used to exercise the
planner's host-side bookkeeping (``advance_time``, logs) without a simulator."""
import numpy as np


class ToyEnv:
    def __init__(self, n=4, m=3):
        self.env_name = "toy"
        self.observation_dim, self.action_dim = n, m
        self.alpha = 1.0
        self.real_step = True
        self.env_timestep = 0
        self.x = np.linspace(-1.0, 1.0, n)
        self.goal_dist = np.array([0.0])
        self.n_set = 0

    def get_env_state(self):
        return dict(x=self.x.copy(), timestep=self.env_timestep)

    def set_env_state(self, s):
        self.x = s["x"].copy()
        self.env_timestep = s["timestep"]
        self.n_set += 1

    def _get_obs(self, goal_dist=False):
        if goal_dist:
            return np.concatenate([self.x, self.goal_dist])
        return self.x.copy()

    def get_env_infos(self):
        return dict(state=self.get_env_state())

    def reset_model(self, seed=None, goal=None, alpha=None):
        self.x = np.linspace(-1.0, 1.0, self.observation_dim)
        self.env_timestep = 0
        return self._get_obs()

    def step(self, a):
        a = np.asarray(a, dtype=np.float64)
        k = min(len(a), len(self.x))
        self.x = 0.9 * self.x
        self.x[:k] += 0.1 * np.tanh(a[:k])
        self.env_timestep += 1
        r = -float(np.sum(np.abs(self.x)))
        self.goal_dist = np.array([abs(r)])
        return self._get_obs(), r, False, self.get_env_infos()
