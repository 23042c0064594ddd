"""`ContinualReacher7DOFEnv` (trajopt/envs/continual_reacher_env.py:5-23): dense reward and a
new target every 50 real steps."""
from .reacher_env import Reacher7DOFEnv


class ContinualReacher7DOFEnv(Reacher7DOFEnv):
    def __init__(self):
        super().__init__()
        self.env_name = 'continual_reacher_7dof'
        self._ctor = {}

    def _step(self, a):
        import numpy as np
        reward = self._advance(np.asarray(a, dtype=np.float64), 'dense')
        ob = self._get_obs()
        self.env_timestep += 1
        if self.env_timestep % 50 == 0 and self.env_timestep > 0 and self.real_step is True:
            self.target_reset()
        return ob, reward, False, self.get_env_infos()
