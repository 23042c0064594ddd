"""`ContinualReacher7DOFEnv` (reference: trajopt/envs/continual_reacher_env.py:5-23): the dense-reward
reacher whose target is redrawn after every 50th *real* step; rollout copies (`real_step` False)
keep their target."""
from .reacher_env import Reacher7DOFEnv

RETARGET_PERIOD = 50


class ContinualReacher7DOFEnv(Reacher7DOFEnv):
    def __init__(self):
        Reacher7DOFEnv.__init__(self)        # always the dense reward, no reference trajectory
        self.env_name = 'continual_reacher_7dof'
        self._ctor = {}

    def _step(self, a):
        # observation and reward are taken before a possible target redraw, the returned infos after it
        # (trajopt/envs/continual_reacher_env.py:15-23)
        ob, reward, done, _ = self._dense_step(a)
        due = self.env_timestep > 0 and self.env_timestep % RETARGET_PERIOD == 0
        if due and self.real_step is True:
            self.target_reset()
        return ob, reward, done, self.get_env_infos()
