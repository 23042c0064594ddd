"""`get_environment` (trajopt/envs/utils.py:8-19).  The simulators behind 'hopper' and the
PyBullet humanoid are not part of this package."""


def get_environment(env_name, reward_type='dense', reference=None, renders=False):
    if env_name == 'reacher_7dof':
        from .reacher_env import Reacher7DOFEnv
        return Reacher7DOFEnv(reward_type=reward_type, reference=reference)
    if env_name == 'continual_reacher_7dof':
        from .continual_reacher_env import ContinualReacher7DOFEnv
        return ContinualReacher7DOFEnv()
    print("Unknown Environment Name")
    return None
