"""Environment factory with the reference's entry point (trajopt/envs/utils.py:8-19).

Only environments whose dynamics have a compiled rollout kernel are registered; the reference's
'hopper' and PyBullet humanoid need other simulators and resolve to "unknown", which -- as in the
reference -- prints a message and yields None instead of raising.
"""


def _reacher(reward_type, reference, renders):
    from .reacher_env import Reacher7DOFEnv
    return Reacher7DOFEnv(reward_type=reward_type, reference=reference)


def _continual_reacher(reward_type, reference, renders):
    from .continual_reacher_env import ContinualReacher7DOFEnv
    return ContinualReacher7DOFEnv()


REGISTRY = {
    'reacher_7dof': _reacher,
    'continual_reacher_7dof': _continual_reacher,
}


def get_environment(env_name, reward_type='dense', reference=None, renders=False):
    make = REGISTRY.get(env_name)
    if make is None:
        print("Unknown Environment Name")
        return None
    return make(reward_type, reference, renders)
