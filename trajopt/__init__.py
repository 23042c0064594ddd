"""Import-name compatibility: `trajopt.*` resolves to the modules of `trajopt_amd`, so code
written against the reference package (`from trajopt.algos.mppi import MPPI`,
`from trajopt.envs.utils import get_environment`, pickles that name `trajopt.*` classes)
runs on this implementation unchanged."""
import importlib
import sys

import trajopt_amd

_NAMES = ["tensor_utils", "envs", "envs.reacher_env", "envs.continual_reacher_env", "envs.utils", "utils",
          "algos", "algos.trajopt_base", "algos.mppi", "models", "models.critic_nets"]

for _n in _NAMES:
    _m = importlib.import_module("trajopt_amd." + _n)
    sys.modules["trajopt." + _n] = _m
    _parent, _, _leaf = ("trajopt." + _n).rpartition(".")
    setattr(sys.modules[_parent], _leaf, _m)
