"""Closed-loop totals of the example configuration over noise seeds (how much one run of 100 planning steps with
40 samples moves with rounding): python tools/planning_spread.py"""
import sys, numpy as np
sys.path.insert(0, '.')
from trajopt.algos.mppi import MPPI
from trajopt.envs.utils import get_environment
FILTER = [np.ones(7), 0.25, 0.8, 0.0]
for seed in range(1, 9):
    e = get_environment('reacher_7dof')
    e.reset_model(seed=3, goal=(0, 0, 0), alpha=1.0)
    ag = MPPI(e, H=16, paths_per_cpu=40, num_cpu=1, kappa=25.0, gamma=1.0, mean=np.zeros(7), filter_coefs=FILTER, default_act='mean', seed=seed)
    for _ in range(100):
        ag.train_step(niter=5, goal=(0, 0, 0))
    print(seed, round(float(np.sum(ag.sol_reward)), 1), round(float(ag.sol_obs[-1][-1]), 4), flush=True)
