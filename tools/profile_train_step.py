import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt.algos.mppi import MPPI
from trajopt.envs.utils import get_environment
e = get_environment('reacher_7dof'); e.reset_model(seed=3, goal=(0, 0, 0))
ag = MPPI(e, H=16, paths_per_cpu=40, num_cpu=1, kappa=25.0, gamma=1.0, mean=np.zeros(7), filter_coefs=[np.ones(7), 0.25, 0.8, 0.0],
          default_act='mean', seed=3, noise=sys.argv[1] if len(sys.argv) > 1 else 'host')
for _ in range(5): ag.train_step(niter=5)
pr = cProfile.Profile(); pr.enable()
for _ in range(100): ag.train_step(niter=5)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
