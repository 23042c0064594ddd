"""Where a drop-in MPPI.train_step spends its host time (cProfile): python tools/profile_train_step.py [host|philox] [H K niter]"""
import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt.algos.mppi import MPPI
from trajopt.envs.utils import get_environment
noise = sys.argv[1] if len(sys.argv) > 1 else 'host'
H, K, niter = [int(x) for x in sys.argv[2:5]] if len(sys.argv) > 4 else (16, 40, 5)
e = get_environment('reacher_7dof'); e.reset_model(seed=3, goal=(0, 0, 0))
ag = MPPI(e, H=H, paths_per_cpu=K, num_cpu=1, kappa=25.0, gamma=1.0, mean=np.zeros(7), filter_coefs=[np.ones(7), 0.25, 0.8, 0.0],
          default_act='mean', seed=3, noise=noise)
for _ in range(5): ag.train_step(niter=niter)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): ag.train_step(niter=niter)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
