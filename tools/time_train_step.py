"""Wall time of the drop-in `MPPI.train_step` (Python API) for a few configurations."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt.algos.mppi import MPPI
from trajopt.envs.utils import get_environment
FILTER = [np.ones(7), 0.25, 0.8, 0.0]
for (H, ppc, ncpu, niter, noise) in [(16, 40, 1, 5, "host"), (16, 40, 1, 5, "philox"), (64, 4096, 1, 1, "philox"),
                                      (64, 4096, 1, 1, "host"), (64, 512, 8, 1, "host")]:
    e = get_environment('reacher_7dof'); e.reset_model(seed=3, goal=(0, 0, 0))
    ag = MPPI(e, H=H, paths_per_cpu=ppc, num_cpu=ncpu, kappa=25.0, gamma=1.0, mean=np.zeros(7), filter_coefs=FILTER,
              default_act='mean', seed=3, noise=noise)
    for _ in range(3):
        ag.train_step(niter=niter)
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        ag.train_step(niter=niter)
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"H": H, "K": ppc * ncpu, "niter": niter, "noise": noise, "train_step_ms": dt * 1e3,
                      "env_steps_per_s": ppc * ncpu * H * niter / dt}), flush=True)
