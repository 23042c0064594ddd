"""A long closed loop through the drop-in class: MPPI on the continual reacher (the target moves every 50 real steps),
thousands of planning steps without a break; every state must stay finite, the status word clean, and the hand must
keep arriving.      python tools/soak.py [steps] [K] [noise]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt.algos.mppi import MPPI
from trajopt.envs.utils import get_environment
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
noise = sys.argv[3] if len(sys.argv) > 3 else "philox"
e = get_environment('continual_reacher_7dof')
e.reset_model(seed=1)
ag = MPPI(e, H=32, paths_per_cpu=K, num_cpu=1, kappa=25.0, gamma=1.0, mean=np.zeros(7), filter_coefs=[np.ones(7), 0.25, 0.8, 0.0],
          default_act='mean', seed=7, noise=noise)
t0 = time.time()
for _ in range(steps):
    ag.train_step(niter=1)
dt = time.time() - t0
obs = np.array(ag.sol_obs)
dist = np.linalg.norm(obs[:, 14:17] - obs[:, 17:20], axis=1)       # hand to target
tp = np.array([s["target_pos"] for s in ag.sol_state])
moves = np.flatnonzero(np.any(tp[1:] != tp[:-1], axis=1)) + 1          # first state with the new target
ends = [dist[m - 1] for m in moves]                                    # distance just before every move
print(json.dumps({"steps": steps, "K": K, "noise": noise, "ms_per_train_step": 1e3 * dt / steps,
                  "all_finite": bool(np.isfinite(obs).all() and np.isfinite(ag.sol_reward).all()),
                  "status": ag._planner.status(), "target_moves": len(moves),
                  "median_distance_before_a_move": float(np.median(ends)), "worst_distance_before_a_move": float(np.max(ends)),
                  "within_5cm_before_a_move": float(np.mean(np.array(ends) < 0.05)),
                  "max_abs_qvel": float(np.abs(obs[:, 7:14]).max())}))
