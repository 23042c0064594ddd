import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
np.set_printoptions(precision=4, linewidth=220, suppress=False)
from oracle import ref_mppi
from oracle.sim import OracleSim
from trajopt_amd.model import get_model
from trajopt_amd.device import DeviceModel, DevicePlanner
FILTER = [np.ones(7), 0.25, 0.8, 0.0]
sim = OracleSim(get_model("reacher_7dof"))
dm = DeviceModel.get("reacher_7dof")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 8
raw = ref_mppi.all_raw_normals(123, 40, 4, H, 7)
act = ref_mppi.filter_noise(raw.copy(), FILTER)
for dtype in ("f32", "f64"):
  for K in (1, 160):
    R_o, nobs_o, rew_o = sim.rollout(np.zeros(7), np.zeros(7), np.zeros(3), 0, act[:K], dump=True)
    p = DevicePlanner(dm, K, H, kappa=25.0, betas=FILTER[1:], dtype=dtype, dump=True)
    p.set_start_state(np.zeros(7), np.zeros(7), np.zeros(3), 0)
    p.set_base_actions(np.zeros((H, 7)))
    p.upload_noise(raw[:K])
    p.rollout()
    nobs = p.next_observations(); R = p.returns(); a = p.actions()
    err = np.abs(nobs - nobs_o)
    print(dtype, "K", K, "act err", np.abs(a - act[:K]).max(), "R err", np.abs(R - R_o).max())
    print("   err by column at t=0 (max over k):", err[:, 0].max(axis=0))
    print("   err by column at t=H-1:", err[:, -1].max(axis=0))
    print("   max err per t:", err.max(axis=(0, 2)))
