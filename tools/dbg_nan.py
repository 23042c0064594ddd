import numpy as np, sys
sys.path.insert(0, '.')
from trajopt_amd.device import DeviceModel, DevicePlanner
dm = DeviceModel.get("reacher_7dof")
for dtype in ("f32", "f64"):
    p = DevicePlanner(dm, 128, 4, dtype=dtype, kappa=25.0, betas=(0.25, 0.8, 0.0))
    q = np.zeros(7); q[3] = np.nan
    p.set_start_state(q, np.zeros(7), np.zeros(3), 0)
    p.set_base_actions(np.zeros((4, 7)))
    p.generate_noise(1)
    p.rollout()
    print(dtype, p.status(), p.returns()[:6])
    p.set_start_state(np.zeros(7), np.full(7, 3e38), np.zeros(3), 0)
    p.rollout()
    print(dtype, p.status(), p.returns()[:6])
