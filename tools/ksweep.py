"""Rollout-kernel time versus K (samples per GPU): where the lane-per-sample kernel saturates.
MPPI_KSWEEP_MODEL selects the compiled model (default reacher_7dof; reacher_7dof_free is the same arm without limit
and contact rows; the FLOP-based columns are those of reacher_7dof's operation count and mean nothing for other models)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt_amd.device import DeviceModel, DevicePlanner
MODEL = os.environ.get("MPPI_KSWEEP_MODEL", "reacher_7dof")
dm = DeviceModel.get(MODEL)
H = 64
FLOP = float(os.environ.get("FLOP_PER_ENV_STEP", "8870"))
for K in [int(x) for x in (sys.argv[1:] or ["4096", "16384", "65536", "131072", "262144", "524288"])]:
    p = DevicePlanner(dm, K, H, kappa=25.0, sigma=np.ones(7), betas=(0.25, 0.8, 0.0), default_act="mean")
    p.set_start_state(np.zeros(7), np.zeros(7), np.zeros(3), 0)
    p.set_base_actions(np.zeros((H, 7)))
    p.generate_noise(1)
    for _ in range(2):
        p.rollout()
    p.synchronize()
    p.profile(True); p.profile_read()
    for _ in range(5):
        p.rollout()
    ms, n = p.profile_read()
    ms /= n
    print(json.dumps({"model": MODEL, "K": K, "H": H, "rollout_ms": ms, "env_steps_per_s": K * H / ms * 1e3,
                      "tflops": FLOP * K * H / ms / 1e9, "frac_fp32_valu": FLOP * K * H / ms / 1e9 / 157.3}), flush=True)
    p.close()
