"""fp32 versus fp64 rollouts on the device: relative state / return error at t = 32, 64, 128."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt_amd.device import DeviceModel, DevicePlanner
FILTER = [np.ones(7), 0.25, 0.8, 0.0]
for model in ("reacher_7dof", "reacher_7dof_free"):
    dm = DeviceModel.get(model)
    K, H = 8192, 128
    out = {}
    for dtype in ("f64", "f32"):
        p = DevicePlanner(dm, K, H, sigma=FILTER[0], betas=FILTER[1:], dtype=dtype, dump=True)
        p.set_start_state(np.zeros(7), np.zeros(7), np.zeros(3), 0)
        p.set_base_actions(np.zeros((H, 7)))
        p.generate_noise(7)
        p.rollout()
        out[dtype] = (p.next_observations(), p.rewards())
        p.close()
    o64, r64 = out["f64"]; o32, r32 = out["f32"]
    rel = np.abs(o32[:, :, :14] - o64[:, :, :14]) / (1.0 + np.abs(o64[:, :, :14]))
    res = {"model": model, "K": K}
    for t in (32, 64, 128):
        e = rel[:, t - 1].max(axis=1)
        R64, R32 = r64[:, :t].sum(1), r32[:, :t].sum(1)
        er = np.abs(R32 - R64) / np.abs(R64)
        res["t=%d" % t] = {"state_rel_median": float(np.median(e)), "state_rel_p99": float(np.percentile(e, 99)),
                           "state_rel_max": float(e.max()), "frac_above_1e-4": float((e > 1e-4).mean()),
                           "return_rel_max": float(er.max()), "argmax_equal": bool(np.argmax(R64) == np.argmax(R32))}
    print(json.dumps(res), flush=True)
