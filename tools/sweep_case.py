"""One case of the randomised parity sweep looked at closely: python tools/sweep_case.py <case> -- where oracle and device
(fp64, fp32) returns are finite, and the samples on which they differ most."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle
oracle.build()
from oracle.sim import OracleSim
from trajopt_amd.model import get_model
from trajopt_amd.device import DeviceModel, DevicePlanner
from sweep_cases import cases
want = int(sys.argv[1])
md = get_model("reacher_7dof")
sim = OracleSim(md)
dm = DeviceModel.get("reacher_7dof")
K, H = 256, 32
for case, q0, v0, tgt, raw, base, act in cases(md, want + 1, K, H):
    if case != want:
        continue
    print("q0", q0, "\nv0", v0, "\nlo", md.range_lo, "\nhi", md.range_hi)
    R_o, nobs_o, rew_o = sim.rollout(q0, v0, tgt, 0, act, dump=True, nthreads=8)
    print("oracle: non-finite returns", int((~np.isfinite(R_o)).sum()), "first bad step per bad sample",
          [int(np.argmax(~np.isfinite(nobs_o[k]).all(axis=1))) for k in np.where(~np.isfinite(R_o))[0][:8]])
    for dt in ("f64", "f32"):
        p = DevicePlanner(dm, K, H, dtype=dt, dump=True, kappa=25.0, sigma=np.ones(7), betas=(0.25, 0.8, 0.0))
        p.set_start_state(q0, v0, tgt, 0); p.set_base_actions(base); p.upload_noise(raw); p.rollout()
        R, nobs = p.returns(), p.next_observations()
        bad = np.where(~np.isfinite(R))[0]
        print(dt, "status", p.status(), "non-finite returns", len(bad), "samples", bad[:8],
              "first bad step", [int(np.argmax(~np.isfinite(nobs[k]).all(axis=1))) for k in bad[:8]])
        ok = np.isfinite(R) & np.isfinite(R_o)
        if ok.any():
            print("   finite samples: max |dR| / max |R|", np.abs(R[ok] - R_o[ok]).max() / np.abs(R_o[ok]).max())
        for k in bad[:2]:
            t = int(np.argmax(~np.isfinite(nobs[k]).all(axis=1)))
            print("   sample", k, "step", t, "oracle obs before", nobs_o[k, max(t - 1, 0), :14], "\n      device obs before", nobs[k, max(t - 1, 0), :14],
                  "\n      action", act[k, t], "\n      oracle obs at", nobs_o[k, t, :14])
        p.close()
