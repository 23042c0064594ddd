"""Randomised parity sweep: device rollouts (fp64 and fp32) against the oracle from the 40 random states of
tests/sweep_cases.py; K = 256, H = 32.  fp64 must agree to 1e-8 everywhere; for fp32 every case's figures are printed
(tests/test_gpu_sweep.py holds the same sweep to stated bands; tools/outlier_trace.py replays it on the host build of
the kernel arithmetic and says where the precisions part ways)."""
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
md = get_model("reacher_7dof")
sim = OracleSim(md)
dm = DeviceModel.get("reacher_7dof")
K, H = 256, 32
worst = {"f64": 0.0, "f32_R": 0.0}
pl = {dt: DevicePlanner(dm, K, H, dtype=dt, dump=True, kappa=25.0, sigma=np.ones(7), betas=(0.25, 0.8, 0.0)) for dt in ("f64", "f32")}
for case, q0, v0, tgt, raw, base, act in cases(md, 40, K, H, "--as-swept" in sys.argv):
    R_o, nobs_o, rew_o = sim.rollout(q0, v0, tgt, 0, act, dump=True, nthreads=8)
    for dt, p in pl.items():
        p.set_start_state(q0, v0, tgt, 0); p.set_base_actions(base); p.upload_noise(raw); p.rollout()
        R, nobs = p.returns(), p.next_observations()
        st = p.status()
        if dt == "f64":
            e = max(np.abs(R - R_o).max(), np.abs(nobs - nobs_o).max())
            worst["f64"] = max(worst["f64"], e)
            assert e < 1e-8 and st == 0, (case, e, st)
        else:
            eR = np.abs(R - R_o).max() / np.abs(R_o).max()
            # final state of every sample, relative to the size of the state
            es = np.abs(nobs[:, -1, :14] - nobs_o[:, -1, :14]).max(axis=1) / (1 + np.abs(nobs_o[:, -1, :14]).max(axis=1))
            worst["f32_R"] = max(worst["f32_R"], eR)
            assert st & 1 == 0 and np.isfinite(R).all(), (case, st)
            print("case %2d |v0| %4.1f  f32: rel dR %.2e; final state: median %.2e, 99th pct %.2e, samples > 1e-4: %d of %d%s"
                  % (case, np.abs(v0).max(), eR, np.median(es), np.percentile(es, 99), int((es > 1e-4).sum()), K,
                     "   <-- returns above 1e-4" if eR > 1e-4 else ""), flush=True)
print("worst", worst)
