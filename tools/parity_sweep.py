"""Randomised parity sweep: device rollouts (fp64 and fp32) against the oracle from 40 random states near and beyond
the joint limits (some exactly on a limit value, some pitched into the table), random velocities, targets, action scales
and base actions; K = 256, H = 32.  fp64 must agree to 1e-8 everywhere; fp32 cases whose returns differ by more than 1e-4
are listed (trajectories that start on a discontinuity of the dynamics and take the other side in fp32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
oracle.build()
from oracle.sim import OracleSim
from oracle import ref_mppi
from trajopt_amd.model import get_model
from trajopt_amd.device import DeviceModel, DevicePlanner
md = get_model("reacher_7dof")
sim = OracleSim(md)
dm = DeviceModel.get("reacher_7dof")
rs = np.random.RandomState(0)
K, H = 256, 32
worst = {"f64": 0.0, "f32_R": 0.0, "f32_state": 0.0}
pl = {dt: DevicePlanner(dm, K, H, dtype=dt, dump=True, kappa=25.0, sigma=np.ones(7), betas=(0.25, 0.8, 0.0)) for dt in ("f64", "f32")}
lo, hi = np.array(md.range_lo), np.array(md.range_hi)
for case in range(40):
    span = hi - lo
    q0 = lo + span * rs.uniform(-0.05, 1.05, 7)          # also beyond the limits
    if case % 4 == 0:
        upper, j = rs.rand() < 0.5, rs.randint(7)
        q0[j] = hi[j] if upper else lo[j]                      # one joint exactly on one of its own limit values
    v0 = rs.normal(0, [0.0, 0.5, 2.0, 6.0][case % 4], 7)
    tgt = rs.uniform(-0.3, 0.3, 3)
    if case % 5 == 0:
        q0[1] = 1.2; q0[3] = -0.3                          # arm pitched towards the table: contact rows
    scale = [0.3, 1.0, 3.0][case % 3]
    raw = rs.normal(0, scale, (K, H, 7))
    base = rs.normal(0, 0.5, (H, 7))
    act = base + ref_mppi.filter_noise(raw.copy(), [np.ones(7), 0.25, 0.8, 0.0])
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
            es = np.abs(nobs[:, :, :14] - nobs_o[:, :, :14]).max(axis=(1, 2))
            frac_bad = float((es > 1e-2).mean())
            worst["f32_R"] = max(worst["f32_R"], eR); worst["f32_state"] = max(worst["f32_state"], float(np.median(es)))
            assert st & 1 == 0 and np.isfinite(R).all(), (case, st)
            print("case %2d f32 rel dR %.2e median state err %.2e p99 %.2e frac>1e-4 %.4f" % (case, eR, np.median(es), np.percentile(es, 99), float((es > 1e-4).mean())), flush=True)
            if eR > 1e-4: print("case %2d  |v0| %.1f scale %.1f  f32: rel dR %.2e, median state err %.2e, samples off by > 1e-2: %.3f, status %d" % (case, np.abs(v0).max(), scale, eR, np.median(es), frac_bad, st), flush=True)
print("worst", worst)
