"""Randomised parity sweep: device rollouts (fp64 and fp32) against the oracle from the random states of
tests/sweep_cases.py (40 of them, or the number given as the first argument: the generator's stream continues, so a
longer sweep starts with the same cases); K = 256, H = 32.  fp64 must agree to 1e-8 everywhere; for fp32 every case's figures are printed
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
N = next((int(a) for a in sys.argv[1:] if a.isdigit()), 40)
WILD = 60.0
worst = {"f64": 0.0, "f32_R": 0.0}
bands = {"cases": 0, "returns_above_1e-4": 0, "samples": 0, "samples_above_1e-4": 0, "samples_above_1e-3": 0, "samples_above_1e-2": 0,
         "diverged_samples": 0, "wild_samples": 0, "wild_disagree_f32": 0}
pl = {dt: DevicePlanner(dm, K, H, dtype=dt, dump=True, kappa=25.0, sigma=np.ones(7), betas=(0.25, 0.8, 0.0)) for dt in ("f64", "f32")}
for case, q0, v0, tgt, raw, base, act in cases(md, N, K, H, "--as-swept" in sys.argv):
    R_o, nobs_o, rew_o = sim.rollout(q0, v0, tgt, 0, act, dump=True, nthreads=8)
    # A start pose with the contact sphere deep inside the table throws the arm so hard that the explicit integrator
    # diverges for some samples, in the oracle as on the device.  Those samples (and the ones on their way there, whose
    # states have left every physical range: a joint angle or rate above WILD = 60 rad or rad/s, three times the fastest
    # joint in the reference's stored runs; rounding differences grow by orders of magnitude per step there) are compared
    # for agreeing that they diverge, not digit by digit.
    wild_o = ~np.isfinite(R_o) | ~(np.abs(nobs_o[:, :, :14]).max(axis=(1, 2)) < WILD)
    for dt, p in pl.items():
        p.set_start_state(q0, v0, tgt, 0); p.set_base_actions(base); p.upload_noise(raw); p.rollout()
        R, nobs = p.returns(), p.next_observations()
        st = p.status()
        wild = ~np.isfinite(R) | ~(np.abs(nobs[:, :, :14]).max(axis=(1, 2)) < WILD)
        ok = ~wild_o & ~wild
        if dt == "f64":
            assert np.array_equal(np.isfinite(R), np.isfinite(R_o)) and (wild != wild_o).sum() <= 2, (case, wild.sum(), wild_o.sum())
            e = max(np.abs(R[ok] - R_o[ok]).max(), np.abs(nobs[ok] - nobs_o[ok]).max()) if ok.any() else 0.0
            worst["f64"] = max(worst["f64"], e)
            # (the status word is sticky: bit 0 stays set once a case has diverged)
            assert e < 1e-8 and (st & ~1) == 0 and (bool(st & 1) or np.isfinite(R_o).all()), (case, e, st)
            assert bool(st & 1) == (bands["diverged_samples"] + int((~np.isfinite(R_o)).sum()) > 0), (case, st)
            bands["diverged_samples"] += int((~np.isfinite(R_o)).sum()); bands["wild_samples"] += int(wild_o.sum())
        else:
            assert (st & ~1) == 0, (case, st)
            bands["wild_disagree_f32"] += int((wild != wild_o).sum())
            if not ok.any():
                continue
            eR = np.abs(R[ok] - R_o[ok]).max() / np.abs(R_o[ok]).max()
            # final state of every sample, relative to the size of the state
            es = np.abs(nobs[ok, -1, :14] - nobs_o[ok, -1, :14]).max(axis=1) / (1 + np.abs(nobs_o[ok, -1, :14]).max(axis=1))
            worst["f32_R"] = max(worst["f32_R"], eR)
            bands["cases"] += 1; bands["returns_above_1e-4"] += int(eR > 1e-4); bands["samples"] += int(ok.sum())
            for thr in ("1e-4", "1e-3", "1e-2"):
                bands["samples_above_" + thr] += int((es > float(thr)).sum())
            print("case %2d |v0| %4.1f  f32: rel dR %.2e; final state: median %.2e, 99th pct %.2e, samples > 1e-4: %d of %d%s%s"
                  % (case, np.abs(v0).max(), eR, np.median(es), np.percentile(es, 99), int((es > 1e-4).sum()), int(ok.sum()),
                     "   <-- returns above 1e-4" if eR > 1e-4 else "", "   (%d samples left every physical range)" % wild_o.sum() if wild_o.any() else ""),
                  flush=True)
print("worst", worst)
print("fp32 bands", bands)
