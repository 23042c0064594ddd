"""The randomised parity sweep as a test: 40 start states near and beyond the joint limits, on a limit value, pitched
into the table, with velocities up to ~12 rad/s and action scales up to 3 (tests/sweep_cases.py), K = 256 samples of
H = 32 steps each, device rollouts in fp64 and fp32 against the oracle.

Bands.  fp64: returns and every state of every sample within 1e-8, status word clean.  fp32: returns within the planner's
tolerance (relative 1e-4 of the largest return) in at least 38 of the 40 cases and within 5e-3 in all; per sample, the
final state within 1e-4 (relative to the size of the state) for at least 99 % of all samples.  The samples outside are
of three kinds, told apart by tools/outlier_trace.py on the host build of the same arithmetic: a joint that reaches its
limit a substep earlier or later in one precision (a row exists iff q is outside its range: a discontinuity of the
simulator), a joint that lands inside the one-milliradian impedance ramp at first contact (the ramp turns a 1e-6
position difference into a 1e-3 difference of the row's weight), and smooth growth in stiff phases (velocities of
10-40 rad/s on the light wrist links, where the 7 x 7 systems have condition numbers of 300-500 and more with rows)."""
import numpy as np
import pytest

from sweep_cases import cases

pytestmark = pytest.mark.gpu


def test_randomised_sweep_against_the_oracle(oracle_sim):
    from trajopt_amd.device import DeviceModel, DevicePlanner
    from trajopt_amd.model import get_model
    md = get_model("reacher_7dof")
    dm = DeviceModel.get("reacher_7dof")
    K, H = 256, 32
    pl = {dt: DevicePlanner(dm, K, H, dtype=dt, dump=True, kappa=25.0, sigma=np.ones(7), betas=(0.25, 0.8, 0.0))
          for dt in ("f64", "f32")}
    above, worst_R, finals = [], 0.0, []
    for case, q0, v0, tgt, raw, base, act in cases(md, 40, K, H):
        R_o, nobs_o, _ = oracle_sim.rollout(q0, v0, tgt, 0, act, dump=True, nthreads=8)
        for dt, p in pl.items():
            p.set_start_state(q0, v0, tgt, 0)
            p.set_base_actions(base)
            p.upload_noise(raw)
            p.rollout()
            R, nobs = p.returns(), p.next_observations()
            if dt == "f64":
                assert max(np.abs(R - R_o).max(), np.abs(nobs - nobs_o).max()) < 1e-8, case
                assert p.status() == 0, case
            else:
                assert np.isfinite(R).all() and p.status() & 1 == 0, case
                eR = np.abs(R - R_o).max() / np.abs(R_o).max()
                worst_R = max(worst_R, eR)
                if eR > 1e-4:
                    above.append((case, float(eR)))
                finals.append(np.abs(nobs[:, -1, :14] - nobs_o[:, -1, :14]).max(axis=1) / (1 + np.abs(nobs_o[:, -1, :14]).max(axis=1)))
    for p in pl.values():
        p.close()
    finals = np.concatenate(finals)
    assert len(above) <= 2, above
    assert worst_R < 5e-3, above
    assert (finals < 1e-4).mean() >= 0.99, ((finals < 1e-4).mean(), np.percentile(finals, 99))
    assert np.median(finals) < 5e-6
