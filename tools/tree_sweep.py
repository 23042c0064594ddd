"""Randomised parity sweep of the compiled non-reacher models (reacher_7dof_free and the three 18-DoF trees): device
rollouts in fp64 against the oracle, and single fp64 env steps, from N random states per model (joint angles up to
+-2 rad, random unit quaternions, rates up to ~5 rad/s, actions well past the control range).

    python tools/tree_sweep.py [N]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
oracle.build()
from oracle.sim import OracleSim
from trajopt_amd.model import get_model
from trajopt_amd.device import DeviceModel, DevicePlanner
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
K, H = 64, 24
for name in ("reacher_7dof_free", "tree_18dof", "tree_18dof_quat", "tree_18dof_rk4"):
    md = get_model(name)
    sim, dm = OracleSim(md), DeviceModel.get(name)
    nq, nv = sim.nq, sim.nv
    rs = np.random.RandomState(11)
    p = DevicePlanner(dm, K, H, kappa=5.0, gamma=0.99, sigma=np.ones(nv), betas=(0.25, 0.8, 0.0), dtype="f64", dump=True)
    worst_roll = worst_step = 0.0
    nwild = violent = 0
    for case in range(N):
        q0 = rs.uniform(-2.0, 2.0, nq) * rs.choice([0.1, 0.5, 1.0])
        if nq > nv:                                  # a free joint: position, unit quaternion
            quat = rs.randn(4); quat /= np.linalg.norm(quat)
            jt = list(md.jtype)
            a = md.qadr[jt.index(2)]
            q0[a:a + 4] = quat
        v0 = rs.randn(nv) * rs.choice([0.0, 1.0, 5.0])
        tgt = rs.uniform(-1, 1, 3)
        base = rs.randn(H, nv) * rs.choice([0.2, 2.0])
        p.set_start_state(q0, v0, tgt, 0); p.set_base_actions(base); p.generate_noise(case); p.rollout()
        act, nobs, R = p.actions(), p.next_observations(), p.returns()
        R_o, nobs_o, _ = sim.rollout(q0, v0, tgt, 0, act, gamma=0.99, dump=True, nthreads=8)
        # (a tumbling tree started at 10 rad/s can outrun the explicit integrator: samples whose states leave every
        # physical range -- above 60 rad or rad/s -- amplify rounding without bound, in the oracle as on the device, and
        # are compared for leaving it in both)
        wild_o = ~(np.abs(nobs_o).max(axis=(1, 2)) < 60.0)
        wild = ~(np.abs(nobs).max(axis=(1, 2)) < 60.0)
        assert (wild != wild_o).sum() <= 1, (name, case, wild.sum(), wild_o.sum())
        ok = ~wild & ~wild_o
        nwild += int(wild_o.sum())
        if wild_o.any():             # the whole case sits next to the instability: its tame samples amplify rounding too
            violent += 1
            assert np.abs(nobs[ok] - nobs_o[ok]).max() < 1e-3 * (1.0 + np.abs(nobs_o[ok]).max()) if ok.any() else True, (name, case)
            continue
        e = np.abs(nobs[ok] - nobs_o[ok]).max() / (1.0 + np.abs(nobs_o[ok]).max())
        worst_roll = max(worst_roll, e)
        assert e < 1e-8 and (p.status() & ~1) == 0, (name, case, e, p.status())
        a1 = rs.randn(nv) * 3
        q2, v2, obs, r = dm.env_step(q0, v0, tgt, a1, 0)
        qo, vo, obs_o, ro, _ = sim.env_step(q0, v0, tgt, a1)
        e1 = max(np.abs(q2 - qo).max(), np.abs(v2 - vo).max(), abs(r - ro)) / (1.0 + np.abs(vo).max())
        worst_step = max(worst_step, e1)
        assert e1 < 1e-10, (name, case, e1)
    p.close()
    print("%-18s %d cases x %d samples x %d steps: rollouts within %.2e of the oracle (relative to the largest state), single steps within %.2e; "
          "%d cases next to the integrator's instability set aside (%d samples left the physical range, in oracle and device alike)"
          % (name, N, K, H, worst_roll, worst_step, violent, nwild), flush=True)
