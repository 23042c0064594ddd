"""Randomised start states and action sets for the parity sweep (tests/test_gpu_sweep.py, tools/parity_sweep.py,
tools/outlier_trace.py): states near and beyond the joint limits (every fourth case with one joint exactly on one of
its own limit values, every fifth with the arm pitched into the table), random velocities up to ~12 rad/s, targets,
action scales and base actions."""
import numpy as np

from oracle import ref_mppi

FILTER = [np.ones(7), 0.25, 0.8, 0.0]


def cases(md, n=40, K=256, H=32, as_swept_in_round_2=False):
    """Yields (case, q0, v0, target, raw draws (K, H, 7), base actions (H, 7), actions (K, H, 7)).
    as_swept_in_round_2: the draw order of an earlier version of the sweep, which picked the joint that receives a limit
    value and the joint whose limit it is separately (kept to reproduce profiles/r2_parity_sweep.txt)."""
    lo, hi = np.array(md.range_lo), np.array(md.range_hi)
    rs = np.random.RandomState(0)
    for case in range(n):
        span = hi - lo
        q0 = lo + span * rs.uniform(-0.05, 1.05, 7)          # also beyond the limits
        if case % 4 == 0:
            if as_swept_in_round_2:
                upper = rs.rand() < 0.5
                src, dst = rs.randint(7), rs.randint(7)
                q0[dst] = hi[src] if upper else lo[src]
            else:
                upper, j = rs.rand() < 0.5, rs.randint(7)
                q0[j] = hi[j] if upper else lo[j]            # one joint exactly on one of its own limit values
        v0 = rs.normal(0, [0.0, 0.5, 2.0, 6.0][case % 4], 7)
        tgt = rs.uniform(-0.3, 0.3, 3)
        if case % 5 == 0:
            q0[1] = 1.2; q0[3] = -0.3                        # arm pitched towards the table: contact rows
        scale = [0.3, 1.0, 3.0][case % 3]
        raw = rs.normal(0, scale, (K, H, 7))
        base = rs.normal(0, 0.5, (H, 7))
        act = base + ref_mppi.filter_noise(raw.copy(), FILTER)
        yield case, q0, v0, tgt, raw, base, act
