"""Random walks over the planner's entry points.  The library offers several ways to make the same planning step --
`mppi_plan` (captured single steps, captured runs of eight), `generate_noise_step` + `iterate` + `advance`,
`advance_noise` that draws the next step's noise in the advance launch, host-uploaded noise -- with host-side bookkeeping
behind them (which noise the device holds, which graph is current, whether the host mirror of the last step is valid).
A planner driven by a random mix of them, with parameters, start state and act_sequence changed at random moments, must
log exactly what a second planner logs that makes every step the plainest way.  Bit for bit, fp32 and fp64.

MPPI_FUZZ_WALKS=<n> runs n walks per precision instead of the default few (tools: a soak run on the GPU box)."""
import os

import numpy as np
import pytest

from oracle import ref_mppi

pytestmark = pytest.mark.gpu
FILTER = [np.ones(7), 0.25, 0.8, 0.0]
WALKS = int(os.environ.get("MPPI_FUZZ_WALKS", "4"))


def _planner(dm, K, H, dtype, default_act):
    from trajopt_amd.device import DevicePlanner
    return DevicePlanner(dm, K, H, kappa=25.0, gamma=1.0, sigma=FILTER[0], betas=FILTER[1:], mean=np.zeros(7),
                         default_act=default_act, dtype=dtype)


class _Plain:
    """The comparison planner: every step is generate_noise_step / upload_noise, iterate, advance."""

    def __init__(self, p, seed):
        self.p, self.seed, self.niter = p, seed, 1

    def step(self, raw=None):
        if raw is None:
            self.p.generate_noise_step(self.seed)
        else:
            self.p.upload_noise(raw)
        self.p.iterate(self.niter)
        self.p.advance()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("walk", range(WALKS))
def test_random_walk_over_the_entry_points(dtype, walk):
    from trajopt_amd.device import DeviceModel
    dm = DeviceModel.get("reacher_7dof")
    rs = np.random.RandomState(1000 * walk + (dtype == "f32"))
    K = int(rs.choice([40, 64, 200, 4096]))
    H = int(rs.choice([4, 9, 16]))
    default_act = str(rs.choice(["mean", "repeat"]))
    seed = int(rs.randint(1, 10 ** 6))
    a, b = _planner(dm, K, H, dtype, default_act), _planner(dm, K, H, dtype, default_act)
    plain = _Plain(b, seed)
    q0, v0, tgt = 0.3 * rs.randn(7), rs.randn(7), rs.uniform(-0.2, 0.2, 3)
    for p in (a, b):
        p.set_start_state(q0, v0, tgt, 0)
        p.set_base_actions(np.zeros((H, 7)))
    niter, steps, trace = 1, 0, []

    def same(what):
        trace.append(what)
        la, lb = a.read_log(max_rows=64), b.read_log(max_rows=64)
        where = "K=%d H=%d %s: %s" % (K, H, default_act, " | ".join(trace))
        assert la["total"] == lb["total"] == steps, (where, la["total"], lb["total"], steps)
        for k in ("qpos", "qvel", "hand", "reward", "action"):
            assert np.array_equal(la[k], lb[k]), (where, k, int(np.argmax((la[k] != lb[k]).any(axis=-1))), len(la[k]))
        assert np.array_equal(a.act_sequence(), b.act_sequence()), where
        assert a.status() == b.status() == 0, where

    for _ in range(30):
        op = rs.randint(10)
        if op <= 2:                                   # mppi_plan: single steps, runs, runs with a remainder
            n = int(rs.choice([1, 1, 2, 3, 8, 9, 17]))
            a.plan(seed, niter, n)
            for _ in range(n):
                plain.step()
            steps += n
            same("plan %d" % n)
            if rs.rand() < 0.5:                       # what a caller of single steps reads back
                rows, seq = a.read_step()
                lb = b.read_log(max_rows=1)
                assert np.array_equal(rows[:7], lb["qpos"][0]) and np.array_equal(seq, b.act_sequence()), trace
        elif op == 3:                                 # the plain sequence on both
            a.generate_noise_step(seed); a.iterate(niter); a.advance()
            plain.step()
            steps += 1
            same("noise, iterate, advance")
        elif op == 4:                                 # the next step's noise drawn in the advance launch
            a.generate_noise_step(seed); a.iterate(niter); a.advance_noise(seed)
            plain.step()
            steps += 1
            same("noise, iterate, advance_noise")
            if rs.rand() < 0.5:                       # ... and used by a step that does not draw again
                a.iterate(niter); a.advance()
                plain.step()
                steps += 1
                same("iterate on the noise drawn ahead, advance")
        elif op == 5:                                 # host noise for one step
            raw = rs.randn(K, H, 7)
            a.upload_noise(raw); a.iterate(niter); a.advance()
            plain.step(raw)
            steps += 1
            same("uploaded noise")
        elif op == 6:                                 # parameters (captured graphs must follow)
            kappa, gamma = float(rs.choice([1.0, 25.0, 60.0])), float(rs.choice([1.0, 0.95]))
            rt = str(rs.choice(["dense", "sparse"]))
            a.set_params(kappa, gamma, rt); b.set_params(kappa, gamma, rt)
            trace.append("params %g %g %s" % (kappa, gamma, rt))
        elif op == 7:                                 # updates per step
            niter = plain.niter = int(rs.choice([1, 2, 3]))
            trace.append("niter %d" % niter)
        elif op == 8:                                 # the caller moves the start state (time step included)
            q, v, t = 0.3 * rs.randn(7), rs.randn(7), int(rs.randint(0, 50))
            a.set_start_state(q, v, tgt, t); b.set_start_state(q, v, tgt, t)
            trace.append("start state, t = %d" % t)
        else:                                         # ... or the action sequence
            seq = 0.2 * rs.randn(H, 7)
            a.set_base_actions(seq); b.set_base_actions(seq)
            rows, got = a.read_step()                 # the mirror of the last step is stale now: the device's copy counts
            assert np.array_equal(got, seq), trace
            trace.append("act_sequence")
    assert steps > 0 or True
    a.close(); b.close()
