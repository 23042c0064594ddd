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


def _planner(dm, K, H, dtype, default_act, P=1):
    from trajopt_amd.device import DevicePlanner
    return DevicePlanner(dm, K, H, kappa=25.0, gamma=1.0, sigma=FILTER[0], betas=FILTER[1:], mean=np.zeros(7),
                         default_act=default_act, dtype=dtype, planners=P)


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
    K = int(rs.choice([1, 8, 40, 63, 64, 65, 200, 4096]))
    H = int(rs.choice([1, 2, 4, 9, 16]))
    default_act = str(rs.choice(["mean", "repeat"]))
    seed = int(rs.randint(1, 10 ** 6))
    # every third walk drives a handle of three planners (own states, own noise planes, own logs)
    P = 3 if walk % 3 == 2 else 1
    lead = (P,) if P > 1 else ()
    a, b = _planner(dm, K, H, dtype, default_act, P), _planner(dm, K, H, dtype, default_act, P)
    plain = _Plain(b, seed)
    q0, v0, tgt = 0.3 * rs.randn(*lead, 7), rs.randn(*lead, 7), rs.uniform(-0.2, 0.2, lead + (3,))
    for p in (a, b):
        p.set_start_state(q0, v0, tgt, 0)
        p.set_base_actions(np.zeros(lead + (H, 7)))
    niter, steps, trace = 1, 0, []

    def same(what):
        trace.append(what)
        where = "P=%d K=%d H=%d %s: %s" % (P, K, H, default_act, " | ".join(trace))
        for pl in range(P):
            la, lb = a.read_log(max_rows=64, planner=pl), b.read_log(max_rows=64, planner=pl)
            assert la["total"] == lb["total"] == steps, (where, la["total"], lb["total"], steps)
            for k in ("qpos", "qvel", "hand", "reward", "action"):
                assert np.array_equal(la[k], lb[k]), (where, pl, k, int(np.argmax((la[k] != lb[k]).any(axis=-1))), len(la[k]))
        assert np.array_equal(a.act_sequence(), b.act_sequence()), where
        assert a.status() == b.status() == 0, where

    reference = np.cumsum(0.002 * rs.randn(2000, 3), axis=0) + np.array([0.6, -0.3, 0.2])
    crit = [(0.3 * rs.randn(128, 14)).astype(np.float32), (0.1 * rs.randn(128)).astype(np.float32),
            (0.1 * rs.randn(128, 128)).astype(np.float32), (0.1 * rs.randn(128)).astype(np.float32),
            (3.0 * rs.randn(1, 128)).astype(np.float32), np.array([0.5], np.float32)]
    pending = None                                    # seed and worker count of host draws made ahead and not yet used
    for _ in range(30):
        op = rs.randint(17)
        if op <= 2:                                   # mppi_plan: single steps, runs, runs with a remainder
            n = int(rs.choice([1, 1, 2, 3, 8, 9, 17]))
            a.plan(seed, niter, n)
            for _ in range(n):
                plain.step()
            steps += n
            same("plan %d" % n)
            if rs.rand() < 0.5:                       # what a caller of single steps reads back
                rows, seq = a.read_step()
                lb = b.read_log(max_rows=1, planner=P - 1)
                assert np.array_equal(rows[..., :7].reshape(P, 7)[P - 1], lb["qpos"][0]) and np.array_equal(seq, b.act_sequence()), trace
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
            raw = rs.randn(*lead, K, H, 7)
            a.upload_noise(raw); a.iterate(niter); a.advance()
            pending = None                            # (the upload went through the staging area the draws waited in)
            plain.step(raw)
            steps += 1
            same("uploaded noise")
        elif op >= 15:                                # numpy's worker streams drawn by the library (single planners)
            if P > 1:
                continue
            from trajopt_amd.utils import worker_seeds
            if pending is None:
                pending = (int(rs.randint(0, 2 ** 31)), int(rs.choice([c for c in (1, 2, 4, 8) if K % c == 0])))
                a.draw_noise(worker_seeds(pending[0], K // pending[1], pending[1]), int(rs.choice([1, 3, 16])))
            sd, ncpu = pending
            a.commit_noise(); a.iterate(niter); a.advance()
            pending = None
            if rs.rand() < 0.5:                       # the next step's draws made while the device still works on this one
                pending = (int(rs.randint(0, 2 ** 31)), int(rs.choice([c for c in (1, 2, 4, 8) if K % c == 0])))
                a.draw_noise(worker_seeds(pending[0], K // pending[1], pending[1]), int(rs.choice([1, 3, 16])))
            plain.step(ref_mppi.all_raw_normals(sd, K // ncpu, ncpu, H, 7))       # numpy's own streams, uploaded
            steps += 1
            same("library-drawn host noise (%d workers)%s" % (ncpu, ", next drawn ahead" if pending else ""))
        elif op == 6:                                 # parameters (captured graphs must follow)
            kappa, gamma = float(rs.choice([1.0, 25.0, 60.0])), float(rs.choice([1.0, 0.95]))
            rt = str(rs.choice(["dense", "sparse"]))
            a.set_params(kappa, gamma, rt); b.set_params(kappa, gamma, rt)
            trace.append("params %g %g %s" % (kappa, gamma, rt))
        elif op == 7:                                 # updates per step
            niter = plain.niter = int(rs.choice([1, 2, 3]))
            trace.append("niter %d" % niter)
        elif op == 8:                                 # the caller moves the start state (time step included)
            q, v, t = 0.3 * rs.randn(*lead, 7), rs.randn(*lead, 7), int(rs.randint(0, 50))
            a.set_start_state(q, v, tgt, t); b.set_start_state(q, v, tgt, t)
            trace.append("start state, t = %d" % t)
        elif op == 10:                                # rewards that read a reference trajectory; the real step's own reward
            rt = str(rs.choice(["tracking", "cooling", "dense"]))
            alpha = float(rs.choice([1.0, 0.6]))
            env_rt = rs.choice(["sparse", "dense", None])
            for p in (a, b):
                p.set_reference(reference if rt != "dense" else None, alpha)
                p.set_params(25.0, 1.0, rt)
                p.set_env_reward(None if env_rt is None else str(env_rt))
            trace.append("reward %s alpha %g env %s" % (rt, alpha, env_rt))
        elif op == 11:                                # terminal-value critic on / off
            on = rs.rand() < 0.6
            for p in (a, b):
                p.set_critic(14 if on else 0, crit if on else None)
            trace.append("critic %s" % on)
        elif op == 12:                                # rollouts alone: returns change, the log and act_sequence do not
            a.generate_noise_step(seed); a.rollout()
            same("rollout only")
        elif op == 13:                                # the update as shard statistics + merge (one shard)
            a.generate_noise_step(seed)
            for _ in range(niter):
                a.local_partials()
                ptr, nbytes = a.partials_ptr()
                a.merge(ptr, 1)
            a.advance()
            plain.step()
            steps += 1
            same("partials, merge, advance")
        else:                                         # ... or the action sequence
            seq = 0.2 * rs.randn(*lead, H, 7)
            a.set_base_actions(seq); b.set_base_actions(seq)
            rows, got = a.read_step()                 # the mirror of the last step is stale now: the device's copy counts
            assert np.array_equal(got, seq), trace
            trace.append("act_sequence")
    assert steps > 0 or True
    a.close(); b.close()


@pytest.mark.parametrize("noise", ["philox", "host"])
@pytest.mark.parametrize("walk", range(max(WALKS // 2, 2)))
def test_random_walk_over_the_mppi_class(noise, walk):
    """The same idea one level up: an `MPPI` agent whose `train_step` stays on the device (state and act_sequence are
    re-sent only when the caller changed them, the real environment's step is made by the planner) against a twin that
    sends everything at every step and steps its environment object itself, while the caller edits `act_sequence`,
    `sol_state[-1]`, parameters and the critic between steps, or pickles the agent and goes on with the copy.  fp64, bit
    for bit."""
    import pickle
    import torch
    from trajopt.algos.mppi import MPPI, _critic_weights
    from trajopt.envs.utils import get_environment
    from trajopt.models.critic_nets import Critic
    rs = np.random.RandomState(77 * walk + (noise == "host"))
    H, K = int(rs.choice([6, 12])), int(rs.choice([40, 96]))
    seed = int(rs.randint(1, 10 ** 5))
    env_rt, rt = [("dense", "dense"), ("sparse", "cooling"), ("tracking", "tracking")][rs.randint(3)]
    reference = np.cumsum(0.002 * rs.randn(400, 3), axis=0) + np.array([0.6, -0.3, 0.2])
    goal = rs.uniform(-0.1, 0.1, 3)
    torch.manual_seed(walk)
    critic = Critic(input_dim=14)
    with torch.no_grad():
        critic.linear3.weight.mul_(20.0)

    def agent():
        e = get_environment('reacher_7dof', reward_type=env_rt, reference=reference if env_rt == "tracking" else None)
        e.reset_model(seed=5, goal=goal, alpha=0.8)
        return MPPI(e, H=H, paths_per_cpu=K, kappa=25.0, gamma=1.0, mean=np.zeros(7), filter_coefs=FILTER,
                    default_act=str(rs.choice(["mean", "repeat"])), seed=seed, noise=noise, dtype="f64", reward_type=rt,
                    reference=reference if rt != "dense" else None)
    state = rs.get_state()
    a = agent()
    rs.set_state(state)
    b = agent()
    niter, use_critic, trace = 1, False, []
    for _ in range(25):
        op = rs.randint(9)
        if op <= 3:
            c = critic if use_critic else None
            a.train_step(critic=c, niter=niter, goal=goal)
            # the twin: everything sent again, iterations on the device, advance_time through the environment object
            b._dev_state = b._dev_seq = None
            b._plan_on_device(b.seed + len(b.sol_state) - 1, niter, 14 if use_critic else 0, _critic_weights(c, 14) if use_critic else None)
            b.advance_time()
            trace.append("step")
            where = "%s %s/%s H=%d K=%d: %s" % (noise, env_rt, rt, H, K, " | ".join(trace))
            for k in ("qp", "qv", "target_pos"):
                assert np.array_equal(a.sol_state[-1][k], b.sol_state[-1][k]), (where, k)
            assert a.sol_state[-1]["timestep"] == b.sol_state[-1]["timestep"] and a.sol_reward[-1] == b.sol_reward[-1], where
            assert np.array_equal(a.act_sequence, b.act_sequence), (
                where, np.abs(a.act_sequence - b.act_sequence).max(), np.argwhere(a.act_sequence != b.act_sequence).tolist(),
                "returns equal: %s" % np.array_equal(a._planner.returns(), b._planner.returns()),
                "noise equal: %s" % np.array_equal(a._planner.actions(), b._planner.actions()))
            assert np.array_equal(a.sol_obs[-1], b.sol_obs[-1]), (where, a.sol_obs[-1] - b.sol_obs[-1], a.env.alpha, b.env.alpha)
            assert np.array_equal(a.sol_act[-1], b.sol_act[-1]) and len(a.sol_state) == len(b.sol_state), where
            sa, sb = a.env.get_env_state(), b.env.get_env_state()
            assert np.array_equal(sa["qp"], sb["qp"]) and sa["timestep"] == sb["timestep"], where
        elif op == 4:
            d = 0.3 * rs.randn(2, 7)
            a.act_sequence[1:3] += d; b.act_sequence[1:3] += d
            trace.append("act_sequence edited in place")
        elif op == 5:
            new = 0.2 * rs.randn(H, 7)
            a.act_sequence = new.copy(); b.act_sequence = new.copy()
            trace.append("act_sequence rebound")
        elif op == 6:
            d = 0.05 * rs.randn(7)
            if rs.rand() < 0.5:
                a.sol_state[-1]["qp"][:] = a.sol_state[-1]["qp"] + d; b.sol_state[-1]["qp"][:] = b.sol_state[-1]["qp"] + d
                trace.append("state edited in place")
            else:
                for ag in (a, b):
                    s = ag.sol_state[-1]
                    ag.sol_state[-1] = dict(qp=s["qp"] + d, qv=s["qv"] * 0.5, target_pos=s["target_pos"].copy(), timestep=s["timestep"])
                trace.append("state replaced")
        elif op == 7:
            niter = int(rs.choice([1, 2, 3]))
            use_critic = bool(rs.rand() < 0.5)
            a.kappa = b.kappa = float(rs.choice([5.0, 25.0]))
            trace.append("niter %d critic %s kappa %g" % (niter, use_critic, a.kappa))
        else:
            a = pickle.loads(pickle.dumps(a))
            # (the environment is rebuilt from its constructor arguments, as gym's EzPickle does for the reference's:
            # what `reset_model` had set -- alpha -- is the caller's to set again)
            assert a.env.alpha == 1.0
            a.env.alpha = b.env.alpha
            trace.append("pickled")
    assert len(a.sol_state) == len(b.sol_state)


@pytest.mark.parametrize("case", range(max(2 * WALKS, 8)))
def test_random_configurations_of_gather_paths_parallel(oracle_sim, case):
    """`gather_paths_parallel` (trajopt/utils.py:148-163) over random horizons, worker layouts, filters (per-dimension
    sigma, all three taps), reward types, start states and time steps: the paths the reference's list-of-dicts schema
    carries, value by value against the oracle (fp64 rollouts, 1e-10)."""
    from trajopt import utils
    rs = np.random.RandomState(4242 + case)
    H, ppc, ncpu = int(rs.randint(1, 40)), int(rs.randint(1, 70)), int(rs.choice([1, 2, 3, 5]))
    filt = [rs.uniform(0.2, 1.5, 7), float(rs.uniform(0.1, 0.6)), float(rs.uniform(0.0, 0.9)), float(rs.choice([0.0, 0.15]))]
    rt = str(rs.choice(["dense", "sparse", "tracking", "cooling"]))
    t0 = int(rs.randint(0, 30))
    reference = np.cumsum(0.004 * rs.randn(t0 + H + 5, 3), axis=0) + np.array([0.6, -0.3, 0.2]) if rt in ("tracking", "cooling") else None
    alpha = float(rs.choice([1.0, 0.4]))
    s = dict(qp=rs.uniform(-0.8, 0.8, 7) * np.array([1, 0.5, 1, 1, 1, 0.5, 1]), qv=rs.randn(7) * rs.choice([0.0, 2.0]),
             target_pos=rs.uniform(-0.2, 0.2, 3), timestep=t0)
    base = rs.randn(H, 7) * 0.3
    seed = int(rs.randint(0, 2 ** 31))
    utils.ROLLOUT_DTYPE = "f64"
    try:
        paths = utils.gather_paths_parallel('reacher_7dof', s, base, filt, seed, None, rt, reference, alpha, ppc, ncpu)
    finally:
        utils.ROLLOUT_DTYPE = "f32"
    assert len(paths) == ppc * ncpu
    act = base + ref_mppi.filter_noise(ref_mppi.all_raw_normals(seed, ppc, ncpu, H, 7), filt)
    R_o, nobs_o, rew_o = oracle_sim.rollout(s["qp"], s["qv"], s["target_pos"], t0, act, reward_type=rt, reference=reference,
                                            alpha=alpha, dump=True, nthreads=4)
    what = (case, H, ppc, ncpu, rt)
    for k in rs.choice(len(paths), size=min(len(paths), 12), replace=False):
        p = paths[k]
        assert np.array_equal(p["actions"], act[k]), what
        assert np.abs(np.array(p["next_observations"]) - nobs_o[k]).max() < 1e-10, what
        assert np.abs(p["rewards"] - rew_o[k]).max() < 1e-10 * (1 + np.abs(rew_o[k]).max()), what
        assert p["observations"].shape == (H, 20) and (H == 1 or np.abs(p["observations"][1:] - nobs_o[k][:-1]).max() < 1e-10), what
        assert np.array_equal(p["env_infos"]["state"]["timestep"], t0 + np.arange(H)) and len(p["states"]) == H, what


@pytest.mark.parametrize("case", range(max(2 * WALKS, 8)))
def test_random_shapes_of_the_update(case):
    """The planner's update (maximum, first arg-max, weights, weighted sums; the launch shape changes with K, H and
    the number of planners) over random sample counts from one to 150 000, horizons, planners per handle and kappa from
    0 to 300, against the numpy update of the device's own returns and actions; and `mppi_update_from_returns` for
    caller-supplied (actions, returns) of random shape."""
    from trajopt_amd.device import DeviceModel, DevicePlanner, update_from_returns
    rs = np.random.RandomState(999 + case)
    dm = DeviceModel.get("reacher_7dof")
    H = int(rs.randint(1, 33))
    P = int(rs.choice([1, 1, 1, 2, 5, 40, 200]))         # (40 and 200: the batched shapes, a sub-block of waves per step)
    kmax = min(150000, 2_000_000 // (H * P))
    K = int(np.exp(rs.uniform(0, np.log(kmax))))
    kappa = float(rs.choice([0.0, 1.0, 25.0, 300.0]))
    dtype = str(rs.choice(["f32", "f64"]))
    lead = (P,) if P > 1 else ()
    p = DevicePlanner(dm, K, H, kappa=kappa, gamma=float(rs.choice([1.0, 0.9])), sigma=np.ones(7), betas=(0.25, 0.8, 0.0),
                      dtype=dtype, planners=P, default_act="mean")
    p.set_start_state(0.3 * rs.randn(*lead, 7), rs.randn(*lead, 7), rs.uniform(-0.2, 0.2, lead + (3,)), 0)
    base = 0.3 * rs.randn(*lead, H, 7)
    p.set_base_actions(base)
    p.generate_noise(int(rs.randint(1, 10 ** 6)))
    act = p.actions().reshape(P, K, H, 7)             # (read before the update moves the base they are relative to)
    p.iterate(1)
    R, seq = p.returns().reshape(P, K), p.act_sequence().reshape(P, H, 7)
    am, rmax = np.atleast_1d(p.argmax()), np.atleast_1d(p.rmax())
    what = (case, K, H, P, kappa, dtype)
    assert p.status() == 0, what
    for b in range(P):
        want = ref_mppi.update(act[b], R[b].astype(np.float64), kappa)
        # (an fp32 planner reports the fp32 actions its rollouts used; the update sums the perturbations in fp64)
        tol = 1e-9 if dtype == "f64" else 1e-6
        assert np.abs(seq[b] - want).max() < tol * (1 + np.abs(want).max()), (what, b, np.abs(seq[b] - want).max())
        assert int(am[b]) == int(np.argmax(R[b])) and rmax[b] == R[b].max(), (what, b)
    p.close()
    # caller-supplied paths
    K2, HM = int(np.exp(rs.uniform(0, np.log(60000)))), int(rs.randint(1, 300))
    K2 = min(K2, 3_000_000 // HM)
    A, R2 = rs.randn(K2, HM), rs.randn(K2) * rs.choice([0.01, 1.0, 50.0])
    got = update_from_returns(A.reshape(K2, HM, 1), R2, kappa).reshape(HM)
    want = ref_mppi.update(A, R2, kappa)
    assert np.abs(got - want).max() < 1e-11 * (1 + np.abs(want).max()), (what, K2, HM)


@pytest.mark.parametrize("walk", range(max(WALKS, 4)))
def test_random_walk_over_the_environment(reacher_model, walk):
    """`Reacher7DOFEnv` (steps on the device, site positions computed when somebody looks) against the oracle's twin of
    the reference class through random sequences of reset_model / step / set_env_state / set_state / sim.forward /
    target_reset and reads of the observation, the state dict and `data.site_xpos` at random moments: 1e-10."""
    from oracle.ref_rollout import OracleReacherEnv
    from trajopt.envs.utils import get_environment
    rs = np.random.RandomState(31337 + walk)
    rt = str(rs.choice(["dense", "sparse", "tracking", "cooling"]))
    reference = np.cumsum(0.004 * rs.randn(400, 3), axis=0) + np.array([0.6, -0.3, 0.2]) if rt in ("tracking", "cooling") else None
    e = get_environment('reacher_7dof', reward_type=rt, reference=reference)
    o = OracleReacherEnv(reacher_model, reward_type=rt, reference=reference)
    e.reset_model(goal=(0.1, 0.0, 0.1), alpha=0.7)
    o.reset_model(goal=(0.1, 0.0, 0.1), alpha=0.7)
    trace = []

    def same():
        where = "%s: %s" % (rt, " | ".join(trace[-12:]))
        look = rs.randint(4)
        if look == 0:
            assert np.abs(e._get_obs() - o._get_obs()).max() < 1e-10, where
        elif look == 1:
            assert np.abs(e.data.site_xpos[e.hand_sid] - o.hand).max() < 1e-10 and np.abs(e.data.site_xpos[e.target_sid] - o.target).max() < 1e-12, where
        elif look == 2:
            se, so = e.get_env_state(), o.get_env_state()
            assert np.abs(se["qp"] - so["qp"]).max() < 1e-10 and np.abs(se["qv"] - so["qv"]).max() < 1e-10, where
            assert np.array_equal(se["target_pos"], so["target_pos"]) and se["timestep"] == so["timestep"], where
        # (look == 3: nobody looks -- the lazy site positions must still be right at the next look)

    for _ in range(60):
        op = rs.randint(8)
        if op <= 2:
            if e.env_timestep > 300:
                continue
            a = rs.randn(7) * rs.choice([0.3, 1.0, 3.0])
            ob, r, done, info = e.step(a)
            ob_o, r_o, _, info_o = o.step(a)
            trace.append("step")
            assert np.abs(ob - ob_o).max() < 1e-10 and abs(r - r_o) < 1e-10 * (1 + abs(r_o)) and done is False, (rt, trace[-12:])
            assert info["state"]["timestep"] == info_o["state"]["timestep"]
        elif op == 3:
            s = dict(qp=rs.uniform(-1, 1, 7) * np.array([1, 0.5, 1, 1, 1, 0.5, 1]), qv=rs.randn(7), target_pos=rs.uniform(-0.2, 0.2, 3),
                     timestep=int(rs.randint(0, 100)))
            e.set_env_state(s); o.set_env_state(s)
            trace.append("set_env_state")
        elif op == 4:
            g, al = rs.uniform(-0.2, 0.2, 3), float(rs.choice([1.0, 0.3]))
            ob, ob_o = e.reset_model(goal=g, alpha=al), o.reset_model(goal=g, alpha=al)
            trace.append("reset_model")
            assert np.abs(ob - ob_o).max() < 1e-10, (rt, trace[-12:])
        elif op == 5:
            q, v = rs.uniform(-1, 1, 7) * 0.5, rs.randn(7)
            e.set_state(q, v)
            o.qpos, o.qvel, o.hand = q.copy(), v.copy(), o.sim.fk_hand(q)
            trace.append("set_state")
        elif op == 6:
            e.sim.forward()
            o.hand = o.sim.fk_hand(o.qpos)
            trace.append("forward")
        else:
            g = rs.uniform(-0.2, 0.2, 3)
            e.target_reset(g)
            o.target = np.array(g)
            o.hand = o.sim.fk_hand(o.qpos)         # target_reset ends in sim.forward()
            trace.append("target_reset")
        same()


@pytest.mark.parametrize("case", range(max(WALKS, 4)))
def test_random_shardings_of_one_planner(case):
    """A planner's samples cut into shards of random sizes (global sample index = the shard's offset + the local one, so
    the Philox stream does not depend on the cut): every shard's returns are its slice of the unsharded planner's, and
    the merged statistics give the unsharded act_sequence, arg-max and maximum on every shard -- over several planning
    steps, with the real environment stepped on every shard alike.  Cuts at multiples of 64 keep every wavefront's
    samples together and reproduce the returns bit for bit; a ragged cut regroups samples into other wavefronts, and a
    sample's active-set solve runs as many sweeps as its slowest wave-mate needs (a repeated sweep solves the same set
    again through the other factorisation routine), so its returns then agree to rounding, not to the bit."""
    import torch
    from trajopt_amd.device import DeviceModel, DevicePlanner
    from trajopt_amd import sharding
    rs = np.random.RandomState(2718 + case)
    dm = DeviceModel.get("reacher_7dof")
    K, H = int(rs.randint(2, 3000)), int(rs.randint(2, 20))
    aligned = case % 2 == 0 and K > 128
    G = int(rs.randint(2, min(6, K // 64 if aligned else K) + 1))
    cuts = np.sort(rs.choice(np.arange(64, K, 64) if aligned else np.arange(1, K), size=G - 1, replace=False))
    lo, hi = np.r_[0, cuts], np.r_[cuts, K]
    dtype, kappa, seed = str(rs.choice(["f32", "f64"])), float(rs.choice([1.0, 25.0])), int(rs.randint(1, 10 ** 6))
    q0, v0, tgt = 0.3 * rs.randn(7), rs.randn(7), rs.uniform(-0.2, 0.2, 3)

    def make(k, off):
        p = DevicePlanner(dm, k, H, kappa=kappa, sigma=np.ones(7), betas=(0.25, 0.8, 0.0), dtype=dtype, default_act="mean", k_offset=off)
        p.set_start_state(q0, v0, tgt, 0)
        p.set_base_actions(np.zeros((H, 7)))
        return p
    full = make(K, 0)
    shards = [make(int(hi[g] - lo[g]), int(lo[g])) for g in range(G)]
    what = (case, K, H, list(map(int, cuts)), dtype)
    for step in range(3):
        full.generate_noise_step(seed); full.iterate(1)
        R, want = full.returns(), full.act_sequence()
        for s in shards:
            s.generate_noise_step(seed); s.local_partials(); s.synchronize()
        for g, s in enumerate(shards):
            if aligned:
                assert np.array_equal(s.returns(), R[lo[g]:hi[g]]), (what, step, g)
            else:
                d = np.abs(s.returns() - R[lo[g]:hi[g]]).max() / np.abs(R).max()
                assert d < (1e-10 if dtype == "f64" else 2e-4), (what, step, g, d)
        rows = np.stack([s.partials() for s in shards])
        gathered = torch.from_numpy(rows).cuda()
        torch.cuda.synchronize()
        host = sharding.merge_rows(rows, np.zeros((H, 7)) if step == 0 else base_before, kappa)
        for s in shards:
            s.merge(gathered.data_ptr(), G)
            tol = 1e-11 if aligned else (1e-8 if dtype == "f64" else 5e-2)      # (kappa (R - Rmax) in the exponent)
            assert np.abs(s.act_sequence() - want).max() < tol * (1 + np.abs(want).max()), (what, step)
            if aligned:
                assert s.argmax() == full.argmax() and s.rmax() == full.rmax(), (what, step)
        assert np.abs(host[0] - shards[0].act_sequence()).max() < 1e-11 * (1 + np.abs(want).max()), (what, step)
        full.advance()
        for s in shards:
            s.advance()
        base_before = full.act_sequence()
        for s in shards:                              # the shards drift apart from the full planner only in rounding
            assert np.abs(s.act_sequence() - base_before).max() < (1e-10 if aligned else (1e-7 if dtype == "f64" else 0.5)), (what, step)
            s.set_base_actions(base_before)           # ... which this test removes, to compare the next step exactly
            lf, ls = full.read_log(max_rows=1), s.read_log(max_rows=1)
            assert np.abs(ls["qpos"] - lf["qpos"]).max() < (1e-9 if aligned or dtype == "f64" else 1e-2), (what, step)
            s.set_start_state(lf["qpos"][0], lf["qvel"][0], tgt, step + 1)
    for p in [full] + shards:
        p.close()


@pytest.mark.parametrize("case", range(max(2 * WALKS, 8)))
def test_reference_boundaries_raise_where_the_reference_raises(reacher_model, case):
    """Tracking / cooling rewards index the reference trajectory with the time step after every env step
    (trajopt/envs/reacher_env.py:61,75): a rollout of H steps from time step t reads rows t + 1 .. t + H and raises
    IndexError when the trajectory is shorter; a missing reference raises TypeError.  Random (t, H, length) around the
    boundary through `do_env_rollout`, `gather_paths_parallel`, `env.step` and `MPPI.train_step`, against the oracle's
    twin of the reference code (which raises from numpy's own indexing)."""
    from oracle import ref_rollout
    from trajopt import utils
    from trajopt.algos.mppi import MPPI
    from trajopt.envs.utils import get_environment
    rs = np.random.RandomState(555 + case)
    rt = str(rs.choice(["tracking", "cooling"]))
    t0, H = int(rs.randint(0, 20)), int(rs.randint(1, 12))
    n = t0 + H + int(rs.randint(-2, 3))
    reference = (np.cumsum(0.01 * rs.randn(max(n, 1), 3), axis=0) + np.array([0.6, -0.3, 0.2]))[:max(n, 0)]
    s = dict(qp=0.2 * rs.randn(7), qv=rs.randn(7), target_pos=rs.uniform(-0.1, 0.1, 3), timestep=t0)
    acts = [0.5 * rs.randn(H, 7) for _ in range(3)]
    what = (case, rt, t0, H, n)

    def outcome(f):
        try:
            return ("ok", f())
        except (IndexError, TypeError) as ex:
            return (type(ex).__name__, None)
    want = outcome(lambda: ref_rollout.do_env_rollout(reacher_model, s, acts, None, rt, reference, 0.5))
    got = outcome(lambda: utils.do_env_rollout('reacher_7dof', s, acts, None, rt, reference, 0.5))
    assert got[0] == want[0] == ("ok" if n > t0 + H else "IndexError"), (what, got[0], want[0])
    if got[0] == "ok":
        for pw, pg in zip(want[1], got[1]):
            assert np.abs(pg["rewards"] - pw["rewards"]).max() < 2e-3 * (1 + np.abs(pw["rewards"]).max()), what      # fp32 rollouts
    got = outcome(lambda: utils.gather_paths_parallel('reacher_7dof', s, acts[0], FILTER, 7, None, rt, reference, 0.5, 4, 2))
    assert got[0] == want[0], (what, got[0])
    # no reference at all: indexing None
    assert outcome(lambda: utils.do_env_rollout('reacher_7dof', s, acts, None, rt, None, 0.5))[0] == "TypeError", what
    # one real step needs row t + 1 only; the constructor itself takes one from time step 0 (trajopt/envs/mujoco_env.py:53)
    if n <= 1:
        assert outcome(lambda: get_environment('reacher_7dof', reward_type=rt, reference=reference))[0] == "IndexError", what
        return
    e = get_environment('reacher_7dof', reward_type=rt, reference=reference)
    e.reset_model(goal=s["target_pos"], alpha=0.5)
    e.set_env_state(s)
    o = ref_rollout.OracleReacherEnv(reacher_model, reward_type=rt, reference=reference)
    o.reset_model(goal=s["target_pos"], alpha=0.5)
    o.set_env_state(s)
    a, b = outcome(lambda: e.step(acts[0][0])), outcome(lambda: o.step(acts[0][0]))
    assert a[0] == b[0] == ("ok" if n > t0 + 1 else "IndexError"), (what, a[0], b[0])
    if a[0] == "ok":
        assert abs(a[1][1] - b[1][1]) < 1e-10
    # the planner: rollouts from the agent's state need t + H, checked before anything is launched
    e.set_env_state(s)
    ag = MPPI(e, H=H, paths_per_cpu=8, kappa=25.0, filter_coefs=FILTER, default_act='mean', seed=3, reward_type=rt, reference=reference)
    before = len(ag.sol_state)
    got = outcome(lambda: ag.train_step(niter=1))
    assert got[0] == ("ok" if n > t0 + H else "IndexError"), (what, got[0])
    assert len(ag.sol_state) == before + (got[0] == "ok"), what


@pytest.mark.parametrize("case", range(max(WALKS, 4)))
def test_random_critics_against_torch(oracle_sim, case):
    """The terminal-value kernel (matrix cores, hardware exponentials) for random input slices, sample counts (ragged
    last blocks), discounts and weight scales from tiny to saturating, against torch's fp32 module on the oracle's final
    observations: within 2e-5 of the value range (fp64 rollouts, so only the network differs)."""
    import torch
    from trajopt.models.critic_nets import Critic
    from trajopt_amd.algos.mppi import _critic_weights
    from trajopt_amd.device import DeviceModel, DevicePlanner
    rs = np.random.RandomState(8080 + case)
    dim = int(rs.choice([14, 15, 17, 3, 6]))
    K, H, gamma = int(rs.randint(1, 1500)), int(rs.randint(1, 9)), float(rs.choice([1.0, 0.9]))
    torch.manual_seed(case)
    critic = Critic(input_dim=dim)
    scale = float(rs.choice([0.01, 1.0, 8.0]))          # 8: most hidden units saturate
    with torch.no_grad():
        critic.linear1.weight.mul_(scale); critic.linear2.weight.mul_(scale)
        critic.linear3.weight.mul_(float(rs.choice([1.0, 40.0])))
    goal, q0 = rs.uniform(-0.2, 0.2, 3), 0.4 * rs.randn(7) * np.array([1, 0.5, 1, 1, 1, 0.5, 1])
    raw = rs.randn(K, H, 7)
    p = DevicePlanner(DeviceModel.get("reacher_7dof"), K, H, kappa=5.0, gamma=gamma, sigma=FILTER[0], betas=FILTER[1:], dtype="f64")
    p.set_start_state(q0, np.zeros(7), goal, 0)
    p.set_base_actions(np.zeros((H, 7)))
    p.upload_noise(raw)
    p.set_critic(dim, _critic_weights(critic, dim))
    p.rollout()
    R = p.returns()
    _, nobs, rew = oracle_sim.rollout(q0, np.zeros(7), goal, 0, ref_mppi.filter_noise(raw.copy(), FILTER), gamma=gamma, dump=True)
    obs = torch.tensor(nobs[:, -1], dtype=torch.float32)
    x = {14: obs[:, :14], 15: torch.cat((obs[:, :14], torch.linalg.norm(obs[:, -6:-3] - obs[:, -3:], dim=1, keepdim=True)), 1),
         17: torch.cat((obs[:, :14], obs[:, -3:]), 1), 3: obs[:, -6:-3], 6: obs[:, -6:]}[dim]
    with torch.no_grad():
        V = critic(x).numpy().reshape(-1).astype(np.float64)
    want = (rew[:, :-1] * gamma ** np.arange(H - 1)).sum(1) + gamma ** (H - 1) * V
    span = max(np.abs(V).max(), 1.0)
    assert np.abs(R - want).max() < 2e-5 * span, (case, dim, K, H, scale, np.abs(R - want).max(), span)
    assert p.status() == 0
    p.close()


def test_planners_driven_from_several_host_threads():
    """Distinct planners may be driven from distinct host threads (ctypes releases the interpreter lock inside every
    call): four threads, each planning and reading back step by step on planners of its own -- device noise, host
    noise drawn by the library, different sizes, several planner lifetimes per thread, so that one thread allocates,
    fills and reads while another captures its graphs -- give what the same planners give one after the other.
    (The captures are made in relaxed mode: a capture in thread-local mode made every allocation-time fill and every
    synchronous read of the other threads fail, and spoiled the capture too.)"""
    import threading
    from trajopt_amd.device import DeviceModel, DevicePlanner
    from trajopt_amd.utils import worker_seeds
    dm = DeviceModel.get("reacher_7dof")
    specs = [(4096, 16, "f32", "philox"), (300, 8, "f64", "philox"), (512, 12, "f32", "host"), (64, 20, "f64", "host")]

    def run(spec, out, idx):
        K, H, dtype, noise = spec
        rows, status = [], 0
        for life in range(5):
            p = DevicePlanner(dm, K, H, kappa=25.0, sigma=FILTER[0], betas=FILTER[1:], mean=np.zeros(7), default_act="mean", dtype=dtype)
            p.set_start_state(np.full(7, 0.1 * idx + 0.02 * life), np.zeros(7), np.array([0.1, 0.0, 0.1]), 0)
            p.set_base_actions(np.zeros((H, 7)))
            for t in range(16):
                if noise == "philox":
                    p.plan(11 + idx, 1 + life % 2, 1 if t % 5 else 9)       # (a run of nine captures a second graph)
                else:
                    p.draw_noise(worker_seeds(1000 * idx + t, K // 2, 2), 2)
                    p.commit_noise(); p.iterate(1 + life % 2); p.advance()
                row, seq = p.read_step()
                rows.append(np.r_[row, seq.ravel(), p.returns()[:4], p.read_log(max_rows=1)["reward"]])
            status |= p.status()
            p.close()
        out[idx] = (np.array(rows), status)

    serial, threaded = {}, {}
    for i, spec in enumerate(specs):
        run(spec, serial, i)
    threads = [threading.Thread(target=run, args=(spec, threaded, i)) for i, spec in enumerate(specs)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(len(specs)):
        assert threaded[i][1] == serial[i][1] == 0
        assert np.array_equal(threaded[i][0], serial[i][0]), i


def test_agents_driven_from_several_host_threads():
    """Three `MPPI` agents (own environment, own planner; the compiled model handle is shared) trained step by step in
    three threads at once give the trajectories they give one after the other -- device noise, host noise, a critic."""
    import threading
    import torch
    from trajopt.algos.mppi import MPPI
    from trajopt.envs.utils import get_environment
    from trajopt.models.critic_nets import Critic
    torch.manual_seed(0)
    critic = Critic(input_dim=14)
    specs = [("philox", "f32", None, 512), ("host", "f64", None, 64), ("philox", "f64", critic, 128)]

    def run(spec, out, idx):
        noise, dtype, c, K = spec
        e = get_environment('reacher_7dof', reward_type='sparse' if idx == 2 else 'dense')
        e.reset_model(seed=idx, goal=(0.05 * idx, 0.0, 0.1))
        ag = MPPI(e, H=10, paths_per_cpu=K, kappa=25.0, filter_coefs=FILTER, default_act='mean', seed=3 + idx, noise=noise, dtype=dtype)
        for _ in range(40):
            ag.train_step(critic=c, niter=2)
        out[idx] = (np.array([s["qp"] for s in ag.sol_state]), np.array(ag.sol_reward), ag.act_sequence.copy())

    serial, threaded = {}, {}
    for i, spec in enumerate(specs):
        run(spec, serial, i)
    threads = [threading.Thread(target=run, args=(spec, threaded, i)) for i, spec in enumerate(specs)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(len(specs)):
        for a, b in zip(threaded[i], serial[i]):
            assert np.array_equal(a, b), i
