"""The reference's Python API on the HIP path: env duck-type, rollout helpers, MPPI.train_step.
Needs a GPU (`-m gpu`).  The oracle is only the checker."""
import os
import pickle

import numpy as np
import pytest

from oracle import ref_mppi

pytestmark = pytest.mark.gpu
FILTER = [np.ones(7), 0.25, 0.8, 0.0]


def test_env_duck_type_and_transitions(transitions, oracle_sim):
    from trajopt.envs.utils import get_environment
    e = get_environment('reacher_7dof')
    assert e.env_name == 'reacher_7dof' and e.action_dim == 7 and e.observation_dim == 26
    assert e.env_timestep == 1                      # the constructor takes one zero-action step
    # gym-style spaces and metadata (trajopt/envs/mujoco_env.py:45-65): control range of the seven motors, unbounded
    # 20-dim observations, 1 / dt frames per second
    assert e.action_space.shape == (7,) and e.action_space.dtype == np.float32 and e.observation_space.shape == (20,)
    assert np.array_equal(e.action_space.low, e.model.actuator_ctrlrange[:, 0].astype(np.float32))
    assert np.all(np.isinf(e.observation_space.high)) and e.action_space.contains(e.action_space.sample())
    assert e.metadata['video.frames_per_second'] == int(np.round(1.0 / e.dt)) and e.obs_dim == 20
    ob = e.reset_model(goal=(0.05, 0.0, -0.1))
    assert e.env_timestep == 0 and ob.shape == (20,)
    assert np.allclose(ob[14:17], [0.821, -0.6, 0.0], atol=1e-12) and np.allclose(ob[17:], [0.05, 0.0, -0.1])
    d = transitions
    for i in range(0, len(d["reward"]), 97):
        s = dict(qp=d["qp"][i], qv=d["qv"][i], target_pos=d["target"][i], timestep=int(d["timestep"][i]))
        e.set_env_state(s)
        assert np.allclose(e.data.site_xpos[e.hand_sid], oracle_sim.fk_hand(d["qp"][i]), atol=1e-13)   # forward()
        ob, r, done, info = e.step(d["act"][i])
        assert done is False and e.env_timestep == s["timestep"] + 1
        assert np.abs(ob[:17] - d["obs_next"][i][:17]).max() < 1e-10
        st = e.get_env_state()
        assert np.abs(st["qp"] - d["qp_next"][i]).max() < 1e-10 and np.array_equal(st["target_pos"], d["target"][i])
        assert info["state"]["timestep"] == s["timestep"] + 1
        assert e._get_obs(goal_dist=True).shape == (21,)
    e2 = pickle.loads(pickle.dumps(e))
    assert e2.env_name == 'reacher_7dof' and e2.env_timestep == 1


def test_site_positions_belong_to_the_forward_call(oracle_sim):
    """`sim.forward()` fixes `site_xpos` for the configuration it was called with (trajopt/envs/mujoco_env.py:120-128);
    here the kernel launch behind it is put off until somebody reads the positions, and must still answer for that
    configuration, not for whatever qpos holds by then."""
    from trajopt.envs.utils import get_environment
    e = get_environment('reacher_7dof')
    q1, q2 = np.linspace(-0.4, 0.3, 7), np.linspace(0.2, -0.5, 7)
    e.set_state(q1, np.zeros(7))
    e.data.qpos[:] = q2                                   # no forward() after this
    assert np.allclose(e.data.site_xpos[e.hand_sid], oracle_sim.fk_hand(q1), atol=1e-13)
    e.sim.forward()
    assert np.allclose(e.data.site_xpos[e.hand_sid], oracle_sim.fk_hand(q2), atol=1e-13)
    # a step leaves its own (one substep old) hand position; a forward() asked for before it is forgotten
    e.set_state(q1, np.zeros(7))
    ob, _, _, _ = e.step(np.zeros(7))
    assert np.array_equal(e.data.site_xpos[e.hand_sid], ob[14:17])
    # assigning the array is allowed, as with the simulator's own buffer
    e.data.site_xpos[e.target_sid] = (0.1, 0.2, 0.3)
    assert np.array_equal(e._get_obs()[17:20], (0.1, 0.2, 0.3))


def test_reward_types(oracle_sim):
    from trajopt.envs.utils import get_environment
    rs = np.random.RandomState(0)
    ref = rs.randn(40, 3) * 0.3
    s = dict(qp=rs.uniform(-1, 1, 7), qv=rs.randn(7), target_pos=np.array([0.1, 0.0, 0.1]), timestep=7)
    a = rs.randn(7) * 2
    for rt, alpha in (("dense", 1.0), ("sparse", 1.0), ("tracking", 1.0), ("cooling", 0.3)):
        e = get_environment('reacher_7dof', reward_type=rt, reference=ref)
        e.reset_model(goal=s["target_pos"], alpha=alpha)
        e.set_env_state(s)
        _, r, _, _ = e.step(a)
        want = oracle_sim.env_step(s["qp"], s["qv"], s["target_pos"], a, 7, rt, ref, alpha)[3]
        assert abs(r - want) < 1e-10, rt
    e = get_environment('reacher_7dof', reward_type='tracking', reference=ref[:8])
    e.set_env_state(s)
    with pytest.raises(IndexError):
        e.step(a)


def test_gather_paths_parallel_schema_and_values(oracle_sim, planner_vectors):
    from trajopt import utils
    utils.ROLLOUT_DTYPE = "f64"
    try:
        H, ppc, ncpu = 12, 5, 3
        base = np.random.RandomState(3).randn(H, 7) * 0.2
        s = dict(qp=np.full(7, 0.1), qv=np.zeros(7), target_pos=np.array([0.0, 0.1, 0.0]), timestep=4)
        paths = utils.gather_paths_parallel('reacher_7dof', s, base, FILTER, 500, None, 'dense', None, 1.0, ppc, ncpu)
    finally:
        utils.ROLLOUT_DTYPE = "f32"
    assert len(paths) == ppc * ncpu
    assert sorted(paths[0].keys()) == ['actions', 'env_infos', 'next_observations', 'observations', 'rewards', 'states']
    assert sorted(paths[0].keys()) == list(planner_vectors["g_keys"])      # the keys the reference's paths carry
    raw = ref_mppi.all_raw_normals(500, ppc, ncpu, H, 7)
    act = base + ref_mppi.filter_noise(raw, FILTER)
    R_o, nobs_o, rew_o = oracle_sim.rollout(s["qp"], s["qv"], s["target_pos"], 4, act, dump=True)
    for k, p in enumerate(paths):
        assert np.array_equal(p["actions"], act[k])           # sample order and worker streams
        assert p["observations"].shape == (H, 20) and p["rewards"].shape == (H,)
        assert np.abs(np.array(p["next_observations"]) - nobs_o[k]).max() < 1e-10
        assert np.abs(p["rewards"] - rew_o[k]).max() < 1e-10
        assert np.abs(p["observations"][1:] - nobs_o[k][:-1]).max() < 1e-10
        assert np.allclose(p["observations"][0][14:17], oracle_sim.fk_hand(s["qp"]), atol=1e-13)   # fresh at t = 0
        assert np.array_equal(p["env_infos"]["state"]["timestep"], 4 + np.arange(H))
        assert p["env_infos"]["state"]["qp"].shape == (H, 7) and len(p["states"]) == H
        assert p["states"][3]["timestep"] == 7 and np.array_equal(p["states"][3]["qp"], p["observations"][3][:7])
    # overlapping worker streams: worker 1 is seeded 5 later, its first sample differs from worker 0's
    assert not np.array_equal(paths[5]["actions"], paths[0]["actions"])


def test_try_multiprocess_runs_the_worker_argument_lists(oracle_sim):
    """`_try_multiprocess(args_list, num_cpu, max_process_time, max_timeouts)` (trajopt/utils.py:166-192): one
    result list per worker argument list, in worker order; None once the retries are used up (max_timeouts == 0).
    Here every worker's share is one batched launch in this process."""
    from trajopt import utils
    s = dict(qp=np.zeros(7), qv=np.zeros(7), target_pos=np.array([0.1, 0.0, 0.1]), timestep=2)
    base = np.zeros((5, 7))
    args = [['reacher_7dof', s, 3, base, FILTER, 40 + 3 * i, None, 'dense', None, 1.0] for i in range(2)]
    results = utils._try_multiprocess(args, 2, 300, 4)
    assert len(results) == 2 and all(len(r) == 3 for r in results)
    want = base + ref_mppi.filter_noise(ref_mppi.all_raw_normals(40, 3, 2, 5, 7), FILTER)
    got = np.array([p["actions"] for r in results for p in r])
    assert np.abs(got - want).max() < 1e-6          # f32 rollouts hand back the fp32 actions they used
    R_o = oracle_sim.rollout(s["qp"], s["qv"], s["target_pos"], 2, want)
    R = np.array([p["rewards"].sum() for r in results for p in r])
    assert np.abs(R - R_o).max() < 1e-4 * np.abs(R_o).max()
    assert utils._try_multiprocess(args, 2, 300, 0) is None
    # and gather_paths_parallel is the same samples in the same order
    paths = utils.gather_paths_parallel('reacher_7dof', s, base, FILTER, 40, None, 'dense', None, 1.0, 3, 2)
    assert np.array_equal(np.array([p["actions"] for p in paths]), got)


def test_do_env_rollout_and_generate_paths(oracle_sim):
    from trajopt import utils
    rs = np.random.RandomState(5)
    s = dict(qp=np.zeros(7), qv=np.zeros(7), target_pos=np.zeros(3), timestep=0)
    acts = [rs.randn(6, 7) for _ in range(3)]
    ref = rs.randn(10, 3)
    paths = utils.do_env_rollout('reacher_7dof', s, acts, None, 'tracking', ref, 1.0)
    R_o, nobs_o, rew_o = oracle_sim.rollout(s["qp"], s["qv"], s["target_pos"], 0, np.stack(acts), "tracking", ref, dump=True)
    for k in range(3):
        assert np.abs(paths[k]["rewards"] - rew_o[k]).max() < 1e-4 * (1 + np.abs(rew_o[k]).max())
        assert np.abs(paths[k]["actions"] - acts[k]).max() < 1e-6
    with pytest.raises(IndexError):
        utils.do_env_rollout('reacher_7dof', s, acts, None, 'tracking', ref[:6], 1.0)
    g = utils.generate_paths('reacher_7dof', s, 4, np.zeros((6, 7)), FILTER, 42, None, 'dense', None, 1.0)
    want = ref_mppi.filter_noise(ref_mppi.raw_normals(42, 4, 6, 7), FILTER)
    assert np.abs(np.array([p["actions"] for p in g]) - want).max() < 1e-6


def _oracle_train_steps(oracle_sim, steps, niter, H, ppc, ncpu, seed, kappa, gamma, goal, mean, default_act):
    """The reference's train_step loop restated on the oracle (trajopt/algos/mppi.py:125-196)."""
    q, v, t = np.zeros(7), np.zeros(7), 0
    seq = np.ones((H, 7)) * mean
    log = []
    for _ in range(steps):
        raw = ref_mppi.all_raw_normals(seed + t, ppc, ncpu, H, 7)
        eps = ref_mppi.filter_noise(raw, FILTER)
        for _ in range(niter):
            R = oracle_sim.rollout(q, v, goal, t, seq + eps, gamma=gamma)
            seq = ref_mppi.update(seq + eps, R, kappa)
        q, v, obs, r, _ = oracle_sim.env_step(q, v, goal, seq[0], t)
        t += 1
        seq = ref_mppi.shift(seq, default_act, mean)
        log.append((q.copy(), v.copy(), r, seq.copy()))
    return log


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_train_step_matches_oracle_loop(oracle_sim, dtype):
    from trajopt.algos.mppi import MPPI
    from trajopt.envs.utils import get_environment
    goal = np.array([0.05, -0.05, 0.1])
    e = get_environment('reacher_7dof')
    e.reset_model(seed=3, goal=goal)
    H, ppc, ncpu, niter, steps = 16, 20, 2, 2, 12
    ag = MPPI(e, H=H, paths_per_cpu=ppc, num_cpu=ncpu, kappa=25.0, gamma=0.98, mean=np.zeros(7),
              filter_coefs=FILTER, default_act='mean', seed=11, dtype=dtype)
    for _ in range(steps):
        assert ag.train_step(niter=niter, goal=goal) == []
    log = _oracle_train_steps(oracle_sim, steps, niter, H, ppc, ncpu, 11, 25.0, 0.98, goal, np.zeros(7), 'mean')
    tol = 1e-7 if dtype == "f64" else 5e-2       # exp(25 dR) amplifies fp32 return error over 24 updates
    for k in range(steps):
        assert np.abs(ag.sol_state[k + 1]["qp"] - log[k][0]).max() < tol
        assert abs(ag.sol_reward[k] - log[k][2]) < tol * 10
    assert np.abs(ag.act_sequence - log[-1][3]).max() < tol * 20
    assert len(ag.sol_act) == steps and np.all(np.array(ag.sol_act)[:10] == 0) and np.any(ag.sol_act[11] != 0)
    assert ag.sol_obs[-1].shape == (21,) and ag.sol_state[-1]["timestep"] == steps
    ag2 = pickle.loads(pickle.dumps(ag))             # picklable with a live device planner
    ag2.train_step(niter=1, goal=goal)
    assert len(ag2.sol_state) == steps + 2


def test_train_step_follows_what_the_caller_changes_between_steps(oracle_sim):
    """`train_step` keeps start state and act_sequence on the device from one call to the next.  A caller who writes to
    `act_sequence`, to `sol_state[-1]` or to the environment between two calls must get the step the reference would
    take from what the attributes hold then: every step is checked against the oracle restarted from the agent's own
    attributes as they were when the step was asked for."""
    from trajopt.algos.mppi import MPPI
    from trajopt.envs.utils import get_environment
    goal = np.array([0.05, -0.05, 0.1])
    e = get_environment('reacher_7dof')
    e.reset_model(seed=3, goal=goal)
    H, K, niter, seed = 12, 48, 2, 21
    ag = MPPI(e, H=H, paths_per_cpu=K, kappa=25.0, gamma=0.98, mean=np.zeros(7), filter_coefs=FILTER, default_act='mean',
              seed=seed, dtype="f64")
    rs = np.random.RandomState(5)
    for step in range(14):
        if step == 3:
            ag.act_sequence[2:5] += 0.3 * rs.randn(3, 7)              # in place
        if step == 5:
            ag.act_sequence = 0.1 * rs.randn(H, 7)                    # rebound
        if step == 7:
            ag.sol_state[-1]['qp'][:] = ag.sol_state[-1]['qp'] + 0.05 * rs.randn(7)      # in place
        if step == 9:
            ag.sol_state[-1] = dict(qp=0.2 * rs.randn(7), qv=rs.randn(7), target_pos=goal.copy(), timestep=ag.sol_state[-1]['timestep'])
        if step == 11:
            e.set_env_state(dict(qp=np.zeros(7), qv=np.ones(7), target_pos=goal.copy(), timestep=40))     # advance_time re-seats the env
            e.step(np.ones(7))
        s0, seq = {k: np.copy(v) for k, v in ag.sol_state[-1].items()}, ag.act_sequence.copy()
        t = len(ag.sol_state) - 1
        ag.train_step(niter=niter, goal=goal)
        eps = ref_mppi.filter_noise(ref_mppi.all_raw_normals(seed + t, K, 1, H, 7), FILTER)
        for _ in range(niter):
            R = oracle_sim.rollout(s0["qp"], s0["qv"], goal, int(s0["timestep"]), seq + eps, gamma=0.98)
            seq = ref_mppi.update(seq + eps, R, 25.0)
        q, v, obs, r, _ = oracle_sim.env_step(s0["qp"], s0["qv"], goal, seq[0], int(s0["timestep"]))
        assert np.abs(ag.sol_state[-1]["qp"] - q).max() < 1e-8 and np.abs(ag.sol_state[-1]["qv"] - v).max() < 1e-8, step
        assert abs(ag.sol_reward[-1] - r) < 1e-8 and ag.sol_state[-1]["timestep"] == s0["timestep"] + 1
        assert np.abs(ag.act_sequence - ref_mppi.shift(seq, "mean", np.zeros(7))).max() < 1e-6, step
        # the environment is left where `set_env_state(sol_state[-2]); step(act_sequence[0])` leaves it
        st = e.get_env_state()
        assert np.array_equal(st["qp"], ag.sol_state[-1]["qp"]) and st["timestep"] == ag.sol_state[-1]["timestep"]
        assert np.abs(ag.sol_obs[-1][:20] - np.r_[q, v, obs[14:17], goal]).max() < 1e-8
        assert abs(ag.sol_obs[-1][20] - np.linalg.norm(obs[14:17] - goal)) < 1e-8 and np.abs(e.data.ctrl - seq[0]).max() < 1e-6
    assert len(ag.sol_act) == 14 and np.all(np.array(ag.sol_act)[:10] == 0) and np.any(ag.sol_act[12] != 0)


@pytest.mark.parametrize("noise,rewards", [("philox", ("dense", "dense")), ("host", ("dense", "dense")),
                                           ("philox", ("sparse", "cooling")), ("host", ("tracking", "cooling"))])
def test_resident_train_step_equals_the_step_by_step_one(noise, rewards):
    """The device-resident step (state, act_sequence and the real environment's step stay on the device) against the
    same planner driven piece by piece with everything sent again at every call (`advance_time` on the host side, which
    an `act_sequence=` argument asks for): same trajectory, bit for bit in fp64."""
    from trajopt.algos.mppi import MPPI
    from trajopt.envs.utils import get_environment
    goal = (0.05, 0.0, 0.05)
    # the environment's own reward need not be the one the rollouts are scored with (the actor-critic scripts log the
    # sparse reward of rollouts scored with 'cooling'): the device's step reports the environment's
    env_reward, rollout_reward = rewards
    ref = np.array([[0.2 - 0.01 * t, 0.1, 0.3 - 0.01 * t] for t in range(40)])
    logs = []
    for resident in (True, False):
        e = get_environment('reacher_7dof', reward_type=env_reward, reference=ref if env_reward == "tracking" else None)
        e.reset_model(seed=2, goal=goal, alpha=0.7)
        ag = MPPI(e, H=10, paths_per_cpu=64, kappa=25.0, filter_coefs=FILTER, default_act='repeat', seed=4, noise=noise,
                  dtype="f64", reward_type=rollout_reward, reference=ref if rollout_reward != "dense" else None)
        assert ag._real_env_steps_on_the_device()
        for step in range(6):
            if resident:
                ag.train_step(niter=2)
            else:
                # the piecewise path: iterations on the device, then advance_time through the environment object
                ag._plan_on_device(ag.seed + len(ag.sol_state) - 1, 2)
                ag.advance_time()
        logs.append((np.array([s["qp"] for s in ag.sol_state]), np.array(ag.sol_reward), ag.act_sequence.copy(),
                     np.array(ag.sol_obs[1:]), np.array(ag.sol_act)))
    for a, b in zip(*logs):
        assert np.array_equal(a, b)


def test_update_with_paths_and_critic():
    import torch
    from trajopt.algos.mppi import MPPI
    from trajopt.envs.utils import get_environment
    from trajopt.models.critic_nets import Critic
    torch.manual_seed(0)
    e = get_environment('reacher_7dof')
    e.reset_model(goal=(0, 0, 0))
    ag = MPPI(e, H=8, paths_per_cpu=16, kappa=5.0, gamma=0.9, filter_coefs=FILTER, default_act='mean', seed=5)
    paths = ag.do_rollouts(5)
    ag.update(paths)
    R = ref_mppi.score(np.array([p["rewards"] for p in paths]), 0.9)
    want = ref_mppi.update(np.array([p["actions"] for p in paths]), R, 5.0)
    assert np.abs(ag.act_sequence - want).max() < 1e-12
    for dim in (14, 15, 17, 3, 6):
        critic = Critic(input_dim=dim)
        before = len(ag.sol_state)
        ag.train_step(critic=critic, niter=1, dim=dim)
        assert len(ag.sol_state) == before + 1


def _example_agent(seed, dtype):
    """The agent of the reference's example script (sandbox/examples/reacher_7dof_mppi.py:11-44, dense reward)."""
    from trajopt.algos.mppi import MPPI
    from trajopt.envs.utils import get_environment
    e = get_environment('reacher_7dof')
    e.reset_model(seed=3, goal=(0, 0, 0), alpha=1.0)
    return MPPI(e, H=16, paths_per_cpu=40, num_cpu=1, kappa=25.0, gamma=1.0, mean=np.zeros(7), filter_coefs=FILTER,
                default_act='mean', seed=seed, dtype=dtype)


@pytest.fixture(scope="module")
def closed_loop():
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "closed_loop_oracle.npz"))
    return {k: d[k] for k in d.files}


def test_fp64_planning_against_the_oracle_loop_for_100_steps(closed_loop, oracle_sim):
    """Acceptance, exact form: the example script's loop (H=16, K=40, niter=5, 100 steps) in fp64 against the same
    loop on the CPU oracle (tests/golden/make_closed_loop_oracle.py).  The closed loop is chaotic -- five updates
    per step with weights exp(25 dR) on 40 samples multiply a rounding difference by 10 to 100 per step, so two
    fp64 implementations with different operation order agree to 1e-9 for about ten steps and are different
    trajectories after twenty.  Hence two checks: the free-running trajectory for the steps it can be compared, and
    EVERY one of the 100 steps with the oracle restarted from the device run's own state and act_sequence."""
    g = closed_loop
    goal = np.zeros(3)
    for i, seed in enumerate(g["seeds"][:2]):
        ag = _example_agent(int(seed), "f64")
        worst_q = worst_seq = 0.0
        for t in range(100):
            s0, seq = ag.sol_state[-1], ag.act_sequence.copy()
            ag.train_step(niter=5, goal=(0, 0, 0))
            eps = ref_mppi.filter_noise(ref_mppi.all_raw_normals(int(seed) + t, 40, 1, 16, 7), FILTER)
            for _ in range(5):
                R = oracle_sim.rollout(s0["qp"], s0["qv"], goal, t, seq + eps, gamma=1.0)
                seq = ref_mppi.update(seq + eps, R, 25.0)
            q, v, _, r, _ = oracle_sim.env_step(s0["qp"], s0["qv"], goal, seq[0], t)
            worst_q = max(worst_q, np.abs(ag.sol_state[-1]["qp"] - q).max(), np.abs(ag.sol_state[-1]["qv"] - v).max(),
                          abs(ag.sol_reward[-1] - r))
            worst_seq = max(worst_seq, np.abs(ag.act_sequence - ref_mppi.shift(seq, "mean", np.zeros(7))).max())
        assert worst_q < 1e-8 and worst_seq < 1e-6, (int(seed), worst_q, worst_seq)
        qp = np.array([s["qp"] for s in ag.sol_state[1:]])
        err = np.abs(qp - g["qpos"][i]).max(axis=1)
        assert err[:8].max() < 1e-8, (int(seed), err[:12])
        # free running it stays a run of the same planner: inside the oracle's seed-to-seed spread
        assert g["spread_totals"].min() - g["spread_totals"].std() < np.sum(ag.sol_reward) < g["spread_totals"].max() + g["spread_totals"].std()


def test_fp32_planning_stays_inside_the_oracle_seed_spread(closed_loop):
    """Acceptance, statistical form.  With 40 samples per step the closed loop is chaotic in the noise seed: the fp64
    ORACLE loop itself collects between -354 and -264 over 16 seeds (mean -293, std 25; the reference's stored
    agent: -274.76, 0.018 m from the target at the end).  An fp32 run decorrelates from the fp64 one after a few
    dozen steps, so it is held to that distribution, not to a trajectory: every total within one standard
    deviation of the oracle's range, the four-seed mean within one standard deviation of the oracle's mean."""
    g = closed_loop
    lo, hi = g["spread_totals"].min(), g["spread_totals"].max()
    mean, std = g["spread_totals"].mean(), g["spread_totals"].std()
    totals = []
    for seed in g["seeds"]:
        ag = _example_agent(int(seed), "f32")
        for _ in range(100):
            ag.train_step(niter=5, goal=(0, 0, 0))
        totals.append(float(np.sum(ag.sol_reward)))
        assert float(ag.sol_obs[-1][-1]) < g["spread_final_dist"].max() + 0.02, (int(seed), float(ag.sol_obs[-1][-1]))
        assert lo - std < totals[-1] < hi + std, (int(seed), totals, (lo, hi, std))
    assert abs(np.mean(totals) - mean) < std, (totals, mean, std)
    # the first steps are still the oracle's trajectory (fp32 tolerance)
    qp = np.array([s["qp"] for s in ag.sol_state[1:6]])
    assert np.abs(qp - g["qpos"][-1][:5]).max() < 1e-3


def test_continual_env_resets_target_every_50_real_steps():
    from trajopt.envs.utils import get_environment
    e = get_environment('continual_reacher_7dof')
    assert e.env_name == 'continual_reacher_7dof'
    e.reset_model(seed=5)
    first = e.get_env_state()['target_pos'].copy()
    for _ in range(49):
        e.step(np.zeros(7))
    assert np.array_equal(e.get_env_state()['target_pos'], first)
    ob, _, _, infos = e.step(np.zeros(7))                 # 50th real step draws a new target
    assert not np.array_equal(e.get_env_state()['target_pos'], first)
    # the observation predates the redraw, the infos carry the new target (continual_reacher_env.py:15-23)
    assert np.array_equal(ob[-3:], first) and np.array_equal(infos['state']['target_pos'], e.get_env_state()['target_pos'])
    e.real_step = False                                    # rollout envs never move the target
    t = e.get_env_state()['target_pos'].copy()
    for _ in range(50):
        e.step(np.zeros(7))
    assert np.array_equal(e.get_env_state()['target_pos'], t)


def test_continual_env_plans_on_the_dense_reward_whatever_reward_type_says():
    """`get_environment('continual_reacher_7dof')` builds the env without arguments and its step ignores
    reward_type (trajopt/envs/utils.py:11-12): rollouts of an MPPI(reward_type='sparse') on it are dense."""
    from trajopt.algos.mppi import MPPI
    from trajopt.envs.utils import get_environment
    from trajopt.utils import gather_paths_parallel
    seqs = []
    for rt in ('dense', 'sparse'):
        e = get_environment('continual_reacher_7dof')
        e.reset_model(goal=(0.05, 0.0, 0.1))
        ag = MPPI(e, H=8, paths_per_cpu=32, kappa=25.0, filter_coefs=[np.ones(7), 0.25, 0.8, 0.0], default_act='mean',
                  seed=3, reward_type=rt)
        ag.train_step(niter=2)
        seqs.append(ag.act_sequence.copy())
        paths = gather_paths_parallel(e.env_name, ag.sol_state[-1], ag.act_sequence, ag.filter_coefs, 3, None, rt, None,
                                      1.0, 16, 1)
        assert not np.isin(paths[0]['rewards'], (-10.0, -100.0, 100.0)).any()
    assert np.array_equal(seqs[0], seqs[1])


def test_worker_streams_leave_the_global_numpy_stream_alone_with_several_workers():
    """num_cpu == 1 re-seeds the process-global stream (the reference runs that worker in-process,
    trajopt/utils.py:171-172); with several workers only pool children are re-seeded there."""
    from trajopt.utils import all_worker_normals
    np.random.seed(77)
    want = np.random.RandomState(77).uniform(size=3)
    a = all_worker_normals(5, 4, 3, 6, 7)
    assert np.array_equal(np.random.uniform(size=3), want)
    b = all_worker_normals(5, 12, 1, 6, 7)
    assert not np.array_equal(np.random.uniform(size=3), want)
    # overlapping streams: worker i starts at seed + i * paths_per_cpu, same legacy generator either way
    assert np.array_equal(a[:4], b[:4]) and np.array_equal(a[4:8], all_worker_normals(9, 4, 1, 6, 7))


@pytest.mark.parametrize("num_cpu", [1, 3])
def test_host_noise_drawn_ahead_is_invisible(num_cpu):
    """`train_step` draws the host noise of the next planning step while the device works on the current one.  A caller
    cannot tell: the same act_sequence and the same process-global numpy generator after every step as with the
    draws made on demand, also when the seed is changed between two steps (what was drawn ahead is then dropped)."""
    from trajopt.algos.mppi import MPPI
    from trajopt.envs.utils import get_environment
    runs = []
    for ahead in (True, False):
        e = get_environment('reacher_7dof')
        e.reset_model(seed=2, goal=(0.05, 0.0, 0.05))
        ag = MPPI(e, H=8, paths_per_cpu=24, num_cpu=num_cpu, kappa=25.0, filter_coefs=FILTER, default_act='mean', seed=11)
        ag.draw_ahead = ahead
        np.random.seed(1)
        trace = []
        for step in range(7):
            if step == 3:
                ag.seed = 500                   # the stream drawn ahead for step 3 is no longer the one asked for
            if step == 5:
                ag.kappa = 10.0                 # a parameter that leaves the stream alone
            ag.train_step(niter=2)
            trace.append((ag.act_sequence.copy(), np.random.get_state()[1].copy(), np.random.get_state()[2], np.random.uniform()))
        runs.append(trace)
    for (a1, k1, p1, u1), (a2, k2, p2, u2) in zip(*runs):
        assert np.array_equal(a1, a2) and np.array_equal(k1, k2) and p1 == p2 and u1 == u2


def test_pickled_agent_resolves_through_the_trajopt_names(tmp_path):
    """An agent pickled under the reference's module names loads back onto this implementation."""
    import pickletools
    from trajopt.algos.mppi import MPPI
    from trajopt.envs.utils import get_environment
    e = get_environment('reacher_7dof')
    e.reset_model(goal=(0, 0, 0))
    ag = MPPI(e, H=8, paths_per_cpu=16, kappa=25.0, filter_coefs=FILTER, default_act='mean', seed=2)
    ag.train_step(niter=1)
    blob = pickle.dumps(ag, protocol=3)
    blob = blob.replace(b"trajopt_amd.algos.mppi", b"trajopt.algos.mppi").replace(b"trajopt_amd.envs.reacher_env", b"trajopt.envs.reacher_env")
    names = {arg for op, arg, _ in pickletools.genops(blob) if op.name == "GLOBAL"}
    assert "trajopt.algos.mppi MPPI" in names and "trajopt.envs.reacher_env Reacher7DOFEnv" in names
    ag2 = pickle.loads(blob)
    assert type(ag2).__name__ == "MPPI" and len(ag2.sol_state) == 2
    ag2.env.set_env_state(ag2.sol_state[1])
    assert np.allclose(ag2.env.data.site_xpos[ag2.env.hand_sid], ag2.env._dev.forward_hand(ag2.sol_state[1]['qp']))
    ag2.train_step(niter=1)
    assert len(ag2.sol_state) == 3


@pytest.mark.parametrize("dim", [14, 15, 17, 3, 6])
def test_device_critic_equals_host_critic(oracle_sim, dim):
    """train_step(critic=Critic) scores the last step with the critic on the device; an equivalent
    critic that is not a `Critic` instance takes the reference's route (paths back to the host, torch
    forward, update).  Same noise, same result; and the returns are the oracle's plus the torch value."""
    import torch
    from trajopt.algos.mppi import MPPI
    from trajopt.envs.utils import get_environment
    from trajopt.models.critic_nets import Critic
    from trajopt_amd.device import DeviceModel, DevicePlanner
    torch.manual_seed(dim)
    critic = Critic(input_dim=dim)
    with torch.no_grad():
        critic.linear3.weight.mul_(40.0)                   # make the terminal value matter
    host_only = lambda x: critic(x)                         # a plain callable: no device kernel for it
    goal = np.array([0.05, 0.0, 0.1])
    agents = []
    for c in (critic, host_only):
        e = get_environment('reacher_7dof')
        e.reset_model(goal=goal)
        ag = MPPI(e, H=8, paths_per_cpu=32, num_cpu=2, kappa=5.0, gamma=0.9, filter_coefs=FILTER, default_act='mean', seed=4)
        ag.train_step(critic=c, niter=2, dim=dim, goal=goal)
        agents.append(ag)
    assert np.abs(agents[0].act_sequence - agents[1].act_sequence).max() < 5e-4
    assert np.abs(agents[0].sol_state[1]['qp'] - agents[1].sol_state[1]['qp']).max() < 1e-4
    # returns: oracle rewards for t < H-1 plus gamma^(H-1) * critic(final observation)
    K, H = 64, 8
    raw = ref_mppi.all_raw_normals(4, 32, 2, H, 7)
    eps = ref_mppi.filter_noise(raw.copy(), FILTER)
    p = DevicePlanner(DeviceModel.get("reacher_7dof"), K, H, kappa=5.0, gamma=0.9, sigma=FILTER[0], betas=FILTER[1:])
    p.set_start_state(np.zeros(7), np.zeros(7), goal, 0)
    p.set_base_actions(np.zeros((H, 7)))
    p.upload_noise(raw)
    from trajopt_amd.algos.mppi import _critic_weights
    p.set_critic(dim, _critic_weights(critic, dim))
    p.rollout()
    R = p.returns()
    _, nobs, rew = oracle_sim.rollout(np.zeros(7), np.zeros(7), goal, 0, eps, gamma=0.9, dump=True)
    obs = torch.tensor(nobs[:, -1], dtype=torch.float32)
    x = {14: obs[:, :14], 15: torch.cat((obs[:, :14], torch.linalg.norm(obs[:, -6:-3] - obs[:, -3:], dim=1, keepdim=True)), 1),
         17: torch.cat((obs[:, :14], obs[:, -3:]), 1), 3: obs[:, -6:-3], 6: obs[:, -6:]}[dim]
    V = critic(x).detach().numpy().reshape(-1).astype(np.float64)
    want = (rew[:, :-1] * 0.9 ** np.arange(H - 1)).sum(1) + 0.9 ** (H - 1) * V
    assert np.abs(R - want).max() < 1e-3 * (1 + np.abs(want).max())
    p.set_critic(0)
    p.rollout()
    assert np.abs(p.returns() - (rew * 0.9 ** np.arange(H)).sum(1)).max() < 1e-3


@pytest.mark.parametrize("dim", [6, 14, 17])
@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_device_critic_with_the_reference_trained_weights(oracle_sim, dim, dtype):
    """The matrix-core critic kernel with weights the reference trained and stored (tests/golden/critic_vectors.npz,
    whose values pin this package's torch `Critic` to the reference's): terminal values of K = 1000 rollouts (a
    ragged last block) against torch on the oracle's final observations.  fp32 network, tolerance 1e-4 of the
    value range (rollout fp32 error included for f32; 2e-5 for fp64 rollouts, where only the network differs)."""
    import torch
    from trajopt.models.critic_nets import Critic
    from trajopt_amd.algos.mppi import _critic_weights
    from trajopt_amd.device import DeviceModel, DevicePlanner
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "critic_vectors.npz"))
    critic = Critic(input_dim=dim)
    critic.load_state_dict({"linear1.weight": torch.tensor(g["w%d_W1" % dim]), "linear1.bias": torch.tensor(g["w%d_b1" % dim]),
                            "linear2.weight": torch.tensor(g["w%d_W2" % dim]), "linear2.bias": torch.tensor(g["w%d_b2" % dim]),
                            "linear3.weight": torch.tensor(g["w%d_W3" % dim]), "linear3.bias": torch.tensor(g["w%d_b3" % dim])})
    K, H, gamma = 1000, 6, 0.95
    goal = np.array([0.05, -0.05, 0.1])
    q0 = np.array([0.2, 0.4, -0.3, -0.9, 0.2, -0.4, 0.1])
    raw = ref_mppi.raw_normals(6, K, H, 7)
    eps = ref_mppi.filter_noise(raw.copy(), FILTER)
    p = DevicePlanner(DeviceModel.get("reacher_7dof"), K, H, kappa=5.0, gamma=gamma, sigma=FILTER[0], betas=FILTER[1:], dtype=dtype)
    p.set_start_state(q0, np.zeros(7), goal, 0)
    p.set_base_actions(np.zeros((H, 7)))
    p.upload_noise(raw)
    p.set_critic(dim, _critic_weights(critic, dim))
    p.rollout()
    R = p.returns()
    _, nobs, rew = oracle_sim.rollout(q0, np.zeros(7), goal, 0, eps, gamma=gamma, dump=True)
    obs = torch.tensor(nobs[:, -1], dtype=torch.float32)
    x = {14: obs[:, :14], 17: torch.cat((obs[:, :14], obs[:, -3:]), 1), 6: obs[:, -6:]}[dim]
    with torch.no_grad():
        V = critic(x).numpy().reshape(-1).astype(np.float64)
    want = (rew[:, :-1] * gamma ** np.arange(H - 1)).sum(1) + gamma ** (H - 1) * V
    scale = np.abs(V).max()
    assert scale > 10.0                              # trained critics: values of the order of the returns
    assert np.abs(R - want).max() < (1e-4 if dtype == "f32" else 2e-5) * scale, np.abs(R - want).max()
    p.close()


def test_example_script_flow():
    """examples/reacher_goals.py: random goals, 116 steps, pickle round trip, sparse replay."""
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "reacher_goals.py")
    spec = importlib.util.spec_from_file_location("reacher_goals", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    results = mod.main(["--steps", "116", "--seeds", "10", "13"])
    for ret, hit, sparse, dt in results:
        assert ret > -400.0 and sparse > -116 * 100.0
    assert sum(r[1] for r in results) >= 1           # the hand gets within 2.5 cm of at least one of the goals


def test_stored_agent_loads_without_unpickling_and_replays(tmp_path):
    """play_mppi_agent flow (sandbox/examples/play_mppi_agent.py:22-31) on a stored agent read by the
    non-executing loader: logged states and actions step to the logged next states, and the run continues."""
    import pickle
    from trajopt_amd.agent_io import load_agent
    from trajopt_amd.algos.mppi import MPPI
    from trajopt_amd.envs.utils import get_environment
    e = get_environment('reacher_7dof', reward_type='dense', reference=None)
    e.reset_model(seed=3, goal=None)
    agent = MPPI(e, H=8, paths_per_cpu=32, num_cpu=1, kappa=25.0, gamma=1.0, mean=np.zeros(7),
                 filter_coefs=[np.ones(7), 0.25, 0.8, 0.0], default_act='mean', seed=5)
    for _ in range(14):
        agent.train_step(niter=2)
    path = tmp_path / "agent.pickle"
    with open(path, "wb") as f:
        pickle.dump(agent, f)
    stored = load_agent(str(path))
    assert stored.H == 8 and stored.kappa == 25.0 and len(stored.sol_state) == 15 and len(stored.sol_act) == 14
    assert np.array_equal(stored.act_sequence, agent.act_sequence)
    env = get_environment('reacher_7dof', reward_type='dense', reference=None)
    env.reset_model(goal=(0, 0, 0), alpha=0.0)
    for k in range(10, 14):                  # the first ten logged actions are zeros by the reference's rule
        env.set_env_state(stored.sol_state[k])
        _, r, _, _ = env.step(stored.sol_act[k])
        nxt = env.get_env_state()
        assert np.array_equal(nxt['qp'], stored.sol_state[k + 1]['qp']) and r == stored.sol_reward[k]
    a2, b2 = stored, agent
    a2.train_step(niter=2); b2.train_step(niter=2)
    assert np.array_equal(a2.act_sequence, b2.act_sequence) and np.array_equal(a2.sol_state[-1]['qp'], b2.sol_state[-1]['qp'])


def test_inputs_may_be_device_memory(oracle_sim):
    """Noise, actions, base actions, start state and reference handed over as addresses of torch tensors on the GPU
    (`tensor.data_ptr()`) give what the same values give as numpy arrays."""
    import ctypes as C
    import torch
    from trajopt_amd import _lib as L
    from trajopt_amd.device import DeviceModel, DevicePlanner
    lib = L.lib()
    K, H = 192, 9
    rs = np.random.RandomState(3)
    raw, base = rs.randn(K, H, 7), 0.1 * rs.randn(H, 7)
    q0, v0, tgt = 0.3 * rs.randn(7), rs.randn(7), np.array([0.1, -0.1, 0.05])
    ref = 0.3 * rs.randn(H + 4, 3)
    res = []
    for on_device in (False, True):
        p = DevicePlanner(DeviceModel.get("reacher_7dof"), K, H, kappa=25.0, sigma=FILTER[0], betas=FILTER[1:],
                          default_act="mean", reward_type="tracking", dtype="f64")
        p.set_params(25.0, 0.97, "tracking")
        if on_device:
            dev = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")
            t_raw, t_base, t_q, t_v, t_t, t_ref = dev(raw), dev(base), dev(q0), dev(v0), dev(tgt), dev(ref)
            t_ts = dev(np.array([2]), torch.int32)
            vp = lambda t: C.c_void_p(t.data_ptr())
            L.check(lib.mppi_set_start_state(p._h, vp(t_q), vp(t_v), vp(t_t), vp(t_ts)))
            L.check(lib.mppi_set_reference(p._h, vp(t_ref), len(ref), 1.0))
            L.check(lib.mppi_set_base_actions(p._h, vp(t_base)))
            L.check(lib.mppi_upload_noise(p._h, vp(t_raw), None))
        else:
            p.set_start_state(q0, v0, tgt, 2)
            p.set_reference(ref, 1.0)
            p.set_base_actions(base)
            p.upload_noise(raw)
        p.iterate(2)
        out = [p.returns(), p.act_sequence()]
        acts = base + ref_mppi.filter_noise(raw.copy(), FILTER)
        if on_device:
            t_act = dev(acts)
            L.check(lib.mppi_upload_actions(p._h, vp(t_act), None))
        else:
            p.upload_actions(acts)
        p.rollout()
        out.append(p.returns())
        if on_device:
            L.check(lib.mppi_shift(p._h, vp(t_base), None))
        else:
            p.shift(base)
        out.append(p.act_sequence())
        res.append(out)
        p.close()
    for a, b in zip(*res):
        assert np.array_equal(a, b)
    R_o = oracle_sim.rollout(q0, v0, tgt, 2, acts, "tracking", ref, 1.0, gamma=0.97)
    assert np.abs(res[1][2] - R_o).max() < 1e-8


def test_environments_step_from_several_threads(oracle_sim):
    """Environments of one process share the compiled model handle; two host threads stepping an environment each
    (ctypes releases the interpreter lock during the call) take turns inside the library and get their own results."""
    import threading
    from trajopt.envs.utils import get_environment
    n = 1000
    rs = np.random.RandomState(8)
    acts = rs.randn(2, n, 7)
    goals = [np.array([0.05, 0.0, 0.1]), np.array([-0.05, 0.1, 0.0])]
    envs = [get_environment('reacher_7dof') for _ in range(2)]
    logs = [[], []]

    def run(i):
        e = envs[i]
        e.reset_model(goal=goals[i])
        for t in range(n):
            ob, r, _, _ = e.step(acts[i, t])
            logs[i].append((ob.copy(), r))

    threads = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for i in range(2):
        q, v = np.zeros(7), np.zeros(7)
        for t in range(n):
            # every step from the state the environment itself was in (a thousand free-running steps drift apart)
            _, _, obs, r, _ = oracle_sim.env_step(q, v, goals[i], acts[i, t], t)
            assert np.abs(logs[i][t][0] - obs).max() < 1e-9 and abs(logs[i][t][1] - r) < 1e-9, (i, t)
            q, v = logs[i][t][0][:7], logs[i][t][0][7:14]
