"""The CPU oracle against the reference's own stored data (CPU only)."""
import os

import numpy as np
import pytest

from oracle import ref_mppi


def test_model_constants(reacher_model, oracle_sim):
    md = reacher_model
    # body masses and M(q=0) recovered from the reference's stored transitions
    assert abs(md.mass.sum() - 44.41427) < 1e-4
    M, _ = oracle_sim.mass_matrix(np.zeros(7))
    diag = [3.948927837632, 2.632981109256, 0.024277153219, 0.362836726801,
            0.012897047991, 0.011421311022, 0.009490331097]
    assert np.allclose(np.diag(M), diag, atol=1e-11)
    assert abs(M[1, 3] - 0.817835665702) < 1e-11 and abs(M[1, 5] - 0.053810319472) < 1e-11
    assert abs(M[3, 5] - 0.028074392454) < 1e-11 and abs(M[2, 4] - 0.008897047991) < 1e-11
    assert np.allclose(M, M.T)
    assert abs(md.contact_invweight - 0.17519767695555766) < 1e-13
    assert abs(md.solref_k - 2770.083102493) < 1e-6 and abs(md.solref_b - 105.263157895) < 1e-6
    assert np.allclose(oracle_sim.fk_hand(np.zeros(7)), [0.821, -0.6, 0.0])


def test_known_answer_transition(oracle_sim):
    # dense_agent.pickle, k = 11 (two joints at / beyond their limits)
    qp = [-0.08576798151090545, 0.23126764266362093, -1.0230690632873827, -1.1921611441050324,
          0.5262111563393779, 0.02089510755764203, 1.539217323685832]
    qv = [0.17062929776795926, 1.1430693610240867, -1.108644570162189, -6.521657828483809,
          5.519114474620228, 0.10169600292945252, -0.4094532456316935]
    act = [2.5125439365380164, 2.4757245442364018, -4.152327571429472, -1.4085549582193855,
           1.953301697880774, 7.469762751227673, 3.0656787719285434]
    q2, v2, obs, r, _ = oracle_sim.env_step(qp, qv, np.zeros(3), act, 11)
    assert np.allclose(q2, [-0.07996868613916472, 0.25094341425495537, -1.057085811722091,
                            -1.3200940276758009, 0.7046288982730131, 0.02215861297915984,
                            1.543394691499017], atol=1e-12)
    assert np.allclose(v2, [0.33583181093117936, 0.9398921174398053, -1.8603459028201599,
                            -6.358260351004002, 9.752708484599061, 0.04704815973075774,
                            0.2565637483688191], atol=1e-11)
    assert abs(r - (-10.471024876893228)) < 1e-12
    assert np.allclose(obs[14:17], [0.6415364404329132, -0.3896284850475088, 0.03102098538043795], atol=1e-14)


def test_stored_transitions(oracle_sim, transitions):
    d = transitions
    n = len(d["reward"])
    worst = 0.0
    rows = 0
    for i in range(n):
        q2, v2, obs, r, info = oracle_sim.env_step(d["qp"][i], d["qv"][i], d["target"][i], d["act"][i],
                                                   d["timestep"][i])
        worst = max(worst, np.abs(q2 - d["qp_next"][i]).max(), np.abs(v2 - d["qv_next"][i]).max(),
                    np.abs(obs[:17] - d["obs_next"][i][:17]).max())
        rows += info[0] > 0
        if d["file_id"][i] == 0:     # dense_agent.pickle: rewards follow the current dense rule
            assert abs(r - d["reward"][i]) < 1e-11
    assert worst < 1e-10, worst
    assert rows > 0.5 * n          # most stored states sit on a joint limit


@pytest.mark.skipif(not os.path.isdir("/root/reference/trajopt"), reason="reference tree not mounted")
def test_all_stored_transitions_when_reference_is_mounted(oracle_sim):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from extract_reference_transitions import load_all
    d, _ = load_all("/root/reference")
    assert len(d["reward"]) > 20000
    worst = 0.0
    for i in range(len(d["reward"])):
        q2, v2, obs, _, _ = oracle_sim.env_step(d["qp"][i], d["qv"][i], d["target"][i], d["act"][i], d["timestep"][i])
        worst = max(worst, np.abs(q2 - d["qp_next"][i]).max(), np.abs(v2 - d["qv_next"][i]).max())
    assert worst < 1e-10, worst


def test_reward_types_follow_the_reference_rules(oracle_sim, transitions):
    # trajopt/envs/reacher_env.py:47-81 applied to the oracle's own (stale) hand position.
    # The stored sparse rewards come from older radius variants, so they are not used.
    d = transitions
    rs = np.random.RandomState(0)
    ref = rs.randn(int(d["timestep"].max()) + 2, 3) * 0.3
    seen = set()
    for i in range(0, len(d["reward"]), 7):
        args = (d["qp"][i], d["qv"][i], d["target"][i], d["act"][i], int(d["timestep"][i]))
        q2, v2, obs, r_dense, _ = oracle_sim.env_step(*args)
        hand, tgt = obs[14:17], obs[17:20]
        dist = np.linalg.norm(hand - tgt)
        vn = np.linalg.norm(v2)
        assert abs(r_dense - (-10.0 * dist - 0.25 * vn)) < 1e-12
        sparse = 100.0 if dist < 0.025 else (-100.0 if dist > 0.8 else -10.0)
        seen.add(sparse)
        assert oracle_sim.env_step(*args, reward_type="sparse")[3] == sparse
        track = -10.0 * np.linalg.norm(hand - ref[args[4] + 1]) - 0.25 * vn
        assert abs(oracle_sim.env_step(*args, reward_type="tracking", reference=ref)[3] - track) < 1e-12
        cool = 0.3 * track + 0.7 * sparse
        assert abs(oracle_sim.env_step(*args, reward_type="cooling", reference=ref, alpha=0.3)[3] - cool) < 1e-12
    assert seen == {100.0, -100.0, -10.0}


def test_rollout_matches_repeated_env_step(oracle_sim):
    rs = np.random.RandomState(1)
    act = rs.randn(3, 8, 7) * 2
    q0, v0, tgt = rs.randn(7) * 0.3, rs.randn(7), np.array([0.1, -0.05, 0.02])
    R, nobs, rew = oracle_sim.rollout(q0, v0, tgt, 5, act, gamma=0.9, dump=True, nthreads=2)
    for k in range(3):
        q, v, tot = q0, v0, 0.0
        for t in range(8):
            q, v, obs, r, _ = oracle_sim.env_step(q, v, tgt, act[k, t], 5 + t)
            assert np.allclose(obs, nobs[k, t], atol=0) and r == rew[k, t]
            tot += 0.9 ** t * r
        assert abs(tot - R[k]) < 1e-12


# ---- planner arithmetic against vectors produced by the reference's numpy functions ----
@pytest.mark.parametrize("tag", ["p0", "p1", "p2", "p3", "p4"])
def test_perturbations(planner_vectors, tag):
    v = planner_vectors
    seed, H, m, N = v[tag + "_cfg"]
    fc = [v[tag + "_sigma"], *v[tag + "_betas"]]
    np.random.seed(int(seed))
    got = np.array([ref_mppi.perturbed_actions(v[tag + "_base"], fc) for _ in range(N)])
    assert np.array_equal(got, v[tag + "_acts"])
    raw = ref_mppi.raw_normals(int(seed), int(N), int(H), int(m))
    assert np.array_equal(v[tag + "_base"] + ref_mppi.filter_noise(raw, fc), v[tag + "_acts"])


def test_gather_order(planner_vectors):
    v = planner_vectors
    raw = ref_mppi.all_raw_normals(500, 4, 3, 6, 3)
    acts = ref_mppi.filter_noise(raw, [np.ones(3), 0.25, 0.8, 0.0])
    assert np.array_equal(acts, v["g_actions"])
    # overlapping worker streams: worker i+1 starts paths_per_cpu seeds later, not draws later
    assert not np.array_equal(acts[4], acts[0])


@pytest.mark.parametrize("tag", ["u0", "u1"])
def test_score_and_update(planner_vectors, tag):
    v = planner_vectors
    kappa, gamma = v[tag + "_kg"]
    R = ref_mppi.score(v["u_rewards"], gamma)
    assert np.array_equal(R, v[tag + "_scores"])
    Rc = ref_mppi.score(v["u_rewards"], gamma, v["u_critic_rewards"])
    assert np.array_equal(Rc, v[tag + "_scores_critic"])
    assert np.array_equal(ref_mppi.update(v["u_actions"], R, kappa), v[tag + "_act_sequence"])
    assert np.array_equal(ref_mppi.update(v["u_actions"], Rc, kappa), v[tag + "_act_sequence_critic"])
    # shard statistics merge back to the same update
    parts = [ref_mppi.partials(v["u_actions"][s], R[s], kappa) for s in (slice(0, 5), slice(5, 12))]
    assert np.allclose(ref_mppi.merge_partials(parts, kappa), v[tag + "_act_sequence"], rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("tag,mode", [("a_repeat", "repeat"), ("a_mean", "mean")])
def test_shift(planner_vectors, tag, mode):
    v = planner_vectors
    seq = v[tag + "_init"]
    for k in range(12):
        seq = ref_mppi.shift(seq, mode, np.array([0.1, 0.2, 0.3]))
        assert np.array_equal(seq, v[tag + "_seqs"][k])


# ---- the planner loop restated from ref_mppi pieces, against the reference's own train_step on a toy env ----
def _toy_rollouts(start_state, acts):
    """`do_env_rollout` bookkeeping (trajopt/utils.py:72-99) on the toy env: observation, state and info are
    recorded before every step, reward and next observation after it."""
    from toy_env import ToyEnv
    e = ToyEnv()
    out = []
    for a in acts:
        e.set_env_state(start_state)
        obs, rew, nobs, ts = [], [], [], []
        for k in range(a.shape[0]):
            obs.append(e._get_obs())
            ts.append(e.get_env_infos()["state"]["timestep"])
            s, r, _, _ = e.step(a[k])
            rew.append(r)
            nobs.append(s)
        out.append((np.array(obs), np.array(rew), np.array(nobs), np.array(ts)))
    return out


def test_gathered_paths_content(planner_vectors):
    """What `gather_paths_parallel` returned in the reference (toy env, 3 workers x 4 paths): per-path observations,
    rewards, next observations and info time steps follow from the recorded actions by the bookkeeping above."""
    from toy_env import ToyEnv
    v = planner_vectors
    paths = _toy_rollouts(ToyEnv().get_env_state(), v["g_actions"])
    assert np.array_equal(np.array([p[0] for p in paths]), v["g_observations"])
    assert np.array_equal(np.array([p[1] for p in paths]), v["g_rewards"])
    assert np.array_equal(np.array([p[2] for p in paths]), v["g_next_observations"])
    assert np.array_equal(np.array([p[3] for p in paths]), v["g_info_timestep"])
    assert list(v["g_keys"]) == sorted(["observations", "actions", "rewards", "env_infos", "states", "next_observations"])


def test_train_step_loop_restatement(planner_vectors):
    """Three `MPPI.train_step(niter=2)` calls of the reference on the toy env (trajopt/algos/mppi.py:125-196):
    noise seeded with seed + t and shared by the iterations of a step, score, update, real step with
    act_sequence[0], shift with the 'mean' tail.  The same loop built from ref_mppi's pieces -- the loop the
    closed-loop golden data and the acceptance tests are made with -- reproduces it bit for bit."""
    from toy_env import ToyEnv
    v = planner_vectors
    fc = [np.ones(3), 0.25, 0.8, 0.0]
    e = ToyEnv()
    state, seq = e.get_env_state(), np.zeros((6, 3))
    xs, rewards = [state["x"]], []
    for t in range(3):
        eps = ref_mppi.filter_noise(ref_mppi.all_raw_normals(77 + t, 8, 1, 6, 3), fc)
        for _ in range(2):
            acts = seq + eps
            R = ref_mppi.score(np.array([p[1] for p in _toy_rollouts(state, acts)]), 0.95)
            seq = ref_mppi.update(acts, R, 5.0)
        e.set_env_state(state)
        _, r, _, _ = e.step(seq[0])
        state = e.get_env_state()
        seq = ref_mppi.shift(seq, "mean", np.zeros(3))
        assert np.array_equal(seq, v["t_seqs"][t])
        xs.append(state["x"]); rewards.append(r)
    assert np.array_equal(np.array(rewards), v["t_sol_reward"]) and np.array_equal(np.array(xs), v["t_sol_x"])
