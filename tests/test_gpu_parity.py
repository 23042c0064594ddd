"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference's
stored transitions.  Needs a GPU: run with `-m gpu`.

Tolerances
  fp64 kernels : 1e-10 absolute on states / rewards / returns (same algorithm, different
                 operation order than the oracle)
  fp32 kernels : relative 1e-4 on states and returns at H = 32 (the tolerance the planner
                 is specified to), arg-max index exact
"""
import os

import numpy as np
import pytest

from oracle import ref_mppi

pytestmark = pytest.mark.gpu

FILTER = [np.ones(7), 0.25, 0.8, 0.0]


@pytest.fixture(scope="module")
def dev_model():
    from trajopt_amd.device import DeviceModel
    return DeviceModel.get("reacher_7dof")


def _planner(dev_model, K, H, dtype, **kw):
    from trajopt_amd.device import DevicePlanner
    kw.setdefault("kappa", 25.0)
    kw.setdefault("sigma", FILTER[0])
    kw.setdefault("betas", FILTER[1:])
    kw.setdefault("default_act", "mean")
    return DevicePlanner(dev_model, K, H, dtype=dtype, **kw)


def test_env_step_fp64_reproduces_stored_transitions(dev_model, transitions):
    d = transitions
    n = len(d["reward"])
    worst = 0.0
    for i in range(0, n, 3):
        q2, v2, obs, r = dev_model.env_step(d["qp"][i], d["qv"][i], d["target"][i], d["act"][i], int(d["timestep"][i]))
        worst = max(worst, np.abs(q2 - d["qp_next"][i]).max(), np.abs(v2 - d["qv_next"][i]).max(),
                    np.abs(obs[:17] - d["obs_next"][i][:17]).max())
        if d["file_id"][i] == 0:
            assert abs(r - d["reward"][i]) < 1e-10
    assert worst < 1e-10, worst


def test_forward_hand(dev_model, oracle_sim):
    rs = np.random.RandomState(0)
    for _ in range(5):
        q = rs.uniform(-1.5, 1.5, 7)
        assert np.allclose(dev_model.forward_hand(q), oracle_sim.fk_hand(q), atol=1e-13)


@pytest.mark.parametrize("start", ["rest", "limits"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_rollouts_K160_H32(dev_model, oracle_sim, transitions, dtype, start):
    K, H = 160, 32
    raw = ref_mppi.all_raw_normals(123, 40, 4, H, 7)
    if start == "rest":
        q0, v0, tgt, t0 = np.zeros(7), np.zeros(7), np.zeros(3), 0
    else:   # step 50 of the stored dense agent: joints on their limits
        i = int(np.where((transitions["file_id"] == 0) & (transitions["step"] == 50))[0][0])
        q0, v0, tgt, t0 = transitions["qp"][i], transitions["qv"][i], transitions["target"][i], 50
    base = np.zeros((H, 7))
    act = base + ref_mppi.filter_noise(raw.copy(), FILTER)
    R_o, nobs_o, rew_o = oracle_sim.rollout(q0, v0, tgt, t0, act, dump=True, nthreads=4)

    p = _planner(dev_model, K, H, dtype, dump=True)
    p.set_start_state(q0, v0, tgt, t0)
    p.set_base_actions(base)
    p.upload_noise(raw)
    p.rollout()
    R, nobs, rew, acts = p.returns(), p.next_observations(), p.rewards(), p.actions()
    assert p.status() == 0
    if dtype == "f64":
        assert np.array_equal(acts, act)                     # filter arithmetic is bit-exact
        assert np.abs(nobs - nobs_o).max() < 1e-10
        assert np.abs(rew - rew_o).max() < 1e-10
        assert np.abs(R - R_o).max() < 1e-10
    else:
        assert np.abs(acts - act).max() < 1e-6
        scale = 1.0 + np.abs(nobs_o)
        assert (np.abs(nobs - nobs_o) / scale).max() < 1e-4
        assert (np.abs(R - R_o) / np.abs(R_o)).max() < 1e-4
    assert int(np.argmax(R)) == int(np.argmax(R_o))
    p.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_rollouts_away_from_every_limit(dev_model, oracle_sim, dtype):
    """Whole waves without a single constraint row (the arm stays inside its joint ranges and off the table):
    the step then takes its no-row path, which a wave of mixed samples never does."""
    from trajopt_amd.model import get_model
    md = get_model("reacher_7dof")
    K, H = 128, 12
    raw = ref_mppi.all_raw_normals(31, 32, 4, H, 7)
    q0 = 0.5 * (md.range_lo + md.range_hi)
    v0, tgt = np.zeros(7), np.array([0.2, -0.3, 0.1])
    filt = [0.05 * np.ones(7), 0.25, 0.8, 0.0]
    act = ref_mppi.filter_noise(raw.copy(), filt)
    R_o, nobs_o, rew_o = oracle_sim.rollout(q0, v0, tgt, 0, act, dump=True, nthreads=4)
    q_all = nobs_o[..., :7]
    assert (q_all > md.range_lo + 1e-3).all() and (q_all < md.range_hi - 1e-3).all()      # no limit row anywhere
    p = _planner(dev_model, K, H, dtype, dump=True, sigma=filt[0])
    p.set_start_state(q0, v0, tgt, 0)
    p.set_base_actions(np.zeros((H, 7)))
    p.upload_noise(raw)
    p.rollout()
    R, nobs = p.returns(), p.next_observations()
    assert p.status() == 0
    tol = 1e-10 if dtype == "f64" else 2e-5
    assert (np.abs(nobs - nobs_o) / (1.0 + np.abs(nobs_o))).max() < tol
    assert (np.abs(R - R_o) / np.abs(R_o)).max() < tol
    p.close()
    # the same through the planning-only kernel variant (nothing dumped)
    p = _planner(dev_model, K, H, dtype, sigma=filt[0])
    p.set_start_state(q0, v0, tgt, 0)
    p.set_base_actions(np.zeros((H, 7)))
    p.upload_noise(raw)
    p.rollout()
    assert (np.abs(p.returns() - R_o) / np.abs(R_o)).max() < tol
    p.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_rollouts_on_the_table(dev_model, oracle_sim, transitions, dtype):
    """The stored transition whose end effector is lowest: the contact sphere rests on the table, so the contact
    row is part of the active-set solve (with joint-limit rows next to it) in a large share of the steps."""
    K, H = 128, 16
    hz = np.array([oracle_sim.fk_hand(q)[2] for q in transitions["qp"]])
    i = int(np.argmin(hz))
    q0, v0, tgt = transitions["qp"][i], transitions["qv"][i], transitions["target"][i]
    raw = ref_mppi.all_raw_normals(5, 32, 4, H, 7)
    act = ref_mppi.filter_noise(raw.copy(), FILTER)
    R_o, nobs_o, rew_o = oracle_sim.rollout(q0, v0, tgt, 0, act, dump=True, nthreads=4)
    assert (nobs_o[..., 16] < -0.343).mean() > 0.3            # sphere bottom within the contact margin of the plane
    tol = 1e-10 if dtype == "f64" else 1e-4
    for dump in (True, False):                                # general kernel, planning-only kernel
        p = _planner(dev_model, K, H, dtype, dump=dump)
        p.set_start_state(q0, v0, tgt, 0)
        p.set_base_actions(np.zeros((H, 7)))
        p.upload_noise(raw)
        p.rollout()
        R = p.returns()
        assert p.status() == 0
        assert (np.abs(R - R_o) / np.abs(R_o)).max() < tol
        if dump:
            nobs = p.next_observations()
            assert (np.abs(nobs - nobs_o) / (1.0 + np.abs(nobs_o))).max() < tol
        p.close()


@pytest.mark.parametrize("H,P", [(45, 1), (64, 1), (75, 1), (21, 3), (40, 2)])
def test_update_block_shapes(dev_model, H, P):
    """The update kernel gives every block one, two or four rows of act_sequence depending on how many there
    are, and the last block of a planner may own fewer: horizons on both sides of every boundary, odd row
    counts included, single and batched planners, against the numpy restatement on the device's own returns."""
    K = 192
    rs = np.random.RandomState(H)
    p = _planner(dev_model, K, H, "f64", planners=P) if P > 1 else _planner(dev_model, K, H, "f64")
    lead = (P,) if P > 1 else ()
    base = rs.randn(*(lead + (H, 7))) * 0.3
    p.set_start_state(np.zeros(lead + (7,)), np.zeros(lead + (7,)), np.zeros(lead + (3,)) + 0.1, 0)
    p.set_base_actions(base)
    p.generate_noise(11)
    acts = p.actions()                 # base + eps, read before the update replaces base
    p.iterate(1)
    R, seq = p.returns(), p.act_sequence()
    for b in range(P):
        Rb, ab, sb = (R[b], acts[b], seq[b]) if P > 1 else (R, acts, seq)
        want = ref_mppi.update(ab, Rb, 25.0)
        assert np.abs(sb - want).max() < 1e-11, (H, P, b)
    p.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_iterate_matches_update(dev_model, oracle_sim, dtype):
    K, H = 160, 32
    raw = ref_mppi.all_raw_normals(7, 40, 4, H, 7)
    base = np.random.RandomState(1).randn(H, 7) * 0.3
    eps = ref_mppi.filter_noise(raw.copy(), FILTER)
    p = _planner(dev_model, K, H, dtype)
    p.set_start_state(np.zeros(7), np.zeros(7), np.array([0.1, -0.05, 0.05]), 0)
    p.set_base_actions(base)
    p.upload_noise(raw)
    p.iterate(1)
    R = p.returns()
    seq = p.act_sequence()
    # the update, given the device's own returns and actions, against the numpy restatement
    acts = base + (eps if dtype == "f64" else eps.astype(np.float32).astype(np.float64))
    want = ref_mppi.update(acts, R, 25.0)
    assert np.abs(seq - want).max() < (1e-11 if dtype == "f64" else 1e-6)
    assert p.argmax() == int(np.argmax(R))
    assert p.rmax() == R.max()
    # end to end against the oracle's rollouts
    R_o = oracle_sim.rollout(np.zeros(7), np.zeros(7), np.array([0.1, -0.05, 0.05]), 0, base + eps, nthreads=4)
    want_o = ref_mppi.update(base + eps, R_o, 25.0)
    assert p.argmax() == int(np.argmax(R_o))
    tol = 1e-9 if dtype == "f64" else 2e-3      # weights are exp(25 * dR): fp32 return error is amplified
    assert np.abs(seq - want_o).max() < tol
    # second iteration uses the updated sequence with the same noise
    p.iterate(1)
    R2_o = oracle_sim.rollout(np.zeros(7), np.zeros(7), np.array([0.1, -0.05, 0.05]), 0, want_o + eps, nthreads=4)
    want2 = ref_mppi.update(want_o + eps, R2_o, 25.0)
    assert np.abs(p.act_sequence() - want2).max() < (1e-8 if dtype == "f64" else 5e-3)
    p.close()


def test_sharded_partials_merge_to_single_rank_result(dev_model):
    import torch
    K, H, G = 256, 16, 4
    raw = ref_mppi.all_raw_normals(3, K, 1, H, 7)
    base = np.zeros((H, 7))
    full = _planner(dev_model, K, H, "f32")
    full.set_start_state(np.zeros(7), np.zeros(7), np.zeros(3), 0)
    full.set_base_actions(base)
    full.upload_noise(raw)
    full.iterate(1)
    want = full.act_sequence()
    shards = []
    for r in range(G):
        s = _planner(dev_model, K // G, H, "f32", k_offset=r * (K // G))
        s.set_start_state(np.zeros(7), np.zeros(7), np.zeros(3), 0)
        s.set_base_actions(base)
        s.upload_noise(raw[r * (K // G):(r + 1) * (K // G)])
        s.local_partials()
        s.synchronize()
        shards.append(s)
    gathered = torch.from_numpy(np.stack([s.partials() for s in shards])).cuda()
    torch.cuda.synchronize()
    for s in shards:
        s.merge(gathered.data_ptr(), G)
        assert np.abs(s.act_sequence() - want).max() < 1e-12
        assert s.argmax() == full.argmax()


def test_merge_kernel_equals_the_host_merge_rule(dev_model):
    """mppi_merge on gathered shard rows against trajopt_amd.sharding.merge_rows (what the gloo ranks apply)."""
    import torch
    from trajopt_amd import sharding
    K, H, G = 192, 10, 3
    base = np.random.RandomState(4).randn(H, 7) * 0.2
    shards = []
    for r in range(G):
        s = _planner(dev_model, K // G, H, "f32", k_offset=r * (K // G))
        s.set_start_state(np.zeros(7), np.zeros(7), np.array([0.1, 0.0, 0.1]), 0)
        s.set_base_actions(base)
        s.generate_noise(17)
        s.local_partials()
        s.synchronize()
        shards.append(s)
    rows = np.stack([s.partials() for s in shards])
    want, rmax, amax = sharding.merge_rows(rows, base, 25.0)
    gathered = torch.from_numpy(rows).cuda()
    torch.cuda.synchronize()
    for s in shards:
        s.merge(gathered.data_ptr(), G)
        assert np.abs(s.act_sequence() - want).max() < 1e-13
        assert s.argmax() == amax and s.rmax() == rmax
        s.close()


def test_in_library_exchange_on_one_rank_equals_the_unsharded_planner(dev_model):
    """After mppi_comm_init (RCCL, one rank) iterate() and plan() take the sharded route -- statistics,
    all-gather on the planner's stream, merge -- inside the captured graphs; results must equal the planner that
    owns all samples, bit for bit (same kernels, same order of sums)."""
    from trajopt_amd import sharding
    K, H, steps, niter = 512, 12, 5, 2
    tgt = np.array([0.05, 0.05, 0.1])
    a = _planner(dev_model, K, H, "f32")
    b = _planner(dev_model, K, H, "f32")
    sharding.init_library_exchange(b, 0, 1)
    for p in (a, b):
        p.set_start_state(np.zeros(7), np.zeros(7), tgt, 0)
        p.set_base_actions(np.zeros((H, 7)))
        p.generate_noise(3)
        p.iterate(2)
    assert np.array_equal(a.act_sequence(), b.act_sequence()) and a.argmax() == b.argmax() and a.rmax() == b.rmax()
    for p in (a, b):
        p.plan(91, niter=niter, nsteps=steps)
    la, lb = a.read_log(), b.read_log()
    assert la["total"] == lb["total"] == steps
    for key in ("qpos", "qvel", "hand", "reward", "action"):
        assert np.array_equal(la[key], lb[key]), key
    assert np.array_equal(a.act_sequence(), b.act_sequence())
    from trajopt_amd._lib import MppiError
    with pytest.raises(MppiError):
        sharding.init_library_exchange(b, 0, 1)          # one exchange per planner
    a.close(); b.close()


def test_shift_and_tail(dev_model):
    H = 6
    seq = np.random.RandomState(2).randn(H, 7)
    for mode, mean in (("repeat", np.zeros(7)), ("mean", np.arange(7) * 0.1)):
        p = _planner(dev_model, 64, H, "f32", default_act=mode, mean=mean)
        p.set_base_actions(seq)
        p.shift()
        assert np.array_equal(p.act_sequence(), ref_mppi.shift(seq, mode, mean))
        other = seq[::-1].copy()
        p.shift(other)
        assert np.array_equal(p.act_sequence(), ref_mppi.shift(other, mode, mean))
        p.close()


def test_update_from_returns(dev_model, planner_vectors):
    v = planner_vectors
    K, H, m = v["u_actions"].shape
    from trajopt_amd.device import update_from_returns
    got = update_from_returns(v["u_actions"], v["u0_scores"], 25.0)
    assert np.abs(got - v["u0_act_sequence"]).max() < 1e-13


def test_philox_noise_statistics(dev_model):
    K, H = 4096, 64
    p = _planner(dev_model, K, H, "f32", sigma=np.ones(7), betas=(1.0, 0.0, 0.0))
    p.set_base_actions(np.zeros((H, 7)))
    p.generate_noise(2024)
    z = p.actions()
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1.0) < 5e-3
    assert abs(np.mean(z ** 3)) < 2e-2 and abs(np.mean(z ** 4) - 3.0) < 5e-2
    c = np.corrcoef(z.reshape(K, -1)[:, :20].T)
    assert np.abs(c - np.eye(20)).max() < 0.08
    p.generate_noise(2024)
    assert np.array_equal(z, p.actions())          # counter based: same seed, same stream
    p.generate_noise(2025)
    assert not np.array_equal(z, p.actions())
    # sharding does not change the stream: samples are addressed by their global index
    s = _planner(dev_model, 1024, H, "f32", sigma=np.ones(7), betas=(1.0, 0.0, 0.0), k_offset=2048)
    s.set_base_actions(np.zeros((H, 7)))
    s.generate_noise(2024)
    assert np.array_equal(s.actions(), z[2048:3072])


def test_device_resident_advance_matches_host_driven_loop(dev_model, oracle_sim):
    """generate_noise_step / iterate / advance with no host round trip equals the same loop driven
    from the host (explicit seed + t, env_step through the ABI, shift)."""
    K, H, steps = 128, 8, 5
    tgt = np.array([0.05, 0.0, 0.05])
    a = _planner(dev_model, K, H, "f32", gamma=0.97)
    b = _planner(dev_model, K, H, "f32", gamma=0.97)
    for p in (a, b):
        p.set_start_state(np.zeros(7), np.zeros(7), tgt, 3)
        p.set_base_actions(np.zeros((H, 7)))
    q, v = np.zeros(7), np.zeros(7)
    host = []
    for t in range(steps):
        a.generate_noise_step(77)
        a.iterate(2)
        a.advance()
        b.set_start_state(q, v, tgt, 3 + t)
        b.generate_noise(77 + 3 + t)
        b.iterate(2)
        seq = b.act_sequence()
        q, v, obs, r = dev_model.env_step(q, v, tgt, seq[0], 3 + t)
        host.append((q, v, obs, r, seq[0]))
        b.shift()
    log = a.read_log()
    assert log["total"] == steps and len(log["reward"]) == steps
    for t in range(steps):
        assert np.array_equal(log["qpos"][t], host[t][0]) and np.array_equal(log["qvel"][t], host[t][1])
        assert np.array_equal(log["hand"][t], host[t][2][14:17]) and log["reward"][t] == host[t][3]
        assert np.array_equal(log["action"][t], host[t][4])
    assert np.array_equal(a.act_sequence(), b.act_sequence())
    # and the logged transitions are the oracle's
    q2, v2, _, r2, _ = oracle_sim.env_step(host[-2][0], host[-2][1], tgt, host[-1][4], 3 + steps - 1)
    assert np.abs(q2 - log["qpos"][-1]).max() < 1e-10 and abs(r2 - log["reward"][-1]) < 1e-10


@pytest.mark.parametrize("P,K,H,dtype", [(1, 4097, 6, "f64"), (1, 5000, 6, "f64"), (3, 4608, 6, "f64"), (2, 9001, 6, "f64"),
                                          (1, 70001, 64, "f64"), (1, 65536, 64, "f32"), (1, 8200, 300, "f64"),
                                          (2, 4352, 1, "f64"), (3, 12345, 100, "f32"), (1, 16384, 37, "f32")])
def test_update_with_many_samples_per_planner(dev_model, P, K, H, dtype):
    """Beyond 4096 samples a planner's update is cut into chunks of samples as well as rows (maximum, weights, the
    chunks' sums, their sum in the order of the chunks): ragged sample counts, chunk counts from 1 to 64, several
    planners per handle, all against the numpy update on the device's own returns and actions."""
    rs = np.random.RandomState(K)
    # (fp32 actions are the fp32 sum of base and noise, the update adds the weighted noise to the fp64 base)
    tol = 1e-12 if dtype == "f64" else 2e-6
    b = _planner(dev_model, K, H, dtype, planners=P)
    q0 = rs.uniform(-0.3, 0.3, (P, 7)) if P > 1 else rs.uniform(-0.3, 0.3, 7)
    v0 = np.zeros_like(q0)
    tgt = rs.uniform(-0.1, 0.1, (P, 3)) if P > 1 else rs.uniform(-0.1, 0.1, 3)
    base = rs.randn(*((P, H, 7) if P > 1 else (H, 7))) * 0.1
    b.set_start_state(q0, v0, tgt, 0 if P == 1 else np.zeros(P, dtype=np.int32))
    b.set_base_actions(base)
    b.generate_noise(11)
    b.rollout()
    R, act = b.returns(), b.actions()
    b.iterate(1)
    seq, am = b.act_sequence(), b.argmax()
    for pl in range(P):
        Rp, ap = (R[pl], act[pl]) if P > 1 else (R, act)
        want = ref_mppi.update(ap, Rp, 25.0)
        got = seq[pl] if P > 1 else seq
        assert np.abs(got - want).max() < tol, (pl, np.abs(got - want).max())
        assert (am[pl] if P > 1 else am) == int(np.argmax(Rp))
    # the sparse reward takes three values: returns repeat, often the best too, and the first of the best is the arg-max
    b.set_params(25.0, 1.0, "sparse")
    b.set_base_actions(base)
    b.generate_noise(12)
    b.rollout()
    R, act = b.returns(), b.actions()
    b.iterate(1)
    seq, am = b.act_sequence(), b.argmax()
    for pl in range(P):
        Rp, ap = (R[pl], act[pl]) if P > 1 else (R, act)
        assert len(np.unique(Rp)) < K // 4
        assert (am[pl] if P > 1 else am) == int(np.argmax(Rp))
        assert np.abs((seq[pl] if P > 1 else seq) - ref_mppi.update(ap, Rp, 25.0)).max() < tol
    b.close()


@pytest.mark.parametrize("K", [64, 65, 100, 128, 129, 250, 1000])
def test_update_of_many_planners_with_few_samples(dev_model, K):
    """Many planners, at most 1024 samples each: a block of the update is cut into sub-blocks of one, two or four waves
    and each owns a step (H = 30 leaves the last block's sub-blocks idle).  Against the numpy update, planner by
    planner, and bit for bit against the update that draws the noise again."""
    P, H = 40, 30
    rs = np.random.RandomState(K)
    q0, tgt = rs.uniform(-0.3, 0.3, (P, 7)), rs.uniform(-0.1, 0.1, (P, 3))
    base = rs.randn(P, H, 7) * 0.1
    out = []
    for regen in ("0", "1"):
        os.environ["MPPI_UPDATE_REGEN"] = regen
        try:
            b = _planner(dev_model, K, H, "f64", planners=P)
            b.set_start_state(q0, np.zeros_like(q0), tgt, np.zeros(P, dtype=np.int32))
            b.set_base_actions(base)
            b.generate_noise(5)
            b.rollout()
            R, act = b.returns(), b.actions()
            b.iterate(1)
            seq, am = b.act_sequence(), b.argmax()
            out.append(seq)
            for pl in range(P):
                assert np.abs(seq[pl] - ref_mppi.update(act[pl], R[pl], 25.0)).max() < 1e-12, (regen, pl)
                assert am[pl] == int(np.argmax(R[pl]))
            b.close()
        finally:
            os.environ.pop("MPPI_UPDATE_REGEN", None)
    assert np.array_equal(out[0], out[1])


def test_batched_planners_equal_independent_planners(dev_model, oracle_sim):
    """One handle with P planners (own start state, base actions, noise) gives, planner by planner,
    what P separate handles give; and its rollouts are the oracle's."""
    P, K, H = 5, 96, 10
    rs = np.random.RandomState(8)
    q0, v0 = rs.uniform(-0.5, 0.5, (P, 7)), rs.randn(P, 7) * 0.5
    tgt = rs.uniform(-0.1, 0.1, (P, 3))
    ts = np.arange(P) * 3
    base = rs.randn(P, H, 7) * 0.2
    raw = np.stack([ref_mppi.raw_normals(100 + b, K, H, 7) for b in range(P)])
    batch = _planner(dev_model, K, H, "f32", planners=P, gamma=0.95)
    batch.set_start_state(q0, v0, tgt, ts)
    batch.set_base_actions(base)
    batch.upload_noise(raw)
    batch.iterate(2)
    R, seq, am = batch.returns(), batch.act_sequence(), batch.argmax()
    assert R.shape == (P, K) and seq.shape == (P, H, 7)
    for b in range(P):
        one = _planner(dev_model, K, H, "f32", gamma=0.95)
        one.set_start_state(q0[b], v0[b], tgt[b], int(ts[b]))
        one.set_base_actions(base[b])
        one.upload_noise(raw[b])
        one.iterate(2)
        assert np.array_equal(one.returns(), R[b]) and np.array_equal(one.act_sequence(), seq[b])
        assert one.argmax() == am[b]
        one.close()
    # first-iteration returns against the oracle for one of the planners
    chk = _planner(dev_model, K, H, "f32", planners=P, gamma=0.95)
    chk.set_start_state(q0, v0, tgt, ts)
    chk.set_base_actions(base)
    chk.upload_noise(raw)
    chk.rollout()
    eps = ref_mppi.filter_noise(raw[3].copy(), FILTER)
    R_o = oracle_sim.rollout(q0[3], v0[3], tgt[3], int(ts[3]), base[3] + eps, gamma=0.95)
    assert (np.abs(chk.returns()[3] - R_o) / np.abs(R_o)).max() < 1e-4
    # device-resident advance runs every planner's real env and log
    batch.advance()
    for b in (0, P - 1):
        log = batch.read_log(planner=b)
        q2, v2, _, r2, _ = oracle_sim.env_step(q0[b], v0[b], tgt[b], seq[b][0], int(ts[b]))
        assert log["total"] == 1 and np.abs(log["qpos"][0] - q2).max() < 1e-10 and abs(log["reward"][0] - r2) < 1e-10
    assert np.array_equal(batch.act_sequence()[1], ref_mppi.shift(seq[1], "mean", np.zeros(7)))
    # Philox streams differ between planners of one handle
    batch.set_base_actions(np.zeros((P, H, 7)))
    batch.generate_noise(5)
    z = batch.actions()
    assert not np.array_equal(z[0], z[1])


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_constraint_free_arm(dtype):
    """The same arm with joint limits and the table contact stripped (the 'no contacts' serial arm).
    Not pinned by any reference data: the checker is this repository's own fp64 oracle."""
    from oracle.sim import OracleSim
    from trajopt_amd.device import DeviceModel
    from trajopt_amd.model import get_model
    sim = OracleSim(get_model("reacher_7dof_free"))
    dm = DeviceModel.get("reacher_7dof_free")
    K, H = 128, 32
    raw = ref_mppi.all_raw_normals(21, K, 1, H, 7)
    act = ref_mppi.filter_noise(raw.copy(), FILTER) * 2.0
    R_o, nobs_o, _ = sim.rollout(np.zeros(7), np.zeros(7), np.zeros(3), 0, act, dump=True, nthreads=4)
    assert np.abs(nobs_o[:, :, :7]).max() > 2.5            # nothing holds the joints inside their former ranges
    p = _planner(dm, K, H, dtype, dump=True, sigma=2.0 * np.ones(7))
    p.set_start_state(np.zeros(7), np.zeros(7), np.zeros(3), 0)
    p.set_base_actions(np.zeros((H, 7)))
    p.upload_noise(raw)
    p.rollout()
    nobs, R = p.next_observations(), p.returns()
    if dtype == "f64":
        assert np.abs(nobs - nobs_o).max() < 1e-9 and np.abs(R - R_o).max() < 1e-9
    else:
        assert (np.abs(nobs - nobs_o) / (1 + np.abs(nobs_o))).max() < 1e-4
        assert (np.abs(R - R_o) / np.abs(R_o)).max() < 1e-4
    assert int(np.argmax(R)) == int(np.argmax(R_o))


@pytest.mark.parametrize("steps", [6, 19])
def test_whole_step_graph_equals_separate_calls(dev_model, steps):
    """mppi_plan against the same steps call by call, bit for bit; 19 steps = two replays of the eight-step graph and
    three of the one-step graph."""
    K, H, niter = 128, 8, 3
    tgt = np.array([0.0, 0.05, 0.05])
    a = _planner(dev_model, K, H, "f32")
    b = _planner(dev_model, K, H, "f32")
    for p in (a, b):
        p.set_start_state(np.zeros(7), np.zeros(7), tgt, 0)
        p.set_base_actions(np.zeros((H, 7)))
    a.plan(91, niter=niter, nsteps=steps)
    for _ in range(steps):
        b.generate_noise_step(91)
        b.iterate(niter)
        b.advance()
    la, lb = a.read_log(), b.read_log()
    assert la["total"] == lb["total"] == steps
    for key in ("qpos", "qvel", "hand", "reward", "action"):
        assert np.array_equal(la[key], lb[key]), key
    assert np.array_equal(a.act_sequence(), b.act_sequence())
    a.plan(91, niter=niter, nsteps=2)              # replays the cached graph
    assert a.read_log()["total"] == steps + 2


@pytest.mark.parametrize("planners", [1, 3])
def test_noise_drawn_in_the_advance_launch_gives_the_same_trajectory(dev_model, planners):
    """advance_noise (real-env step and the next step's Philox noise in one launch) against the separate calls, and
    mppi_plan against both: identical real-env logs, bit for bit."""
    K, H, steps, seed = 192, 12, 6, 31
    rs = np.random.RandomState(8)
    lead = (planners,) if planners > 1 else ()
    tgt = rs.randn(*lead, 3) * 0.1
    ps = [_planner(dev_model, K, H, "f32", gamma=0.97, planners=planners) for _ in range(3)]
    for p in ps:
        p.set_start_state(np.zeros(lead + (7,)), np.zeros(lead + (7,)), tgt, 2)
        p.set_base_actions(np.zeros(lead + (H, 7)))
    a, b, c = ps
    for t in range(steps):
        a.generate_noise_step(seed)
        a.iterate(2)
        a.advance()
    b.generate_noise_step(seed)
    for t in range(steps):
        b.iterate(2)
        b.advance_noise(seed)
    c.plan(seed, niter=2, nsteps=2)
    c.plan(seed, niter=2, nsteps=steps - 2)          # the second call finds its first step's noise already drawn
    for pl in range(planners):
        la, lb, lc = a.read_log(planner=pl), b.read_log(planner=pl), c.read_log(planner=pl)
        assert la["total"] == lb["total"] == lc["total"] == steps
        for key in ("qpos", "qvel", "hand", "reward", "action"):
            assert np.array_equal(la[key], lb[key]), key
            assert np.array_equal(la[key], lc[key]), key
    assert np.array_equal(a.act_sequence(), b.act_sequence()) and np.array_equal(a.act_sequence(), c.act_sequence())
    # a change of state between two plans invalidates nothing that matters: the noise key follows the new timestep
    c.set_start_state(np.zeros(lead + (7,)), np.zeros(lead + (7,)), tgt, 40)
    a.set_start_state(np.zeros(lead + (7,)), np.zeros(lead + (7,)), tgt, 40)
    c.set_base_actions(np.zeros(lead + (H, 7))); a.set_base_actions(np.zeros(lead + (H, 7)))
    c.plan(seed, niter=1, nsteps=1)
    a.generate_noise_step(seed); a.iterate(1); a.advance()
    assert np.array_equal(a.act_sequence(), c.act_sequence())
    for p in ps:
        p.close()


@pytest.mark.parametrize("dtype,H,P,K", [("f32", 16, 64, 512), ("f64", 32, 32, 192), ("f32", 70, 16, 1000)])
def test_update_from_noise_drawn_again_equals_the_update_from_stored_noise(dev_model, dtype, H, P, K):
    """Batched planners on device noise update their act_sequence without reading the noise back: every block draws
    its samples' perturbations again (same counters, same key) and sums them in the order the stored-noise kernel
    uses.  Both routes must give act_sequence, arg-max, best return and the shard statistics bit for bit: one
    iteration, a second one from the updated sequence, and whole planning steps replayed from the captured graph."""
    import os
    rs = np.random.RandomState(H + P)
    res = []
    for regen in ("1", "0"):
        os.environ["MPPI_UPDATE_REGEN"] = regen
        try:
            p = _planner(dev_model, K, H, dtype, planners=P)
            p.set_params(25.0, 0.97, "dense")
            q0, v0 = 0.3 * rs.randn(P, 7) if regen == "1" else q0, rs.randn(P, 7) if regen == "1" else v0
            tgt = 0.2 * rs.rand(P, 3) if regen == "1" else tgt
            base = 0.2 * rs.randn(P, H, 7) if regen == "1" else base
            p.set_start_state(q0, v0, tgt, 3)
            p.set_base_actions(base)
            p.generate_noise(17)
            p.iterate(2)
            out = [p.act_sequence(), p.argmax(), p.rmax()]
            p.generate_noise_step(5)
            p.local_partials()
            out.append(p.partials())
            p.plan(9, 1, 3)
            out += [p.act_sequence(), p.read_log(3, planner=P - 1)["qpos"]]
            res.append(out)
            p.close()
        finally:
            os.environ.pop("MPPI_UPDATE_REGEN", None)
    for a, b in zip(*res):
        assert np.array_equal(a, b)
