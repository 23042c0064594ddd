"""The multi-GPU, batched and tree configurations at their full sizes (one GPU's view of them), through the C ABI:
  * K = 65 536, H = 64 (sharded 8-way in production): a slice against the oracle, shard invariance of the
    rollouts (global sample index), merged shard statistics = the update of the planner that owns all samples
  * 1024 planners x K = 512, H = 64: whole planning steps as graph replays against the same steps made of separate
    calls, the update against the oracle's formula, slices of the rollouts against the oracle
  * the 18-DoF trees at K = 32 768, H = 128: fp32 against fp64 rollouts within the bounds the design states
Needs a GPU.  fp32 tolerances as elsewhere: relative 1e-4 on returns (H = 64), arg-max exact."""
import numpy as np
import pytest

from oracle import ref_mppi

pytestmark = pytest.mark.gpu
FILTER = [np.ones(7), 0.25, 0.8, 0.0]
H = 64


def _planner(name, K, H_=H, dtype="f32", **kw):
    from trajopt_amd.device import DeviceModel, DevicePlanner
    kw.setdefault("kappa", 25.0)
    kw.setdefault("sigma", FILTER[0])
    kw.setdefault("betas", FILTER[1:])
    kw.setdefault("default_act", "mean")
    return DevicePlanner(DeviceModel.get(name), K, H_, dtype=dtype, **kw)


def _oracle(name):
    from oracle.sim import OracleSim
    from trajopt_amd.model import get_model
    return OracleSim(get_model(name))


@pytest.mark.parametrize("name", ["reacher_7dof", "reacher_7dof_free"])
def test_65536_samples_sharded_eight_ways(name):
    K, G = 65536, 8
    q0 = np.array([0.1, 0.3, -0.2, -0.8, 0.3, -0.5, 0.2])
    tgt = np.array([0.05, -0.05, 0.1])
    base = np.random.RandomState(1).randn(H, 7) * 0.1
    full = _planner(name, K)
    full.set_start_state(q0, np.zeros(7), tgt, 0)
    full.set_base_actions(base)
    full.generate_noise(2024)
    full.rollout()
    R, act = full.returns(), full.actions()
    assert full.status() == 0 and np.isfinite(R).all()
    # a slice of 256 samples spread over the launch against the oracle
    idx = np.r_[np.arange(0, 128), np.arange(K - 128, K)]
    R_o = _oracle(name).rollout(q0, np.zeros(7), tgt, 0, act[idx], nthreads=8)
    assert np.abs(R[idx] - R_o).max() / np.abs(R_o).max() < 1e-4
    full.iterate(1)
    seq, amax = full.act_sequence(), full.argmax()
    assert amax == int(np.argmax(R))
    # (the update adds base and noise in fp64; `actions()` returns the fp32 sums the rollouts used: the two differ
    # by the fp32 rounding of an action, 6e-8 relative)
    want = ref_mppi.update(act, R, 25.0)
    assert np.abs(seq - want).max() < 1e-6
    full.close()
    # eight shards of 8192 (what each GPU of a node runs): same returns sample for sample, and their merged
    # statistics give the same act_sequence and arg-max
    import torch
    from trajopt_amd import sharding
    rows = []
    for r in range(G):
        s = _planner(name, K // G, k_offset=r * (K // G))
        s.set_start_state(q0, np.zeros(7), tgt, 0)
        s.set_base_actions(base)
        s.generate_noise(2024)
        s.local_partials()
        assert np.array_equal(s.returns(), R[r * (K // G):(r + 1) * (K // G)])
        rows.append(s.partials())
        if r < G - 1:
            s.close()
    gathered = torch.from_numpy(np.stack(rows)).cuda()
    torch.cuda.synchronize()
    s.merge(gathered.data_ptr(), G)
    assert np.abs(s.act_sequence() - seq).max() < 1e-12 and s.argmax() == amax
    host_seq, _, host_amax = sharding.merge_rows(np.stack(rows), base, 25.0)
    assert np.abs(host_seq - seq).max() < 1e-12 and host_amax == amax
    s.close()


def test_1024_planners_of_512_samples(oracle_sim):
    P, K = 1024, 512
    rs = np.random.RandomState(0)
    tgt = np.stack([np.random.RandomState(p).uniform(-0.1, 0.1, 3) for p in range(P)])
    q0 = rs.uniform(-0.3, 0.3, (P, 7))
    q0[:, 3] -= 0.8
    q0[:, 5] -= 0.5
    v0 = np.zeros((P, 7))

    def fresh():
        b = _planner("reacher_7dof", K, planners=P)
        b.set_start_state(q0, v0, tgt, 0)
        b.set_base_actions(np.zeros((P, H, 7)))
        return b
    a = fresh()
    a.generate_noise_step(1000)
    a.rollout()
    R, act = a.returns(), a.actions()
    # (67 million fp32 constraint solves: none may end at the sweep cap -- a few used to, alternating at rounding
    # level, before rows were only let join the set in the late sweeps)
    assert a.status() == 0 and R.shape == (P, K) and act.shape == (P, K, H, 7)
    # planners use separate noise streams and see their own start state and target
    assert not np.array_equal(act[0], act[1])
    for pl in (0, 517, 1023):
        R_o = oracle_sim.rollout(q0[pl], v0[pl], tgt[pl], 0, act[pl, :96])
        assert np.abs(R[pl, :96] - R_o).max() / np.abs(R_o).max() < 1e-4
    a.iterate(1)          # same noise: same rollouts, then the update (one block per planner and step at this size)
    seq, amax = a.act_sequence(), a.argmax()
    for pl in (0, 1, 517, 1023):
        assert np.abs(seq[pl] - ref_mppi.update(act[pl], R[pl], 25.0)).max() < 1e-6
        assert amax[pl] == int(np.argmax(R[pl]))
    del act
    # planner 0 of the batch is the single planner with the same seed (its noise stream is plane 0)
    one = _planner("reacher_7dof", K)
    one.set_start_state(q0[0], v0[0], tgt[0], 0)
    one.set_base_actions(np.zeros((H, 7)))
    one.generate_noise_step(1000)
    one.iterate(1)
    assert np.array_equal(one.returns(), R[0]) and np.array_equal(one.act_sequence(), seq[0])
    one.close()
    # whole planning steps: graph replays (noise launch of its own at this size) against separate calls
    steps, niter = 3, 2
    b = fresh()
    b.plan(1000, niter=niter, nsteps=steps)
    c = fresh()
    for _ in range(steps):
        c.generate_noise_step(1000)
        c.iterate(niter)
        c.advance()
    assert np.array_equal(b.act_sequence(), c.act_sequence())
    for pl in (0, 333, 1023):
        lb, lc = b.read_log(planner=pl), c.read_log(planner=pl)
        assert lb["total"] == lc["total"] == steps
        for key in ("qpos", "qvel", "reward", "action"):
            assert np.array_equal(lb[key], lc[key]), (pl, key)
        # the logged real-env transitions are the oracle's
        q, v = q0[pl], v0[pl]
        for i in range(steps):
            q, v, _, r, _ = oracle_sim.env_step(q, v, tgt[pl], lb["action"][i], env_timestep=i)
            assert np.abs(q - lb["qpos"][i]).max() < 1e-10 and abs(r - lb["reward"][i]) < 1e-10
    for x in (a, b, c):
        x.close()


def _tree_start(nq):
    rs = np.random.RandomState(3)
    q0 = np.r_[0.0, 0.0, 0.0, rs.randn(15) * 0.3]
    v0 = rs.randn(18) * 0.3
    if nq == 19:
        ax, ay, az = q0[3:6]

        def qmul(a, b):
            return np.r_[a[0] * b[0] - a[1:] @ b[1:], a[0] * b[1:] + b[0] * a[1:] + np.cross(a[1:], b[1:])]
        hx = np.r_[np.cos(ax / 2), np.sin(ax / 2), 0, 0]
        hy = np.r_[np.cos(ay / 2), 0, np.sin(ay / 2), 0]
        hz = np.r_[np.cos(az / 2), 0, 0, np.sin(az / 2)]
        q0 = np.r_[q0[:3], qmul(qmul(hx, hy), hz), q0[6:]]
    return q0, v0


# bounds on max |x32 - x64| / (1 + |x64|) over all samples at t = 32, 64, 128: twice the measured maxima
@pytest.mark.parametrize("name,nq,bounds", [("tree_18dof", 18, (4e-6, 8e-6, 1.5e-5)),
                                            ("tree_18dof_quat", 19, (4e-6, 8e-6, 1.5e-5)),
                                            ("tree_18dof_rk4", 19, (8e-6, 1.5e-5, 6e-5))])
def test_tree_fp32_against_fp64_at_32768_samples(name, nq, bounds):
    K, Ht = 32768, 128
    q0, v0 = _tree_start(nq)
    out = {}
    for dtype in ("f64", "f32"):
        p = _planner(name, K, Ht, dtype, kappa=5.0, gamma=0.99, sigma=np.ones(18), dump=True)
        p.set_start_state(q0, v0, np.array([0.4, 0.1, 0.8]), 0)
        p.set_base_actions(np.zeros((Ht, 18)))
        p.generate_noise(2024)
        p.rollout()
        nobs = p.next_observations()
        out[dtype] = (p.returns(), [nobs[:, t - 1, :nq + 18].copy() for t in (32, 64, 128)], p.status())
        del nobs
        p.close()
    (R64, s64, st64), (R32, s32, st32) = out["f64"], out["f32"]
    assert st64 == 0 and st32 == 0
    for x32, x64, bound in zip(s32, s64, bounds):
        rel = (np.abs(x32 - x64) / (1.0 + np.abs(x64))).max()
        assert rel < bound, (name, rel, bound)
    assert (np.abs(R32 - R64) / np.abs(R64)).max() < 1e-5
    assert int(np.argmax(R32)) == int(np.argmax(R64))
    # a slice of the fp64 rollouts against the oracle (the fp64 kernels are the comparison's ground truth)
    p = _planner(name, 64, Ht, "f64", kappa=5.0, gamma=0.99, sigma=np.ones(18))
    p.set_start_state(q0, v0, np.array([0.4, 0.1, 0.8]), 0)
    p.set_base_actions(np.zeros((Ht, 18)))
    p.generate_noise(2024)            # the first 64 samples of the same Philox stream
    p.rollout()
    assert np.array_equal(p.returns(), R64[:64])
    R_o = _oracle(name).rollout(q0, v0, np.array([0.4, 0.1, 0.8]), 0, p.actions(), gamma=0.99, nthreads=8)
    assert np.abs(R64[:64] - R_o).max() < 1e-7
    p.close()
