"""The synthetic 18-DoF floating-base tree (slides, branching, gravity; no limits or contacts)
on the HIP path.  The reference has no such model: parity here is unpinned by the reference and
rests on the fp64 oracle (whose bias forces are checked against a finite-difference Lagrangian
and energy conservation in test_oracle_tree.py).

Tolerances
  fp64 kernel : 1e-8 absolute on states after 128 steps of a falling, tumbling tree
  fp32 kernel : state error relative to 1 + |x| against the fp64 kernel, reported per horizon;
                bounds asserted on the median and the 99th percentile
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NV = 18


@pytest.fixture(scope="module")
def tree():
    from trajopt_amd.device import DeviceModel
    return DeviceModel.get("tree_18dof")


@pytest.fixture(scope="module")
def tree_oracle():
    from oracle.sim import OracleSim
    from trajopt_amd.model import get_model
    return OracleSim(get_model("tree_18dof"))


def _start(seed=3):
    rs = np.random.RandomState(seed)
    q0 = np.r_[0.0, 0.0, 0.0, rs.randn(15) * 0.3]
    v0 = rs.randn(NV) * 0.3
    return q0, v0, np.array([0.4, 0.1, 0.8])


def _run(tree, dtype, K, H, q0, v0, tgt, seed, dump=True):
    from trajopt_amd.device import DevicePlanner
    p = DevicePlanner(tree, K, H, kappa=5.0, gamma=0.99, sigma=np.ones(NV), betas=(0.25, 0.8, 0.0), dtype=dtype, dump=dump)
    p.set_start_state(q0, v0, tgt, 0)
    p.set_base_actions(np.zeros((H, NV)))
    p.generate_noise(seed)
    p.rollout()
    return p


def test_tree_fp64_against_oracle(tree, tree_oracle):
    K, H = 128, 128
    q0, v0, tgt = _start()
    p = _run(tree, "f64", K, H, q0, v0, tgt, 5)
    act, nobs, rew, R = p.actions(), p.next_observations(), p.rewards(), p.returns()
    assert p.status() == 0
    R_o, nobs_o, rew_o = tree_oracle.rollout(q0, v0, tgt, 0, act, gamma=0.99, dump=True, nthreads=4)
    assert np.abs(nobs - nobs_o).max() < 1e-8
    assert np.abs(rew - rew_o).max() < 1e-8
    assert np.abs(R - R_o).max() < 1e-7
    # the tree really falls: trunk height after 0.64 s of free fall is about -2 m
    assert nobs[:, -1, 2].mean() < -1.0
    p.close()


def test_tree_single_step_and_site(tree, tree_oracle):
    rs = np.random.RandomState(9)
    for _ in range(4):
        q, v, a = rs.randn(NV) * 0.5, rs.randn(NV), rs.randn(NV) * 2
        tgt = rs.randn(3)
        q2, v2, obs, r = tree.env_step(q, v, tgt, a, 0)
        qo, vo, obs_o, ro, _ = tree_oracle.env_step(q, v, tgt, a)
        assert np.abs(q2 - qo).max() < 1e-12 and np.abs(v2 - vo).max() < 1e-11
        assert np.abs(obs - obs_o).max() < 1e-11 and abs(r - ro) < 1e-11
        assert np.allclose(tree.forward_hand(q), tree_oracle.fk_hand(q), atol=1e-13)


def test_tree_fp32_error_growth(tree):
    """fp32 against fp64 on identical Philox noise: relative state error at t = 32, 64, 128."""
    K, H = 2048, 128
    q0, v0, tgt = _start()
    p64 = _run(tree, "f64", K, H, q0, v0, tgt, 2024)
    p32 = _run(tree, "f32", K, H, q0, v0, tgt, 2024)
    o64, o32 = p64.next_observations()[..., :2 * NV], p32.next_observations()[..., :2 * NV]
    assert p32.status() == 0
    rel = np.abs(o32 - o64) / (1.0 + np.abs(o64))
    med = [float(np.median(rel[:, t - 1].max(axis=1))) for t in (32, 64, 128)]
    p99 = [float(np.percentile(rel[:, t - 1].max(axis=1), 99)) for t in (32, 64, 128)]
    print("fp32 tree error median", med, "p99", p99)
    assert med[0] < 1e-4 and med[1] < 3e-4 and med[2] < 1e-3, med
    assert p99[2] < 2e-2, p99
    R64, R32 = p64.returns(), p32.returns()
    assert np.median(np.abs(R32 - R64) / np.abs(R64)) < 1e-4
    p64.close(); p32.close()


def test_tree_planning_iteration(tree):
    """One whole planning step (noise, rollouts, update, advance) runs on the 18-DoF tree and the
    update equals the host formula applied to the device's own returns and actions."""
    from oracle import ref_mppi
    K, H = 512, 32
    q0, v0, tgt = _start(4)
    p = _run(tree, "f64", K, H, q0, v0, tgt, 77, dump=False)
    act, R = p.actions(), p.returns()
    p.iterate(1)                      # same noise and base actions: same rollouts, then the update
    want = ref_mppi.update(act, R, 5.0)
    assert np.abs(p.act_sequence() - want).max() < 1e-10
    p.advance()
    log = p.read_log()
    assert log["total"] == 1 and np.isfinite(log["qpos"]).all()
    p.close()


# ---- the same tree on a free joint with a quaternion (tree_18dof_quat: nq = 19, nv = 18) ------------------
NQ = 19


@pytest.fixture(scope="module")
def treeq():
    from trajopt_amd.device import DeviceModel
    return DeviceModel.get("tree_18dof_quat")


@pytest.fixture(scope="module")
def treeq_oracle():
    from oracle.sim import OracleSim
    from trajopt_amd.model import get_model
    return OracleSim(get_model("tree_18dof_quat"))


def _start_quat(seed=3):
    rs = np.random.RandomState(seed)
    quat = rs.randn(4)
    q0 = np.r_[0.0, 0.0, 0.0, quat / np.linalg.norm(quat), rs.randn(12) * 0.3]
    return q0, rs.randn(NV) * 0.5, np.array([0.4, 0.1, 0.8])


def test_quat_tree_fp64_against_oracle(treeq, treeq_oracle):
    K, H = 128, 128
    q0, v0, tgt = _start_quat()
    p = _run(treeq, "f64", K, H, q0, v0, tgt, 5)
    act, nobs, rew, R = p.actions(), p.next_observations(), p.rewards(), p.returns()
    assert p.status() == 0 and nobs.shape == (K, H, NQ + NV + 6)
    R_o, nobs_o, rew_o = treeq_oracle.rollout(q0, v0, tgt, 0, act, gamma=0.99, dump=True, nthreads=4)
    assert np.abs(nobs - nobs_o).max() < 1e-8
    assert np.abs(rew - rew_o).max() < 1e-8 and np.abs(R - R_o).max() < 1e-7
    assert np.abs(np.linalg.norm(nobs[..., 3:7], axis=-1) - 1.0).max() < 1e-12
    # it tumbles: the trunk's orientation moves well away from where it started
    assert np.abs(nobs[:, -1, 3:7] - q0[3:7]).max() > 0.2
    p.close()


def test_quat_tree_single_step_and_site(treeq, treeq_oracle):
    rs = np.random.RandomState(9)
    for s in range(4):
        q, v, tgt = _start_quat(20 + s)
        a = rs.randn(NV) * 2
        q2, v2, obs, r = treeq.env_step(q, v, tgt, a, 0)
        qo, vo, obs_o, ro, _ = treeq_oracle.env_step(q, v, tgt, a)
        assert np.abs(q2 - qo).max() < 1e-12 and np.abs(v2 - vo).max() < 1e-11
        assert np.abs(obs - obs_o).max() < 1e-11 and abs(r - ro) < 1e-11
        assert np.allclose(treeq.forward_hand(q), treeq_oracle.fk_hand(q), atol=1e-13)


def test_quat_tree_fp32_error_growth_and_planning(treeq):
    K, H = 2048, 128
    q0, v0, tgt = _start_quat()
    p64 = _run(treeq, "f64", K, H, q0, v0, tgt, 2024)
    p32 = _run(treeq, "f32", K, H, q0, v0, tgt, 2024)
    o64, o32 = p64.next_observations()[..., :NQ + NV], p32.next_observations()[..., :NQ + NV]
    assert p32.status() == 0
    rel = np.abs(o32 - o64) / (1.0 + np.abs(o64))
    med = [float(np.median(rel[:, t - 1].max(axis=1))) for t in (32, 64, 128)]
    p99 = [float(np.percentile(rel[:, t - 1].max(axis=1), 99)) for t in (32, 64, 128)]
    print("fp32 quaternion tree error median", med, "p99", p99)
    assert med[0] < 1e-4 and med[2] < 1e-3 and p99[2] < 2e-2
    assert np.abs(np.linalg.norm(o32[..., 3:7], axis=-1) - 1.0).max() < 1e-6
    # a whole planning step: update, real-env step in fp64, log with the 19-entry qpos
    p32.iterate(1)
    p32.advance()
    log = p32.read_log()
    assert log["total"] == 1 and log["qpos"].shape == (1, NQ) and abs(np.linalg.norm(log["qpos"][0, 3:7]) - 1.0) < 1e-14
    p64.close(); p32.close()


# ---- the quaternion tree stepped with fourth-order Runge-Kutta (tree_18dof_rk4) ----------------------------
@pytest.fixture(scope="module")
def treer():
    from trajopt_amd.device import DeviceModel
    return DeviceModel.get("tree_18dof_rk4")


@pytest.fixture(scope="module")
def treer_oracle():
    from oracle.sim import OracleSim
    from trajopt_amd.model import get_model
    return OracleSim(get_model("tree_18dof_rk4"))


def test_rk4_tree_fp64_against_oracle(treer, treer_oracle):
    K, H = 128, 64
    q0, v0, tgt = _start_quat()
    p = _run(treer, "f64", K, H, q0, v0, tgt, 5)
    act, nobs, rew, R = p.actions(), p.next_observations(), p.rewards(), p.returns()
    assert p.status() == 0 and nobs.shape == (K, H, NQ + NV + 6)
    R_o, nobs_o, rew_o = treer_oracle.rollout(q0, v0, tgt, 0, act, gamma=0.99, dump=True, nthreads=4)
    assert np.abs(nobs - nobs_o).max() < 1e-8
    assert np.abs(rew - rew_o).max() < 1e-8 and np.abs(R - R_o).max() < 1e-7
    assert np.abs(np.linalg.norm(nobs[..., 3:7], axis=-1) - 1.0).max() < 1e-12
    p.close()


def test_rk4_tree_single_step_fp32_and_planning(treer, treer_oracle):
    rs = np.random.RandomState(11)
    for s in range(3):
        q, v, tgt = _start_quat(30 + s)
        a = rs.randn(NV) * 2
        q2, v2, obs, r = treer.env_step(q, v, tgt, a, 0)
        qo, vo, obs_o, ro, _ = treer_oracle.env_step(q, v, tgt, a)
        assert np.abs(q2 - qo).max() < 1e-12 and np.abs(v2 - vo).max() < 1e-11
        assert np.abs(obs - obs_o).max() < 1e-11 and abs(r - ro) < 1e-11
    K, H = 1024, 64
    q0, v0, tgt = _start_quat()
    p64 = _run(treer, "f64", K, H, q0, v0, tgt, 2024)
    p32 = _run(treer, "f32", K, H, q0, v0, tgt, 2024)
    o64, o32 = p64.next_observations()[..., :NQ + NV], p32.next_observations()[..., :NQ + NV]
    assert p32.status() == 0
    rel = (np.abs(o32 - o64) / (1.0 + np.abs(o64)))[:, -1].max(axis=1)
    print("fp32 rk4 tree error at t=64: median %.2e max %.2e" % (np.median(rel), rel.max()))
    assert np.median(rel) < 1e-4 and rel.max() < 1e-2
    p32.iterate(1)
    p32.advance()
    log = p32.read_log()
    assert log["total"] == 1 and abs(np.linalg.norm(log["qpos"][0, 3:7]) - 1.0) < 1e-14
    p64.close(); p32.close()
