"""Physics checks of the oracle on the synthetic 18-DoF tree (slide joints, branching, gravity).
The reference holds no data for such a model, so the oracle is checked against first principles:
a numpy composite-body mass matrix, a finite-difference Lagrangian for the bias forces, and
energy conservation without damping."""
import dataclasses

import numpy as np
import pytest

from oracle.sim import OracleSim
from trajopt_amd.model import get_model, mass_matrix

NV = 18


@pytest.fixture(scope="module")
def md():
    return get_model("tree_18dof")


def _potential(md, q):
    _, (R, p) = mass_matrix(md, q)
    return -sum(md.mass[i] * md.gravity @ (p[i] + R[i] @ md.com[i]) for i in range(md.nv))


def test_structure(md):
    assert md.nv == NV and list(md.jtype[:6]) == [1, 1, 1, 0, 0, 0] and not md.limited.any() and md.contact_link < 0
    assert sorted(np.bincount(md.parent[md.parent >= 0])) [-1] == 4      # four limbs leave the trunk link
    assert md.mass[:5].sum() == 0.0 and md.mass[5] > 10.0


def test_mass_matrix_symmetric_positive_definite(md):
    sim = OracleSim(md)
    rs = np.random.RandomState(0)
    for _ in range(5):
        q = rs.randn(NV)
        M, _ = sim.mass_matrix(q)
        assert np.abs(M - M.T).max() < 1e-12 and np.linalg.eigvalsh(M).min() > 1e-3
        assert np.abs(M - mass_matrix(md, q)[0]).max() < 1e-12
        # translating the base moves the whole mass
        assert abs(M[0, 0] - md.mass.sum()) < 1e-10 and abs(M[2, 2] - md.mass.sum()) < 1e-10


def test_bias_forces_equal_lagrangian(md):
    sim = OracleSim(md)
    rs = np.random.RandomState(1)
    q, v = rs.randn(NV) * 0.5, rs.randn(NV)
    _, bias = sim.mass_matrix(q, v)
    eps = 1e-6
    c = np.zeros(NV)
    for k in range(NV):
        d = np.zeros(NV)
        d[k] = eps
        dM = (mass_matrix(md, q + d)[0] - mass_matrix(md, q - d)[0]) / (2 * eps)
        c += dM @ v * v[k]
        c[k] += -0.5 * v @ dM @ v + (_potential(md, q + d) - _potential(md, q - d)) / (2 * eps)
    assert np.abs(c - bias).max() < 1e-6 * np.abs(bias).max()


def test_energy_is_conserved_without_damping(md):
    """The integrator is first order: the energy drift over a fixed time span shrinks with the step."""
    rs = np.random.RandomState(2)
    q0, v0 = rs.randn(NV) * 0.3, rs.randn(NV) * 0.5
    drift = []
    for h, n in ((1e-3, 500), (2.5e-4, 2000)):
        md0 = dataclasses.replace(md, damping=np.zeros(NV), h=h)
        sim = OracleSim(md0)
        energy = lambda q, v: 0.5 * v @ mass_matrix(md0, q)[0] @ v + _potential(md0, q)
        q, v = q0, v0
        e0 = energy(q, v)
        for _ in range(n):
            q, v, _, _ = sim.substep(q, v, np.zeros(NV))
        drift.append(abs(energy(q, v) - e0) / abs(e0))
    assert drift[1] < 2e-3 and drift[1] < 0.4 * drift[0], drift


def test_free_fall_of_the_base(md):
    """With every joint at rest and no actuation the trunk accelerates at g (first step)."""
    sim = OracleSim(md)
    q, v, _, _ = sim.substep(np.zeros(NV), np.zeros(NV), np.zeros(NV))
    assert abs(v[2] - md.gravity[2] * md.h) < 1e-12 and np.abs(v[[0, 1]]).max() < 1e-12


# ---- free joint with a quaternion (tree_18dof_quat) ---------------------------------------------------
def _momenta(md, q, v):
    """Linear momentum and angular momentum about the world origin from first principles."""
    from trajopt_amd.model import SLIDE
    _, (R, p) = mass_matrix(md, q)
    P, L = np.zeros(3), np.zeros(3)
    om, vo = [None] * md.nv, [None] * md.nv          # angular velocity, velocity of the body point at the world origin
    for i in range(md.nv):
        pi = md.parent[i]
        w = np.zeros(3) if pi < 0 else om[pi].copy()
        u = np.zeros(3) if pi < 0 else vo[pi].copy()
        a = R[i] @ md.axis[i]
        if md.jtype[i] == SLIDE:
            u += a * v[i]
        else:
            w += a * v[i]
            u += np.cross(p[i], a) * v[i]
        om[i], vo[i] = w, u
        c = p[i] + R[i] @ md.com[i]
        cdot = u + np.cross(w, c)
        Ic = R[i] @ md.link_inertia_matrix(i) @ R[i].T
        P += md.mass[i] * cdot
        L += Ic @ w + md.mass[i] * np.cross(c, cdot)
    return P, L


@pytest.fixture(scope="module")
def mdq():
    return get_model("tree_18dof_quat")


def _random_quat_state(md, rs):
    q = md.qpos0()
    q[:3] = rs.randn(3) * 0.2
    quat = rs.randn(4)
    q[3:7] = quat / np.linalg.norm(quat)
    q[7:] = rs.randn(md.nq - 7) * 0.4
    return q, rs.randn(md.nv) * 0.8


def test_quat_structure(mdq):
    assert mdq.nv == NV and mdq.nq == NV + 1 and list(mdq.jtype[:6]) == [1, 1, 1, 2, 3, 4]
    assert list(mdq.qadr[:7]) == [0, 1, 2, 3, 3, 3, 7] and np.array_equal(mdq.qpos0()[3:7], [1, 0, 0, 0])


def test_free_flight_conserves_momentum_and_energy(mdq):
    """No gravity, no damping, no actuation: linear and angular momentum are conserved by the dynamics; this
    pins the velocity-product terms of the ball joint (body-fixed axes) with a tumbling trunk and moving limbs."""
    rs = np.random.RandomState(4)
    q0, v0 = _random_quat_state(mdq, rs)
    drift = []
    for h, n in ((4e-4, 500), (1e-4, 2000)):
        md0 = dataclasses.replace(mdq, damping=np.zeros(NV), gravity=np.zeros(3), h=h)
        sim = OracleSim(md0)
        q, v = q0.copy(), v0.copy()
        P0, L0 = _momenta(md0, q, v)
        e0 = 0.5 * v @ mass_matrix(md0, q)[0] @ v
        for _ in range(n):
            q, v, _, _ = sim.substep(q, v, np.zeros(NV))
        P1, L1 = _momenta(md0, q, v)
        e1 = 0.5 * v @ mass_matrix(md0, q)[0] @ v
        assert abs(np.linalg.norm(q[3:7]) - 1.0) < 1e-12
        drift.append((np.abs(P1 - P0).max() / np.abs(P0).max(), np.abs(L1 - L0).max() / np.abs(L0).max(), abs(e1 - e0) / e0))
    print(drift)
    assert drift[1][0] < 1e-4 and drift[1][0] < 0.4 * drift[0][0], drift
    assert drift[1][1] < 2e-3 and drift[1][1] < 0.4 * drift[0][1], drift
    assert drift[1][2] < 2e-3 and drift[1][2] < 0.4 * drift[0][2], drift


def test_quat_base_equals_hinge_base_for_single_axis_spin(md, mdq):
    """With the trunk upright and spinning about one body axis only, three successive hinges and the ball joint
    describe the same motion with the same coordinates: identical M, bias and first step."""
    rs = np.random.RandomState(6)
    sq, se = OracleSim(mdq), OracleSim(md)
    for axis in range(3):
        qe = np.r_[rs.randn(3) * 0.2, 0.0, 0.0, 0.0, rs.randn(12) * 0.4]
        ve = np.r_[rs.randn(3), 0.0, 0.0, 0.0, rs.randn(12)]
        ve[3 + axis] = 0.7
        qq = np.r_[qe[:3], 1.0, 0.0, 0.0, 0.0, qe[6:]]
        Mq, bq = sq.mass_matrix(qq, ve)
        Me, be = se.mass_matrix(qe, ve)
        assert np.abs(Mq - Me).max() < 1e-12 and np.abs(bq - be).max() < 1e-11
        u = rs.randn(NV)
        q2, v2, hq, _ = sq.substep(qq, ve, u)
        e2, w2, he, _ = se.substep(qe, ve, u)
        assert np.abs(v2 - w2).max() < 1e-11 and np.abs(hq - he).max() < 1e-13
        assert np.abs(q2[:3] - e2[:3]).max() < 1e-14 and np.abs(q2[7:] - e2[6:]).max() < 1e-14


def test_quaternion_integration_of_a_steady_spin(mdq):
    """A lone rigid trunk spinning about a principal axis keeps its rate; the quaternion turns by w t."""
    md0 = dataclasses.replace(mdq, gravity=np.zeros(3), damping=np.zeros(NV))
    sim = OracleSim(md0)
    q, v = md0.qpos0(), np.zeros(NV)
    # fold the limbs' motion away by starting them at rest: spin about the trunk's x axis couples to them only
    # through centrifugal terms, so compare the angle after one step instead of a long run
    v[3] = 2.0
    q2, v2, _, _ = sim.substep(q, v, np.zeros(NV))
    ang = 2.0 * np.arctan2(np.linalg.norm(q2[4:7]), q2[3])
    w_new = np.linalg.norm(v2[3:6])
    assert abs(ang - md0.h * w_new) < 1e-12 and abs(np.linalg.norm(q2[3:7]) - 1.0) < 1e-15
