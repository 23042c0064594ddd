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
