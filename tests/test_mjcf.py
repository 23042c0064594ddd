"""The MJCF front end (trajopt_amd/mjcf.py): the reference's own XML asset compiles to the same flat
model as the hand-written spec the kernels were generated from, and small inline models behave
like their closed forms under the oracle."""
import os

import numpy as np
import pytest

from trajopt_amd.mjcf import load_mjcf
from trajopt_amd.model import compile_model, get_model, mass_matrix

SAWYER = "/root/reference/trajopt/envs/assets/sawyer.xml"

PLANAR = """
<mujoco model="planar">
  <compiler angle="radian" coordinate="local" inertiafromgeom="true"/>
  <option timestep="0.001" gravity="0 0 -9.81" integrator="Euler"/>
  <default><geom contype="0" conaffinity="0"/></default>
  <worldbody>
    <body name="upper" pos="0 0 0">
      <joint name="j0" type="hinge" axis="0 1 0"/>
      <geom type="sphere" size="0.05" pos="0 0 -0.5"/>
      <body name="lower" pos="0 0 -0.5">
        <joint name="j1" type="hinge" axis="0 1 0"/>
        <geom type="sphere" size="0.04" pos="0 0 -0.4"/>
        <site name="finger" pos="0 0 -0.4"/>
      </body>
    </body>
  </worldbody>
  <actuator><motor joint="j0" gear="1" ctrlrange="-1 1" ctrllimited="true"/></actuator>
</mujoco>
"""


@pytest.mark.skipif(not os.path.exists(SAWYER), reason="reference tree not mounted")
def test_reference_asset_compiles_to_the_kernel_model():
    md = compile_model(load_mjcf(SAWYER, "reacher_7dof", frame_skip=2))
    want = get_model("reacher_7dof")
    for f in ("nv", "h", "frame_skip", "contact_link", "hand_link", "contact_radius", "contact_plane_z",
              "contact_margin", "contact_invweight", "solref_k", "solref_b"):
        assert getattr(md, f) == getattr(want, f), f
    for f in ("parent", "pos", "axis", "mass", "com", "inertia", "armature", "damping", "limited", "range_lo",
              "range_hi", "gear", "ctrl_lo", "ctrl_hi", "dof_invweight", "contact_pos", "hand_pos", "solimp",
              "jtype", "gravity"):
        assert np.array_equal(getattr(md, f), getattr(want, f)), f


def test_planar_double_pendulum_matches_closed_form():
    from oracle.sim import OracleSim
    md = compile_model(load_mjcf(PLANAR, "planar", frame_skip=1))
    assert md.nv == 2 and list(md.parent) == [-1, 0] and md.hand_link == 1 and md.contact_link < 0
    assert md.gear[0] == 1.0 and md.gear[1] == 0.0 and not md.limited.any()
    m1, m2 = md.mass
    l1, l2 = 0.5, 0.4
    i1, i2 = 0.4 * m1 * 0.05 ** 2, 0.4 * m2 * 0.04 ** 2          # solid spheres about their centres
    sim = OracleSim(md)
    rs = np.random.RandomState(0)
    for _ in range(5):
        q, v = rs.uniform(-2, 2, 2), rs.randn(2)
        M, bias = sim.mass_matrix(q, v)
        c2, s2 = np.cos(q[1]), np.sin(q[1])
        M_cf = np.array([[m1 * l1 ** 2 + i1 + m2 * (l1 ** 2 + l2 ** 2 + 2 * l1 * l2 * c2) + i2, m2 * (l2 ** 2 + l1 * l2 * c2) + i2],
                         [m2 * (l2 ** 2 + l1 * l2 * c2) + i2, m2 * l2 ** 2 + i2]])
        assert np.abs(M - M_cf).max() < 1e-12
        g = 9.81
        # rotation about +y by q swings the link tip from -z towards -x: height of tip = -l cos q
        G = np.array([(m1 + m2) * g * l1 * np.sin(q[0]) + m2 * g * l2 * np.sin(q[0] + q[1]), m2 * g * l2 * np.sin(q[0] + q[1])])
        h = -m2 * l1 * l2 * s2
        C = np.array([h * (2 * v[0] * v[1] + v[1] ** 2), -h * v[0] ** 2])
        assert np.abs(bias - (C + G)).max() < 1e-11
    assert np.abs(mass_matrix(md, q)[0] - M).max() < 1e-12


def test_unsupported_features_are_rejected():
    for bad, what in ((PLANAR.replace('integrator="Euler"', 'integrator="implicit"'), "integrator"),
                      (PLANAR.replace('<body name="lower" pos="0 0 -0.5">', '<body name="lower" pos="0 0 -0.5" quat="1 0 0 0">'), "orientation"),
                      (PLANAR.replace('name="j1" type="hinge"', 'name="j1" type="screw"'), "joint type"),
                      (PLANAR.replace('name="j0" type="hinge" axis="0 1 0"', 'name="j0" type="ball"'), "motors on ball"),
                      (PLANAR.replace('angle="radian"', 'angle="degree"'), "radian"),
                      (PLANAR.replace('joint="j0"', 'joint="nope"'), "unknown joints")):
        with pytest.raises(ValueError, match=what):
            load_mjcf(bad, "x", 1)


FLOATING = """
<mujoco model="floating">
  <compiler angle="radian" coordinate="local" inertiafromgeom="true"/>
  <option timestep="0.0005" gravity="0 0 -9.81" integrator="Euler"/>
  <default><geom contype="0" conaffinity="0"/></default>
  <worldbody>
    <body name="hull" pos="0 0 1">
      <freejoint/>
      <geom type="capsule" size="0.05" fromto="-0.2 0 0 0.2 0 0"/>
      <geom type="sphere" size="0.06" pos="0.05 0.1 0.02"/>
      <body name="pendant" pos="0.2 0 0">
        <joint name="b" type="ball"/>
        <geom type="capsule" size="0.03" fromto="0 0 0 0 0.1 -0.3"/>
        <site name="finger" pos="0 0.1 -0.3"/>
      </body>
    </body>
  </worldbody>
</mujoco>
"""


def test_free_and_ball_joints_from_xml():
    """<freejoint/> and type="ball": position + quaternion coordinates, and under the oracle the total
    linear momentum follows gravity while the spin angular momentum about the centre of mass is kept."""
    import dataclasses
    from oracle.sim import OracleSim
    from test_oracle_tree import _momenta
    md = compile_model(load_mjcf(FLOATING, "floating", frame_skip=1))
    assert md.nv == 9 and md.nq == 11
    assert list(md.jtype) == [1, 1, 1, 2, 3, 4, 2, 3, 4] and list(md.qadr) == [0, 1, 2, 3, 3, 3, 7, 7, 7]
    q0 = md.qpos0()
    assert np.array_equal(q0, [0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0])
    rs = np.random.RandomState(2)
    q = q0.copy()
    for a in (3, 7):
        quat = rs.randn(4)
        q[a:a + 4] = quat / np.linalg.norm(quat)
    v = rs.randn(md.nv)
    sim = OracleSim(md)
    mtot = md.mass.sum()
    P0, L0 = _momenta(md, q, v)
    n = 400
    for k in range(n):
        q, v, _, _ = sim.substep(q, v, np.zeros(md.nv))
    P1, L1 = _momenta(md, q, v)
    want = P0 + mtot * md.gravity * md.h * n
    assert np.abs(P1 - want).max() < 2e-3 * np.abs(want).max()
    # without gravity the angular momentum about the origin is kept (first order in h)
    md0 = dataclasses.replace(md, gravity=np.zeros(3))
    sim = OracleSim(md0)
    P0, L0 = _momenta(md0, q, v)
    for k in range(n):
        q, v, _, _ = sim.substep(q, v, np.zeros(md.nv))
    P1, L1 = _momenta(md0, q, v)
    assert np.abs(L1 - L0).max() < 5e-3 * np.abs(L0).max() and np.abs(P1 - P0).max() < 1e-3 * np.abs(P0).max()
    assert abs(np.linalg.norm(q[3:7]) - 1) < 1e-12 and abs(np.linalg.norm(q[7:11]) - 1) < 1e-12
