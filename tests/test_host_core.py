"""The product's per-sample step (csrc/dyn_core.hpp, env_core.hpp) compiled for the host with g++
and checked against the oracle and the reference's stored transitions -- the same arithmetic
the GPU lanes execute (link-local formulation), without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_core_harness.cpp")
OUT = os.path.join(ROOT, "tests", "_host_core.so")


@pytest.fixture(scope="module")
def core():
    deps = [SRC] + [os.path.join(ROOT, "trajopt_amd", "csrc", f) for f in ("dyn_core.hpp", "env_core.hpp", "models_gen.hpp")]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", OUT, SRC])
    L = C.CDLL(OUT)

    def run(f64, q0, v0, tgt, act, rt=0, ref=None, t0=0, alpha=1.0, gamma=1.0, model=0):
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        q0, v0, tgt, act = (np.ascontiguousarray(x, dtype=np.float64) for x in (q0, v0, tgt, act))
        H = act.shape[0]
        nv = act.shape[1]
        q, v, hand, rew = np.zeros(len(q0)), np.zeros(nv), np.zeros(3), np.zeros(H)
        R, sw = C.c_double(), C.c_int()
        refp = None if ref is None else p(np.ascontiguousarray(ref, dtype=np.float64))
        L.core_rollout(int(f64), model, p(q0), p(v0), p(tgt), p(act), H, rt, refp, t0, C.c_double(alpha),
                       C.c_double(gamma), p(q), p(v), p(hand), p(rew), C.byref(R), C.byref(sw))
        return q, v, hand, rew, R.value, sw.value
    return run


def test_single_steps_against_stored_transitions(core, transitions):
    d = transitions
    w64 = w32 = 0.0
    for i in range(0, len(d["reward"]), 2):
        q, v, hand, _, _, sw = core(1, d["qp"][i], d["qv"][i], d["target"][i], d["act"][i][None])
        w64 = max(w64, np.abs(q - d["qp_next"][i]).max(), np.abs(v - d["qv_next"][i]).max(),
                  np.abs(hand - d["obs_next"][i][14:17]).max())
        assert sw <= 4
        q, v, _, _, _, _ = core(0, d["qp"][i], d["qv"][i], d["target"][i], d["act"][i][None])
        w32 = max(w32, np.abs(q - d["qp_next"][i]).max(), (np.abs(v - d["qv_next"][i]) / (1 + np.abs(d["qv_next"][i]))).max())
    assert w64 < 1e-10, w64
    assert w32 < 1e-4, w32


@pytest.mark.parametrize("H", [32, 64])
def test_rollouts_against_oracle(core, oracle_sim, H):
    from oracle import ref_mppi
    act = ref_mppi.filter_noise(ref_mppi.raw_normals(9, 60, H, 7), [np.ones(7), 0.25, 0.8, 0.0])
    tgt = np.array([0.05, 0.0, -0.05])
    R_o, nobs_o, rew_o = oracle_sim.rollout(np.zeros(7), np.zeros(7), tgt, 0, act, gamma=0.97, dump=True)
    e64 = e32 = r32 = 0.0
    for k in range(len(act)):
        q, v, hand, rew, R, _ = core(1, np.zeros(7), np.zeros(7), tgt, act[k], gamma=0.97)
        e64 = max(e64, np.abs(q - nobs_o[k, -1, :7]).max(), np.abs(v - nobs_o[k, -1, 7:14]).max(), abs(R - R_o[k]),
                  np.abs(rew - rew_o[k]).max(), np.abs(hand - nobs_o[k, -1, 14:17]).max())
        q, v, _, _, R, _ = core(0, np.zeros(7), np.zeros(7), tgt, act[k], gamma=0.97)
        e32 = max(e32, (np.abs(np.r_[q, v] - nobs_o[k, -1, :14]) / (1 + np.abs(nobs_o[k, -1, :14]))).max())
        r32 = max(r32, abs(R - R_o[k]) / abs(R_o[k]))
    assert e64 < 1e-9, e64
    assert e32 < 1e-4 and r32 < 1e-4, (e32, r32)     # the fp32 tolerance of the planner


def test_reward_types_and_free_model(core, oracle_sim):
    from oracle.sim import OracleSim
    from trajopt_amd.model import get_model
    rs = np.random.RandomState(4)
    act = rs.randn(10, 7)
    ref = rs.randn(40, 3) * 0.2
    q0, v0, tgt = rs.uniform(-1, 1, 7), rs.randn(7), np.array([0.1, 0.1, 0.0])
    for rt, name in enumerate(("dense", "sparse", "tracking", "cooling")):
        _, _, rew_o = oracle_sim.rollout(q0, v0, tgt, 3, act[None], name, ref, 0.4, dump=True)
        rew = core(1, q0, v0, tgt, act, rt=rt, ref=ref, t0=3, alpha=0.4)[3]
        assert np.abs(rew - rew_o[0]).max() < 1e-10, name
    free = OracleSim(get_model("reacher_7dof_free"))
    R_o, nobs_o, _ = free.rollout(q0, v0, tgt, 0, act[None] * 3, dump=True)
    q, v, _, _, R, sw = core(1, q0, v0, tgt, act * 3, model=1)
    assert sw == 0 and np.abs(q - nobs_o[0, -1, :7]).max() < 1e-10 and abs(R - R_o[0]) < 1e-10


def test_tree_model_slides_and_gravity(core):
    """18-DoF floating-base tree (slides, branching, gravity): the link-local kernel arithmetic
    against the world-frame oracle, whose bias forces are themselves checked against a finite-
    difference Lagrangian in test_oracle_tree.py."""
    from oracle.sim import OracleSim
    from trajopt_amd.model import get_model
    sim = OracleSim(get_model("tree_18dof"))
    rs = np.random.RandomState(11)
    H = 64
    q0, v0, tgt = rs.randn(18) * 0.4, rs.randn(18) * 0.5, np.array([0.3, 0.1, 1.0])
    act = rs.randn(H, 18)
    R_o, nobs_o, rew_o = sim.rollout(q0, v0, tgt, 0, act[None], gamma=0.99, dump=True)
    q, v, hand, rew, R, sw = core(1, q0, v0, tgt, act, gamma=0.99, model=2)
    assert sw == 0
    assert np.abs(q - nobs_o[0, -1, :18]).max() < 1e-9 and np.abs(v - nobs_o[0, -1, 18:36]).max() < 1e-9
    assert np.abs(rew - rew_o[0]).max() < 1e-9 and abs(R - R_o[0]) < 1e-8
    assert np.abs(hand - nobs_o[0, -1, 36:39]).max() < 1e-10
    q, v, _, _, R32, _ = core(0, q0, v0, tgt, act, gamma=0.99, model=2)
    assert (np.abs(np.r_[q, v] - nobs_o[0, -1, :36]) / (1 + np.abs(nobs_o[0, -1, :36]))).max() < 2e-3



def test_free_joint_with_quaternion(core):
    """Ball joint in the link-local kernel arithmetic (rotation matrix from the quaternion, body-fixed axes,
    quaternion integration) against the world-frame oracle, on a tumbling, falling tree."""
    from oracle.sim import OracleSim
    from trajopt_amd.model import get_model
    md = get_model("tree_18dof_quat")
    sim = OracleSim(md)
    rs = np.random.RandomState(12)
    H = 64
    q0 = md.qpos0()
    q0[:3] = rs.randn(3) * 0.2
    quat = rs.randn(4)
    q0[3:7] = quat / np.linalg.norm(quat)
    q0[7:] = rs.randn(12) * 0.4
    v0, tgt = rs.randn(18) * 0.8, np.array([0.3, 0.1, 1.0])
    act = rs.randn(H, 18)
    R_o, nobs_o, rew_o = sim.rollout(q0, v0, tgt, 0, act[None], gamma=0.99, dump=True)
    q, v, hand, rew, R, sw = core(1, q0, v0, tgt, act, gamma=0.99, model=3)
    assert np.abs(q - nobs_o[0, -1, :19]).max() < 1e-9 and np.abs(v - nobs_o[0, -1, 19:37]).max() < 1e-9
    assert np.abs(rew - rew_o[0]).max() < 1e-9 and abs(R - R_o[0]) < 1e-8
    assert np.abs(hand - nobs_o[0, -1, 37:40]).max() < 1e-10 and abs(np.linalg.norm(q[3:7]) - 1) < 1e-14
    q, v, _, _, R32, _ = core(0, q0, v0, tgt, act, gamma=0.99, model=3)
    assert (np.abs(np.r_[q, v] - nobs_o[0, -1, :37]) / (1 + np.abs(nobs_o[0, -1, :37]))).max() < 2e-3


def test_runge_kutta_integrator(core):
    """Four stages through one copy of the dynamics, positions advanced on the manifold, end-effector of the
    fourth stage: the kernel arithmetic against the oracle's restatement of the same integrator."""
    from oracle.sim import OracleSim
    from trajopt_amd.model import get_model
    md = get_model("tree_18dof_rk4")
    sim = OracleSim(md)
    rs = np.random.RandomState(13)
    H = 48
    q0 = md.qpos0()
    q0[:3] = rs.randn(3) * 0.2
    quat = rs.randn(4)
    q0[3:7] = quat / np.linalg.norm(quat)
    q0[7:] = rs.randn(12) * 0.4
    v0, tgt = rs.randn(18) * 0.8, np.array([0.3, 0.1, 1.0])
    act = rs.randn(H, 18)
    R_o, nobs_o, rew_o = sim.rollout(q0, v0, tgt, 0, act[None], gamma=0.99, dump=True)
    q, v, hand, rew, R, sw = core(1, q0, v0, tgt, act, gamma=0.99, model=4)
    assert np.abs(q - nobs_o[0, -1, :19]).max() < 1e-9 and np.abs(v - nobs_o[0, -1, 19:37]).max() < 1e-9
    assert np.abs(rew - rew_o[0]).max() < 1e-9 and abs(R - R_o[0]) < 1e-8
    assert np.abs(hand - nobs_o[0, -1, 37:40]).max() < 1e-10 and abs(np.linalg.norm(q[3:7]) - 1) < 1e-14
    q, v, _, _, R32, _ = core(0, q0, v0, tgt, act, gamma=0.99, model=4)
    assert (np.abs(np.r_[q, v] - nobs_o[0, -1, :37]) / (1 + np.abs(nobs_o[0, -1, :37]))).max() < 2e-3
