"""Stored agents are read without running the pickle machine (trajopt_amd/agent_io.py, static_pickle.py)."""
import os
import pickle
import sys
import types

import numpy as np
import pytest

from trajopt_amd import agent_io, static_pickle

DENSE = "/root/reference/trajopt/sandbox/examples/dense_agent.pickle"


@pytest.mark.skipif(not os.path.exists(DENSE), reason="reference tree not mounted")
def test_reference_agent_fields(transitions):
    st = agent_io.read_agent_state(DENSE)
    assert st["H"] == 16 and st["paths_per_cpu"] == 40 and st["kappa"] == 25.0 and st["reward_type"] == "dense"
    assert len(st["sol_state"]) == 117 and len(st["sol_act"]) == 116 and st["act_sequence"].shape == (16, 7)
    d = transitions
    rows = np.where(d["file_id"] == 0)[0]
    for i in rows[::10]:
        k = int(d["step"][i])
        assert np.array_equal(st["sol_state"][k]["qp"], d["qp"][i]) and np.array_equal(st["sol_act"][k], d["act"][i])
        assert float(st["sol_reward"][k]) == d["reward"][i]


@pytest.mark.parametrize("protocol", [2, 3, 4, 5])
def test_modern_pickles_of_planner_like_objects(tmp_path, protocol):
    mod = types.ModuleType("fake_pkg.algos.mppi")

    class MPPI:
        pass
    MPPI.__module__, MPPI.__qualname__ = "fake_pkg.algos.mppi", "MPPI"
    mod.MPPI = MPPI
    names = ("fake_pkg", "fake_pkg.algos", "fake_pkg.algos.mppi")
    for n in names[:2]:
        sys.modules[n] = types.ModuleType(n)
    sys.modules[names[2]] = mod
    try:
        a = MPPI()
        rs = np.random.RandomState(0)
        a.H, a.paths_per_cpu, a.num_cpu, a.kappa, a.gamma, a.seed, a.default_act = 8, 4, 2, 3.5, 0.9, 7, "mean"
        a.mean, a.filter_coefs = np.zeros(7), [np.ones(7), 0.25, 0.8, 0.0]
        a.sol_state = [dict(qp=rs.randn(7), qv=rs.randn(7), target_pos=rs.randn(3), timestep=t) for t in range(3)]
        a.sol_act = [rs.randn(7) for _ in range(2)]
        a.sol_reward = [np.float64(-1.5), -2.0]
        a.sol_obs = [rs.randn(21) for _ in range(3)]
        a.act_sequence = rs.randn(8, 7).astype(np.float32).astype(np.float64)
        a.reward_type, a.reference = "tracking", rs.randn(20, 3)
        path = tmp_path / "agent.pickle"
        with open(path, "wb") as f:
            pickle.dump(a, f, protocol=protocol)
    finally:
        for n in names:
            del sys.modules[n]
    st = agent_io.read_agent_state(str(path))            # the class no longer exists anywhere: nothing is imported
    assert st["H"] == 8 and st["default_act"] == "mean" and st["kappa"] == 3.5
    assert np.array_equal(st["act_sequence"], a.act_sequence) and np.array_equal(st["reference"], a.reference)
    assert np.array_equal(st["sol_state"][2]["qp"], a.sol_state[2]["qp"]) and st["sol_state"][1]["timestep"] == 1
    assert float(st["sol_reward"][0]) == -1.5 and np.array_equal(st["filter_coefs"][0], np.ones(7))


def test_nothing_in_the_file_is_executed(tmp_path):
    class Boom:
        def __reduce__(self):
            return (os.system, ("echo executed > %s" % (tmp_path / "proof"),))
    path = tmp_path / "evil.pickle"
    with open(path, "wb") as f:
        pickle.dump({"H": 1, "x": Boom()}, f)
    node = static_pickle.read(str(path))
    assert not (tmp_path / "proof").exists()
    assert node["x"]["__class__"] in ("posix.system", "os.system", "nt.system")
    with pytest.raises(ValueError):
        agent_io.read_agent_state(str(path))
