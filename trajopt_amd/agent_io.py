"""Stored agents: reading the agent pickles the reference's scripts write (`pickle.dump(agent, ...)`,
trajopt/sandbox/examples/reacher_7dof_mppi.py:63-64) for the replay flow of
trajopt/sandbox/examples/play_mppi_agent.py:22-31, without running the pickle machine on them.

A pickle names classes to import and call; a file from somewhere else is code.  `load_agent` therefore
never unpickles: it reads the opcode stream with the disassembler-based reader of `static_pickle`, takes
the planner's attribute dictionary out of it as plain numbers, lists and numpy arrays, and puts those on a
fresh `MPPI` of this package bound to an environment made here.  Agents written by this package's own
`MPPI` (which pickles without device handles) load the same way.
"""
import numpy as np

from . import static_pickle

# planner attributes that are data (everything else in a stored agent -- the env object -- is rebuilt)
_FIELDS = ("H", "paths_per_cpu", "num_cpu", "kappa", "gamma", "seed", "default_act", "n", "m", "mean", "filter_coefs",
           "sol_state", "sol_act", "sol_reward", "sol_obs", "act_sequence", "reward_type", "reference", "noise", "dtype")


def read_agent_state(path):
    """Attribute dictionary of the stored planner: plain data only."""
    node = static_pickle.read(path)
    if not (isinstance(node, dict) and isinstance(node.get("__state__"), dict)):
        raise ValueError("%s does not hold a pickled object with an attribute dictionary" % (path,))
    cls = node.get("__class__") or ""
    if not cls.endswith(".MPPI"):
        raise ValueError("%s holds a %s, not an MPPI planner" % (path, cls))
    st = node["__state__"]
    return {k: st[k] for k in _FIELDS if k in st}


def load_agent(path, env=None, env_name="reacher_7dof"):
    """The stored planner as an `MPPI` of this package on `env` (made with `get_environment` when not given):
    `sol_state`, `sol_act`, `sol_reward`, `sol_obs`, `act_sequence` and the constructor arguments come from the
    file, so `agent.sol_state[k]` / `agent.sol_act[k]` replay and `agent.train_step()` continues the run."""
    from .algos.mppi import MPPI
    from .envs.utils import get_environment
    st = read_agent_state(path)
    if env is None:
        env = get_environment(env_name, reward_type=st.get("reward_type") or "dense", reference=st.get("reference"))
        env.reset_model()
    fc = st.get("filter_coefs")
    agent = MPPI(env, H=int(st["H"]), paths_per_cpu=int(st["paths_per_cpu"]), num_cpu=int(st.get("num_cpu", 1)),
                 kappa=float(st.get("kappa", 1.0)), gamma=float(st.get("gamma", 1.0)),
                 mean=None if st.get("mean") is None else np.asarray(st["mean"], float),
                 filter_coefs=None if fc is None else [np.asarray(fc[0], float), float(fc[1]), float(fc[2]), float(fc[3])],
                 default_act=st.get("default_act", "repeat"), seed=int(st.get("seed", 123)),
                 reward_type=st.get("reward_type") or "dense", reference=st.get("reference"),
                 noise=st.get("noise", "host"), dtype=st.get("dtype", "f32"))

    def state(d):
        return dict(qp=np.asarray(d["qp"], float), qv=np.asarray(d["qv"], float),
                    target_pos=np.asarray(d["target_pos"], float), timestep=int(d["timestep"]))
    agent.sol_state = [state(d) for d in st["sol_state"]]
    agent.sol_act = [np.asarray(a, float) for a in st["sol_act"]]
    agent.sol_reward = [float(r) for r in st["sol_reward"]]
    agent.sol_obs = [np.asarray(o, float) for o in st["sol_obs"]]
    if st.get("act_sequence") is not None:
        agent.act_sequence = np.array(st["act_sequence"], dtype=float)
    return agent
