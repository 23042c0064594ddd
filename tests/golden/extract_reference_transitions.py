"""Extract the MuJoCo transitions stored in the reference's pickled MPPI agents.

Test infrastructure; runs only where the reference tree is mounted.  The files
are read with ``static_pickle`` (a disassembler-based reader that executes
nothing from them).  Every stored agent holds ``sol_state[k]`` (qp, qv,
target_pos, timestep), ``sol_act[k]``, ``sol_reward[k]`` and ``sol_obs[k]``;
consecutive entries are one real ``env.step`` of the reference's simulator:

    (sol_state[k], sol_act[k]) -> (sol_state[k+1], sol_obs[k+1], sol_reward[k])

Sources (relative to the reference root):
    trajopt/sandbox/examples/dense_agent.pickle, 116_agent.pickle,
    trajopt/sandbox/examples/ac_tests/*/*.pickle

Usage:
    python tests/golden/extract_reference_transitions.py [reference_root] [out.npz]
writes every transition (not committed; ~7 MB).  ``make_transition_fixture.py``
picks the committed subset.
"""
import glob
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import static_pickle  # noqa: E402

# its recorded sol_act are not the actions that were applied (94 transitions
# inconsistent with every other file); excluded from the golden set
EXCLUDED = ("ac_tests/ac_sparse_005_no_init/ac_test_agent_99.pickle",)


def agent_files(root):
    ex = os.path.join(root, "trajopt", "sandbox", "examples")
    files = [os.path.join(ex, "dense_agent.pickle"), os.path.join(ex, "116_agent.pickle")]
    files += sorted(glob.glob(os.path.join(ex, "ac_tests", "*", "*.pickle")))
    return files


def load_all(root="/root/reference"):
    cols = {k: [] for k in ("qp", "qv", "target", "timestep", "act", "qp_next", "qv_next",
                            "obs_next", "reward", "file_id", "step")}
    names, configs = [], []
    for fid, path in enumerate(agent_files(root)):
        rel = os.path.relpath(path, os.path.join(root, "trajopt", "sandbox", "examples"))
        names.append(rel)
        st = static_pickle.read(path)["__state__"]
        configs.append(dict(H=st["H"], paths_per_cpu=st["paths_per_cpu"], num_cpu=st["num_cpu"],
                            kappa=st["kappa"], gamma=st["gamma"], seed=st["seed"],
                            default_act=st["default_act"], reward_type=st.get("reward_type")))
        if rel in EXCLUDED:
            continue
        S, A, R, O = st["sol_state"], st["sol_act"], st["sol_reward"], st["sol_obs"]
        for k in range(len(A)):
            cols["qp"].append(S[k]["qp"]); cols["qv"].append(S[k]["qv"])
            cols["target"].append(np.asarray(S[k]["target_pos"], dtype=np.float64))
            cols["timestep"].append(int(S[k]["timestep"]))
            cols["act"].append(A[k])
            cols["qp_next"].append(S[k + 1]["qp"]); cols["qv_next"].append(S[k + 1]["qv"])
            cols["obs_next"].append(O[k + 1][:20])
            cols["reward"].append(float(R[k]))
            cols["file_id"].append(fid); cols["step"].append(k)
    out = {k: np.asarray(v) for k, v in cols.items()}
    out["file_names"] = np.asarray(names)
    return out, configs


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    dst = sys.argv[2] if len(sys.argv) > 2 else "/tmp/reference_transitions_all.npz"
    data, configs = load_all(root)
    np.savez(dst, **data)
    print("files", len(data["file_names"]), "transitions", len(data["reward"]), "->", dst)
