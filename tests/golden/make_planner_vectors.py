"""Generate known-answer vectors for the planner arithmetic by running the
reference's own pure-numpy functions.

Runs only where the reference tree is mounted (never on the GPU box).  The
reference modules import a simulator stack that is not installed; the modules
named below are replaced by empty stand-ins so that the *numpy-only* functions
(`generate_perturbed_actions`, `generate_paths` seeding, `gather_paths_parallel`
ordering, `MPPI.update`, `MPPI.score_trajectory`, `MPPI.advance_time`,
`MPPI.train_step` control flow, `stack_tensor_dict_list`, `discount_sum`) can be
imported and called.  No simulator arithmetic is produced here.

    python tests/golden/make_planner_vectors.py [reference_root]
writes tests/golden/planner_vectors.npz
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from toy_env import ToyEnv  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference(root):
    class _Any:
        def __init__(self, *a, **k):
            pass

    class _Err(Exception):
        pass
    gym = _stub("gym", Env=_Any)
    gym.utils = _stub("gym.utils", EzPickle=_Any)
    gym.utils.seeding = _stub("gym.utils.seeding", np_random=lambda seed=None: (np.random.RandomState(seed), seed))
    gym.error = _stub("gym.error", DependencyNotInstalled=_Err)
    gym.spaces = _stub("gym.spaces", Box=_Any)
    _stub("mujoco_py", load_model_from_path=None, MjSim=_Any, MjViewer=_Any)
    for name in ("pybullet_envs", "pybullet_envs.deep_mimic", "pybullet_envs.deep_mimic.gym_env",
                 "pybullet_envs.deep_mimic.gym_env.deep_mimic_env"):
        _stub(name, HumanoidDeepMimicBackflipBulletEnv=_Any, HumanoidDeepBulletEnv=_Any)
    sys.path.insert(0, root)
    import trajopt.utils as ru
    import trajopt.algos.mppi as rm
    import trajopt.tensor_utils as rt
    return ru, rm, rt


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    ru, rm, rt = import_reference(root)
    out = {}

    # --- filtered perturbations, seeding and draw order (utils.py:118-141) -----------------
    cases = [
        ("p0", 123, 32, 7, np.ones(7), (0.25, 0.8, 0.0), 5, 0.0),
        ("p1", 7, 5, 3, np.array([0.5, 1.0, 2.0]), (0.5, 0.3, 0.2), 4, 0.3),    # odd H*m: cached variate carries
        ("p2", 2024, 1, 7, np.ones(7), (0.25, 0.8, 0.0), 3, -1.0),
        ("p3", 11, 2, 2, np.array([1.0, 3.0]), (0.25, 0.8, 0.1), 3, 0.0),
        ("p4", 99, 64, 7, 0.7 * np.ones(7), (1.0, 0.0, 0.0), 2, 0.0),
    ]
    for tag, seed, H, m, sigma, betas, N, fill in cases:
        base = np.full((H, m), fill) + np.arange(H * m).reshape(H, m) * 0.01
        np.random.seed(seed)
        acts = [ru.generate_perturbed_actions(base, [sigma, *betas]) for _ in range(N)]
        out[tag + "_cfg"] = np.array([seed, H, m, N], np.int64)
        out[tag + "_sigma"] = sigma
        out[tag + "_betas"] = np.array(betas)
        out[tag + "_base"] = base
        out[tag + "_acts"] = np.array(acts)

    # --- worker seeds and sample order of gather_paths_parallel (utils.py:148-163) ---------
    # a toy env stands in for get_environment; what is recorded is the action tensor of every
    # returned path (= which noise stream produced global sample k) and the path schema
    ru.get_environment = lambda env_name, reward_type=None, reference=None: ToyEnv()
    env = ToyEnv()
    base = np.zeros((6, 3))
    paths = ru.gather_paths_parallel("toy", env.get_env_state(), base, [np.ones(3), 0.25, 0.8, 0.0], 500, None,
                                     "dense", None, 1.0, 4, 3)
    out["g_actions"] = np.array([p["actions"] for p in paths])
    out["g_rewards"] = np.array([p["rewards"] for p in paths])
    out["g_observations"] = np.array([p["observations"] for p in paths])
    out["g_next_observations"] = np.array([np.array(p["next_observations"]) for p in paths])
    out["g_info_timestep"] = np.array([p["env_infos"]["state"]["timestep"] for p in paths])
    out["g_keys"] = np.array(sorted(paths[0].keys()))

    # --- score / update (mppi.py:56-65, 91-108) -----------------------------------------
    rs = np.random.RandomState(5)
    K, H, m = 12, 6, 3
    fab = [dict(actions=rs.randn(H, m), rewards=rs.randn(H) * 3 - 5, critic_rewards=rs.randn(H, 1))
           for _ in range(K)]
    for tag, kappa, gamma in (("u0", 25.0, 1.0), ("u1", 2.0, 0.9)):
        ag = rm.MPPI(ToyEnv(), H=H, paths_per_cpu=K, kappa=kappa, gamma=gamma)
        out[tag + "_kg"] = np.array([kappa, gamma])
        out[tag + "_scores"] = ag.score_trajectory(fab)
        out[tag + "_scores_critic"] = ag.score_trajectory(fab, use_critic=True)
        ag.update(fab)
        out[tag + "_act_sequence"] = ag.act_sequence.copy()
        ag.update(fab, use_critic=True)
        out[tag + "_act_sequence_critic"] = ag.act_sequence.copy()
    out["u_actions"] = np.array([p["actions"] for p in fab])
    out["u_rewards"] = np.array([p["rewards"] for p in fab])
    out["u_critic_rewards"] = np.array([p["critic_rewards"] for p in fab])

    # --- advance_time bookkeeping (mppi.py:67-89) ---------------------------------------
    for tag, default_act in (("a_repeat", "repeat"), ("a_mean", "mean")):
        e = ToyEnv()
        seq = np.random.RandomState(3).randn(5, 3)
        ag = rm.MPPI(e, H=5, paths_per_cpu=2, mean=np.array([0.1, 0.2, 0.3]), default_act=default_act,
                     init_seq=seq.copy())
        seqs = []
        for _ in range(12):
            ag.advance_time()
            seqs.append(ag.act_sequence.copy())
        out[tag + "_init"] = seq
        out[tag + "_seqs"] = np.array(seqs)
        out[tag + "_sol_act"] = np.array(ag.sol_act)
        out[tag + "_sol_reward"] = np.array(ag.sol_reward)
        out[tag + "_sol_obs"] = np.array(ag.sol_obs)
        out[tag + "_sol_x"] = np.array([s["x"] for s in ag.sol_state])

    # --- train_step control flow on the toy env (mppi.py:125-196) -------------------------
    e = ToyEnv()
    ag = rm.MPPI(e, H=6, paths_per_cpu=8, num_cpu=1, kappa=5.0, gamma=0.95, mean=np.zeros(3),
                 filter_coefs=[np.ones(3), 0.25, 0.8, 0.0], default_act="mean", seed=77)
    ts = []
    for _ in range(3):
        ag.train_step(niter=2)
        ts.append(ag.act_sequence.copy())
    out["t_seqs"] = np.array(ts)
    out["t_sol_reward"] = np.array(ag.sol_reward)
    out["t_sol_x"] = np.array([s["x"] for s in ag.sol_state])

    # --- helpers ----------------------------------------------------------------------------
    out["ds_in"] = rs.randn(9)
    out["ds_out"] = ru.discount_sum(out["ds_in"], 0.9, 0.5)
    st = rt.stack_tensor_dict_list([dict(a=np.arange(3) + i, b=dict(c=float(i), d=np.ones((2, 2)) * i))
                                    for i in range(4)])
    out["st_a"], out["st_bc"], out["st_bd"] = st["a"], st["b"]["c"], st["b"]["d"]

    np.savez_compressed(os.path.join(HERE, "planner_vectors.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
