#!/usr/bin/env python3
"""Dense-reward MPPI on reacher_7dof for a few random goals, then a sparse-reward replay of every
solution -- the flows of the reference's `sandbox/examples/test_goals.py` and `play_mppi_agent.py`,
written against the same `trajopt` API (no viewer, no prompts).

    python examples/reacher_goals.py [--steps 116] [--seeds 10 11 13 17] [--noise host|philox]
"""
import argparse
import os
import pickle
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajopt.algos.mppi import MPPI                      # noqa: E402
from trajopt.envs.utils import get_environment           # noqa: E402
from trajopt_amd.agent_io import load_agent              # noqa: E402


def reached(sol_obs, radius=0.025):
    return int(min(np.linalg.norm(o[-6:-3] - o[-3:]) if len(o) == 20 else o[-1] for o in sol_obs) < radius)


def plan(seed, steps, noise, planner_seed=12345, niter=5, H=16):
    e = get_environment('reacher_7dof', reward_type='dense', reference=None)
    e.reset_model(seed=seed, goal=None)                  # seeded: draws a random target
    goal = e.get_env_state()['target_pos']
    agent = MPPI(e, H=H, paths_per_cpu=40, num_cpu=1, kappa=25.0, gamma=1.0, mean=np.zeros(e.action_dim),
                 filter_coefs=[np.ones(e.action_dim), 0.25, 0.8, 0.0], default_act='mean', seed=planner_seed,
                 noise=noise)
    for _ in range(steps):
        agent.train_step(niter=niter, goal=goal)
    return agent


def replay_sparse(agent):
    """Step the logged states with the logged actions under the sparse reward and sum it."""
    e = get_environment('reacher_7dof', reward_type='sparse', reference=None)
    e.reset_model(goal=(0, 0, 0), alpha=0.0)
    total = 0.0
    for state, action in zip(agent.sol_state, agent.sol_act):
        e.set_env_state(state)
        total += e.step(action)[1]
    return total


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=116)
    ap.add_argument('--seeds', type=int, nargs='*', default=[10, 11, 13, 17])
    ap.add_argument('--noise', default='host')
    args = ap.parse_args(argv)
    results = []
    for seed in args.seeds:
        t0 = time.time()
        agent = plan(seed, args.steps, args.noise)
        dt = time.time() - t0
        with tempfile.NamedTemporaryFile(suffix='.pickle') as f:
            pickle.dump(agent, f)
            f.flush()
            agent = pickle.load(open(f.name, 'rb'))      # our own file: fine to unpickle
            stored = load_agent(f.name)                   # the loader for files from elsewhere: reads, never executes
            assert len(stored.sol_act) == len(agent.sol_act) and np.array_equal(stored.act_sequence, agent.act_sequence)
        ret, hit, sparse = float(np.sum(agent.sol_reward)), reached(agent.sol_obs), replay_sparse(agent)
        print("seed %d: trajectory reward %.2f, reached %d, sparse replay %.0f, %.2f s" % (seed, ret, hit, sparse, dt))
        results.append((ret, hit, sparse, dt))
    print("Avg. Reward: %.2f (%.2f)" % (np.mean([r[0] for r in results]), np.std([r[0] for r in results])))
    return results


if __name__ == '__main__':
    main()
