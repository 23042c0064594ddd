#!/usr/bin/env python3
"""MPPI that tracks a stored solution -- the flow of the reference's `sandbox/examples/reacher_7dof_mppi.py`, written
against the same `trajopt` API (no viewer, no prompt).

The hand trajectory of a dense-reward solution (a stored agent pickle given with --agent, read without executing it, or
one planned here first) becomes the `reference` of a 'tracking' environment and planner; the planner then follows it.

    python examples/reacher_tracking.py [--agent dense_agent.pickle] [--steps 100] [--noise host|philox]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajopt.algos.mppi import MPPI                      # noqa: E402
from trajopt.envs.utils import get_environment           # noqa: E402
from trajopt_amd.agent_io import load_agent              # noqa: E402

ENV_NAME, SEED, N_ITER, H = 'reacher_7dof', 3, 5, 16
GOAL = (0, 0, 0)


def planner(e, noise, **kw):
    return MPPI(e, H=H, paths_per_cpu=40, num_cpu=1, kappa=25.0, gamma=1.0, mean=np.zeros(e.action_dim),
                filter_coefs=[np.ones(e.action_dim), 0.25, 0.8, 0.0], default_act='mean', seed=SEED, noise=noise, **kw)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--agent', default=None, help="stored agent whose solution is tracked (default: plan one here)")
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--noise', default='host')
    args = ap.parse_args(argv)

    if args.agent:
        reference_agent = load_agent(args.agent, env_name=ENV_NAME)
    else:
        e0 = get_environment(ENV_NAME)
        e0.reset_model(seed=SEED, goal=GOAL, alpha=1.0)
        reference_agent = planner(e0, args.noise)
        for _ in range(args.steps + H):
            reference_agent.train_step(niter=N_ITER, goal=GOAL)
    reference_pos = []
    for s in reference_agent.sol_state:
        reference_agent.env.set_env_state(s)
        reference_pos.append(np.array(reference_agent.env.data.site_xpos[reference_agent.env.hand_sid]))
    reference = np.array(reference_pos)
    steps = min(args.steps, len(reference) - H - 1)

    e = get_environment(ENV_NAME, reward_type='tracking', reference=reference)
    e.reset_model(seed=SEED, goal=GOAL, alpha=1.0)
    agent = planner(e, args.noise, reward_type='tracking', reference=reference)
    ts = time.time()
    for _ in range(steps):
        agent.train_step(niter=N_ITER, goal=GOAL)
    dt = time.time() - ts
    hand = np.array([o[14:17] for o in agent.sol_obs[1:]])
    gap = np.linalg.norm(hand - reference[1:steps + 1], axis=1)
    print("Trajectory reward = %f" % np.sum(agent.sol_reward))
    print("Time for trajectory optimization = %f seconds (%d steps)" % (dt, steps))
    print("mean / final distance to the tracked hand position: %.3f / %.3f m" % (gap.mean(), gap[-1]))
    return float(np.sum(agent.sol_reward)), float(gap.mean()), float(gap[-1])


if __name__ == '__main__':
    main()
