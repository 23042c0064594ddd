#!/usr/bin/env python3
"""Actor-critic MPPI on reacher_7dof with a cooled tracking reward -- the flow of the reference's
`sandbox/examples/ac_reacher_7dof_mppi.py`, written against the same `trajopt` API (no tensorboard, no files
unless asked for).

A dense-reward planner first produces a solution trajectory; its hand positions become the tracked reference.  Every
epoch then builds a new `MPPI` whose rollouts are scored with the 'cooling' reward (alpha tracking + (1 - alpha)
sparse, alpha cooled by `eta` per epoch) and whose last reward is replaced by the critic's value, while the real
environment reports the sparse reward; the solution goes into a replay buffer twice, as in the reference, and the
critic takes temporal-difference steps against a target copy of itself.

    python examples/actor_critic_reacher.py [--epochs 4] [--steps 100] [--iters 200] [--noise host|philox]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajopt.algos.mppi import MPPI                      # noqa: E402
from trajopt.envs.utils import get_environment           # noqa: E402
from trajopt.models.critic_nets import Critic            # noqa: E402
from trajopt.utils import ReplayBuffer                   # noqa: E402

ENV_NAME = 'reacher_7dof'
H, PATHS, N_ITER, SEED = 16, 40, 5, 12345
GOAL = (0.0, 0.0, 0.0)


def planner(e, noise, **kw):
    return MPPI(e, H=H, paths_per_cpu=PATHS, num_cpu=1, kappa=25.0, gamma=1.0, mean=np.zeros(e.action_dim),
                filter_coefs=[np.ones(e.action_dim), 0.25, 0.8, 0.0], default_act='mean', seed=SEED, noise=noise, **kw)


def hand_positions(agent):
    """Hand site at every logged state, read the way the reference script reads it from a stored agent."""
    out = []
    for s in agent.sol_state:
        agent.env.set_env_state(s)
        out.append(np.array(agent.env.data.site_xpos[agent.env.hand_sid]))
    return np.array(out)


def dense_solution(steps, noise):
    e = get_environment(ENV_NAME)
    e.reset_model(seed=None, goal=GOAL)
    agent = planner(e, noise)
    for _ in range(steps):
        agent.train_step(niter=N_ITER, goal=GOAL)
    return agent


def critic_steps(critic, target, replay, optimizer, iters):
    """Temporal-difference regression of the critic on minibatches of the replay buffer; rewards above 10 mark the
    goal and end the bootstrap.  The target copy follows the critic eight times per call."""
    loss_fn = torch.nn.MSELoss()
    loss = None
    for i in range(iters):
        batch = replay.get_minibatch(size=min(critic.batch_size, len(replay.buffer)))
        state = torch.tensor(np.array([d.state for d in batch]), dtype=torch.float32)
        nxt = torch.tensor(np.array([d.next_state for d in batch]), dtype=torch.float32)
        reward = torch.tensor([float(d.reward) for d in batch], dtype=torch.float32).unsqueeze(1)
        with torch.no_grad():
            boot = torch.where(reward > 10.0, torch.zeros_like(reward), critic.gamma * target(nxt))
        loss = loss_fn(critic(state), reward + boot)
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        if i % max(iters // 8, 1) == 0:
            target.load_state_dict(critic.state_dict())
    return None if loss is None else float(loss.detach())


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--epochs', type=int, default=4)
    ap.add_argument('--steps', type=int, default=100, help="real-env steps per epoch (at most the dense solution's length minus H)")
    ap.add_argument('--iters', type=int, default=200, help="critic steps per epoch")
    ap.add_argument('--eta', type=float, default=0.9)
    ap.add_argument('--lr', type=float, default=1e-2)
    ap.add_argument('--dim', type=int, default=14)
    ap.add_argument('--noise', default='host')
    args = ap.parse_args(argv)

    t0 = time.time()
    reference = hand_positions(dense_solution(args.steps + H, args.noise))
    print("dense solution: %d states, hand ends %.3f m from the goal, %.2f s"
          % (len(reference), np.linalg.norm(reference[-1] - np.array(GOAL)), time.time() - t0))

    e = get_environment(ENV_NAME, reward_type='sparse')          # the logged reward is the sparse one
    torch.manual_seed(0)
    critic = Critic(num_iters=args.iters, input_dim=args.dim, inner_layer=128, batch_size=128, gamma=0.9)
    target = Critic(num_iters=args.iters, input_dim=args.dim, inner_layer=128)
    target.load_state_dict(critic.state_dict())
    for net in (critic, target):
        net.eval()
        net.float()
    optimizer = torch.optim.Adam(critic.parameters(), lr=args.lr)
    replay = ReplayBuffer(max_size=10000)

    alpha, results = 1.0, []
    for epoch in range(args.epochs):
        t0 = time.time()
        e.reset_model(seed=None, goal=GOAL, alpha=alpha)
        goal = e.get_env_state()['target_pos']
        agent = planner(e, args.noise, reward_type='cooling', reference=reference)
        for _ in range(args.steps):
            agent.train_step(critic=critic, niter=N_ITER, goal=goal, dim=args.dim)
        samples = critic.compress_agent(agent, dim=args.dim)
        samples += critic.compress_agent(agent, dim=args.dim)   # the solution goes in twice
        replay.concatenate(samples)
        loss = critic_steps(critic, target, replay, optimizer, critic.num_iters)
        ret, dist = float(np.sum(agent.sol_reward)), float(agent.sol_obs[-1][-1])
        print("epoch %d: alpha %.3f, sparse return %.0f, final distance %.3f m, critic loss %s, %.2f s"
              % (epoch, alpha, ret, dist, "%.3g" % loss if loss is not None else "-", time.time() - t0))
        results.append((alpha, ret, dist, loss))
        alpha *= args.eta
    return results


if __name__ == '__main__':
    main()
