"""`Critic`: the terminal-value network of the reference (trajopt/models/critic_nets.py:12-115),
a tanh MLP input_dim -> inner -> inner -> 1, with the observation slicing helpers the example
scripts call.  Layer names match the reference so its stored `state_dict`s load."""
import numpy as np
import torch
import torch.nn as nn

from ..utils import Tuple

STATE_DIM = 14


def _slice(state, dim):
    """Critic input of one 20-dim observation [qpos7, qvel7, hand3, target3] by `dim`."""
    if dim == 14:
        return state[:14]
    if dim == 15:
        return np.concatenate((state[:14], [np.linalg.norm(state[-6:-3] - state[-3:])]))
    if dim == 17:
        return np.concatenate((state[:14], state[-3:]))
    if dim == 3:
        return state[-6:-3]
    if dim == 6:
        return state[-6:]
    return None


class Critic(nn.Module):
    def __init__(self, gamma=1.0, num_iters=10000, batch_size=128, input_dim=14, inner_layer=128):
        super().__init__()
        self.gamma, self.num_iters, self.batch_size, self.input_dim = gamma, num_iters, batch_size, input_dim
        self.tanh = torch.tanh
        self.softplus = nn.Softplus()
        self.relu = nn.ReLU()
        self.linear1 = nn.Linear(input_dim, inner_layer)
        self.linear2 = nn.Linear(inner_layer, inner_layer)
        self.linear3 = nn.Linear(inner_layer, 1)

    def forward(self, x):
        return self.linear3(self.tanh(self.linear2(self.tanh(self.linear1(x)))))

    def compress_state(self, state, dim=14):
        return _slice(state, dim)

    def compress_states(self, tuples, dim=14):
        for i, t in enumerate(tuples):
            tuples[i] = Tuple(_slice(t.state, dim), t.action, t.reward, _slice(t.next_state, dim))
        return tuples

    def compress_agent(self, agent, dim=14):
        obs = agent.sol_obs
        return [Tuple(_slice(obs[i], dim), agent.sol_act[i], agent.sol_reward[i], _slice(obs[i + 1], dim))
                for i in range(len(obs) - 1)]
