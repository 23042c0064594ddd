"""numpy restatement of the reference planner arithmetic -- TEST INFRASTRUCTURE.

Checker only (see oracle/__init__.py).  Pinned by tests/golden/planner_vectors.npz,
which was produced by the reference's own functions.
"""
import numpy as np


def perturbed_actions(base_act, filter_coefs, normal=None):
    """`generate_perturbed_actions` (trajopt/utils.py:118-126): N(0,1)*sigma, then the
    in-place recursion eps[t] = b0*eps[t] + b1*eps[t-1] + b2*eps[t-2] for t >= 2 (rows 0
    and 1 stay unfiltered, later rows see already-filtered predecessors), plus base."""
    sigma, b0, b1, b2 = filter_coefs
    draw = np.random.normal if normal is None else normal
    e = draw(loc=0, scale=1.0, size=base_act.shape) * sigma
    for t in range(2, e.shape[0]):
        e[t] = b0 * e[t] + b1 * e[t - 1] + b2 * e[t - 2]
    return base_act + e


def raw_normals(base_seed, n_paths, H, m):
    """The N(0,1) block one worker consumes (`generate_paths`, trajopt/utils.py:135-139):
    legacy global stream seeded once, then n_paths draws of shape (H, m) in order."""
    np.random.seed(base_seed)
    return np.stack([np.random.normal(loc=0, scale=1.0, size=(H, m)) for _ in range(n_paths)])


def worker_seeds(base_seed, paths_per_cpu, num_cpu):
    """`gather_paths_parallel` (trajopt/utils.py:151-152): overlapping streams by design."""
    return [base_seed + i * paths_per_cpu for i in range(num_cpu)]


def all_raw_normals(base_seed, paths_per_cpu, num_cpu, H, m):
    """(K,H,m) raw normals in the global sample order k = worker*paths_per_cpu + j."""
    return np.concatenate([raw_normals(s, paths_per_cpu, H, m)
                           for s in worker_seeds(base_seed, paths_per_cpu, num_cpu)])


def filter_noise(raw, filter_coefs):
    """sigma scaling + recursive filter applied to a (K,H,m) block of raw normals."""
    sigma, b0, b1, b2 = filter_coefs
    e = raw * sigma
    for t in range(2, e.shape[1]):
        e[:, t] = b0 * e[:, t] + b1 * e[:, t - 1] + b2 * e[:, t - 2]
    return e


def score(rewards, gamma, critic_rewards=None):
    """`MPPI.score_trajectory` (trajopt/algos/mppi.py:91-108): left-to-right sum of
    gamma**t * r_t; with a critic the last term uses critic_rewards[T-1] instead."""
    rewards = np.asarray(rewards, dtype=np.float64)
    K, T = rewards.shape
    out = np.zeros(K)
    for i in range(K):
        s = 0.0
        for t in range(T):
            r = rewards[i][t]
            if critic_rewards is not None and t == T - 1:
                r = critic_rewards[i][t]
            s = s + (gamma ** t) * r
        out[i] = s if critic_rewards is None else np.asarray(s).reshape(-1)[0]
    return out


def update(actions, R, kappa):
    """`MPPI.update` (trajopt/algos/mppi.py:56-65): exponentiated-return weights and the
    weighted action mean with the 1e-6 guard in the denominator."""
    actions = np.asarray(actions, dtype=np.float64)
    S = np.exp(kappa * (R - np.max(R)))
    num = np.sum((S * actions.T).T, axis=0)
    return num / (np.sum(S) + 1e-6)


def shift(act_sequence, default_act, mean):
    """tail of `MPPI.advance_time` (trajopt/algos/mppi.py:85-89); returns a new array."""
    out = act_sequence.copy()
    out[:-1] = act_sequence[1:]
    out[-1] = out[-2] if default_act == "repeat" else mean
    return out


def partials(actions, R, kappa):
    """Per-shard sufficient statistics of the update: (Rmax, argmax, sum S*act, sum S) with
    S relative to the shard's own maximum."""
    Rmax = np.max(R)
    S = np.exp(kappa * (R - Rmax))
    return Rmax, int(np.argmax(R)), np.tensordot(S, actions, axes=(0, 0)), np.sum(S)


def merge_partials(parts, kappa):
    """Rank-ordered merge of shard statistics; equals `update` on the concatenated shards."""
    gmax = max(p[0] for p in parts)
    num = sum(np.exp(kappa * (p[0] - gmax)) * p[2] for p in parts)
    den = sum(np.exp(kappa * (p[0] - gmax)) * p[3] for p in parts)
    return num / (den + 1e-6)
