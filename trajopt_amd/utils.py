"""Rollout helpers with the reference's signatures (trajopt/utils.py:12-192), executing on
the GPU through the C ABI.

What moved: the `multiprocessing.Pool` of per-worker MuJoCo simulations
(`_try_multiprocess`, `do_env_rollout`) is one batched rollout kernel; the K samples of
all "workers" are lanes of the same launch.  What stayed on the host: the legacy numpy
random stream, so that sample k sees exactly the noise the reference's worker would have
drawn (`np.random.seed(base_seed + i*paths_per_cpu)` per worker, `generate_paths` order).
"""
import random
from collections import namedtuple
from collections.abc import Sequence

import numpy as np

from . import tensor_utils
from .device import DeviceModel, DevicePlanner
from .envs.utils import get_environment  # noqa: F401  (re-exported, as in the reference module)

Tuple = namedtuple("Tuple", ["state", "action", "reward", "next_state"])


class ReplayBuffer(object):
    """List-backed FIFO of `Tuple`s (trajopt/utils.py:15-56)."""

    def __init__(self, max_size=1000000):
        self.max_size = max_size
        self.buffer = []

    def _trim(self):
        extra = len(self.buffer) - self.max_size
        if extra > 0:
            self.buffer = self.buffer[extra:]

    def append(self, tup):
        assert type(tup) == Tuple
        self.buffer.append(tup)
        self._trim()

    def concatenate(self, tuples):
        assert (type(tuples) == list and type(tuples[0]) == Tuple) or type(tuples) == Tuple
        self.buffer.extend(tuples)
        self._trim()

    def append_dict(self, history):
        assert type(history) == dict
        n = len(history['states'])
        for i in range(n - 1):
            self.buffer.append(Tuple(history['states'][i], history['actions'][i], history['rewards'][i],
                                     history['states'][i + 1]))
        self._trim()

    def get_minibatch(self, size=32):
        return random.sample(self.buffer, size)


def discount_sum(x, gamma, discounted_terminal=0.0):
    """Discounted suffix sums with a terminal value (trajopt/utils.py:105-115)."""
    out = np.empty(len(x))
    run = discounted_terminal
    for t in range(len(x) - 1, -1, -1):
        run = x[t] + gamma * run
        out[t] = run
    return out


def generate_perturbed_actions(base_act, filter_coefs):
    """One filtered Gaussian perturbation of `base_act` from the global numpy stream
    (trajopt/utils.py:118-126).  The device applies the same recursion to uploaded draws;
    this host form exists because callers use it directly."""
    sigma, beta_0, beta_1, beta_2 = filter_coefs
    eps = np.random.normal(loc=0, scale=1.0, size=base_act.shape) * sigma
    for i in range(2, eps.shape[0]):
        eps[i] = beta_0 * eps[i] + beta_1 * eps[i - 1] + beta_2 * eps[i - 2]
    return base_act + eps


def score_trajectory(paths, gamma, use_critic=False):
    """Discounted return of every path, accumulated left to right with gamma**t as in
    `MPPI.score_trajectory` (trajopt/algos/mppi.py:91-108); with `use_critic` the last term
    uses critic_rewards[T-1] instead of rewards[T-1]."""
    scores = np.zeros(len(paths))
    for i, p in enumerate(paths):
        r = np.asarray(p["rewards"], dtype=np.float64)
        T = r.shape[0]
        s = 0.0
        for t in range(T):
            term = r[t]
            if use_critic and t == T - 1:
                term = np.asarray(p["critic_rewards"][t]).reshape(-1)[0]
            s = s + (gamma ** t) * term
        scores[i] = s
    return scores


# ---------------------------------------------------------------------------------------------
# device planners are cached per configuration: the reference re-creates env + simulator on
# every call (trajopt/utils.py:66); here the compiled model and its buffers are reused
# ---------------------------------------------------------------------------------------------
_PLANNERS = {}
_MAX_CACHED = 8
ROLLOUT_DTYPE = "f32"        # arithmetic of the batched rollouts ('f64' for exact comparison runs)


def _model_name(env_name):
    if env_name in ('reacher_7dof', 'continual_reacher_7dof'):
        return 'reacher_7dof'
    raise ValueError("no compiled rollout kernel for environment %r" % (env_name,))


def rollout_objective(env_name, reward_type, reference):
    """(reward_type, reference) the rollout copies of an environment score with.  The continual reacher is
    built without arguments by `get_environment` and its step ignores `reward_type`
    (trajopt/envs/utils.py:11-12, trajopt/envs/continual_reacher_env.py:10-17): always dense, no reference."""
    if env_name == 'continual_reacher_7dof':
        return 'dense', None
    return (reward_type if reward_type in ('sparse', 'tracking', 'cooling') else 'dense'), reference


def _rollout_planner(env_name, K, H, reward_type, dtype):
    key = (env_name, K, H, dtype)
    p = _PLANNERS.get(key)
    if p is None:
        while len(_PLANNERS) >= _MAX_CACHED:
            _PLANNERS.pop(next(iter(_PLANNERS))).close()
        p = DevicePlanner(DeviceModel.get(_model_name(env_name)), K, H, dtype=dtype, dump=True)
        _PLANNERS[key] = p
    p.set_params(1.0, 1.0, reward_type if reward_type in ('sparse', 'tracking', 'cooling') else 'dense')
    return p


def _check_reference(reward_type, reference, timestep, H):
    if reward_type in ('tracking', 'cooling'):
        if reference is None:
            raise TypeError("'NoneType' object is not subscriptable")      # what indexing None raises
        if timestep + H >= len(reference):
            raise IndexError("index %d is out of bounds for axis 0 with size %d" % (timestep + H, len(reference)))


class _StatesView(Sequence):
    """`states` of one path (trajopt/utils.py:89,98): H state dicts {qp, qv, target_pos, timestep},
    materialised on access from the rollout's observation block instead of being built eagerly
    for all K x H steps."""

    def __init__(self, obs, target, t0, nv):
        self._obs, self._target, self._t0, self._nv = obs, target, t0, nv

    def __len__(self):
        return self._obs.shape[0]

    def __getitem__(self, t):
        if isinstance(t, slice):
            return [self[i] for i in range(*t.indices(len(self)))]
        if t < 0:
            t += len(self)
        if not 0 <= t < len(self):
            raise IndexError("list index out of range")
        nv = self._nv
        return dict(qp=self._obs[t, :nv].copy(), qv=self._obs[t, nv:2 * nv].copy(), target_pos=self._target.copy(),
                    timestep=self._t0 + t)


class _RowsView(Sequence):
    """`next_observations` of one path (trajopt/utils.py:91,99: a list of H observation vectors) over the rows of the
    block read back from the device: rows come out as arrays on access, `np.asarray(view)` is the block itself."""

    def __init__(self, block):
        self._block = block

    def __len__(self):
        return self._block.shape[0]

    def __getitem__(self, t):
        if isinstance(t, slice):
            return list(self._block[t])
        return self._block[t]

    def __iter__(self):
        return iter(self._block)

    def __array__(self, dtype=None, copy=None):
        return self._block if dtype is None else self._block.astype(dtype)

    def __eq__(self, other):
        return len(other) == len(self) and all(np.array_equal(a, b) for a, b in zip(self, other))


def _paths_from_device(p, start_state, act, env_name):
    """Build the reference's list-of-dict `paths` (trajopt/utils.py:94-99) from device buffers.
    Arrays are per-sample views of the K-sample blocks read back from the device."""
    K, H, nv = act.shape
    nobs = p.next_observations()                 # (K, H, 2nv+6): obs after every step
    rew = p.rewards()
    dm = p.model
    target = np.asarray(start_state['target_pos'], dtype=np.float64)
    qp0, qv0 = np.asarray(start_state['qp'], float), np.asarray(start_state['qv'], float)
    # the first observation follows set_env_state -> forward(): fresh end-effector position
    obs0 = np.concatenate([qp0, qv0, dm.forward_hand(qp0), target])
    obs = np.concatenate([np.broadcast_to(obs0, (K, 1, obs0.size)), nobs[:, :-1]], axis=1)
    t0 = int(start_state['timestep'])
    timesteps = t0 + np.arange(H)
    tp = np.broadcast_to(target, (H, 3))
    paths = []
    for k in range(K):
        ok = obs[k]
        paths.append(dict(observations=ok,
                          actions=act[k],
                          rewards=rew[k],
                          env_infos=dict(state=dict(qp=ok[:, :nv], qv=ok[:, nv:2 * nv], target_pos=tp, timestep=timesteps)),
                          states=_StatesView(ok, target, t0, nv),
                          next_observations=_RowsView(nobs[k])))
    return paths


def do_env_rollout(env_name, start_state, act_list, goal, reward_type, reference, alpha):
    """Roll every action sequence of `act_list` (N arrays of shape (H, m)) out from
    `start_state`; returns N path dicts (trajopt/utils.py:59-102).  One kernel launch."""
    if isinstance(goal, int) and not isinstance(goal, bool):
        raise ValueError("integer goals select saved PyBullet states of the humanoid env, which has no kernel")
    act = np.ascontiguousarray(np.stack([np.asarray(a, dtype=np.float64) for a in act_list]))
    K, H, _ = act.shape
    reward_type, reference = rollout_objective(env_name, reward_type, reference)
    _check_reference(reward_type, reference, int(start_state['timestep']), H)
    p = _rollout_planner(env_name, K, H, reward_type, ROLLOUT_DTYPE)
    p.set_start_state(start_state['qp'], start_state['qv'], start_state['target_pos'], start_state['timestep'])
    p.set_reference(reference if reward_type in ('tracking', 'cooling') else None,
                    1.0 if alpha is None else alpha)
    p.upload_actions(act)
    p.rollout()
    return _paths_from_device(p, start_state, p.actions(), env_name)


_HOST_THREADS = None


def _host_threads():
    global _HOST_THREADS
    if _HOST_THREADS is None:
        import os
        n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:                # OMP_NUM_THREADS may be a list ("4,2": one value per nesting level) or empty
            omp = int(os.environ.get("OMP_NUM_THREADS", "16").split(",")[0])
        except ValueError:
            omp = 16
        _HOST_THREADS = max(1, min(16, n, omp))
    return _HOST_THREADS


def worker_seeds(base_seed, paths_per_cpu, num_cpu):
    """Seeds of the pool workers, worker i: base_seed + i * paths_per_cpu (trajopt/utils.py:151-152)."""
    seeds = [int(base_seed) + i * paths_per_cpu for i in range(num_cpu)]
    if not all(0 <= s < 2 ** 32 for s in seeds):
        raise ValueError("Seed must be between 0 and 2**32 - 1")
    return np.array(seeds, dtype=np.uint32)


def _set_numpy_state(state):
    """Put the process-global legacy generator where the library's 628-word state record says."""
    np.random.set_state(("MT19937", state[:624], int(state[624]), int(state[625]), float(state[626:628].view(np.float64)[0])))


def worker_normals(base_seed, n_paths, H, m, in_process=True, out=None, defer_state=False):
    """Raw N(0,1) draws of one reference worker: seed once, then n_paths blocks of (H, m) in
    order (`generate_paths`, trajopt/utils.py:135-139) -- numpy's legacy stream, produced in bulk by the
    library's generator (same bits as `np.random.seed(base_seed)` + `np.random.normal(size=(H, m))` calls).
    in_process: the worker is the calling process itself (the reference's num_cpu == 1 path,
    trajopt/utils.py:171-172), which leaves the process-global numpy generator re-seeded and advanced past the
    draws: the global state is set to exactly that.  Pool workers re-seed only their own copy, so with several
    workers the caller's generator is left alone.
    out: a C-contiguous float64 array of shape (n_paths, H, m) to fill instead of a new one.
    defer_state: leave the global generator alone and return (draws, state record) instead; the caller applies the
    record with `_set_numpy_state` when the draws are consumed (draws made ahead of time)."""
    from . import _lib as L
    seed = int(base_seed)
    if not 0 <= seed < 2 ** 32:
        raise ValueError("Seed must be between 0 and 2**32 - 1")
    if out is None:
        out = np.empty((n_paths, H, m))
    assert out.shape == (n_paths, H, m) and out.dtype == np.float64 and out.flags.c_contiguous
    state = np.empty(628, dtype=np.uint32) if in_process else None
    rc = L.lib().mppi_numpy_normals(seed, out.size, L.ptr(out), _host_threads(), L.ptr(state))
    if rc != 0:
        raise RuntimeError("mppi_numpy_normals failed")
    if defer_state:
        return out, state
    if in_process:
        _set_numpy_state(state)
    return out


def all_worker_normals(base_seed, paths_per_cpu, num_cpu, H, m, out=None, defer_state=False):
    """(K, H, m) draws in global sample order k = worker * paths_per_cpu + j, worker i seeded
    with base_seed + i * paths_per_cpu (trajopt/utils.py:151-152: overlapping streams).
    out: a (K, H, m) float64 array to fill (a planner's page-locked staging buffer) instead of a new one."""
    if out is None:
        out = np.empty((paths_per_cpu * num_cpu, H, m))
    assert out.shape == (paths_per_cpu * num_cpu, H, m) and out.dtype == np.float64 and out.flags.c_contiguous
    if num_cpu == 1:
        return worker_normals(base_seed, paths_per_cpu, H, m, in_process=True, out=out, defer_state=defer_state)
    from . import _lib as L
    seeds = [int(base_seed) + i * paths_per_cpu for i in range(num_cpu)]
    if not all(0 <= s < 2 ** 32 for s in seeds):
        raise ValueError("Seed must be between 0 and 2**32 - 1")
    seeds = np.array(seeds, dtype=np.uint32)
    if L.lib().mppi_numpy_normals_many(L.ptr(seeds), num_cpu, paths_per_cpu * H * m, L.ptr(out), _host_threads()) != 0:
        raise RuntimeError("mppi_numpy_normals_many failed")
    return (out, None) if defer_state else out


def generate_paths(env_name, start_state, N, base_act, filter_coefs, base_seed, goal, reward_type, reference, alpha):
    """One worker's share: seed, N perturbations, rollouts (trajopt/utils.py:129-141)."""
    np.random.seed(base_seed)
    act_list = [generate_perturbed_actions(base_act, filter_coefs) for _ in range(N)]
    return do_env_rollout(env_name, start_state, act_list, goal, reward_type, reference, alpha)


def generate_paths_star(args_list):
    return generate_paths(*args_list)


def gather_paths_parallel(env_name, start_state, base_act, filter_coefs, base_seed, goal, reward_type, reference,
                          alpha, paths_per_cpu, num_cpu=None):
    """All K = paths_per_cpu * num_cpu perturbed rollouts in one launch, in the reference's
    sample order and with the reference's per-worker noise streams (trajopt/utils.py:148-163)."""
    if num_cpu is None:
        import multiprocessing as mp
        num_cpu = mp.cpu_count()
    base_act = np.asarray(base_act, dtype=np.float64)
    H, m = base_act.shape
    K = paths_per_cpu * num_cpu
    reward_type, reference = rollout_objective(env_name, reward_type, reference)
    _check_reference(reward_type, reference, int(start_state['timestep']), H)
    sigma, b0, b1, b2 = filter_coefs
    key = (env_name, K, H, ROLLOUT_DTYPE, 'noise', tuple(np.broadcast_to(np.asarray(sigma, float), (m,))), b0, b1, b2)
    p = _PLANNERS.get(key)
    if p is None:
        while len(_PLANNERS) >= _MAX_CACHED:
            _PLANNERS.pop(next(iter(_PLANNERS))).close()
        p = DevicePlanner(DeviceModel.get(_model_name(env_name)), K, H, sigma=sigma, betas=(b0, b1, b2),
                          dtype=ROLLOUT_DTYPE, dump=True)
        _PLANNERS[key] = p
    p.set_params(1.0, 1.0, reward_type if reward_type in ('sparse', 'tracking', 'cooling') else 'dense')
    p.set_start_state(start_state['qp'], start_state['qv'], start_state['target_pos'], start_state['timestep'])
    p.set_reference(reference if reward_type in ('tracking', 'cooling') else None, 1.0 if alpha is None else alpha)
    p.set_base_actions(base_act)
    p.upload_noise(all_worker_normals(base_seed, paths_per_cpu, num_cpu, H, m, out=p.host_staging()))
    p.rollout()
    return _paths_from_device(p, start_state, p.actions(), env_name)


def _try_multiprocess(args_list, num_cpu, max_process_time, max_timeouts):
    """Kept for callers of the reference helper (trajopt/utils.py:166-192): the per-worker
    argument lists run back to back in this process, each as one batched launch."""
    if max_timeouts == 0:
        return None
    return [generate_paths_star(a) for a in args_list[:num_cpu]]


def close_cached_planners():
    while _PLANNERS:
        _PLANNERS.popitem()[1].close()
