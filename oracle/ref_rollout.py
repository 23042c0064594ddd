"""CPU restatement of the reference's rollout path, structured like the reference --
TEST INFRASTRUCTURE / reported CPU baseline only (see oracle/__init__.py).

The reference's path is a Python loop around a C simulator: a process pool is created for
every call (`_try_multiprocess`, trajopt/utils.py:166-192), every worker builds an env
(`do_env_rollout`, :66), and every env step does ~6 small numpy operations around two
`sim.step()` calls (`Reacher7DOFEnv._step`, trajopt/envs/reacher_env.py:43-85).  This module
keeps that structure and puts the fp64 C++ restatement of the simulator step
(oracle/mjstep.hpp) where `sim.step()` was, so that timing it is a fair stand-in for timing
the reference on the same host when MuJoCo is not installed.
"""
import multiprocessing as mp

import numpy as np

from . import ref_mppi
from .sim import OracleSim


class OracleReacherEnv:
    """Python-level twin of `Reacher7DOFEnv` on the oracle simulator."""

    def __init__(self, md, reward_type='dense', reference=None):
        self.sim = OracleSim(md)
        self.nv = md.nv
        self.frame_skip = md.frame_skip
        self.reward_type, self.reference = reward_type, reference
        self.alpha = 1.0
        self.real_step = True
        self.env_timestep = 0
        self.qpos, self.qvel = np.zeros(self.nv), np.zeros(self.nv)
        self.target = np.zeros(3)
        self.hand = self.sim.fk_hand(self.qpos)

    def reset_model(self, goal=None, alpha=None):
        if alpha is not None:
            self.alpha = alpha
        self.qpos, self.qvel = np.zeros(self.nv), np.zeros(self.nv)
        self.target = np.zeros(3) if goal is None else np.array(goal, dtype=np.float64)
        self.hand = self.sim.fk_hand(self.qpos)
        self.env_timestep = 0
        return self._get_obs()

    def get_env_state(self):
        return dict(qp=self.qpos.copy(), qv=self.qvel.copy(), target_pos=self.target.copy(),
                    timestep=self.env_timestep)

    def set_env_state(self, state):
        self.qpos, self.qvel = state['qp'].copy(), state['qv'].copy()
        self.target = np.asarray(state['target_pos'], dtype=np.float64)
        self.env_timestep = state['timestep']
        self.hand = self.sim.fk_hand(self.qpos)

    def get_env_infos(self):
        return dict(state=self.get_env_state())

    def _get_obs(self):
        return np.concatenate([self.qpos, self.qvel, self.hand, self.target])

    def step(self, a):
        for _ in range(self.frame_skip):
            self.qpos, self.qvel, self.hand, _ = self.sim.substep(self.qpos, self.qvel, a)
        self.env_timestep += 1
        dist = np.linalg.norm(self.hand - self.target)
        reward = -10.0 * dist - 0.25 * np.linalg.norm(self.qvel)
        if self.reward_type == 'sparse':
            reward = 100.0 if dist < 0.025 else (-100.0 if dist > 0.8 else -10.0)
        elif self.reward_type in ('tracking', 'cooling'):
            sparse = 100.0 if dist < 0.025 else (-100.0 if dist > 0.8 else -10.0)
            track = -10.0 * np.linalg.norm(self.hand - self.reference[self.env_timestep]) \
                - 0.25 * np.linalg.norm(self.qvel)
            reward = track if self.reward_type == 'tracking' else self.alpha * track + (1.0 - self.alpha) * sparse
        return self._get_obs(), reward, False, self.get_env_infos()


def do_env_rollout(md, start_state, act_list, goal, reward_type, reference, alpha):
    """`do_env_rollout` (trajopt/utils.py:59-102) with the same per-step bookkeeping."""
    e = OracleReacherEnv(md, reward_type=reward_type, reference=reference)
    e.reset_model(goal=goal, alpha=alpha)
    e.real_step = False
    paths = []
    H = act_list[0].shape[0]
    for act_seq in act_list:
        e.set_env_state(start_state)
        obs, act, rewards, infos, states, nobs = [], [], [], [], [], []
        for k in range(H):
            obs.append(e._get_obs())
            act.append(act_seq[k])
            infos.append(e.get_env_infos())
            states.append(e.get_env_state())
            s, r, _, _ = e.step(act[-1])
            rewards.append(r)
            nobs.append(s)
        paths.append(dict(observations=np.array(obs), actions=np.array(act), rewards=np.array(rewards),
                          env_infos=infos, states=states, next_observations=nobs))
    return paths


def _worker(args):
    md, start_state, N, base_act, filter_coefs, seed, goal, reward_type, reference, alpha = args
    np.random.seed(seed)
    act_list = [ref_mppi.perturbed_actions(base_act, filter_coefs) for _ in range(N)]
    return do_env_rollout(md, start_state, act_list, goal, reward_type, reference, alpha)


def gather_paths_parallel(md, start_state, base_act, filter_coefs, base_seed, goal, reward_type, reference, alpha,
                          paths_per_cpu, num_cpu):
    """`gather_paths_parallel` + `_try_multiprocess` (trajopt/utils.py:148-192): a fresh pool
    per call, one task per worker, results concatenated in worker order."""
    args = [(md, start_state, paths_per_cpu, base_act, filter_coefs, base_seed + i * paths_per_cpu, goal,
             reward_type, reference, alpha) for i in range(num_cpu)]
    if num_cpu == 1:
        results = [_worker(args[0])]
    else:
        pool = mp.get_context("fork").Pool(processes=num_cpu, maxtasksperchild=1)
        try:
            jobs = [pool.apply_async(_worker, args=(a,)) for a in args]
            results = [j.get(timeout=300) for j in jobs]
        finally:
            pool.close()
            pool.terminate()
            pool.join()
    return [p for res in results for p in res]


def planning_iteration(md, start_state, base_act, filter_coefs, seed, kappa, gamma, paths_per_cpu, num_cpu):
    """One iteration of `MPPI.train_step`: rollouts, scoring, update (trajopt/algos/mppi.py:131-193)."""
    paths = gather_paths_parallel(md, start_state, base_act, filter_coefs, seed, None, 'dense', None, 1.0,
                                  paths_per_cpu, num_cpu)
    R = ref_mppi.score(np.array([p["rewards"] for p in paths]), gamma)
    return ref_mppi.update(np.array([p["actions"] for p in paths]), R, kappa)
