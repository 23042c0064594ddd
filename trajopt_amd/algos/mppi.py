"""`MPPI` with the reference's constructor, methods and attributes
(trajopt/algos/mppi.py:15-196), planning on the GPU.

`train_step` keeps the whole planning step on the device: the K perturbed rollouts, their
discounted returns, the exponentiated-return weights, the weighted action update, the step of
the real environment with the first action and the horizon shift run as kernels behind
`mppi_plan` (device noise) or `mppi_iterate` + `mppi_advance` (uploaded numpy noise); one log
row and the (H, m) `act_sequence` come back.  State and `act_sequence` stay on the device from
one call to the next and are sent again only when the caller has changed `sol_state[-1]` or
`act_sequence` in between (they are compared with what the device was last seen to hold).  The quirks that
callers observe are kept: every iteration of one `train_step` sees the same noise
(seed + t), worker streams overlap (seed + i*paths_per_cpu), the denominator carries 1e-6,
`sol_act` logs zeros for the first ten steps, `update` rebinds `act_sequence` while
`advance_time` shifts it in place.

Two keyword arguments are additions with behaviour-preserving defaults:
    noise  'host'   draw the reference's exact numpy stream on the host and upload it
           'philox' generate on the device (same distribution, different stream)
    dtype  'f32' | 'f64' arithmetic of the batched rollouts
"""
import numpy as np

from ..device import DeviceModel, DevicePlanner, update_from_returns
from ..utils import (ReplayBuffer, Tuple, _host_threads, _set_numpy_state, all_worker_normals,  # noqa: F401
                     gather_paths_parallel, rollout_objective, score_trajectory, worker_seeds)
from .trajopt_base import Trajectory

_KERNEL_ENVS = {'reacher_7dof': 'reacher_7dof', 'continual_reacher_7dof': 'reacher_7dof'}


def _critic_weights(critic, dim):
    """(W1, b1, W2, b2, W3, b3) of a `Critic` whose terminal value has a device kernel, else None."""
    from ..models.critic_nets import Critic
    if type(critic) is not Critic or dim not in (14, 15, 17, 3, 6):
        return None
    w = [critic.linear1.weight, critic.linear1.bias, critic.linear2.weight, critic.linear2.bias,
         critic.linear3.weight, critic.linear3.bias]
    w = [x.detach().cpu().numpy().astype(np.float32) for x in w]
    if w[0].shape != (128, dim) or w[2].shape != (128, 128) or w[4].shape != (1, 128):
        return None
    return w


class MPPI(Trajectory):
    def __init__(self, env, H, paths_per_cpu,
                 num_cpu=1,
                 kappa=1.0,
                 gamma=1.0,
                 mean=None,
                 filter_coefs=None,
                 default_act='repeat',
                 seed=123,
                 init_seq=None,
                 reward_type='dense',
                 reference=None,
                 noise='host',
                 dtype='f32'):
        # like the reference, the base constructor is not run: the env is not reset here
        self.env, self.seed = env, seed
        self.n, self.m = env.observation_dim, env.action_dim
        self.H, self.paths_per_cpu, self.num_cpu = H, paths_per_cpu, num_cpu
        self.kappa, self.gamma = kappa, gamma
        self.mean = np.zeros(self.m) if mean is None else mean
        self.filter_coefs = [np.ones(self.m), 1.0, 0.0, 0.0] if filter_coefs is None else filter_coefs
        self.default_act = default_act

        self.sol_state = [self.env.get_env_state().copy()]
        self.sol_act = []
        self.sol_reward = []
        self.sol_obs = [self.env._get_obs(goal_dist=True)]
        self.act_sequence = np.ones((self.H, self.m)) * self.mean
        if init_seq is not None and init_seq.shape == self.act_sequence.shape:
            self.act_sequence = init_seq            # aliased on purpose, as in the reference

        self.reward_type = reward_type
        self.reference = reference
        self.noise = noise
        self.dtype = dtype
        self._planner = None
        self._planner_key = None
        self._ahead = None              # (key, numpy state record) of host noise drawn for the next planning step
        self.draw_ahead = True
        self._dev_state = None          # (planner id, qp, qv, target_pos, timestep) the device planner holds ...
        self._dev_seq = None            # ... and its act_sequence, as of the last step read back

    # ---- pickling: device handles are dropped and rebuilt on demand -------------------------
    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop('_planner', None)
        d.pop('_planner_key', None)
        d.pop('_ahead', None)
        d.pop('_dev_state', None)
        d.pop('_dev_seq', None)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self.__dict__.setdefault('noise', 'host')
        self.__dict__.setdefault('dtype', 'f32')
        self._planner = None
        self._planner_key = None
        self._ahead = None
        self._dev_state = None
        self._dev_seq = None
        self.__dict__.setdefault('draw_ahead', True)

    # ---- device planner -----------------------------------------------------------------
    def _device_planner(self):
        sigma, b0, b1, b2 = self.filter_coefs
        # (compared as raw bytes: this runs at every planning step, and the arrays are seven numbers long)
        key = (self.env.env_name, self.H, self.paths_per_cpu, self.num_cpu, np.asarray(sigma, dtype=np.float64).tobytes(),
               b0, b1, b2, self.dtype, np.asarray(self.mean, dtype=np.float64).tobytes(), self.default_act)
        if self._planner is None or key != self._planner_key:
            if self._planner is not None:
                self._planner.close()
            self._ahead = None
            self._dev_state = self._dev_seq = None
            model = DeviceModel.get(_KERNEL_ENVS[self.env.env_name])
            sig = tuple(np.broadcast_to(np.asarray(sigma, dtype=np.float64), (self.m,)))
            self._planner = DevicePlanner(model, self.paths_per_cpu * self.num_cpu, self.H, kappa=self.kappa,
                                          gamma=self.gamma, sigma=sig, betas=(b0, b1, b2), mean=self.mean,
                                          default_act=self.default_act, dtype=self.dtype)
            self._planner_key = key
        rt, _ = rollout_objective(self.env.env_name, self.reward_type, self.reference)
        self._planner.set_params(self.kappa, self.gamma, rt)
        return self._planner

    # ---- reference API ------------------------------------------------------------------
    def update(self, paths, use_critic=False):
        """Weighted action update from caller-supplied paths (trajopt/algos/mppi.py:56-65)."""
        act = np.array([p["actions"] for p in paths], dtype=np.float64)
        R = self.score_trajectory(paths, use_critic=use_critic)
        self.act_sequence = update_from_returns(act, R, self.kappa)

    def advance_time(self, act_sequence=None):
        """Apply the first action to the real env, log, shift the horizon
        (trajopt/algos/mppi.py:67-89)."""
        seq = self.act_sequence if act_sequence is None else act_sequence
        first = seq[0]
        self.sol_act.append(np.zeros_like(first) if len(self.sol_act) < 10 else first.copy())
        self.env.set_env_state(self.sol_state[-1].copy())
        _, r, _, _ = self.env.step(first)
        self.sol_state.append(self.env.get_env_state().copy())
        self.sol_obs.append(self.env._get_obs(goal_dist=True))
        self.sol_reward.append(r)
        self.act_sequence[:-1] = seq[1:]
        self.act_sequence[-1] = self.act_sequence[-2] if self.default_act == 'repeat' else self.mean.copy()

    def score_trajectory(self, paths, use_critic=False):
        return score_trajectory(paths, self.gamma, use_critic=use_critic)

    def do_rollouts(self, seed, goal=None):
        return gather_paths_parallel(self.env.env_name, self.sol_state[-1], self.act_sequence, self.filter_coefs,
                                     seed, goal, self.reward_type, self.reference, self.env.alpha,
                                     self.paths_per_cpu, self.num_cpu)

    def train_step(self, critic=None, niter=1, act_sequence=None, goal=None, dim=14):
        t = len(self.sol_state) - 1
        weights = _critic_weights(critic, dim)
        if self.env.env_name in _KERNEL_ENVS and (critic is None or weights is not None):
            if act_sequence is None and self._real_env_steps_on_the_device():
                # the whole step, the real environment's included, without leaving the device
                self._plan_on_device(self.seed + t, niter, dim if critic is not None else 0, weights, advance=True)
                return []
            self._plan_on_device(self.seed + t, niter, dim if critic is not None else 0, weights)
        else:
            # an arbitrary Python critic can only be evaluated on the host: rollouts come back as paths
            for _ in range(niter):
                paths = self.do_rollouts(self.seed + t, goal=goal)
                self._attach_critic_rewards(paths, critic, dim)
                self.update(paths, use_critic=(critic is not None))
        self.advance_time(act_sequence=act_sequence)
        return []

    # ---- device-resident iteration ------------------------------------------------------
    def _real_env_steps_on_the_device(self):
        """True if the device planner's own fp64 step of the real environment is what `self.env.step` would do: a
        plain reacher, scored with its own reward type (the planner is told which: it need not be the one the rollouts
        are scored with -- the actor-critic scripts plan on 'cooling' and log 'sparse').  An environment that tracks a
        reference must track the one the rollouts use: the device holds one."""
        from ..envs.reacher_env import Reacher7DOFEnv
        e = self.env
        if type(e) is not Reacher7DOFEnv:
            return False
        if self.H * self.m > 4096:           # the advance kernel holds the shifted sequence in registers: 16 values x 256 threads
            return False
        rt, reference = rollout_objective(e.env_name, self.reward_type, self.reference)
        if self._env_reward() in ('tracking', 'cooling'):
            if rt not in ('tracking', 'cooling') or e.reference is None or reference is None:
                return False
            if e.reference is not reference and not np.array_equal(np.asarray(e.reference), np.asarray(reference)):
                return False
        return True

    def _env_reward(self):
        rt = self.env.reward_type
        return rt if rt in ('sparse', 'tracking', 'cooling') else 'dense'

    @staticmethod
    def _state_record(p, s):
        """What the device holds after being given (or having produced) state s, in a form that compares fast."""
        return (id(p), np.asarray(s['qp'], dtype=np.float64).tobytes(), np.asarray(s['qv'], dtype=np.float64).tobytes(),
                np.asarray(s['target_pos'], dtype=np.float64).tobytes(), int(s['timestep']))

    def _device_holds(self, p, s):
        return self._dev_state is not None and self._dev_state == self._state_record(p, s)

    def _device_holds_sequence(self, p):
        d = self._dev_seq
        return (d is not None and d[0] == id(p) and np.shape(self.act_sequence) == (self.H, self.m)
                and np.asarray(self.act_sequence, dtype=np.float64).tobytes() == d[1])

    def _plan_on_device(self, seed, niter, critic_dim=0, critic_weights=None, advance=False):
        """niter iterations on the device from sol_state[-1] and act_sequence.  advance: also `advance_time` there
        (real-env step with the new act_sequence[0], logs, shift); otherwise act_sequence is read back and the
        caller advances."""
        s = self.sol_state[-1]
        H = self.H
        rt, reference = rollout_objective(self.env.env_name, self.reward_type, self.reference)
        tracked = rt in ('tracking', 'cooling')
        if tracked:
            if reference is None:
                raise TypeError("'NoneType' object is not subscriptable")
            if int(s['timestep']) + H >= len(reference):
                raise IndexError("index %d is out of bounds for axis 0 with size %d"
                                 % (int(s['timestep']) + H, len(reference)))
        p = self._device_planner()
        if not self._device_holds(p, s):
            p.set_start_state(s['qp'], s['qv'], s['target_pos'], s['timestep'])
        p.set_reference(reference if tracked else None, self.env.alpha)
        if not self._device_holds_sequence(p):
            p.set_base_actions(self.act_sequence)
        self._dev_state = self._dev_seq = None           # until the outcome of this step has been read back
        p.set_critic(critic_dim, critic_weights)         # a no-op while dim and weights stay what they were
        if advance:
            p.set_env_reward(self._env_reward())
        if self.noise == 'philox':
            if advance:
                # the device keys its stream with seed argument + time step of its state; one launch of the captured step
                p.plan(seed - int(s['timestep']), niter, 1)
            else:
                p.generate_noise(seed)
                p.iterate(niter)
        else:
            # The host stream of planning step t + 1 depends on nothing but its seed, so it is drawn -- and sent to the
            # device, piece by piece behind the generator -- while the device works on step t.  What was drawn ahead
            # is used only if the next call asks for exactly that stream from the same planner; the process-global
            # numpy generator is moved when the draws are consumed, as if they had been made then (one worker runs
            # in-process in the reference, trajopt/utils.py:171-172, and leaves the global generator behind its draws;
            # pool workers re-seed only their own copies).
            def draw(sd):
                return p.draw_noise(worker_seeds(sd, self.paths_per_cpu, self.num_cpu), _host_threads(),
                                    want_state=self.num_cpu == 1)
            want = (id(p), seed, self.paths_per_cpu, self.num_cpu, H, self.m)
            if self._ahead is not None and self._ahead[0] == want:
                state = self._ahead[1]
            else:
                state = draw(seed)
            self._ahead = None
            if state is not None:
                _set_numpy_state(state)
            p.commit_noise()
            p.iterate(niter)
            if advance:
                p.advance()
            if self.draw_ahead and 0 <= seed + 1 + (self.num_cpu - 1) * self.paths_per_cpu < 2 ** 32:
                self._ahead = ((id(p), seed + 1, self.paths_per_cpu, self.num_cpu, H, self.m), draw(seed + 1))
        if not advance:
            self.act_sequence = p.act_sequence()
            self._dev_seq = (id(p), self.act_sequence.tobytes())
            self._dev_state = self._state_record(p, s)
            return
        # `advance_time` (trajopt/algos/mppi.py:67-89) from the device's log row
        row, seq = p.read_step()
        nq, nv = p.nq, p.nv
        q2, v2, hand, r, action = row[:nq], row[nq:nq + nv], row[nq + nv:nq + nv + 3], row[nq + nv + 6], row[nq + nv + 7:]
        self.sol_act.append(np.zeros_like(action) if len(self.sol_act) < 10 else action.copy())
        after = dict(qp=q2.copy(), qv=v2.copy(), target_pos=np.array(s['target_pos'], dtype=float), timestep=s['timestep'] + 1)
        self.env._adopt_step(after, action, hand)
        self.sol_state.append(self.env.get_env_state().copy())
        self.sol_obs.append(self.env._get_obs(goal_dist=True))
        self.sol_reward.append(float(r))
        self.act_sequence = seq
        self._dev_seq = (id(p), seq.tobytes())
        self._dev_state = self._state_record(p, after)

    def _attach_critic_rewards(self, paths, critic, dim):
        """Terminal-value inputs by `dim` (trajopt/algos/mppi.py:134-159, 191)."""
        import torch
        for path in paths:
            values = []
            if critic is not None:
                obs = torch.tensor(np.asarray(path["next_observations"]), dtype=torch.float32)
                if dim == 14:
                    x = obs[:, :14]
                elif dim == 15:
                    d = torch.linalg.norm(obs[:, -6:-3] - obs[:, -3:], dim=1, keepdim=True)
                    x = torch.cat((obs[:, :14], d), dim=1)
                elif dim == 17:
                    x = torch.cat((obs[:, :14], obs[:, -3:]), dim=1)
                elif dim == 3:
                    x = obs[:, -6:-3]
                elif dim == 6:
                    x = obs[:, -6:]
                else:
                    raise ValueError("unsupported critic input dim %r" % (dim,))
                values = critic(x).detach().numpy()
            path["critic_rewards"] = np.array(values)
