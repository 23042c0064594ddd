"""`Reacher7DOFEnv` with the reference's duck-type (trajopt/envs/reacher_env.py:8-174,
trajopt/envs/mujoco_env.py:26-140), stepping on the GPU.

The reference builds a MuJoCo simulation of `assets/sawyer.xml`; here the same model is
compiled by `trajopt_amd.model` and every `step` / `forward` is one fp64 kernel launch
through the C ABI (`mppi_env_step`, `mppi_env_forward`).  There is no CPU simulator in
this package.
"""
import numpy as np

from ..device import DeviceModel


class _Data:
    """`sim.data` look-alike: qpos, qvel and the two site positions the examples read.

    `site_xpos` after `sim.forward()` is the end-effector position of the current qpos, which costs a kernel launch
    and a read-back; it is computed when somebody looks at it, not when forward() is called (a `set_env_state`
    followed by `step`, as in `MPPI.advance_time`, never looks)."""

    def __init__(self, nv):
        self.qpos = np.zeros(nv)
        self.qvel = np.zeros(nv)
        self.ctrl = np.zeros(nv)
        self._site_xpos = np.zeros((2, 3))
        self._refresh = None                # callable that brings _site_xpos up to date, or None when it is

    @property
    def site_xpos(self):
        if self._refresh is not None:
            refresh, self._refresh = self._refresh, None
            refresh()
        return self._site_xpos

    @site_xpos.setter
    def site_xpos(self, value):
        self._refresh = None
        self._site_xpos = np.asarray(value, dtype=np.float64)


class _Model:
    """`sim.model` look-alike: `site_pos` (the target is moved through it), name lookups."""

    def __init__(self, md):
        self.nq = self.nv = self.nu = md.nv
        self.site_pos = np.zeros((2, 3))
        self.site_pos[1] = (1.0, 0.0, 0.0)            # `target` site of the model file
        self.actuator_ctrlrange = np.stack([md.ctrl_lo, md.ctrl_hi], axis=1)
        self._sites = {"finger": 0, "target": 1}

    def site_name2id(self, name):
        return self._sites[name]


class _Box:
    """`gym.spaces.Box` look-alike (gym is not a dependency here): the bounds of `action_space` and
    `observation_space` (trajopt/envs/mujoco_env.py:58-65), which policy-learning code reads for shapes and limits."""

    def __init__(self, low, high, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)
        self.np_random = np.random.RandomState()

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def sample(self):
        """Uniform inside bounded dimensions, standard normal along unbounded ones."""
        bounded = np.isfinite(self.low) & np.isfinite(self.high)
        x = self.np_random.normal(size=self.shape)
        u = self.np_random.uniform(size=self.shape)
        lo, hi = np.where(bounded, self.low, 0.0), np.where(bounded, self.high, 0.0)
        return np.where(bounded, lo + u * (hi - lo), x).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    __contains__ = contains

    def __repr__(self):
        return "Box%s" % (self.shape,)

    def __eq__(self, other):
        return isinstance(other, _Box) and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high)


class _Sim:
    def __init__(self, env):
        self._env = env
        self.data = env.data
        self.model = env.model

    def forward(self):
        self._env._forward()

    def reset(self):
        e = self._env
        e.data.qpos[:] = e.init_qpos
        e.data.qvel[:] = e.init_qvel

    def step(self):
        raise NotImplementedError("single physics substeps are not exposed; use env.step")


class Reacher7DOFEnv:
    MODEL_NAME = "reacher_7dof"

    def __init__(self, reward_type='dense', reference=None):
        self.env_name = 'reacher_7dof'
        self.seeding = False
        self.real_step = True
        self.env_timestep = 0
        self.alpha = 1.0
        self.reward_type = reward_type
        self.reference = reference
        self._dev = DeviceModel.get(self.MODEL_NAME)
        md = self._dev.md
        self.frame_skip = md.frame_skip
        self.model = _Model(md)
        self.data = _Data(md.nv)
        self.sim = _Sim(self)
        self.hand_sid, self.target_sid = 0, 1
        self.init_qpos = np.zeros(md.nv)
        self.init_qvel = np.zeros(md.nv)
        self.observation_dim = 26          # value the reference declares (reacher_env.py:31)
        self.action_dim = md.nv
        self.obs_dim = 2 * md.nv + 6
        self.mujoco_render_frames = False
        self.viewer = None
        self.metadata = {'render.modes': ['human', 'rgb_array'],
                         'video.frames_per_second': int(np.round(1.0 / (md.h * md.frame_skip)))}
        self.action_space = _Box(self.model.actuator_ctrlrange[:, 0], self.model.actuator_ctrlrange[:, 1])
        self.observation_space = _Box(-np.inf * np.ones(self.obs_dim), np.inf * np.ones(self.obs_dim))
        self.np_random = np.random.RandomState()
        self.goal_dist = np.array([0.0])
        self._ctor = dict(reward_type=reward_type, reference=reference)
        # the reference constructor takes one zero-action step (mujoco_env.py:53)
        self.data.site_xpos[self.target_sid] = self.model.site_pos[self.target_sid]
        self._forward()
        self._step(np.zeros(md.nv))

    # ---- pickling: rebuilt from constructor arguments, like gym's EzPickle --------------
    def __getstate__(self):
        return {"_ezpickle_args": (), "_ezpickle_kwargs": dict(self._ctor)}

    def __setstate__(self, d):
        kw = d.get("_ezpickle_kwargs", {})
        self.__init__(*d.get("_ezpickle_args", ()), **kw)

    @property
    def dt(self):
        return self._dev.md.h * self.frame_skip

    # ---- simulation -----------------------------------------------------------------
    def _forward(self):
        # the positions belong to the configuration at the time of the call, whenever they are looked at
        qpos, target = self.data.qpos.copy(), self.model.site_pos[self.target_sid].copy()
        self.data._refresh = lambda: self._forward_now(qpos, target)

    def _forward_now(self, qpos, target):
        self.data._site_xpos[self.hand_sid] = self._dev.forward_hand(qpos)
        self.data._site_xpos[self.target_sid] = target

    def _seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def do_simulation(self, ctrl, n_frames):
        if n_frames != self.frame_skip:
            raise ValueError("the compiled kernel advances exactly frame_skip substeps")
        self._advance(np.asarray(ctrl, dtype=np.float64), reward_type='dense')

    def _advance(self, a, reward_type):
        ref = self.reference if reward_type in ('tracking', 'cooling') else None
        q2, v2, obs, r = self._dev.env_step(self.data.qpos, self.data.qvel, self.model.site_pos[self.target_sid],
                                            a, self.env_timestep, reward_type, ref, self.alpha)
        self.data.ctrl[:] = a
        self.data.qpos[:] = q2
        self.data.qvel[:] = v2
        self.data._refresh = None                              # the step leaves its own site positions
        self.data._site_xpos[self.hand_sid] = obs[-6:-3]       # position before the last substep's integration
        self.data._site_xpos[self.target_sid] = self.model.site_pos[self.target_sid]
        return r

    def _step(self, a):
        rt = self.reward_type if self.reward_type in ('sparse', 'tracking', 'cooling') else 'dense'
        reward = self._advance(np.asarray(a, dtype=np.float64), rt)
        self.env_timestep += 1
        hand, target = self.data.site_xpos[self.hand_sid], self.data.site_xpos[self.target_sid]
        self.goal_dist = np.array([np.linalg.norm(hand - target)])
        return self._get_obs(), reward, False, self.get_env_infos()

    def step(self, a):
        return self._step(a)

    def _adopt_step(self, state, action, hand):
        """Take over the outcome of a `step(action)` that a device-resident planner made from `state` with this
        environment's own reward: what `set_env_state(state); step(action)` leaves behind, without a launch."""
        self.sim.reset()
        self.model.site_pos[self.target_sid] = state['target_pos']
        self.env_timestep = state['timestep']
        self.data.qpos[:] = state['qp']
        self.data.qvel[:] = state['qv']
        self.data.ctrl[:] = action
        self.data._refresh = None
        self.data._site_xpos[self.hand_sid] = hand
        self.data._site_xpos[self.target_sid] = self.model.site_pos[self.target_sid]
        self.goal_dist = np.array([np.linalg.norm(self.data._site_xpos[self.hand_sid] - self.data._site_xpos[self.target_sid])])

    def _dense_step(self, a):
        """One step with the dense reward whatever `reward_type` says; goal_dist is left untouched."""
        reward = self._advance(np.asarray(a, dtype=np.float64), 'dense')
        ob = self._get_obs()
        self.env_timestep += 1
        return ob, reward, False, self.get_env_infos()

    def _get_obs(self, goal_dist=False):
        parts = [self.data.qpos.flat, self.data.qvel.flat, self.data.site_xpos[self.hand_sid],
                 self.data.site_xpos[self.target_sid]]
        if goal_dist:
            parts.append(self.goal_dist)
        return np.concatenate(parts)

    # ---- resets -----------------------------------------------------------------------
    def set_state(self, qpos, qvel):
        assert qpos.shape == (self.model.nq,) and qvel.shape == (self.model.nv,)
        self.data.qpos[:] = qpos
        self.data.qvel[:] = qvel
        self._forward()

    def robot_reset(self):
        self.set_state(self.init_qpos, self.init_qvel)

    def target_reset(self, goal=None):
        target_pos = np.zeros(3)
        if self.seeding is True:
            target_pos[0] = self.np_random.uniform(low=-0.1, high=0.1)
            target_pos[1] = self.np_random.uniform(low=-0.1, high=0.1)
            target_pos[2] = self.np_random.uniform(low=-0.1, high=0.1)
        if goal is not None:
            target_pos = np.array(goal)
        self.model.site_pos[self.target_sid] = target_pos
        self.data.site_xpos[self.target_sid] = target_pos
        self._forward()

    def reset_model(self, seed=None, goal=None, alpha=None):
        if seed is not None:
            self.seeding = True
            self._seed(seed)
        if alpha is not None:
            self.alpha = alpha
        self.robot_reset()
        self.target_reset(goal)
        self.env_timestep = 0
        return self._get_obs()

    def reset(self):
        return self.reset_model()

    def _reset(self):
        """gym's hook (trajopt/envs/mujoco_env.py:114-118): simulator reset, forward, model reset."""
        self.sim.reset()
        self.sim.forward()
        return self.reset_model()

    # ---- state dicts ---------------------------------------------------------------------
    def get_env_state(self):
        return dict(qp=self.data.qpos.copy(), qv=self.data.qvel.copy(),
                    target_pos=self.model.site_pos[self.target_sid].copy(), timestep=self.env_timestep)

    def set_env_state(self, state):
        self.sim.reset()
        self.model.site_pos[self.target_sid] = state['target_pos']
        self.env_timestep = state['timestep']
        self.set_state(state['qp'].copy(), state['qv'].copy())

    def get_env_infos(self):
        return dict(state=self.get_env_state())

    # ---- headless stand-ins for the viewer hooks --------------------------------------------
    def mj_viewer_setup(self):
        self.viewer = None

    def viewer_setup(self):
        pass

    def _get_viewer(self, mode=None):
        return None

    def mj_render(self):
        pass

    def render(self, *a, **k):
        return None

    def close(self):
        self.viewer = None

    def state_vector(self):
        return np.concatenate([self.data.qpos.flat, self.data.qvel.flat])
