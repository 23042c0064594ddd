"""Thin object wrappers over the C ABI: a model bound to its compiled kernels and a
device-resident planner."""
import ctypes as C
import os

import numpy as np

from . import _lib as L
from .model import ModelDesc, get_model


def default_device():
    return int(os.environ.get("LOCAL_RANK", "0"))


class DeviceModel:
    _cache = {}

    def __init__(self, md: ModelDesc):
        self.md = md
        self.nv = md.nv
        self.nq = md.nq
        self._h = L._vp()
        self._c = L.desc_to_c(md)
        L.check(L.lib().mppi_model_create(C.byref(self._c), C.byref(self._h)))

    @classmethod
    def get(cls, name):
        if name not in cls._cache:
            cls._cache[name] = cls(get_model(name))
        return cls._cache[name]

    def env_step(self, qpos, qvel, target, act, timestep=0, reward_type="dense", reference=None, alpha=1.0,
                 device=None):
        """One fp64 `env.step` on the device -> (qpos', qvel', obs[nq+nv+6], reward)."""
        nv, nq = self.nv, self.nq
        qpos, qvel, target, act = L.f64(qpos), L.f64(qvel), L.f64(target), L.f64(act)
        assert qpos.shape == (nq,) and qvel.shape == (nv,) and act.shape == (nv,) and target.shape == (3,)
        ref = None if reference is None else L.f64(reference)
        if ref is not None and len(ref) == 0 and reward_type in ("tracking", "cooling"):
            # an empty trajectory is still a trajectory: indexing it raises IndexError in the reference
            raise IndexError("index %d is out of bounds for the reference trajectory" % (timestep + 1))
        q2, v2, obs = np.empty(nq), np.empty(nv), np.empty(nq + nv + 6)
        r = C.c_double(0.0)
        rc = L.lib().mppi_env_step(self._h, default_device() if device is None else device, L.ptr(qpos), L.ptr(qvel),
                                   L.ptr(target), L.ptr(act), int(timestep), L.REWARD_TYPES[reward_type],
                                   L.ptr(ref), 0 if ref is None else len(ref), float(alpha), L.ptr(q2), L.ptr(v2),
                                   L.ptr(obs), C.cast(C.byref(r), L._vp))
        if rc == -5:
            raise IndexError("index %d is out of bounds for the reference trajectory" % (timestep + 1))
        L.check(rc)
        return q2, v2, obs, r.value

    def forward_hand(self, qpos, device=None):
        """End-effector site position at qpos (what `sim.forward()` leaves in `site_xpos`)."""
        qpos = L.f64(qpos)
        hand = np.empty(3)
        L.check(L.lib().mppi_env_forward(self._h, default_device() if device is None else device, L.ptr(qpos),
                                         L.ptr(hand)))
        return hand


def update_from_returns(actions, R, kappa, device=None):
    """`MPPI.update` arithmetic on the device for caller-supplied (K,H,M) actions and (K,) returns."""
    actions, R = L.f64(actions), L.f64(R)
    K, H, M = actions.shape
    out = np.empty((H, M))
    L.check(L.lib().mppi_update_from_returns(default_device() if device is None else device, L.ptr(actions),
                                             L.ptr(R), K, H * M, float(kappa), L.ptr(out)))
    return out


class DevicePlanner:
    """Device-resident state of one planner (or of one rank's shard of it)."""

    def __init__(self, model: DeviceModel, K, H, kappa=1.0, gamma=1.0, sigma=None, betas=(1.0, 0.0, 0.0),
                 mean=None, default_act="repeat", reward_type="dense", dtype="f32", device=None, k_offset=0,
                 dump=False, planners=1, sweep_cap=0):
        """planners > 1 batches that many independent planners (own start state, base actions, noise
        stream, log) in one handle: every per-planner array gains a leading axis of that size."""
        self.model, self.K, self.H, self.nv, self.nq = model, int(K), int(H), model.nv, model.nq
        self.P = int(planners)
        self.dtype = dtype
        cfg = L.PlannerCfgC()
        cfg.K, cfg.H = self.K, self.H
        cfg.dtype = L.F32 if dtype == "f32" else L.F64
        cfg.device = default_device() if device is None else device
        cfg.reward_type = L.REWARD_TYPES[reward_type]
        cfg.default_act = L.TAIL[default_act] if default_act in L.TAIL else L.TAIL["mean"]
        cfg.k_offset, cfg.dump = int(k_offset), int(bool(dump))
        cfg.P = self.P
        cfg.sweep_cap = int(sweep_cap)
        cfg.kappa, cfg.gamma = float(kappa), float(gamma)
        sigma = np.ones(self.nv) if sigma is None else np.broadcast_to(np.asarray(sigma, float), (self.nv,))
        mean = np.zeros(self.nv) if mean is None else np.asarray(mean, float)
        for j in range(self.nv):
            cfg.sigma[j], cfg.mean[j] = float(sigma[j]), float(mean[j])
        for j in range(3):
            cfg.beta[j] = float(betas[j])
        self.dump = bool(dump)
        self._device = int(cfg.device)
        self._ref_sent = (None, None)       # what the device holds: calls that would not change it are skipped
        self._critic_sent = None
        self._h = L._vp()
        self._staging = None
        L.check(L.lib().mppi_planner_create(model._h, C.byref(cfg), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            L.lib().mppi_planner_destroy(self._h)
            self._h = None
            self._staging = None            # the memory behind it is gone

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inputs ---------------------------------------------------------------------
    def _lead(self, *shape):
        return shape if self.P == 1 else (self.P,) + shape

    def set_start_state(self, qpos, qvel, target, timestep=0):
        qpos, qvel, target = L.f64(qpos), L.f64(qvel), L.f64(target)
        ts = np.ascontiguousarray(np.broadcast_to(np.asarray(timestep, dtype=np.int32), (self.P,)))
        assert qpos.shape == self._lead(self.nq) and qvel.shape == self._lead(self.nv) and target.shape == self._lead(3)
        L.check(L.lib().mppi_set_start_state(self._h, L.ptr(qpos), L.ptr(qvel), L.ptr(target), L.ptr(ts)))

    def set_base_actions(self, act):
        act = L.f64(act)
        assert act.shape == self._lead(self.H, self.nv), act.shape
        L.check(L.lib().mppi_set_base_actions(self._h, L.ptr(act)))

    def set_reference(self, reference, alpha=1.0):
        ref = None if reference is None else L.f64(reference)
        assert ref is None or (ref.ndim == 2 and ref.shape[1] == 3)
        sent = (None if ref is None else ref.tobytes(), float(alpha))
        if sent == self._ref_sent:
            return
        L.check(L.lib().mppi_set_reference(self._h, L.ptr(ref), 0 if ref is None else len(ref), float(alpha)))
        self._ref_sent = sent

    def set_params(self, kappa, gamma, reward_type):
        L.check(L.lib().mppi_set_params(self._h, float(kappa), float(gamma), L.REWARD_TYPES[reward_type]))

    def set_env_reward(self, reward_type):
        """Reward type the planner's own step of the real environment reports (None: the rollouts')."""
        rt = -1 if reward_type is None else L.REWARD_TYPES[reward_type]
        if rt != getattr(self, "_env_reward_sent", -1):
            L.check(L.lib().mppi_set_env_reward(self._h, rt))
            self._env_reward_sent = rt

    def set_critic(self, dim, weights=None):
        """weights = (W1, b1, W2, b2, W3, b3) as float32 arrays in torch's row-major layout, or
        None / dim 0 to remove the critic."""
        if not dim or weights is None:
            if self._critic_sent is not None:
                L.check(L.lib().mppi_set_critic(self._h, 0, 0, None, None, None, None, None, 0.0))
                self._critic_sent = None
            return
        W1, b1, W2, b2, W3, b3 = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        sent = (int(dim),) + tuple(w.tobytes() for w in (W1, b1, W2, b2, W3, b3))
        if sent == self._critic_sent:
            return
        inner = W2.shape[0]
        assert W1.shape == (inner, dim) and W2.shape == (inner, inner) and W3.size == inner and b3.size == 1
        L.check(L.lib().mppi_set_critic(self._h, int(dim), int(inner), L.ptr(W1), L.ptr(b1), L.ptr(W2), L.ptr(b2),
                                        L.ptr(W3), float(b3.reshape(-1)[0])))
        self._critic_sent = sent

    def host_staging(self):
        """The planner's page-locked host buffer as a (K, H, m) float64 array (leading P for batches): draws written
        here upload at the bus rate.  The same memory at every call; it dies with the planner."""
        if self._staging is None:
            ptr = C.c_void_p()
            L.check(L.lib().mppi_host_staging(self._h, C.byref(ptr)))
            shape = self._lead(self.K, self.H, self.nv)
            n = int(np.prod(shape))
            self._staging = np.ctypeslib.as_array((C.c_double * n).from_address(ptr.value)).reshape(shape)
        return self._staging

    def upload_noise(self, raw):
        raw = L.f64(raw)
        assert raw.shape == self._lead(self.K, self.H, self.nv), raw.shape
        L.check(L.lib().mppi_upload_noise(self._h, L.ptr(raw), None))

    def draw_noise(self, seeds, nthreads=16, want_state=False):
        """numpy's legacy normal stream for every worker seed, made on the host into the staging buffer and sent to the
        device while it is being made (on a side stream: the planner's own work goes on).  Returns the 628-word
        generator state record (one seed, want_state) or None.  `commit_noise` makes the draws the planner's noise."""
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        state = np.empty(628, dtype=np.uint32) if want_state else None
        L.check(L.lib().mppi_draw_noise(self._h, L.ptr(seeds), len(seeds), int(nthreads), L.ptr(state)))
        return state

    def commit_noise(self, stream=None):
        L.check(L.lib().mppi_commit_noise(self._h, stream))

    def generate_noise(self, seed, stream=None):
        L.check(L.lib().mppi_generate_noise(self._h, int(seed) & (2 ** 64 - 1), stream))

    def generate_noise_step(self, seed, stream=None):
        """Philox stream keyed by seed + the planner's current timestep (read on the device)."""
        L.check(L.lib().mppi_generate_noise_step(self._h, int(seed) & (2 ** 64 - 1), stream))

    def advance(self, stream=None):
        """Device-resident advance_time: real-env fp64 step with act_sequence[0], log, shift."""
        L.check(L.lib().mppi_advance(self._h, stream))

    def advance_noise(self, seed, stream=None):
        """advance() and, in the same launch, generate_noise_step(seed) for the next planning step."""
        L.check(L.lib().mppi_advance_noise(self._h, int(seed) & (2 ** 64 - 1), stream))

    def plan(self, seed, niter=1, nsteps=1, stream=None):
        """nsteps whole planning steps (noise, niter iterations, real-env step, shift) as graph replays."""
        L.check(L.lib().mppi_plan(self._h, int(seed) & (2 ** 64 - 1), int(niter), int(nsteps), stream))

    def read_log(self, max_rows=4096, planner=0):
        """Most recent rows of one planner's real-env log, oldest first: dict of arrays + total count."""
        nv, nq = self.nv, self.nq
        rows = np.empty((max_rows, nq + 2 * nv + 7))
        n, total = C.c_int32(), C.c_int32()
        L.check(L.lib().mppi_read_log(self._h, int(planner), L.ptr(rows), int(max_rows), C.byref(n), C.byref(total)))
        r = rows[:n.value]
        s = nq + nv
        return dict(qpos=r[:, :nq], qvel=r[:, nq:s], hand=r[:, s:s + 3], target=r[:, s + 3:s + 6],
                    reward=r[:, s + 6], action=r[:, s + 7:], total=total.value)

    def read_step(self):
        """What the last device-resident advance produced: (rows, act_sequence) -- the newest log row of every planner
        (qpos', qvel', hand, target, reward, action: (3M+7,), leading P for batches) and the shifted act_sequence."""
        nv, nq = self.nv, self.nq
        rows = np.empty(self._lead(nq + 2 * nv + 7))
        seq = np.empty(self._lead(self.H, nv))
        L.check(L.lib().mppi_read_step(self._h, L.ptr(rows), L.ptr(seq)))
        return rows, seq

    def upload_actions(self, act):
        act = L.f64(act)
        assert act.shape == self._lead(self.K, self.H, self.nv), act.shape
        L.check(L.lib().mppi_upload_actions(self._h, L.ptr(act), None))

    # ---- compute --------------------------------------------------------------------
    def rollout(self, stream=None):
        L.check(L.lib().mppi_rollout(self._h, stream))

    def iterate(self, niter=1, stream=None):
        L.check(L.lib().mppi_iterate(self._h, int(niter), stream))

    def local_partials(self, stream=None):
        L.check(L.lib().mppi_local_partials(self._h, stream))

    def profile(self, enable=True):
        L.check(L.lib().mppi_profile(self._h, int(bool(enable))))

    def profile_read(self):
        """(summed rollout-kernel milliseconds, launches) since the last read."""
        ms, n = C.c_double(), C.c_int32()
        L.check(L.lib().mppi_profile_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def partials_tensor(self):
        """torch view (no copy) of this shard's statistics buffer, for collectives."""
        import torch
        ptr, nbytes = self.partials_ptr()

        class _View:
            __cuda_array_interface__ = {"data": (ptr, False), "shape": (nbytes // 8,), "typestr": "<f8",
                                        "strides": None, "version": 2}
        return torch.as_tensor(_View(), device="cuda:%d" % self._device)

    def partials_ptr(self):
        p, n = L._vp(), C.c_size_t()
        L.check(L.lib().mppi_partials_ptr(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def merge(self, gathered_ptr, nranks, stream=None):
        L.check(L.lib().mppi_merge(self._h, L._vp(gathered_ptr), int(nranks), stream))

    @staticmethod
    def comm_unique_id():
        """128-byte id of a new exchange group (made on one rank, handed to the others by the caller)."""
        buf = C.create_string_buffer(128)
        L.check(L.lib().mppi_comm_unique_id(buf, 128))
        return buf.raw

    def comm_init(self, unique_id, rank, nranks):
        """Join the exchange group (collective over the ranks): iterate() and plan() then run sharded, with the
        all-gather of the shard statistics made by the library on the planner's stream."""
        assert len(unique_id) == 128
        L.check(L.lib().mppi_comm_init(self._h, C.c_char_p(bytes(unique_id)), 128, int(rank), int(nranks)))

    def shift(self, override=None, stream=None):
        o = None if override is None else L.f64(override)
        L.check(L.lib().mppi_shift(self._h, L.ptr(o), stream))

    def synchronize(self):
        L.check(L.lib().mppi_synchronize(self._h))

    # ---- outputs --------------------------------------------------------------------
    def _read(self, what, shape, dtype=np.float64):
        out = np.empty(shape, dtype=dtype)
        L.check(L.lib().mppi_read(self._h, what, L.ptr(out), out.nbytes))
        return out

    def returns(self):
        return self._read(L.READ_RETURNS, self._lead(self.K))

    def act_sequence(self):
        return self._read(L.READ_ACT_SEQ, self._lead(self.H, self.nv))

    def argmax(self):
        a = self._read(L.READ_ARGMAX, (self.P,), np.int64)
        return int(a[0]) if self.P == 1 else a

    def rmax(self):
        a = self._read(L.READ_RMAX, (self.P,))
        return float(a[0]) if self.P == 1 else a

    def actions(self):
        return self._read(L.READ_ACTIONS, self._lead(self.K, self.H, self.nv))

    def next_observations(self):
        return self._read(L.READ_NEXT_OBS, self._lead(self.K, self.H, self.nq + self.nv + 6))

    def rewards(self):
        return self._read(L.READ_REWARDS, self._lead(self.K, self.H))

    def status(self):
        return int(self._read(L.READ_STATUS, (1,), np.int32)[0])

    def clear_status(self):
        """The status word is sticky; this clears it (behind the work already enqueued)."""
        L.check(L.lib().mppi_clear_status(self._h))

    def status_detail(self):
        """(bits, first sample of the wave + 1, env step, planner) of the first capped solve a general-variant rollout reported."""
        return [int(x) for x in self._read(L.READ_STATUS_DETAIL, (4,), np.int32)]

    def partials(self):
        return self._read(L.READ_PARTIALS, self._lead(3 + self.H * self.nv))
