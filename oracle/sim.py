"""ctypes binding of the fp64 C++ oracle (test infrastructure; see oracle/__init__.py)."""
import ctypes as C
import os

import numpy as np

from . import LIB_PATH, build

MAXV = 32
_d = C.c_double
_i = C.c_int


class CModel(C.Structure):
    _fields_ = [
        ("nv", _i), ("frame_skip", _i), ("parent", _i * MAXV), ("limited", _i * MAXV),
        ("contact_link", _i), ("hand_link", _i), ("h", _d),
        ("pos", (_d * 3) * MAXV), ("axis", (_d * 3) * MAXV), ("mass", _d * MAXV),
        ("com", (_d * 3) * MAXV), ("inertia", (_d * 6) * MAXV), ("armature", _d * MAXV),
        ("damping", _d * MAXV), ("range_lo", _d * MAXV), ("range_hi", _d * MAXV),
        ("gear", _d * MAXV), ("ctrl_lo", _d * MAXV), ("ctrl_hi", _d * MAXV),
        ("dof_invweight", _d * MAXV), ("contact_pos", _d * 3), ("contact_radius", _d),
        ("contact_plane_z", _d), ("contact_margin", _d), ("contact_invweight", _d),
        ("hand_pos", _d * 3), ("solref_k", _d), ("solref_b", _d), ("solimp", _d * 5),
        ("gravity", _d * 3), ("jtype", _i * MAXV), ("nq", _i), ("qadr", _i * MAXV), ("integrator", _i),
    ]


REWARD_TYPES = {"dense": 0, "sparse": 1, "tracking": 2, "cooling": 3}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        assert _lib.oracle_model_size() == C.sizeof(CModel), "oracle struct layout mismatch"
    return _lib


def to_cmodel(md) -> CModel:
    m = CModel()
    m.nv, m.frame_skip, m.h = md.nv, md.frame_skip, md.h
    m.nq = int(md.nq)
    m.integrator = int(md.integrator)
    m.contact_link, m.hand_link = md.contact_link, md.hand_link
    for i in range(md.nv):
        m.parent[i] = int(md.parent[i]); m.limited[i] = int(md.limited[i]); m.jtype[i] = int(md.jtype[i])
        m.qadr[i] = int(md.qadr[i])
        for c in range(3):
            m.pos[i][c] = md.pos[i][c]; m.axis[i][c] = md.axis[i][c]; m.com[i][c] = md.com[i][c]
        for c in range(6):
            m.inertia[i][c] = md.inertia[i][c]
        for f in ("mass", "armature", "damping", "range_lo", "range_hi", "gear", "ctrl_lo",
                  "ctrl_hi", "dof_invweight"):
            getattr(m, f)[i] = float(getattr(md, f)[i])
    for c in range(3):
        m.contact_pos[c] = md.contact_pos[c]; m.hand_pos[c] = md.hand_pos[c]; m.gravity[c] = float(md.gravity[c])
    m.contact_radius, m.contact_plane_z = md.contact_radius, md.contact_plane_z
    m.contact_margin, m.contact_invweight = md.contact_margin, md.contact_invweight
    m.solref_k, m.solref_b = md.solref_k, md.solref_b
    for c in range(5):
        m.solimp[c] = md.solimp[c]
    return m


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleSim:
    """fp64 simulator restatement bound to one compiled model."""

    def __init__(self, md):
        self.md = md
        self.cm = to_cmodel(md)
        self.nv = md.nv
        self.nq = md.nq
        self.L = lib()

    def substep(self, q, v, ctrl):
        q, v, ctrl = _f64(q), _f64(v), _f64(ctrl)
        q2, v2, hand = np.zeros(self.nq), np.zeros(self.nv), np.zeros(3)
        info = np.zeros(3, np.int32)
        self.L.oracle_substep(C.byref(self.cm), _p(q), _p(v), _p(ctrl), _p(q2), _p(v2), _p(hand), _p(info))
        return q2, v2, hand, info

    def fk_hand(self, q):
        q = _f64(q)
        hand = np.zeros(3)
        self.L.oracle_fk_hand(C.byref(self.cm), _p(q), _p(hand))
        return hand

    def mass_matrix(self, q, qd=None):
        q = _f64(q)
        qd = np.zeros(self.nv) if qd is None else _f64(qd)
        M, b = np.zeros((self.nv, self.nv)), np.zeros(self.nv)
        self.L.oracle_mass_matrix(C.byref(self.cm), _p(q), _p(qd), _p(M), _p(b))
        return M, b

    def env_step(self, q, v, target, act, env_timestep=0, reward_type="dense", reference=None, alpha=1.0):
        q, v, target, act = _f64(q), _f64(v), _f64(target), _f64(act)
        ref = None if reference is None else _f64(reference)
        q2, v2, obs = np.zeros(self.nq), np.zeros(self.nv), np.zeros(self.nq + self.nv + 6)
        r = C.c_double(0.0)
        info = np.zeros(3, np.int32)
        self.L.oracle_env_step(C.byref(self.cm), _p(q), _p(v), _p(target), _p(act), _i(int(env_timestep)),
                               _i(REWARD_TYPES[reward_type]), None if ref is None else _p(ref),
                               _i(0 if ref is None else len(ref)), _d(alpha), _p(q2), _p(v2), _p(obs),
                               C.byref(r), _p(info))
        return q2, v2, obs, r.value, info

    def rollout(self, q0, v0, target, timestep0, act, reward_type="dense", reference=None, alpha=1.0,
                gamma=1.0, dump=False, nthreads=1):
        """act (K,H,nv) -> returns (K,), and if dump: next_obs (K,H,2nv+6), rewards (K,H)."""
        q0, v0, target, act = _f64(q0), _f64(v0), _f64(target), _f64(act)
        K, H, _ = act.shape
        ref = None if reference is None else _f64(reference)
        R = np.zeros(K)
        nobs = np.zeros((K, H, self.nq + self.nv + 6)) if dump else None
        rew = np.zeros((K, H)) if dump else None
        self.L.oracle_rollout(C.byref(self.cm), _p(q0), _p(v0), _p(target), _i(int(timestep0)), _p(act),
                              _i(K), _i(H), _i(REWARD_TYPES[reward_type]), None if ref is None else _p(ref),
                              _i(0 if ref is None else len(ref)), _d(alpha), _d(gamma), _p(R),
                              None if nobs is None else _p(nobs), None if rew is None else _p(rew),
                              _i(int(nthreads)))
        return (R, nobs, rew) if dump else R

    def count_ops(self, q, v, ctrl):
        q, v, ctrl = _f64(q), _f64(v), _f64(ctrl)
        c = np.zeros(6, np.int64)
        self.L.oracle_count_ops(C.byref(self.cm), _p(q), _p(v), _p(ctrl), _p(c))
        return dict(add=int(c[0]), mul=int(c[1]), div=int(c[2]), sqrt=int(c[3]), trig=int(c[4]), cmp=int(c[5]))
