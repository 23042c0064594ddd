"""ctypes binding of the C ABI declared in include/mppi_hip.h.

There is no CPU fallback: if the HIP library is missing or cannot be loaded the
import of anything that needs it raises.
"""
import ctypes as C
import os

import numpy as np

MAXV = 32
NOALLOC_MAX_K = 262144
_d, _i32, _u64, _vp = C.c_double, C.c_int32, C.c_uint64, C.c_void_p

F32, F64 = 0, 1
REWARD_TYPES = {"dense": 0, "sparse": 1, "tracking": 2, "cooling": 3}
TAIL = {"repeat": 0, "mean": 1}
(READ_RETURNS, READ_ACT_SEQ, READ_ARGMAX, READ_ACTIONS, READ_NEXT_OBS, READ_REWARDS, READ_STATUS, READ_RMAX,
 READ_PARTIALS, READ_STATUS_DETAIL) = range(10)

LIB_PATH = os.environ.get("MPPI_HIP_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmppi_hip.so")


class ModelDescC(C.Structure):
    _fields_ = [
        ("name", C.c_char * 64), ("nv", _i32), ("frame_skip", _i32), ("parent", _i32 * MAXV),
        ("limited", _i32 * MAXV), ("contact_link", _i32), ("hand_link", _i32), ("h", _d),
        ("pos", (_d * 3) * MAXV), ("axis", (_d * 3) * MAXV), ("mass", _d * MAXV), ("com", (_d * 3) * MAXV),
        ("inertia", (_d * 6) * MAXV), ("armature", _d * MAXV), ("damping", _d * MAXV),
        ("range_lo", _d * MAXV), ("range_hi", _d * MAXV), ("gear", _d * MAXV), ("ctrl_lo", _d * MAXV),
        ("ctrl_hi", _d * MAXV), ("dof_invweight", _d * MAXV), ("contact_pos", _d * 3),
        ("contact_radius", _d), ("contact_plane_z", _d), ("contact_margin", _d), ("contact_invweight", _d),
        ("hand_pos", _d * 3), ("solref_k", _d), ("solref_b", _d), ("solimp", _d * 5),
        ("gravity", _d * 3), ("jtype", _i32 * MAXV), ("nq", _i32), ("qadr", _i32 * MAXV), ("integrator", _i32),
    ]


class PlannerCfgC(C.Structure):
    _fields_ = [
        ("K", _i32), ("H", _i32), ("dtype", _i32), ("device", _i32), ("reward_type", _i32),
        ("default_act", _i32), ("k_offset", _i32), ("dump", _i32), ("P", _i32), ("sweep_cap", _i32),
        ("kappa", _d), ("gamma", _d),
        ("sigma", _d * MAXV), ("beta", _d * 3), ("mean", _d * MAXV),
    ]


class MppiError(RuntimeError):
    pass


_lib = None

_PROTOS = {
    "mppi_version": (C.c_int, []),
    "mppi_last_error": (C.c_char_p, []),
    "mppi_model_create": (C.c_int, [C.POINTER(ModelDescC), C.POINTER(_vp)]),
    "mppi_model_destroy": (None, [_vp]),
    "mppi_planner_create": (C.c_int, [_vp, C.POINTER(PlannerCfgC), C.POINTER(_vp)]),
    "mppi_planner_destroy": (None, [_vp]),
    "mppi_set_start_state": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "mppi_set_base_actions": (C.c_int, [_vp, _vp]),
    "mppi_set_reference": (C.c_int, [_vp, _vp, _i32, _d]),
    "mppi_set_params": (C.c_int, [_vp, _d, _d, _i32]),
    "mppi_set_env_reward": (C.c_int, [_vp, _i32]),
    "mppi_clear_status": (C.c_int, [_vp]),
    "mppi_set_critic": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, C.c_float]),
    "mppi_upload_noise": (C.c_int, [_vp, _vp, _vp]),
    "mppi_host_staging": (C.c_int, [_vp, C.POINTER(C.c_void_p)]),
    "mppi_draw_noise": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "mppi_commit_noise": (C.c_int, [_vp, _vp]),
    "mppi_generate_noise": (C.c_int, [_vp, _u64, _vp]),
    "mppi_generate_noise_step": (C.c_int, [_vp, _u64, _vp]),
    "mppi_advance": (C.c_int, [_vp, _vp]),
    "mppi_advance_noise": (C.c_int, [_vp, C.c_uint64, _vp]),
    "mppi_plan": (C.c_int, [_vp, _u64, _i32, _i32, _vp]),
    "mppi_read_log": (C.c_int, [_vp, _i32, _vp, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "mppi_read_step": (C.c_int, [_vp, _vp, _vp]),
    "mppi_upload_actions": (C.c_int, [_vp, _vp, _vp]),
    "mppi_rollout": (C.c_int, [_vp, _vp]),
    "mppi_iterate": (C.c_int, [_vp, _i32, _vp]),
    "mppi_local_partials": (C.c_int, [_vp, _vp]),
    "mppi_partials_ptr": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "mppi_merge": (C.c_int, [_vp, _vp, _i32, _vp]),
    "mppi_comm_unique_id": (C.c_int, [_vp, C.c_size_t]),
    "mppi_comm_init": (C.c_int, [_vp, _vp, C.c_size_t, _i32, _i32]),
    "mppi_shift": (C.c_int, [_vp, _vp, _vp]),
    "mppi_read": (C.c_int, [_vp, _i32, _vp, C.c_size_t]),
    "mppi_profile": (C.c_int, [_vp, _i32]),
    "mppi_profile_read": (C.c_int, [_vp, C.POINTER(_d), C.POINTER(_i32)]),
    "mppi_synchronize": (C.c_int, [_vp]),
    "mppi_env_step": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _d, _vp, _vp, _vp, _vp]),
    "mppi_env_forward": (C.c_int, [_vp, _i32, _vp, _vp]),
    "mppi_update_from_returns": (C.c_int, [_i32, _vp, _vp, _i32, _i32, _d, _vp]),
    "mppi_update_from_returns_noalloc": (C.c_int, [_i32, _vp, _vp, _i32, _i32, _d, _vp, _vp]),
    "mppi_numpy_normals": (C.c_int, [C.c_uint32, C.c_size_t, _vp, _i32, _vp]),
    "mppi_numpy_normals_many": (C.c_int, [_vp, _i32, C.c_size_t, _vp, _i32]),
    "mppi_numpy_normals_test_margin": (None, [_d]),
    "mppi_philox4x32": (None, [_vp, _vp, _vp]),
}
EXPORTS = tuple(_PROTOS)


def lib():
    """The loaded HIP library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MppiError("HIP library %s is missing: build it with `python -c 'import __graft_entry__ as g; "
                            "g.build()'` or `make -C trajopt_amd/csrc`" % LIB_PATH)
        # PyTorch-ROCm bundles its own HIP runtime under the same soname as the system one; a
        # process that ends up with both cannot open the GPU from the second.  Loading torch
        # first makes this library bind to the runtime torch uses, so tensors, streams and
        # RCCL buffers are shared with it.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().mppi_last_error()
        raise MppiError("mppi_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a):
    """Address of a (contiguous) numpy array as c_void_p; `ndarray.ctypes` is an order of magnitude slower."""
    return None if a is None else _vp(a.__array_interface__["data"][0])


def desc_to_c(md) -> ModelDescC:
    m = ModelDescC()
    m.name = md.name.encode()
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
        for f in ("mass", "armature", "damping", "range_lo", "range_hi", "gear", "ctrl_lo", "ctrl_hi",
                  "dof_invweight"):
            getattr(m, f)[i] = float(getattr(md, f)[i])
    for c in range(3):
        m.contact_pos[c] = md.contact_pos[c]; m.hand_pos[c] = md.hand_pos[c]; m.gravity[c] = float(md.gravity[c])
    m.contact_radius, m.contact_plane_z = md.contact_radius, md.contact_plane_z
    m.contact_margin, m.contact_invweight = md.contact_margin, md.contact_invweight
    m.solref_k, m.solref_b = md.solref_k, md.solref_b
    for c in range(5):
        m.solimp[c] = md.solimp[c]
    return m
