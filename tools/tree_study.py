#!/usr/bin/env python3
"""Precision and timing study on the synthetic 18-DoF floating-base tree at K = 32768, H = 128:
fp32 rollouts against fp64 rollouts of the same Philox noise (relative state error
|x32 - x64| / (1 + |x64|), worst coordinate per sample, at t = 32 / 64 / 128), and the
rollout-kernel time of both.  Writes one JSON object (stdout, and the path given as argv[1]).
argv[2] names the model: tree_18dof (default; Euler-angle base) or tree_18dof_quat (quaternion base, same
start pose and velocity: the error then covers position, the four quaternion components and velocities)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajopt_amd.device import DeviceModel, DevicePlanner  # noqa: E402

NV, K, H = 18, 32768, 128


def run(model, dtype, dump, reps):
    p = DevicePlanner(model, K, H, kappa=5.0, gamma=0.99, sigma=np.ones(NV), betas=(0.25, 0.8, 0.0), dtype=dtype, dump=dump)
    rs = np.random.RandomState(3)
    q0 = np.r_[0.0, 0.0, 0.0, rs.randn(15) * 0.3]
    if model.nq > NV:
        # the same trunk attitude as the successive x, y, z hinges of the other model
        ax, ay, az = q0[3:6]
        hx = np.r_[np.cos(ax / 2), np.sin(ax / 2), 0, 0]
        hy = np.r_[np.cos(ay / 2), 0, np.sin(ay / 2), 0]
        hz = np.r_[np.cos(az / 2), 0, 0, np.sin(az / 2)]

        def qmul(a, b):
            return np.r_[a[0] * b[0] - a[1:] @ b[1:], a[0] * b[1:] + b[0] * a[1:] + np.cross(a[1:], b[1:])]
        q0 = np.r_[q0[:3], qmul(qmul(hx, hy), hz), q0[6:]]
    p.set_start_state(q0, rs.randn(NV) * 0.3, np.array([0.4, 0.1, 0.8]), 0)
    p.set_base_actions(np.zeros((H, NV)))
    p.generate_noise(2024)
    p.rollout()
    p.returns()
    p.profile(True)
    for _ in range(reps):
        p.rollout()
    p.returns()
    ms, n = p.profile_read()
    p.profile(False)
    return p, ms / max(n, 1)


def main():
    name = sys.argv[2] if len(sys.argv) > 2 else "tree_18dof"
    model = DeviceModel.get(name)
    ND = model.nq + NV
    out = dict(model=name, K=K, H=H, frame_skip=1, noise="philox seed 2024")
    p64, t64 = run(model, "f64", True, 3)
    o64 = p64.next_observations()[..., :ND]
    R64 = p64.returns()
    p64.close()
    p32, t32d = run(model, "f32", True, 3)
    o32 = p32.next_observations()[..., :ND]
    R32 = p32.returns()
    status = p32.status()
    p32.close()
    p32, t32 = run(model, "f32", False, 10)
    p32.close()
    p64, t64n = run(model, "f64", False, 3)
    p64.close()
    err = {}
    for t in (32, 64, 128):
        rel = (np.abs(o32[:, t - 1] - o64[:, t - 1]) / (1.0 + np.abs(o64[:, t - 1]))).max(axis=1)
        err["t%d" % t] = dict(median=float(np.median(rel)), p99=float(np.percentile(rel, 99)),
                              p999=float(np.percentile(rel, 99.9)), max=float(rel.max()))
    out["fp32_vs_fp64_rel_state_error"] = err
    out["return_rel_error"] = dict(median=float(np.median(np.abs(R32 - R64) / np.abs(R64))),
                                   max=float((np.abs(R32 - R64) / np.abs(R64)).max()))
    out["argmax_equal"] = bool(np.argmax(R32) == np.argmax(R64))
    out["status_f32"] = int(status)
    out["rollout_kernel_ms"] = dict(f32=t32, f32_with_dump=t32d, f64=t64n, f64_with_dump=t64)
    out["env_steps_per_s"] = dict(f32=K * H / (t32 * 1e-3), f64=K * H / (t64n * 1e-3))
    text = json.dumps(out, indent=1)
    print(text)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
