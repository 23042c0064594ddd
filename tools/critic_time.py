#!/usr/bin/env python3
"""Time of the terminal-value critic kernel at K = 4096 (and 65536): rollouts with the critic set, kernel by kernel,
for `rocprofv3 --kernel-trace --stats` to pick up the critic_kernel row; prints the step time with and without it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajopt_amd.device import DeviceModel, DevicePlanner
rs = np.random.RandomState(0)
w = [rs.randn(128, 17).astype(np.float32) * 0.2, rs.randn(128).astype(np.float32) * 0.1, rs.randn(128, 128).astype(np.float32) * 0.1,
     rs.randn(128).astype(np.float32) * 0.1, rs.randn(1, 128).astype(np.float32), rs.randn(1).astype(np.float32)]
for K in (4096, 65536):
    p = DevicePlanner(DeviceModel.get("reacher_7dof"), K, 8, kappa=25.0, sigma=np.ones(7), betas=(0.25, 0.8, 0.0), default_act="mean")
    p.set_start_state(np.zeros(7), np.zeros(7), np.zeros(3), 0)
    p.set_base_actions(np.zeros((8, 7)))
    p.generate_noise(1)
    out = {}
    for on in (False, True):
        p.set_critic(17 if on else 0, w if on else None)
        for _ in range(10): p.rollout()
        p.synchronize(); t0 = time.perf_counter()
        for _ in range(100): p.rollout()
        p.synchronize(); out[on] = (time.perf_counter() - t0) / 100 * 1e6
    print("K=%d H=8: rollout %.1f us, with critic %.1f us, critic %.1f us" % (K, out[False], out[True], out[True] - out[False]), flush=True)
    p.close()
