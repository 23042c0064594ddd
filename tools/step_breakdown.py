#!/usr/bin/env python3
"""Steady-state time of every stage of one planning step, each stage repeated alone.
    python tools/step_breakdown.py [K [planners]]       (default 4096 samples, one planner; H = 64)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trajopt_amd.device import DeviceModel, DevicePlanner
dm = DeviceModel.get("reacher_7dof")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1
p = DevicePlanner(dm, K, 64, kappa=25.0, gamma=1.0, sigma=np.ones(7), betas=(0.25, 0.8, 0.0), mean=np.zeros(7), default_act="mean", dtype="f32", planners=P)
lead = (P,) if P > 1 else ()
p.set_start_state(np.zeros(lead + (7,)), np.zeros(lead + (7,)), np.zeros(lead + (3,)), 0)
p.set_base_actions(np.zeros(lead + (64, 7)))
s = torch.cuda.current_stream().cuda_stream
print("K = %d, planners = %d" % (K, P))
def timeit(name, fn, n=300 if P == 1 else 30):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print("%-28s %.2f us" % (name, (time.perf_counter() - t0) / n * 1e6))
timeit("generate_noise_step", lambda: p.generate_noise_step(2024, s))
timeit("rollout", lambda: p.rollout(s))
timeit("iterate (rollout + update)", lambda: p.iterate(1, s))
timeit("advance", lambda: p.advance(s))
timeit("whole step (3 calls)", lambda: (p.generate_noise_step(2024, s), p.iterate(1, s), p.advance(s)))
timeit("whole step (graph)", lambda: p.plan(2024, 1, 1, s))
