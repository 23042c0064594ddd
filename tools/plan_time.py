import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
from trajopt_amd.device import DeviceModel, DevicePlanner
dm = DeviceModel.get("reacher_7dof")
p = DevicePlanner(dm, 4096, 64, kappa=25.0, gamma=1.0, sigma=np.ones(7), betas=(0.25, 0.8, 0.0), mean=np.zeros(7), default_act="mean", dtype="f32")
p.set_start_state(np.zeros(7), np.zeros(7), np.zeros(3), 0)
p.set_base_actions(np.zeros((64, 7)))
s = torch.cuda.current_stream().cuda_stream
for mode in ("graph", "calls", "graph"):
    for _ in range(20):
        if mode == "graph": p.plan(2024, 1, 1, s)
        else:
            p.generate_noise_step(2024, s); p.iterate(1, s); p.advance(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    if mode == "graph":
        for _ in range(n): p.plan(2024, 1, 1, s)
    else:
        for _ in range(n):
            p.generate_noise_step(2024, s); p.iterate(1, s); p.advance(s)
    torch.cuda.synchronize()
    print(mode, (time.perf_counter() - t0) / n * 1e3, "ms/step")
t0 = time.perf_counter(); p.plan(2024, 1, 300, s); torch.cuda.synchronize(); print("one call, 300 steps", (time.perf_counter() - t0) / 300 * 1e3)
# runs of eight steps per replay (a call that asks for at least eight steps) against one step per replay
for label, chunk in (("300 steps, one call (8 per replay)", 300), ("300 steps, 1 per call", 1), ("300 steps, one call (8 per replay)", 300)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300 // chunk): p.plan(2024, 1, chunk, s)
    torch.cuda.synchronize()
    print(label, (time.perf_counter() - t0) / 300 * 1e3, "ms/step")
