"""Batched independent planners (1024 planners x K=512, H=64 is the reference workload of the
'many planners' configuration): planning steps per second of one handle holding P planners."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from trajopt_amd.device import DeviceModel, DevicePlanner
dm = DeviceModel.get("reacher_7dof")
K, H, steps = 512, 64, 30
FLOP = 8870.0
for P in [int(x) for x in (sys.argv[1:] or ["128", "1024"])]:
    p = DevicePlanner(dm, K, H, kappa=25.0, sigma=np.ones(7), betas=(0.25, 0.8, 0.0), default_act="mean", planners=P)
    rs = np.random.RandomState(0)
    tgt = np.stack([np.random.RandomState(b).uniform(-0.1, 0.1, 3) for b in range(P)])
    p.set_start_state(np.zeros((P, 7)), np.zeros((P, 7)), tgt, 0)
    p.set_base_actions(np.zeros((P, H, 7)))
    s = torch.cuda.current_stream().cuda_stream
    def step():
        p.generate_noise_step(1000, s); p.iterate(1, s); p.advance(s)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    p.profile(True); p.profile_read()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, n = p.profile_read()
    # the same steps as one call (captured graphs, eight steps per replay; no events around the rollouts)
    p.profile(False)
    p.plan(1000, 1, 8, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p.plan(1000, 1, 4 * steps, s)
    torch.cuda.synchronize()
    dt_plan = (time.perf_counter() - t0) / (4 * steps)
    d = [float(np.linalg.norm(p.read_log(planner=b)["hand"][-1] - tgt[b])) for b in (0, P // 2, P - 1)]
    print(json.dumps({"planners": P, "K": K, "H": H, "ms_per_step": 1e3 * dt / steps, "ms_per_step_one_call": 1e3 * dt_plan, "rollout_ms": ms / n,
                      "planner_steps_per_s": P * steps / dt, "env_steps_per_s": P * K * H * steps / dt,
                      "rollout_frac_fp32_valu": FLOP * P * K * H / (ms / n) / 1e9 / 157.3,
                      "hand_target_dist_after_%d_steps" % (5 * steps + 11): d}), flush=True)
    p.close()
