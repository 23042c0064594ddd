"""Where a host-noise `train_step` spends its time (K = 4096, H = 64): the pieces one by one, then the whole step."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt_amd.device import DeviceModel, DevicePlanner
from trajopt_amd.utils import _host_threads, worker_seeds
K, H = 4096, 64
p = DevicePlanner(DeviceModel.get("reacher_7dof"), K, H, kappa=25.0, sigma=np.ones(7), betas=(0.25, 0.8, 0.0), default_act="mean")
p.set_start_state(np.zeros(7), np.zeros(7), np.zeros(3), 0)
p.set_base_actions(np.zeros((H, 7)))
def timed(f, n=30):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return 1e3 * (time.perf_counter() - t0) / n
out = {"threads": _host_threads()}
for th in (8, 12, 15, 16):
    out["draw_noise_ms_%d_threads" % th] = timed(lambda: (p.draw_noise(worker_seeds(5, K, 1), th, True), p.synchronize()))
def rest():
    p.commit_noise(); p.iterate(1); p.advance(); p.read_step()
p.draw_noise(worker_seeds(5, K, 1), 16, True)
def rest_once():
    p.draw_noise(worker_seeds(5, K, 1), 16, True)
    t0 = time.perf_counter(); rest(); return time.perf_counter() - t0
rest_once(); rest_once()
out["commit_iterate_advance_read_ms"] = 1e3 * np.median([rest_once() for _ in range(20)])
def overlapped():
    p.commit_noise(); p.iterate(1); p.advance(); p.draw_noise(worker_seeds(6, K, 1), 16, True); p.read_step()
p.draw_noise(worker_seeds(6, K, 1), 16, True)
out["whole_step_ms"] = timed(overlapped)
print(json.dumps(out))
