"""Single planning steps as a C-ABI caller makes them: mppi_plan(nsteps = 1) followed by mppi_read_step, timed on the host
(K = 4096, H = 64).  MPPI_GRAPH=0 times the same calls without the captured step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt_amd.device import DeviceModel, DevicePlanner
K, H = 4096, 64
p = DevicePlanner(DeviceModel.get("reacher_7dof"), K, H, kappa=25.0, gamma=1.0, sigma=np.ones(7), betas=(0.25, 0.8, 0.0),
                  mean=np.zeros(7), default_act="mean", dtype="f32")
p.set_start_state(np.zeros(7), np.zeros(7), np.zeros(3), 0)
p.set_base_actions(np.zeros((H, 7)))
for _ in range(20):
    p.plan(3, 1, 1); p.read_step()
n = 300
a = b = 0.0
for _ in range(n):
    t0 = time.perf_counter(); p.plan(3, 1, 1); t1 = time.perf_counter(); p.read_step(); t2 = time.perf_counter()
    a += t1 - t0; b += t2 - t0
print("graph=%s plan call %.1f us, plan+read_step %.1f us" % (os.environ.get("MPPI_GRAPH", "1"), a / n * 1e6, b / n * 1e6))
