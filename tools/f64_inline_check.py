"""fp64 / fp32 rollouts of one study build (MPPI_HIP_LIBRARY) against the CPU oracle: prints the largest return
and state errors.  Used with tools/f64_inline_build.sh.  argv[1]: model name (default tree_18dof_quat)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle.sim import OracleSim                 # noqa: E402
from trajopt_amd.device import DeviceModel, DevicePlanner   # noqa: E402
from trajopt_amd.model import get_model          # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "tree_18dof_quat"
md = get_model(name)
K, H = 128, 16
rs = np.random.RandomState(3)
if md.nq > md.nv:
    quat = rs.randn(4)
    q0 = np.r_[0.0, 0.0, 0.0, quat / np.linalg.norm(quat), rs.randn(md.nv - 6) * 0.3]
else:
    q0 = rs.randn(md.nq) * 0.3
v0, tgt = rs.randn(md.nv) * 0.5, np.array([0.4, 0.1, 0.8])
sim = OracleSim(md)
for dtype in (("f64",) if os.environ.get("MPPI_CHECK_F64_ONLY") else ("f64", "f32")):
    p = DevicePlanner(DeviceModel.get(name), K, H, dtype=dtype, kappa=5.0, gamma=0.99, sigma=np.full(md.nv, 0.5),
                      betas=(0.25, 0.8, 0.0), dump=True)
    p.set_start_state(q0, v0, tgt, 0)
    p.set_base_actions(np.zeros((H, md.nv)))
    p.generate_noise(5)
    p.rollout()
    act, R, nobs = p.actions(), p.returns(), p.next_observations()
    R_o, nobs_o, _ = sim.rollout(q0, v0, tgt, 0, act, gamma=0.99, dump=True, nthreads=4)
    print("%s %s %s: max |dR| %.3e   max |dstate| %.3e (first step %.3e)" % (
        os.path.basename(os.environ.get("MPPI_HIP_LIBRARY", "shipped")), name, dtype, np.abs(R - R_o).max(),
        np.abs(nobs - nobs_o).max(), np.abs(nobs[:, 0] - nobs_o[:, 0]).max()), flush=True)
    p.close()
