#!/usr/bin/env python3
"""Finds a constraint solve that runs into the sweep cap on the GPU (1024 planners x 512 samples from random poses,
sparse reward so that the general kernel variant runs and reports where), then replays the 64 samples of that wave
on the host in fp32 and fp64 (tests/_host_core.so) and prints their sweep counts at that step and whether the
returns still agree with the oracle."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trajopt_amd.device import DeviceModel, DevicePlanner
from oracle.sim import OracleSim
from trajopt_amd.model import get_model
P, K, H = 1024, 512, 64
rs = np.random.RandomState(0)
tgt = np.stack([np.random.RandomState(p).uniform(-0.1, 0.1, 3) for p in range(P)])
q0 = rs.uniform(-0.3, 0.3, (P, 7)); q0[:, 3] -= 0.8; q0[:, 5] -= 0.5
p = DevicePlanner(DeviceModel.get("reacher_7dof"), K, H, kappa=25.0, sigma=np.ones(7), betas=(0.25, 0.8, 0.0), default_act="mean",
                  reward_type="sparse", planners=P)
p.set_start_state(q0, np.zeros((P, 7)), tgt, 0)
p.set_base_actions(np.zeros((P, H, 7)))
p.generate_noise_step(1000)
p.rollout()
bits, k1, step, pl = p.status_detail()
print("status bits %d; first report: planner %d, wave starting at sample %d, env step %d" % (bits, pl, k1 - 1, step))
if not bits & 2:
    sys.exit(0)
act = p.actions()[pl, k1 - 1:k1 + 63]
R_gpu = p.returns()[pl, k1 - 1:k1 + 63]
sim = OracleSim(get_model("reacher_7dof"))
R_o, nobs, _ = sim.rollout(q0[pl], np.zeros(7), tgt[pl], 0, act, reward_type="sparse", dump=True)
print("returns of the wave: max |gpu - oracle| = %.3g (sparse reward: steps of 90 / 110 mean a flipped level)" % np.abs(R_gpu - R_o).max())
L = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "_host_core.so"))
pp = lambda a: a.ctypes.data_as(C.c_void_p)
for f64 in (0, 1):
    worst = []
    for k in range(64):
        # state just before the reported step, from the oracle's trajectory; one env step from there on the host build
        s = nobs[k, step - 1] if step > 0 else np.r_[q0[pl], np.zeros(7)]
        qq, vv = np.ascontiguousarray(s[:7]), np.ascontiguousarray(s[7:14])
        q, v, hand, rew = np.zeros(7), np.zeros(7), np.zeros(3), np.zeros(1)
        R, sw = C.c_double(), C.c_int()
        a = np.ascontiguousarray(act[k, step][None])
        L.core_rollout(f64, 0, pp(qq), pp(vv), pp(np.ascontiguousarray(tgt[pl])), pp(a), 1, 1, None, 0, C.c_double(1.0), C.c_double(1.0),
                       pp(q), pp(v), pp(hand), pp(rew), C.byref(R), C.byref(sw))
        worst.append(sw.value)
    print("host %s: sweeps of the 64 samples at that step (cold start of the active set): max %d, histogram %s" % (
        "fp64" if f64 else "fp32", max(worst), np.bincount(worst).tolist()))
# the whole rollout up to and including that step on the host, active set carried as on the device
for f64 in (0, 1):
    worst = []
    for k in range(64):
        q, v, hand, rew = np.zeros(7), np.zeros(7), np.zeros(3), np.zeros(step + 1)
        R, sw = C.c_double(), C.c_int()
        a = np.ascontiguousarray(act[k, :step + 1])
        L.core_rollout(f64, 0, pp(np.ascontiguousarray(q0[pl])), pp(np.zeros(7)), pp(np.ascontiguousarray(tgt[pl])), pp(a), step + 1, 1, None, 0,
                       C.c_double(1.0), C.c_double(1.0), pp(q), pp(v), pp(hand), pp(rew), C.byref(R), C.byref(sw))
        worst.append(sw.value)
    print("host %s: most sweeps of any substep up to that step, carried set: max %d, histogram %s" % (
        "fp64" if f64 else "fp32", max(worst), np.bincount(worst).tolist()))
# the dense returns of the same wave: does the capped solve leave a trace?
p.set_params(25.0, 1.0, "dense")
p.rollout()
R_d = p.returns()[pl, k1 - 1:k1 + 63]
R_od = sim.rollout(q0[pl], np.zeros(7), tgt[pl], 0, act)
print("dense returns of the wave: max rel |gpu - oracle| = %.3g" % (np.abs(R_d - R_od).max() / np.abs(R_od).max()))
