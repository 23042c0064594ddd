"""Rehearsal: two ranks sharing GPU 0 try the in-library exchange (RCCL normally refuses two ranks on one device;
if it accepts, the sharded plan() of two shards must equal the single planner).  Run under torch.distributed.run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from trajopt_amd import sharding
from trajopt_amd.device import DeviceModel, DevicePlanner
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
K, H = 512, 8
def mk(k, off):
    p = DevicePlanner(DeviceModel.get("reacher_7dof"), k, H, kappa=25.0, sigma=np.ones(7), betas=(0.25, 0.8, 0.0), default_act="mean", device=0, k_offset=off)
    p.set_start_state(np.zeros(7), np.zeros(7), np.array([0.05, 0, 0.1]), 0); p.set_base_actions(np.zeros((H, 7))); return p
p = mk(K // world, rank * (K // world))
try:
    sharding.init_library_exchange(p, rank, world, device="cuda:0")
except Exception as e:
    print("rank %d: library exchange refused: %s" % (rank, e), flush=True); sys.exit(0)
p.plan(7, 2, 3); p.synchronize()
seq = p.act_sequence()
if rank == 0:
    f = mk(K, 0); f.plan(7, 2, 3); f.synchronize()
    print("two ranks on one GPU: max |d act_sequence| vs single planner = %.3e" % np.abs(seq - f.act_sequence()).max(), flush=True)
dist.barrier(); dist.destroy_process_group()
