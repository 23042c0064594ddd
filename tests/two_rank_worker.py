"""One rank of tests/test_gpu_two_ranks.py (started by torch.distributed.run, one process per GPU): a planner of K samples
sharded over the ranks with the exchange inside the library, three whole planning steps; rank 0 also plans the same K
samples on one handle and writes both results."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trajopt_amd import sharding                             # noqa: E402
from trajopt_amd.device import DeviceModel, DevicePlanner     # noqa: E402

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
K, H = 1024, 8


def make(k, off):
    p = DevicePlanner(DeviceModel.get("reacher_7dof"), k, H, kappa=25.0, sigma=np.ones(7), betas=(0.25, 0.8, 0.0),
                      default_act="mean", device=local, k_offset=off, dtype="f64")
    p.set_start_state(np.zeros(7), np.zeros(7), np.array([0.05, 0.0, 0.1]), 0)
    p.set_base_actions(np.zeros((H, 7)))
    return p


p = make(K // world, rank * (K // world))
sharding.init_library_exchange(p, rank, world, device="cuda:%d" % local)
p.plan(7, 2, 3)
p.synchronize()
out = dict(seq=p.act_sequence(), argmax=p.argmax(), rmax=p.rmax(), log=p.read_log(8)["qpos"])
if rank == 0:
    f = make(K, 0)
    f.plan(7, 2, 3)
    f.synchronize()
    out.update(seq1=f.act_sequence(), argmax1=f.argmax(), rmax1=f.rmax(), log1=f.read_log(8)["qpos"])
np.savez(os.path.join(sys.argv[1], "rank%d.npz" % rank), **out)
dist.barrier()
dist.destroy_process_group()
