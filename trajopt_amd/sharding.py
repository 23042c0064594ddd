"""Sample sharding across ranks (one process per GPU) and the single exchange step of an
iteration.  Replaces the worker-order concatenation of `gather_paths_parallel`
(trajopt/utils.py:158-161): instead of whole trajectories, each rank contributes
3 + H*M doubles -- (Rmax, first arg-max as a global index, sum S, sum S*eps[H*M]) with S
relative to the rank's own maximum -- and every rank merges the gathered rows in rank order.
"""
import torch
import torch.distributed as dist


def shard_range(K_global, rank, world):
    """Contiguous shard [lo, hi) of rank; K_global must divide evenly so that every rank
    launches the same grid.  Shards that are multiples of 64 samples keep every wavefront's samples together and
    reproduce the unsharded returns bit for bit; other sizes regroup samples into other wavefronts, whose active-set
    solves run as many sweeps as their slowest sample needs, and reproduce them to rounding."""
    if K_global % world:
        raise ValueError("K=%d is not divisible by %d ranks" % (K_global, world))
    per = K_global // world
    return rank * per, (rank + 1) * per


def partials_size(H, M):
    return 3 + H * M


def all_gather_partials(part, world, out=None):
    """All-gather the per-rank statistics rows (rank order).  `part` is a 1-D float64 tensor
    on the device the process group works on (CUDA for RCCL, CPU for gloo)."""
    if out is None:
        out = torch.empty(world * part.numel(), dtype=part.dtype, device=part.device)
    if world == 1:
        out.copy_(part)
    else:
        dist.all_gather_into_tensor(out, part)
    return out


def merge_rows(gathered, base, kappa):
    """The merge every rank applies to the gathered statistics rows, on the host (numpy): the arithmetic of the
    device's merge kernel, in rank order.  gathered: (world, 3 + H*M) rows [Rmax, global arg-max, sum S,
    sum S*eps]; base: (H, M) act_sequence the samples were drawn around.  Returns (act_sequence, Rmax, arg-max);
    ties for the maximum go to the lowest rank, whose arg-max is already its first one (`np.argmax` order)."""
    import numpy as np
    g = np.asarray(gathered, dtype=np.float64)
    base = np.asarray(base, dtype=np.float64)
    gmax, gidx = -np.inf, 0
    for row in g:
        if row[0] > gmax:
            gmax, gidx = row[0], int(row[1])
    den = 0.0
    num = np.zeros(base.size)
    for row in g:
        w = np.exp(kappa * (row[0] - gmax))
        den += w * row[2]
        num += w * row[3:]
    seq = (base.ravel() * den + num) / (den + 1e-6)
    return seq.reshape(base.shape), gmax, gidx


def init_library_exchange(planner, rank, world, device=None):
    """Set up the in-library exchange of a sharded DevicePlanner: rank 0 makes the 128-byte group id, the
    process group (any backend) broadcasts it, every rank joins.  With world == 1 no process group is needed."""
    uid = planner.comm_unique_id() if rank == 0 else bytes(128)
    if world > 1:
        on_gpu = dist.get_backend() == "nccl"
        t = torch.tensor(list(uid), dtype=torch.uint8, device=(device if on_gpu else "cpu"))
        dist.broadcast(t, src=0)
        uid = bytes(t.cpu().tolist())
    planner.comm_init(uid, rank, world)
