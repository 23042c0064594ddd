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
    launches the same grid."""
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
