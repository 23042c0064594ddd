"""The N > 1 exchange step on CPU: world_size-2 gloo ranks shard the samples, all-gather their
statistics rows and merge them; the result must equal the single-rank update.  The rank-local
arithmetic is the oracle's (no GPU here); what is under test is the sharding, the row layout,
the collective plumbing and the merge rule."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, K, H, M, kappa, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ref_mppi
    from trajopt_amd import sharding
    rs = np.random.RandomState(0)                       # replicated inputs, as on the GPUs
    eps = rs.randn(K, H, M)
    R = rs.randn(K) * 3 - 50
    R[[5, K - 3]] = R.max() + 1.0                       # tie for the maximum: first index must win
    lo, hi = sharding.shard_range(K, rank, world)
    rmax, amax, num, den = ref_mppi.partials(eps[lo:hi], R[lo:hi], kappa)       # noise-relative, as on the device
    row = np.concatenate([[rmax, lo + amax, den], num.ravel()])
    assert row.size == sharding.partials_size(H, M)
    gathered = sharding.all_gather_partials(torch.from_numpy(row), world).numpy().reshape(world, -1)
    base = rs.randn(H, M)
    seq, _, amax_g = sharding.merge_rows(gathered, base, kappa)      # the product's merge rule
    # rows in the reference's own terms: sum S * (base + eps) = base * sum S + sum S * eps
    parts = [(g[0], int(g[1]), base * g[2] + g[3:].reshape(H, M), g[2]) for g in gathered]
    assert np.allclose(seq, ref_mppi.merge_partials(parts, kappa), rtol=1e-12, atol=1e-13)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), seq=seq, amax=amax_g,
             want=ref_mppi.update(base + eps, R, kappa), want_amax=int(np.argmax(R)))
    dist.destroy_process_group()


def test_two_rank_exchange_equals_single_rank_update(tmp_path):
    world, K, H, M = 2, 64, 6, 7
    mp.spawn(_rank_main, args=(world, _free_port(), K, H, M, 25.0, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    for r in res:
        assert np.allclose(r["seq"], r["want"], rtol=1e-12, atol=1e-13)
        assert int(r["amax"]) == int(r["want_amax"]) == 5
    assert np.array_equal(res[0]["seq"], res[1]["seq"])          # every rank applies the identical update


def test_shard_range():
    import pytest
    from trajopt_amd import sharding
    assert [sharding.shard_range(64, r, 4) for r in range(4)] == [(0, 16), (16, 32), (32, 48), (48, 64)]
    with pytest.raises(ValueError):
        sharding.shard_range(10, 0, 4)
