"""The sharded planner on two GPUs: RCCL all-gather of the shard statistics captured in the step graph, rank-ordered
merge.  Skipped on a box with fewer than two GPUs (RCCL refuses two ranks on one device)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_two_ranks_reproduce_the_single_planner(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                    "127.0.0.1", "--master-port", "29541", os.path.join(here, "two_rank_worker.py"), str(tmp_path)],
                   check=True, env=env, timeout=110)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # every rank ends with the identical act_sequence, arg-max and real-env trajectory ...
    for k in ("seq", "argmax", "rmax", "log"):
        assert np.array_equal(r0[k], r1[k]), k
    # ... and it is the single planner's: same samples (noise addressed by global index), the update summed in another
    # order (two shards, merged), hence to rounding; the arg-max is an index and must be exact
    assert int(r0["argmax"]) == int(r0["argmax1"])
    assert abs(float(r0["rmax"]) - float(r0["rmax1"])) < 1e-9
    assert np.abs(r0["seq"] - r0["seq1"]).max() < 1e-9 and np.abs(r0["log"] - r0["log1"]).max() < 1e-9
