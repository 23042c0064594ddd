"""mppi_update_from_returns_noalloc (device pointers in and out, fixed workspace, no allocation) against
mppi_update_from_returns on the same inputs: two calls in a row, the second with a smaller input, so that the
first call's left-over weights in the shared workspace would show if they were read."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(actions, R, kappa):
    import torch
    from trajopt_amd import _lib as L
    from trajopt_amd.device import update_from_returns
    K, H, M = actions.shape
    want = update_from_returns(actions, R, kappa, device=0)
    dev = torch.device("cuda", 0)
    d_act = torch.from_numpy(np.ascontiguousarray(actions)).to(dev)
    d_R = torch.from_numpy(np.ascontiguousarray(R)).to(dev)
    d_out = torch.full((H * M,), float("nan"), dtype=torch.float64, device=dev)
    torch.cuda.synchronize(dev)
    stream = torch.cuda.current_stream(dev)
    L.check(L.lib().mppi_update_from_returns_noalloc(0, d_act.data_ptr(), d_R.data_ptr(), K, H * M, float(kappa),
                                                     d_out.data_ptr(), stream.cuda_stream))
    stream.synchronize()
    return want, d_out.cpu().numpy().reshape(H, M)


def test_noalloc_update_matches_the_allocating_entry_point():
    from oracle import ref_mppi
    rng = np.random.RandomState(5)
    for K, H, M, kappa in [(4096, 64, 7, 25.0), (300, 16, 7, 3.0), (1, 2, 7, 1.0)]:
        actions = rng.normal(size=(K, H, M))
        R = -50.0 + 10.0 * rng.normal(size=K)
        want, got = _both(actions, R, kappa)
        assert np.array_equal(got, want), np.abs(got - want).max()      # same kernels' arithmetic, same order
        assert np.abs(got - ref_mppi.update(actions, R, kappa)).max() < 1e-12


def test_noalloc_update_rejects_oversized_input():
    import torch
    from trajopt_amd import _lib as L
    d = torch.zeros(8, dtype=torch.float64, device="cuda:0")
    rc = L.lib().mppi_update_from_returns_noalloc(0, d.data_ptr(), d.data_ptr(), L.NOALLOC_MAX_K + 1, 1, 1.0,
                                                  d.data_ptr(), None)
    assert rc == -6 and b"exceeds" in L.lib().mppi_last_error()
