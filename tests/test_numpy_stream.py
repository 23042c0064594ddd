"""The library's bulk generator of numpy's legacy normal stream (mppi_numpy_normals) against numpy itself: same bits
for every seed, length and thread count, and the same generator state afterwards.  Host code only: no GPU needed."""
import numpy as np
import pytest


def _ours(seed, n, threads, want_state=True):
    from trajopt_amd import _lib as L
    out = np.empty(n)
    state = np.empty(628, dtype=np.uint32) if want_state else None
    assert L.lib().mppi_numpy_normals(seed, n, L.ptr(out), threads, L.ptr(state)) == 0
    return out, state


@pytest.mark.parametrize("seed,n,threads", [(0, 1, 1), (1, 2, 1), (123, 7, 2), (12345, 3584, 1), (2, 4481, 3),
                                            (2 ** 32 - 1, 1001, 4), (77, 100000, 8), (5, 250001, 16), (9, 0, 1),
                                            (11, 1000001, 8), (3, 600000, 1), (4, 700000, 2)])   # the last three wrap the generator's ring
def test_stream_and_state_equal_numpy(seed, n, threads):
    got, state = _ours(seed, n, threads)
    np.random.seed(seed)
    want = np.random.normal(loc=0, scale=1.0, size=n)
    assert np.array_equal(got, want)
    name, key, pos, has_gauss, cached = np.random.get_state()
    assert np.array_equal(state[:624], key) and int(state[624]) == pos and int(state[625]) == has_gauss
    assert state[626:628].view(np.float64)[0] == cached
    # and a stream continued from that state goes on like numpy's
    np.random.set_state(("MT19937", state[:624], int(state[624]), int(state[625]), float(state[626:628].view(np.float64)[0])))
    a = np.random.normal(size=5)
    np.random.seed(seed)
    np.random.normal(size=n)
    assert np.array_equal(a, np.random.normal(size=5))


def test_blocks_of_odd_size_carry_the_cached_variate():
    """`generate_paths` draws (H, m) blocks one after the other; with H*m odd the second variate of a pair is handed
    to the next block (trajopt/utils.py:118-139).  One bulk draw of all blocks is the same stream."""
    from trajopt_amd.utils import all_worker_normals, worker_normals
    np.random.seed(31)
    want = np.stack([np.random.normal(loc=0, scale=1.0, size=(5, 3)) for _ in range(9)])
    after = np.random.normal(size=4)
    got = worker_normals(31, 9, 5, 3)
    assert np.array_equal(got, want) and np.array_equal(np.random.normal(size=4), after)
    # several workers: overlapping seeds, the caller's generator untouched
    np.random.seed(8)
    mark = np.random.get_state()[1].copy()
    blocks = all_worker_normals(31, 3, 4, 5, 3)
    assert np.array_equal(np.random.get_state()[1], mark)
    for i in range(4):
        np.random.seed(31 + 3 * i)
        assert np.array_equal(blocks[3 * i:3 * i + 3], np.stack([np.random.normal(size=(5, 3)) for _ in range(3)]))
    with pytest.raises(ValueError):
        worker_normals(-1, 1, 2, 2)


def test_random_seeds_and_lengths():
    rs = np.random.RandomState(0)
    for _ in range(40):
        seed, n, th = int(rs.randint(0, 2 ** 31)), int(rs.randint(1, 30000)), int(rs.randint(1, 9))
        got, state = _ours(seed, n, th)
        ref = np.random.RandomState(seed)
        assert np.array_equal(got, ref.normal(size=n)), (seed, n, th)
        _, key, pos, has_gauss, cached = ref.get_state()
        assert np.array_equal(state[:624], key) and (int(state[624]), int(state[625])) == (pos, has_gauss), (seed, n, th)


@pytest.mark.parametrize("streams,n_each,threads", [(8, 40001, 4), (3, 300000, 8), (20, 1000, 3), (1, 5000, 2), (0, 10, 2)])
def test_several_worker_streams_side_by_side(streams, n_each, threads):
    from trajopt_amd import _lib as L
    seeds = np.array([7 + 13 * i for i in range(streams)], dtype=np.uint32)
    out = np.full(streams * n_each, np.nan)
    assert L.lib().mppi_numpy_normals_many(L.ptr(seeds), streams, n_each, L.ptr(out), threads) == 0
    for i in range(streams):
        assert np.array_equal(out[i * n_each:(i + 1) * n_each], np.random.RandomState(int(seeds[i])).normal(size=n_each)), i


@pytest.mark.parametrize("seed,n,threads", [(21, 100001, 8), (8, 64, 1), (13, 300000, 2)])
def test_a_round_of_attempts_that_falls_short_is_made_again_exactly(seed, n, threads):
    """The rejection sampler makes its attempts in bulk with a surplus; if a round still ends short of the values
    asked for, the stream is made again from the seed with a larger surplus.  The test hook removes the surplus of
    the first round, which forces that path: stream and generator state must be numpy's all the same."""
    from trajopt_amd import _lib as L
    L.lib().mppi_numpy_normals_test_margin(-1.0)
    try:
        got, state = _ours(seed, n, threads)
    finally:
        L.lib().mppi_numpy_normals_test_margin(1.0)
    np.random.seed(seed)
    want = np.random.normal(loc=0, scale=1.0, size=n)
    assert np.array_equal(got, want)
    name, key, pos, has_gauss, cached = np.random.get_state()
    assert np.array_equal(state[:624], key) and int(state[624]) == pos and int(state[625]) == has_gauss
    assert state[626:628].view(np.float64)[0] == cached
