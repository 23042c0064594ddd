"""Edge cases and full-size properties of the HIP path (GPU)."""
import numpy as np
import pytest

from oracle import ref_mppi

pytestmark = pytest.mark.gpu
FILTER = [np.ones(7), 0.25, 0.8, 0.0]


@pytest.fixture(scope="module")
def dm():
    from trajopt_amd.device import DeviceModel
    return DeviceModel.get("reacher_7dof")


def _p(dm, K, H, dtype="f32", **kw):
    from trajopt_amd.device import DevicePlanner
    kw.setdefault("kappa", 25.0)
    kw.setdefault("sigma", FILTER[0])
    kw.setdefault("betas", FILTER[1:])
    return DevicePlanner(dm, K, H, dtype=dtype, **kw)


@pytest.mark.parametrize("K,H", [(1, 1), (1, 2), (3, 5), (65, 3), (127, 16)])
def test_ragged_and_tiny_shapes(dm, oracle_sim, K, H):
    raw = ref_mppi.raw_normals(K + H, K, H, 7)
    eps = ref_mppi.filter_noise(raw.copy(), FILTER)       # H < 3: nothing is filtered
    base = np.full((H, 7), 0.1)
    p = _p(dm, K, H, "f64", dump=True, default_act="repeat")
    p.set_start_state(np.full(7, -0.2), np.zeros(7), np.zeros(3), 0)
    p.set_base_actions(base)
    p.upload_noise(raw)
    p.iterate(1)
    R_o, nobs_o, _ = oracle_sim.rollout(np.full(7, -0.2), np.zeros(7), np.zeros(3), 0, base + eps, dump=True)
    assert np.abs(p.returns() - R_o).max() < 1e-10 and np.abs(p.next_observations() - nobs_o).max() < 1e-10
    assert np.abs(p.act_sequence() - ref_mppi.update(base + eps, R_o, 25.0)).max() < 1e-10
    assert p.argmax() == int(np.argmax(R_o))
    if H > 1:
        seq = p.act_sequence()
        p.shift()
        assert np.array_equal(p.act_sequence(), ref_mppi.shift(seq, "repeat", np.zeros(7)))
    p.close()


def test_bad_arguments_are_refused(dm):
    from trajopt_amd._lib import MppiError
    from trajopt_amd.device import DevicePlanner
    for K, H in ((0, 4), (4, 0), (-1, 4)):
        with pytest.raises(MppiError):
            DevicePlanner(dm, K, H)
    p = _p(dm, 8, 4)
    with pytest.raises(AssertionError):
        p.set_base_actions(np.zeros((5, 7)))
    with pytest.raises(AssertionError):
        p.upload_noise(np.zeros((8, 4, 6)))
    p.upload_actions(np.zeros((8, 4, 7)))
    with pytest.raises(MppiError):       # an update needs noise-relative samples
        p.iterate(1)
    with pytest.raises(MppiError):       # dumps were not requested
        p.next_observations()
    from trajopt_amd import _lib
    assert _lib.lib().mppi_set_params(p._h, 1.0, 1.0, 9) != 0 and b"reward_type" in _lib.lib().mppi_last_error()
    p.close()


def test_full_size_rollouts_against_the_oracle(dm, oracle_sim):
    """The benchmark configuration itself: K = 4096, H = 64, fp32, host-uploaded noise."""
    K, H = 4096, 64
    raw = ref_mppi.all_raw_normals(123, 512, 8, H, 7)           # 8 overlapping worker streams
    eps = ref_mppi.filter_noise(raw.copy(), FILTER)
    p = _p(dm, K, H, default_act="mean")
    p.set_start_state(np.zeros(7), np.zeros(7), np.zeros(3), 0)
    p.set_base_actions(np.zeros((H, 7)))
    p.upload_noise(raw)
    p.iterate(1)
    R = p.returns()
    R_o = oracle_sim.rollout(np.zeros(7), np.zeros(7), np.zeros(3), 0, eps, nthreads=8)
    rel = np.abs(R - R_o) / np.abs(R_o)
    assert rel.max() < 1e-4, rel.max()                            # stated fp32 tolerance
    assert p.argmax() == int(np.argmax(R_o)) == int(np.argmax(R))
    assert p.status() == 0
    # the update against the numpy restatement on the device's own returns
    want = ref_mppi.update(eps.astype(np.float32).astype(np.float64), R, 25.0)
    assert np.abs(p.act_sequence() - want).max() < 1e-6
    # overlapping worker streams: worker i+1's stream is worker i's shifted by 512 seeds, not by 512 draws
    assert np.array_equal(raw[512], ref_mppi.raw_normals(123 + 512, 1, H, 7)[0])
    p.close()


def test_size_independent_properties(dm):
    K, H = 8192, 64
    p = _p(dm, K, H, kappa=0.0)
    p.set_start_state(np.zeros(7), np.zeros(7), np.array([0.1, 0.0, 0.0]), 0)
    base = np.random.RandomState(0).randn(H, 7) * 0.1
    p.set_base_actions(base)
    p.generate_noise(11)
    acts = p.actions()
    p.iterate(1)
    R1, seq1 = p.returns(), p.act_sequence()
    # kappa = 0: every weight is 1, the update is the plain mean of the sampled actions
    assert np.abs(seq1 - acts.sum(0) / (K + 1e-6)).max() < 1e-6
    # bitwise reproducible: same state, same noise, same result
    p.set_base_actions(base)
    p.iterate(1)
    assert np.array_equal(p.returns(), R1) and np.array_equal(p.act_sequence(), seq1)
    # a return depends only on its own sample: shards of the same global stream reproduce their slice
    s = _p(dm, 1024, H, kappa=0.0, k_offset=4096)
    s.set_start_state(np.zeros(7), np.zeros(7), np.array([0.1, 0.0, 0.0]), 0)
    s.set_base_actions(base)
    s.generate_noise(11)
    s.rollout()
    assert np.array_equal(s.returns(), R1[4096:5120])
    # discounting: gamma = 0 keeps only the first reward; all samples share state 0 but not action 0
    p.set_params(0.0, 0.0, "dense")
    p.set_base_actions(base)
    p.rollout()
    R0 = p.returns()
    p.set_params(0.0, 1.0, "sparse")
    p.rollout()
    # sparse reward, gamma = 1: every return is a sum of H rewards out of {100, -10, -100}
    levels = {100.0 * a - 10.0 * b - 100.0 * (H - a - b) for a in range(H + 1) for b in range(H + 1 - a)}
    Rs = p.returns()
    assert set(np.unique(Rs)).issubset(levels) and len(np.unique(Rs)) > 1
    assert np.all(R0 < 0) and R0.std() > 0
    p.close(); s.close()


def test_fp32_tolerance_statement(dm):
    """fp32 rollouts against fp64 rollouts of the same kernels on 8192 samples.  Stated tolerance: at
    H = 32 the relative state error is below 1e-4 for at least 99.8 % of the samples and the return
    error below 1e-5 for all; the few outliers are samples whose joint-limit row switches on one
    substep earlier or later in fp32 (a row exists iff q - lo < 0: a discontinuity no precision removes)."""
    from trajopt_amd.device import DevicePlanner
    K, H = 8192, 32
    res = {}
    for dtype in ("f64", "f32"):
        p = DevicePlanner(dm, K, H, sigma=FILTER[0], betas=FILTER[1:], dtype=dtype, dump=True)
        p.set_start_state(np.zeros(7), np.zeros(7), np.zeros(3), 0)
        p.set_base_actions(np.zeros((H, 7)))
        p.generate_noise(7)
        p.rollout()
        res[dtype] = (p.next_observations()[:, -1, :14], p.returns())
        p.close()
    (s64, R64), (s32, R32) = res["f64"], res["f32"]
    e = (np.abs(s32 - s64) / (1 + np.abs(s64))).max(axis=1)
    assert np.median(e) < 5e-6
    assert (e > 1e-4).mean() < 2e-3, (e > 1e-4).mean()
    assert (np.abs(R32 - R64) / np.abs(R64)).max() < 1e-5
    assert int(np.argmax(R32)) == int(np.argmax(R64))


def test_captured_graphs_follow_parameter_changes(dm):
    """plan / set_params / iterate / plan: each captured graph (one iteration, whole planning step) must be
    rebuilt after a setter changed what it carries, whichever of the two was replayed in between.  The
    comparison handle launches every kernel by itself (profiling switches graph replay off)."""
    K, H = 256, 8

    def run(graphs):
        p = _p(dm, K, H, default_act="mean")
        p.profile(not graphs)
        p.set_start_state(np.full(7, -0.1), np.zeros(7), np.array([0.05, 0.0, 0.1]), 0)
        p.set_base_actions(np.zeros((H, 7)))
        p.plan(7, 2, 1)                       # captures the step graph with kappa = 25, dense reward
        p.set_params(3.0, 0.9, "sparse")
        p.generate_noise(99)
        p.iterate(1)                          # rebuilds the iteration graph only
        mid = p.act_sequence()
        p.plan(7, 2, 1)                       # must not replay the step graph captured with the old parameters
        p.synchronize()
        out = (mid, p.act_sequence(), p.returns(), p.read_log())
        p.close()
        return out
    g, k = run(True), run(False)
    assert np.array_equal(g[0], k[0]) and np.array_equal(g[1], k[1]) and np.array_equal(g[2], k[2])
    assert g[3]["total"] == k[3]["total"] == 2
    assert np.array_equal(g[3]["qpos"], k[3]["qpos"]) and np.array_equal(g[3]["reward"], k[3]["reward"])


def test_status_reports_non_finite_returns(dm):
    """A NaN start state (or an action that overflows fp32) must set MPPI_STATUS_NONFINITE; the library is
    built with -ffinite-math-only, under which a naive test may be folded away."""
    K, H = 128, 4
    for dtype in ("f32", "f64"):
        p = _p(dm, K, H, dtype)
        q = np.zeros(7)
        p.set_start_state(q, np.zeros(7), np.zeros(3), 0)
        p.set_base_actions(np.zeros((H, 7)))
        p.generate_noise(1)
        p.rollout()
        assert p.status() == 0 and np.isfinite(p.returns()).all()
        q[3] = np.nan
        p.set_start_state(q, np.zeros(7), np.zeros(3), 0)
        p.rollout()
        assert p.status() & 1, dtype
        assert not np.isfinite(p.returns()).any()
        p.close()
    # a start velocity that overflows fp32 inside the step
    p = _p(dm, K, H, "f32")
    p.set_start_state(np.zeros(7), np.full(7, 3e38), np.zeros(3), 0)
    p.set_base_actions(np.zeros((H, 7)))
    p.generate_noise(1)
    p.rollout()
    assert p.status() & 1
    p.close()


def test_status_reports_a_capped_active_set_solve(dm, oracle_sim):
    """With the default cap (16 sweeps) no solve of this workload is cut short; with a cap of 2 the substeps
    that need a third sweep are, and MPPI_STATUS_SWEEP_CAP says so."""
    K, H = 2048, 32
    raw = ref_mppi.raw_normals(5, K, H, 7)
    q0 = np.array([0.3, 1.2, 1.6, -0.1, 1.4, -0.05, 1.45])       # close to several limits
    got = {}
    for cap in (0, 2):
        p = _p(dm, K, H, "f32", sweep_cap=cap)
        p.set_start_state(q0, np.zeros(7), np.zeros(3), 0)
        p.set_base_actions(np.zeros((H, 7)))
        p.upload_noise(raw)
        p.rollout()
        got[cap] = (p.status(), p.returns())
        p.close()
    assert got[0][0] == 0
    assert got[2][0] & 2          # (forces of a solve that was cut short may also drive a sample to overflow: bit 0)
    eps = ref_mppi.filter_noise(raw.copy(), FILTER)
    R_o = oracle_sim.rollout(q0, np.zeros(7), np.zeros(3), 0, eps)
    assert np.abs(got[0][1] - R_o).max() / np.abs(R_o).max() < 1e-4


@pytest.mark.gpu
def test_page_locked_staging_uploads_like_any_host_array(dm):
    """The planner's own host buffer (mppi_host_staging) is a faster source for uploads, nothing else: same noise,
    same returns, same memory at every call, and rewriting it after an upload does not disturb the upload."""
    from trajopt_amd.device import DevicePlanner
    from trajopt_amd.utils import all_worker_normals
    K, H = 96, 8
    a = DevicePlanner(dm, K, H, dtype="f64", dump=True)
    b = DevicePlanner(dm, K, H, dtype="f64", dump=True)
    raw = all_worker_normals(3, 48, 2, H, 7)
    buf = a.host_staging()
    assert buf.shape == (K, H, 7) and buf.dtype == np.float64 and a.host_staging() is buf
    got = all_worker_normals(3, 48, 2, H, 7, out=buf)
    assert got is buf and np.array_equal(buf, raw)
    for p, src in ((a, buf), (b, raw)):
        p.set_start_state(np.zeros(7), np.zeros(7), np.array([0.1, 0.0, 0.1]), 0)
        p.set_base_actions(np.zeros((H, 7)))
        p.upload_noise(src)
    buf[:] = 0.0
    for p in (a, b):
        p.rollout()
    assert np.array_equal(a.actions(), b.actions()) and np.array_equal(a.returns(), b.returns())
    assert np.abs(a.actions()).max() > 0
    a.close(); b.close()


@pytest.mark.gpu
def test_log_ring_wraps_and_reads_back_in_order(dm):
    """The real-env log keeps the last 4096 steps; a read that spans the end of the ring comes back oldest first, the
    same rows a read before the wrap returned."""
    from trajopt_amd.device import DevicePlanner
    p = DevicePlanner(dm, 64, 2, kappa=25.0, sigma=np.ones(7), betas=(0.25, 0.8, 0.0), default_act="mean")
    p.set_start_state(np.zeros(7), np.zeros(7), np.array([0.1, 0.0, 0.1]), 0)
    p.set_base_actions(np.zeros((2, 7)))
    p.plan(5, 1, 4090)
    before = p.read_log(max_rows=50)
    assert before["total"] == 4090 and before["qpos"].shape == (50, 7)
    p.plan(5, 1, 16)                                  # steps 4090 .. 4105: the ring's end lies inside the last 50
    after = p.read_log(max_rows=50)
    assert after["total"] == 4106 and after["qpos"].shape == (50, 7)
    for k in ("qpos", "qvel", "hand", "reward", "action"):
        assert np.array_equal(after[k][:34], before[k][16:]), k
    full = p.read_log()
    assert full["qpos"].shape == (4096, 7) and np.array_equal(full["qpos"][-50:], after["qpos"])
    assert p.status() & 1 == 0
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("K", [64, 40000])
def test_read_step_after_every_way_of_advancing(dm, K):
    """`mppi_read_step` hands back the newest log row and the shifted act_sequence.  A single planner's last step may
    have written them to host memory by itself (a lone advance, a single captured step) or not (the steps of a captured
    run; anything that touched act_sequence afterwards): whichever way the step was made, the answer is what the device
    holds.  (K = 40000: the advance launch carries no noise blocks.)"""
    from trajopt_amd.device import DevicePlanner
    H = 6
    p = DevicePlanner(dm, K, H, kappa=25.0, sigma=np.ones(7), betas=(0.25, 0.8, 0.0), default_act="mean")
    p.set_start_state(np.zeros(7), np.zeros(7), np.array([0.1, 0.0, 0.1]), 0)
    p.set_base_actions(np.zeros((H, 7)))
    steps = 0

    def check(what):
        rows, seq = p.read_step()
        log = p.read_log(max_rows=1)
        assert log["total"] == steps, what
        assert np.array_equal(seq, p.act_sequence()), what
        assert np.array_equal(rows[:7], log["qpos"][0]) and np.array_equal(rows[7:14], log["qvel"][0]), what
        assert rows[20] == log["reward"][0] and np.array_equal(rows[21:], log["action"][0]), what
        return rows, seq

    for n in (1, 1, 8, 9, 3, 16):                     # single captured steps, whole runs, runs with a remainder
        p.plan(7, 1, n)
        steps += n
        check("plan %d" % n)
    p.generate_noise(3)
    p.iterate(1)
    p.advance()                                       # a lone advance behind an iteration
    steps += 1
    rows, _ = check("advance")
    p.advance_noise(9)
    steps += 1
    check("advance_noise")
    p.iterate(1)                                      # the update rewrites act_sequence after the mirrored step
    check("iterate after the step")
    new = np.arange(H * 7, dtype=float).reshape(H, 7) / 50.0
    p.plan(7, 1, 1)
    steps += 1
    p.set_base_actions(new)                           # ... and so does the caller
    _, seq = check("set_base_actions after the step")
    assert np.array_equal(seq, new)
    p.plan(7, 1, 1)
    steps += 1
    a = check("plan 1")
    b = check("read twice")
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert p.status() == 0
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_diverged_rollouts_take_no_part_in_the_update(dm, oracle_sim, dtype):
    """From a start pose with the contact sphere deep inside the table plane the soft contact throws the arm so hard
    that the explicit integrator diverges for some samples -- in the fp64 oracle exactly as on the device (same
    samples).  Their returns are NaN and MPPI_STATUS_NONFINITE is set; the update gives them weight zero instead of
    turning the whole action sequence into NaN (the reference's simulator resets itself when an acceleration stops
    being finite, after rewards that have long weighed such a sample out)."""
    K, H = 1024, 32
    q0 = np.array([-2.09561976, 1.43984361, -1.5, 0.01131303, -1.06462017, -0.98820045, -0.73393196])
    rs = np.random.RandomState(3)
    raw = rs.normal(0, 1.0, (K, H, 7))
    base = rs.normal(0, 0.5, (H, 7))
    p = _p(dm, K, H, dtype, kappa=25.0)
    p.set_start_state(q0, np.zeros(7), np.array([0.1, 0.0, 0.2]), 0)
    p.set_base_actions(base)
    p.upload_noise(raw)
    p.iterate(1)
    R = p.returns()
    bad = ~np.isfinite(R)
    assert p.status() & 1 and 0 < bad.sum() < K // 4, bad.sum()
    act = base + ref_mppi.filter_noise(raw.copy(), FILTER)
    R_o = oracle_sim.rollout(q0, np.zeros(7), np.array([0.1, 0.0, 0.2]), 0, act, nthreads=8)
    if dtype == "f64":
        assert np.array_equal(bad, ~np.isfinite(R_o))                    # the oracle diverges on the same samples
        # (samples on their way to divergence carry returns of 1e6 .. 1e9 that no two summation orders share)
        tame = ~bad & (np.abs(R_o) < 1e4)
        assert tame.sum() > K // 2 and np.abs(R[tame] - R_o[tame]).max() < 1e-6 * np.abs(R_o[tame]).max()
    seq = p.act_sequence()
    assert np.isfinite(seq).all()
    # the reference update of the samples that survived (fp32 planners hold the perturbations in fp32)
    want = ref_mppi.update(act[~bad], R[~bad].astype(np.float64), 25.0)
    assert np.abs(seq - want).max() < (1e-9 if dtype == "f64" else 1e-6)
    assert p.argmax() == int(np.flatnonzero(~bad)[np.argmax(R[~bad])])
    # the status word is sticky until cleared: a clean rollout afterwards leaves the bit, clear_status removes it
    p.set_start_state(np.zeros(7), np.zeros(7), np.array([0.1, 0.0, 0.2]), 0)
    p.rollout()
    assert np.isfinite(p.returns()).all() and p.status() & 1
    p.clear_status()
    p.rollout()
    assert p.status() == 0
    p.close()


@pytest.mark.gpu
def test_horizons_beyond_the_device_resident_advance():
    """The device-resident advance serves H * m <= 4096; the class takes the separate-launch route for longer
    horizons (the reference has no limit), the C ABI says -1."""
    from trajopt.algos.mppi import MPPI
    from trajopt.envs.utils import get_environment
    from trajopt_amd._lib import MppiError
    for noise in ("philox", "host"):
        e = get_environment('reacher_7dof')
        e.reset_model(seed=1, goal=(0, 0, 0))
        ag = MPPI(e, H=600, paths_per_cpu=64, kappa=25.0, filter_coefs=FILTER, default_act='mean', seed=1, noise=noise)
        assert not ag._real_env_steps_on_the_device()
        for _ in range(2):
            ag.train_step(niter=1)
        assert len(ag.sol_state) == 3 and ag.act_sequence.shape == (600, 7) and np.isfinite(ag.act_sequence).all()
        assert np.array_equal(ag.act_sequence[-1], np.zeros(7))
        with pytest.raises(MppiError):
            ag._planner.plan(1, 1, 1)
