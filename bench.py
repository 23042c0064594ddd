#!/usr/bin/env python3
"""Headline benchmark: MPPI planning steps on reacher_7dof, H = 64: K = 4096 samples on one GPU; with N > 1 ranks
one planner of K = 65536 samples sharded over the N GPUs (strong scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one planning step of `MPPI.train_step(niter=1)` with everything resident on the
device and enqueued on one stream without host synchronisation: Philox noise + filter,
K x H rollouts with fused reward / discounted return, max / weights / weighted action sum
(N > 1: this shard's statistics, one RCCL all-gather of 3 + H*M doubles per rank made by the library,
rank-ordered merge), the single fp64 step of the real environment from the planner's own state, and
the horizon shift.  The timed steps are one `mppi_plan` call, which replays a captured hipGraph of eight
consecutive steps (the kernels read the time step from the device-resident state).  After the
timed region the same steps are launched piece by piece with HIP events around every rollout
kernel (roofline.kernel_ms), and -- one GPU only -- the other flavours of the step are timed and
reported as extra keys: the step with numpy noise uploaded from the host (`host_noise`) and the
drop-in `MPPI.train_step` through the Python API (`train_step_ms`).
Sample noise is addressed by global index, so the planner's result does not depend on N.  With N > 1
rank 0 first times the same K = 65536 planner on its GPU alone; `strong_scaling.efficiency` is
T(1) / (N * T(N)).

Prints one JSON line (rank 0).  `value` = env steps per second summed over all ranks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K_PER_GPU, H, M = 4096, 64, 7
K_STRONG = 65536                   # the sharded planner of the multi-GPU runs
K_STRONG_LARGE = 524288            # a second, larger sharded planner, reported beside it (N > 1 only)
KAPPA, GAMMA = 25.0, 1.0
FILTER = [np.ones(M), 0.25, 0.8, 0.0]
SEED = 123
PEAK_FP32_VALU_TFLOPS = 157.3      # MI355X vector fp32 peak
PEAK_HBM_GBS = 8000.0


def cpu_baseline(cores):
    """The reference's rollout path restated on the CPU oracle (same Python loop and process
    pool structure), timed on this host for a bounded sample of the same workload."""
    from oracle import ref_rollout
    from oracle.sim import OracleSim
    from trajopt_amd.model import get_model
    md = get_model("reacher_7dof")
    start = dict(qp=np.zeros(M), qv=np.zeros(M), target_pos=np.zeros(3), timestep=0)
    ppc = K_PER_GPU // cores
    iters = 3
    times = []
    for i in range(iters):
        t0 = time.perf_counter()
        ref_rollout.planning_iteration(md, start, np.zeros((H, M)), FILTER, SEED + i, KAPPA, GAMMA, ppc, cores)
        times.append(time.perf_counter() - t0)
    py = ppc * cores * H / sorted(times)[iters // 2]          # median of the repeats
    # the same arithmetic without the Python loop: multithreaded C++ rollouts
    from oracle import ref_mppi
    act = ref_mppi.filter_noise(ref_mppi.raw_normals(SEED, 2048, H, M), FILTER)
    sim = OracleSim(md)
    t0 = time.perf_counter()
    sim.rollout(np.zeros(M), np.zeros(M), np.zeros(3), 0, act, nthreads=cores)
    c_rate = 2048 * H / (time.perf_counter() - t0)
    return {"value": py, "unit": "env_steps/s", "cores": cores, "kind": "port",
            "sample": "median of %d planning iterations of gather_paths_parallel + update at K=%d (=%d workers x %d paths), H=%d, "
                      "Python-structured restatement of the reference path on the fp64 C++ simulator restatement "
                      "(MuJoCo is not installed)" % (iters, ppc * cores, cores, ppc, H),
            "c_rollouts_only_env_steps_per_s": c_rate}


def flops_per_env_step():
    """Algorithmic operation count of one env step (frame_skip substeps), from the
    operation-counting build of the CPU restatement, averaged over states of this workload."""
    from oracle import ref_mppi
    from oracle.sim import OracleSim
    from trajopt_amd.model import get_model
    md = get_model("reacher_7dof")
    sim = OracleSim(md)
    act = ref_mppi.filter_noise(ref_mppi.raw_normals(SEED, 6, H, M), FILTER)
    total, n = 0.0, 0
    for k in range(6):
        q, v = np.zeros(M), np.zeros(M)
        for t in range(H):
            for _ in range(md.frame_skip):
                c = sim.count_ops(q, v, act[k, t])
                total += c["add"] + c["mul"] + c["div"] + c["sqrt"] + c["trig"]
                q, v, _, _ = sim.substep(q, v, act[k, t])
            n += 1
    return total / n


def _profile(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return None


def measured_traffic():
    """HBM bytes per rollout launch from the committed rocprofv3 PMC summary (FETCH_SIZE / WRITE_SIZE
    passes, gfx950 corrections applied); None if the summary is not for this workload."""
    d = _profile("r3_rollout_traffic.json")
    try:
        if d and d.get("K") == K_PER_GPU and d.get("H") == H:
            return float(d["hbm_bytes_per_launch"])
    except (ValueError, KeyError):
        pass
    return None


def issue_statistics(flops_algorithmic):
    """What limits the rollout kernel at this size, from the committed counter summaries of the same kernel and workload
    shape (rocprofv3 --pmc over tools/ksweep.py 4096; profiles/r3_rollout_pmc_sq.json, r3_rollout_flops.json):
    how busy the instruction issue of a wave is, what share of the GPU's vector issue slots the launch uses, and the
    floating-point operations the kernel executes (it folds the model's structural zeros, which the algorithmic count
    of the generic algorithm includes)."""
    out = {}
    sq = _profile("r3_rollout_pmc_sq.json")
    if sq and "per_wave_substep" in sq:
        w = sq["per_wave_substep"]
        waves, simds = (K_PER_GPU + 63) // 64, 1024
        out["valu_per_wave_substep"] = w.get("SQ_INSTS_VALU")
        out["wave_issue_busy_frac"] = w.get("SQ_ACTIVE_INST_ANY", 0.0) / w["SQ_WAVE_CYCLES"]
        # vector instructions x 4 cycles x waves / (SIMDs x kernel cycles); SQ cycle counters are in quad-cycles
        out["issue_slot_frac"] = w["SQ_INSTS_VALU"] * waves / (simds * w["SQ_WAVE_CYCLES"])
        out["waves"], out["simds"] = waves, simds
    fl = _profile("r3_rollout_flops.json")
    if fl and "executed_flop_per_env_step" in fl:
        out["executed_flop_per_env_step"] = fl["executed_flop_per_env_step"]
        out["executed_over_algorithmic"] = fl["executed_flop_per_env_step"] / flops_algorithmic
    return out


def make_planner(K, k_offset, device):
    from trajopt_amd.device import DeviceModel, DevicePlanner
    p = DevicePlanner(DeviceModel.get("reacher_7dof"), K, H, kappa=KAPPA, gamma=GAMMA, sigma=FILTER[0], betas=FILTER[1:],
                      mean=np.zeros(M), default_act="mean", dtype="f32", device=device, k_offset=k_offset)
    p.set_start_state(np.zeros(M), np.zeros(M), np.zeros(3), 0)
    p.set_base_actions(np.zeros((H, M)))
    return p


def other_step_flavours(sptr):
    """The planning step as the reference's configurations word it, timed on one GPU after the headline loop:
    numpy noise uploaded from the host (exact-comparison mode) through the C ABI, and the drop-in
    `MPPI.train_step` of the Python API with device noise and with numpy noise."""
    from trajopt.algos.mppi import MPPI
    from trajopt.envs.utils import get_environment
    from trajopt_amd.utils import _host_threads, worker_seeds
    n = 40
    out = {}
    p = make_planner(K_PER_GPU, 0, int(os.environ.get("LOCAL_RANK", "0")))
    # (a) the draws are there (page-locked staging buffer): upload + filter, rollouts, update, real-env step
    p.draw_noise(worker_seeds(SEED, K_PER_GPU, 1), _host_threads())
    p.commit_noise()
    stage = p.host_staging()
    for timed in (False, True):
        t0 = time.perf_counter()
        for _ in range(n if timed else 3):
            p.upload_noise(stage)
            p.iterate(1, sptr)
            p.advance(sptr)
        p.synchronize()
        up_ms = 1e3 * (time.perf_counter() - t0) / n
    # (b) the draws are made for every step (numpy's legacy stream, bit for bit) and sent while they are made;
    #     the stream of step t + 1 is drawn while the device works on step t, as `train_step` does
    p.draw_noise(worker_seeds(SEED, K_PER_GPU, 1), _host_threads())
    for timed in (False, True):
        t0 = time.perf_counter()
        for i in range(n if timed else 3):
            p.commit_noise()
            p.iterate(1)
            p.advance()
            p.draw_noise(worker_seeds(SEED + 1 + i, K_PER_GPU, 1), _host_threads())
            p.read_step()
        gen_ms = 1e3 * (time.perf_counter() - t0) / n
    p.close()
    out["host_noise"] = {"ms_per_step_noise_uploaded": up_ms, "env_steps_per_sec_noise_uploaded": K_PER_GPU * H / (up_ms * 1e-3),
                         "ms_per_step_noise_drawn_and_uploaded": gen_ms, "host_threads": _host_threads(),
                         "what": "K=%d, H=%d: 14.7 MB of fp64 draws per step from the page-locked staging buffer "
                                 "(mppi_upload_noise: copy + sigma / filter kernel), mppi_iterate, mppi_advance; second figure: "
                                 "numpy's legacy stream generated per step and streamed to the device while it is made "
                                 "(mppi_draw_noise / mppi_commit_noise), state read back every step" % (K_PER_GPU, H)}
    ts = {}
    for noise in ("philox", "host"):
        e = get_environment('reacher_7dof')
        e.reset_model(seed=3, goal=(0, 0, 0))
        ag = MPPI(e, H=H, paths_per_cpu=K_PER_GPU, num_cpu=1, kappa=KAPPA, gamma=GAMMA, mean=np.zeros(M), filter_coefs=FILTER,
                  default_act='mean', seed=SEED, noise=noise)
        for _ in range(3):
            ag.train_step(niter=1)
        t0 = time.perf_counter()
        for _ in range(n):
            ag.train_step(niter=1)
        ts[noise] = 1e3 * (time.perf_counter() - t0) / n
        ag._planner.close()
    out["train_step_ms"] = dict(ts, what="MPPI.train_step(niter=1) of the Python API, K=%d, H=%d: whole call, logs and "
                                         "real-env step included" % (K_PER_GPU, H))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the host-noise and train_step timings (one GPU)")
    args = ap.parse_args()

    # RCCL prints a version banner on standard output when a communicator is made; the result line must be the only
    # thing there, so everything else this process (and the libraries in it) prints goes to standard error
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1:
            raise SystemExit("--gpus %d needs a launcher that starts %d ranks (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))

    base = None
    flops = None
    if rank == 0:
        flops = flops_per_env_step()
        if world == 1 and not args.no_cpu_baseline:
            cores = min(int(os.environ.get("OMP_NUM_THREADS", "16")), os.cpu_count() or 1, 16)
            base = cpu_baseline(cores)      # before the GPU is touched: the pool forks workers

    import torch
    import torch.distributed as dist
    from trajopt_amd import sharding

    # MPPI_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than ranks (ranks share
    # devices round-robin; RCCL refuses two ranks on one GPU, so the exchange then goes through torch.distributed and
    # host memory: a functional check whose timings mean nothing); the measured configuration is RCCL, one rank per GPU
    backend = os.environ.get("MPPI_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    # MPPI_BENCH_SHARDED=1 exercises the sharded code path (statistics, all-gather, merge) on one rank
    sharded = world > 1 or os.environ.get("MPPI_BENCH_SHARDED") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # One GPU: the headline planner, K = 4096.  Several GPUs: ONE planner of K_STRONG samples sharded over the
    # ranks (strong scaling: the work per planning step is fixed, each rank rolls out K_STRONG / N of it).
    K_total = K_PER_GPU if world == 1 else K_STRONG
    K_local = K_total // world
    if K_local * world != K_total:
        raise SystemExit("%d ranks do not divide K = %d" % (world, K_total))
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # strong scaling is quoted against the same planner on one GPU, timed here on rank 0 while the others wait
    t1_ms = None
    if world > 1:
        if rank == 0:
            p1 = make_planner(K_total, 0, local_rank)
            for _ in range(5):
                p1.plan(SEED, 1, 1, sptr)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                p1.plan(SEED, 1, 1, sptr)
            torch.cuda.synchronize()
            t1_ms = 1e3 * (time.perf_counter() - t0) / 50
            p1.close()
        barrier()

    p = make_planner(K_local, rank * K_local, local_rank)
    exchange = "none (single GPU)"
    torch_exchange = False
    if sharded:
        # the exchange lives inside the library: one RCCL all-gather of 3 + H*M doubles per rank, captured with the
        # kernels in the planning step's hipGraph
        exchange = "RCCL all-gather inside the library, captured in the step graph"
        try:
            if backend != "nccl" and world > 1:
                raise RuntimeError("rehearsal backend: ranks share GPUs")
            sharding.init_library_exchange(p, rank, world, device="cuda:%d" % local_rank)
        except Exception as e:             # noqa: BLE001  (reported in the result line, never silent)
            torch_exchange = True
            exchange = "torch.distributed all-gather between library calls (library exchange unavailable: %s)" % (e,)
            part = p.partials_tensor()
            gathered = torch.empty(world * part.numel(), dtype=torch.float64, device="cuda")

    # Every step: noise (drawn ahead in the previous step's advance launch), rollouts, update (sharded: statistics,
    # all-gather, merge), real-env fp64 step, horizon shift; mppi_plan replays them from a captured hipGraph, eight
    # steps per replay when asked for a run of steps.  A graph replay cannot be timed from inside: after the timed region the same steps are launched piece
    # by piece with HIP events around every rollout kernel (roofline.kernel_ms).
    EVENT_PASS_STEPS = 64

    def planning_step(i):
        # everything below is enqueued on one stream; there is no host synchronisation inside a step
        if not torch_exchange:
            p.plan(SEED, 1, 1, sptr)
            return
        p.local_partials(sptr)
        if backend == "nccl":
            dist.all_gather_into_tensor(gathered, part)
        else:
            stream.synchronize()
            host = [torch.empty(part.numel(), dtype=torch.float64) for _ in range(world)]
            dist.all_gather(host, part.cpu())
            gathered.copy_(torch.cat(host))
        p.merge(gathered.data_ptr(), world, sptr)
        p.advance_noise(SEED, sptr)

    # the collective's cost, measured: the same shard planned by a handle of its own without the exchange (update from
    # the local samples only), on rank 0 while the others wait
    local_ms = None
    if world > 1 and rank == 0:
        p0 = make_planner(K_local, rank * K_local, local_rank)
        for _ in range(5):
            p0.plan(SEED, 1, 1, sptr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            p0.plan(SEED, 1, 1, sptr)
        torch.cuda.synchronize()
        local_ms = 1e3 * (time.perf_counter() - t0) / 50
        p0.close()
    if world > 1:
        barrier()

    if sharded and not torch_exchange:
        # first step with the exchange captured in the graph; if the library cannot do it here, every rank fails the
        # same way and the run continues with the exchange made by torch.distributed (said so in the result line)
        try:
            planning_step(0)
            torch.cuda.synchronize()
        except Exception as e:             # noqa: BLE001
            torch_exchange = True
            exchange = "torch.distributed all-gather between library calls (captured library exchange failed: %s)" % (e,)
            p.close()
            p = make_planner(K_local, rank * K_local, local_rank)
            part = p.partials_tensor()
            gathered = torch.empty(world * part.numel(), dtype=torch.float64, device="cuda")
    if torch_exchange:
        p.generate_noise_step(SEED, sptr)
    def planning_steps(n):
        # a run of steps is one library call: mppi_plan replays its captured step, eight steps per replay
        if not torch_exchange:
            p.plan(SEED, 1, n, sptr)
        else:
            for i in range(n):
                planning_step(i)

    planning_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    planning_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    log = p.read_log(args.steps + args.warmup)
    log_total = int(log["total"])
    # rollout-kernel time: HIP events on the kernel's own stream around every launch of a pass of its own
    p.profile(True)
    p.profile_read()
    for i in range(EVENT_PASS_STEPS):
        planning_step(i)
    torch.cuda.synchronize()
    kernel_ms, launches = p.profile_read()
    p.profile(False)
    log_dist = float(np.linalg.norm(log["hand"][-1] - log["target"][-1]))
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # The same measurement for a planner eight times larger, where every GPU still has 65536 samples at N = 8:
    # K = 65536 is one wavefront per SIMD of ONE GPU, so sharding it only empties SIMDs (a rollout takes as long as
    # one sample's instruction stream however few samples a GPU holds); this second point shows the scaling of the
    # same code once there is work to share.  Reported under "strong_scaling_large", outside the timed region above.
    large = None
    # (MPPI_BENCH_LARGE=1 with MPPI_BENCH_SHARDED=1 rehearses this block on one rank)
    if (world > 1 or (sharded and os.environ.get("MPPI_BENCH_LARGE") == "1")) and not torch_exchange:
        p.close()
        t1_large = None
        if rank == 0:
            p1 = make_planner(K_STRONG_LARGE, 0, local_rank)
            for _ in range(3):
                p1.plan(SEED, 1, 1, sptr)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                p1.plan(SEED, 1, 1, sptr)
            torch.cuda.synchronize()
            t1_large = 1e3 * (time.perf_counter() - t0) / 20
            p1.close()
        barrier()
        KL = K_STRONG_LARGE // world
        pl = make_planner(KL, rank * KL, local_rank)
        sharding.init_library_exchange(pl, rank, world, device="cuda:%d" % local_rank)
        for _ in range(5):
            pl.plan(SEED, 1, 1, sptr)
        barrier()
        t0 = time.perf_counter()
        for _ in range(40):
            pl.plan(SEED, 1, 1, sptr)
        barrier()
        tl = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        tn_large = 1e3 * float(tl.item()) / 40
        pl.close()
        if rank == 0:
            large = {"K_total": K_STRONG_LARGE, "one_gpu_ms_per_step": t1_large, "n_gpu_ms_per_step": tn_large,
                     "efficiency": t1_large / (world * tn_large),
                     "env_steps_per_sec": K_STRONG_LARGE * H / (tn_large * 1e-3)}

    # One GPU: the other flavours of the planning step, outside the timed region (extra keys of the result line).
    extras = {}
    if world == 1 and not args.no_extras:
        p.close()
        extras = other_step_flavours(sptr)

    if rank == 0:
        env_steps = K_total * H * args.steps
        k_ms = kernel_ms / max(launches, 1)
        flop_launch = flops * K_local * H
        achieved = flop_launch / (k_ms * 1e-3) / 1e12
        bytes_launch = K_local * H * M * 4 + K_local * 4 + H * M * 8
        ms_step = 1e3 * dt / args.steps
        out = {
            "metric": "env_steps_per_sec", "value": env_steps / dt, "unit": "env_steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak" if world == 1 else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "planning_steps_per_sec": args.steps / dt,
            "config": {"workload": "reacher_7dof MPPI planning step, K=%d samples (%d per GPU), H=%d, niter=1, "
                                   "dense reward, kappa=25, filter (0.25, 0.8, 0), Philox noise on device, "
                                   "real-env fp64 step included" % (K_total, K_local, H),
                       "K_per_gpu": K_local, "K_total": K_total, "H": H, "frame_skip": 2, "niter": 1,
                       "parallelism": ("one planner, samples sharded %d-way by global index; per step one all-gather of "
                                       "%d doubles per rank" % (world, 3 + H * M)) if world > 1 else "single GPU",
                       "exchange": exchange},
            "roofline": {"bound": "valu", "kernel": "rollout_kernel<float, Reacher7Dof, true>",
                         "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                         "traffic": measured_traffic() if K_local == K_PER_GPU else None,
                         "flop_per_env_step": flops, "kernel_ms": k_ms, "launches": launches,
                         "kernel_ms_how": "HIP events on the planner's stream around every rollout launch of %d planning "
                                          "steps issued piece by piece after the timed region" % EVENT_PASS_STEPS,
                         **(issue_statistics(flops) if K_local == K_PER_GPU else {}),
                         "hbm": {"algorithmic_bytes_per_launch": bytes_launch,
                                 "achieved_GBs": bytes_launch / (k_ms * 1e-3) / 1e9, "peak_GBs": PEAK_HBM_GBS,
                                 "frac": bytes_launch / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}},
        }
        if t1_ms is not None:
            # T(1 GPU, K) / (N * T(N GPUs, K)), both measured in this run
            out["strong_scaling"] = {"K_total": K_total, "one_gpu_ms_per_step": t1_ms, "n_gpu_ms_per_step": ms_step,
                                     "efficiency": t1_ms / (world * ms_step)}
        if local_ms is not None:
            # what the collective costs: this rank's shard planned alone (no exchange) against the sharded step
            out["exchange"] = {"shard_alone_ms_per_step": local_ms, "sharded_ms_per_step": ms_step,
                               "exchange_ms": ms_step - local_ms, "how": exchange}
        out.update(extras)
        if large is not None:
            out["strong_scaling_large"] = large
        if base is not None:
            out["cpu_baseline"] = base
        out["config"]["real_env_steps_logged"] = log_total
        out["config"]["final_hand_target_distance"] = log_dist
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
