#!/usr/bin/env python3
"""Headline benchmark: MPPI planning steps on reacher_7dof, K = 4096 samples per GPU, H = 64.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one planning step of `MPPI.train_step(niter=1)` with everything resident on the
device and enqueued on one stream without host synchronisation: Philox noise + filter,
K x H rollouts with fused reward / discounted return, max / weights / weighted action sum,
(N > 1: one RCCL all-gather of the per-rank statistics), merge into act_sequence, the single
fp64 step of the real environment from the planner's own state, and the horizon shift.
On one GPU a step is one `mppi_plan` call (a hipGraph replay; every 8th step is launched kernel
by kernel with HIP events around the rollout kernel, which gives roofline.kernel_ms).
With N ranks each rank rolls out its own K = 4096 shard of a 4096*N-sample planner (weak
scaling); sample noise is addressed by global index so the result does not depend on N.

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


def measured_traffic():
    """HBM bytes per rollout launch from the committed rocprofv3 PMC summary (FETCH_SIZE / WRITE_SIZE
    passes, gfx950 corrections applied); None if the summary is not for this workload."""
    path = os.path.join(ROOT, "profiles", "r1g_rollout_traffic.json")
    try:
        d = json.load(open(path))
        if d.get("K") == K_PER_GPU and d.get("H") == H:
            return float(d["hbm_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

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
    from trajopt_amd.device import DeviceModel, DevicePlanner

    # MPPI_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (ranks share
    # devices round-robin and are time-sliced by the driver, the exchange goes through host memory: a functional
    # check whose timings mean nothing); the measured configuration is RCCL ("nccl"), one rank per GPU
    backend = os.environ.get("MPPI_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    # MPPI_BENCH_SHARDED=1 exercises the sharded code path (partials, all-gather, merge) on one rank
    sharded = world > 1 or os.environ.get("MPPI_BENCH_SHARDED") == "1"
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    dm = DeviceModel.get("reacher_7dof")
    p = DevicePlanner(dm, K_PER_GPU, H, kappa=KAPPA, gamma=GAMMA, sigma=FILTER[0], betas=FILTER[1:],
                      mean=np.zeros(M), default_act="mean", dtype="f32", device=local_rank,
                      k_offset=rank * K_PER_GPU)
    p.set_start_state(np.zeros(M), np.zeros(M), np.zeros(3), 0)
    p.set_base_actions(np.zeros((H, M)))
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream
    if sharded:
        part = p.partials_tensor()
        gathered = torch.empty(world * part.numel(), dtype=torch.float64, device="cuda")

    # One GPU: every step is mppi_plan (noise ahead, rollouts, update, real-env step, shift).  It replays one captured
    # hipGraph per step; on every PROFILE_EVERY-th step it launches the same kernels one by one instead, with HIP
    # events around the rollout kernel (a graph replay cannot be timed from inside), which is where
    # roofline.kernel_ms comes from.  Several ranks: the exchange sits between the kernels, so they are launched
    # one by one on every step, all of them timed.
    PROFILE_EVERY = 8

    def planning_step(i):
        # everything below is enqueued on one stream; there is no host synchronisation inside a step
        if not sharded:
            p.profile(timed[0] and i % PROFILE_EVERY == 0)
            p.plan(SEED, 1, 1, sptr)
            return
        # the Philox noise of this step (key = SEED + timestep, read on the device) was drawn by the previous
        # step's advance launch, next to the real-env step; the first step draws its own (before the loop)
        p.local_partials(sptr)
        if backend == "nccl":
            dist.all_gather_into_tensor(gathered, part)
        else:
            dist.all_gather(list(gathered.view(world, -1).unbind(0)), part)
        p.merge(gathered.data_ptr(), world, sptr)
        p.advance_noise(SEED, sptr)                  # real env fp64 step with act_sequence[0], log, horizon shift,
                                                     # and the next step's noise on the idle CUs of the same launch

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    timed = [False]
    if sharded:
        p.generate_noise_step(SEED, sptr)
    for i in range(args.warmup):
        planning_step(i)
    timed[0] = True
    p.profile(sharded)
    p.profile_read()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        planning_step(i)
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms, launches = p.profile_read()
    p.profile(False)
    if sharded:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        env_steps = world * K_PER_GPU * H * args.steps
        k_ms = kernel_ms / max(launches, 1)
        flop_launch = flops * K_PER_GPU * H
        achieved = flop_launch / (k_ms * 1e-3) / 1e12
        bytes_launch = K_PER_GPU * H * M * 4 + K_PER_GPU * 4 + H * M * 8
        out = {
            "metric": "env_steps_per_sec", "value": env_steps / dt, "unit": "env_steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "planning_steps_per_sec": args.steps / dt,
            "config": {"workload": "reacher_7dof MPPI planning step, K=%d samples per GPU (%d total), H=%d, niter=1, "
                                   "dense reward, kappa=25, filter (0.25, 0.8, 0), Philox noise on device, "
                                   "real-env fp64 step included" % (K_PER_GPU, world * K_PER_GPU, H),
                       "K_per_gpu": K_PER_GPU, "K_total": world * K_PER_GPU, "H": H, "frame_skip": 2, "niter": 1,
                       "parallelism": "samples sharded %d-way, one all-gather of %d doubles per rank per step"
                                      % (world, 3 + H * M) if world > 1 else "single GPU"},
            "roofline": {"bound": "valu", "kernel": "rollout_kernel<float, Reacher7Dof, true>",
                         "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": measured_traffic(),
                         "flop_per_env_step": flops, "kernel_ms": k_ms, "launches": launches,
                         "hbm": {"algorithmic_bytes_per_launch": bytes_launch,
                                 "achieved_GBs": bytes_launch / (k_ms * 1e-3) / 1e9, "peak_GBs": PEAK_HBM_GBS,
                                 "frac": bytes_launch / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}},
        }
        if base is not None:
            out["cpu_baseline"] = base
        log = p.read_log(args.steps + args.warmup)
        out["config"]["real_env_steps_logged"] = int(log["total"])
        out["config"]["final_hand_target_distance"] = float(np.linalg.norm(log["hand"][-1] - log["target"][-1]))
        print(json.dumps(out), flush=True)
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
