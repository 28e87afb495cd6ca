#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched World::integrate() hot path on N MI355X (BASELINE.json metric).

One "step" = one control step of the vectorised env = 4 x integrate() (dt = 0.0025) for 4096 ANYmal-C-like envs
per GPU on flat ground (BASELINE.json configs[1]), i.e. 16384 env-steps per GPU per step, run as ONE fused kernel
launch through the C-ABI (rsb_integrate(w, 4)).  Per step, inside the timed region, every rank also
  - copies the control step's PD targets (nominal + U(-0.3,0.3) rad) into the world (device->device),
  - applies rsg_anymal's termination rule on device (any non-foot contact -> reset that env),
  - gathers the (q, u, foot contact force) observation block and, for N>1, all-gathers it over RCCL/xGMI.
State is resident in HBM when the timed region starts.  Envs are sharded 4096 per rank (weak scaling); the
only collective is the obs all-gather.

Output: ONE JSON line on rank 0 (contract in the task statement) with `roofline` (HBM; algorithmic bytes from
SURVEY.md §8d: 456 B per env-step x 16384 env-steps per launch, over the step kernel's mean launch time measured
with HIP events on the launch stream) and `cpu_baseline` (the in-repo fp64 oracle, OpenMP over envs on the host
cores, same workload recipe, bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
BYTES_PER_ENV_STEP = 456.0        # SURVEY.md §8d contract number (unfused state traffic, fp32)
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy peak)
TARGET_BANK = 16                  # distinct pre-generated PD-target sets cycled through in the timed loop
EVENT_STRIDE = 8                  # every 8th launch of the timed region is bracketed by HIP events


def recorded_traffic(n_envs, substeps):
    """HBM bytes per launch of the step kernel from the newest committed PMC pass (profiles/rNN_pmc_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this same command, tools/collect_profiles.sh).
    The counters cannot be collected from inside this process, so the committed measurement is reported - only
    when it was taken on this workload - together with its provenance; otherwise null."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files or n_envs != ENVS_PER_GPU or substeps != 4:
        return None, None, None
    try:
        rec = json.load(open(files[-1]))
        return float(rec["hbm_bytes_per_launch_raw"]), os.path.relpath(files[-1], ROOT) + \
            " (FETCH_SIZE+WRITE_SIZE per launch, raw: gfx950 factors for 4 B/lane row accesses are uncalibrated)", rec
    except Exception:
        return None, None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--max-iter", type=int, default=0, help="contact-solver iteration cap (0 = library default)")
    ap.add_argument("--lanes-per-env", type=int, default=0)
    ap.add_argument("--no-reset", action="store_true", help="disable the non-foot-contact termination rule")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--stall-window", type=int, default=-1, help="diagnostic: solver stagnation window (library default 4)")
    ap.add_argument("--freeze-after", type=int, default=-1, help="diagnostic: sweeps before friction directions lag (default 5)")
    ap.add_argument("--settle-tol", type=float, default=-1.0, help="diagnostic: settled-direction tolerance (default 1e-4 rad)")
    ap.add_argument("--early-termination", action="store_true",
                    help="NOT the headline workload: envs stop integrating at the sub-step of their first non-foot contact")
    ap.add_argument("--overlap-collective", action="store_true",
                    help="double-buffer the obs block and overlap the all-gather of step k with the kernel of step k+1 "
                         "(default: in line on the launch stream; the overlap could not be tried on >1 GPU here)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="diagnostic: do not bracket the step kernel with HIP events (roofline fields become null)")
    ap.add_argument("--target-amplitude", type=float, default=0.3,
                    help="diagnostic: amplitude (rad) of the uniform PD-target noise around the nominal pose (config 2: 0.3)")
    ap.add_argument("--force-collective", action="store_true",
                    help="diagnostic: run the obs all-gather (RCCL) even with one rank, to see its per-step cost")
    return ap.parse_args()


def cpu_baseline(model, feet, max_iter, reset, budget_s):
    """Time the fp64 oracle (OpenMP over envs) on a bounded sample of the same workload.

    The container's visible core count can exceed its CPU quota, so the leg first probes a few thread counts for
    ~1 s each and then spends the budget at the fastest one; `cores` reports the threads actually used.
    """
    from oracle.pyoracle import Oracle  # the CPU baseline leg is one of the three allowed oracle users
    from raisimlib_amd import workload
    orc = Oracle(model.blob)
    if max_iter > 0:
        orc.p.max_iter = max_iter
    n = 4096
    gc0, gv0 = workload.anymal_initial_state(n)
    kp, kd = workload.anymal_gains()
    kp, kd = kp.astype(np.float64), kd.astype(np.float64)
    dtg = np.zeros((n, model.nv))
    feet_set = np.zeros(model.ncol, bool)
    feet_set[feet] = True
    state = {"q": gc0.astype(np.float32).astype(np.float64), "u": gv0.copy(), "cs": 0}
    warm = orc.new_warm_state(n)      # the solver warm state the device keeps per env (cleared on reset, as there)

    def run(threads, seconds):
        spent, steps = 0.0, 0
        while spent < seconds:
            pt = workload.anymal_targets(n, state["cs"]).astype(np.float32).astype(np.float64)
            t0 = time.perf_counter()
            r = orc.step_batch(state["q"], state["u"], workload.SUBSTEPS, kp, kd, pt, dtg, nthreads=threads,
                               want_contacts=reset, lam_warm=warm)
            spent += time.perf_counter() - t0
            q, u = r["q"], r["u"]
            if reset:
                con, ncs = r["contacts"], r["n_contacts"]
                valid = np.arange(con.shape[1])[None, :] < ncs[:, None]
                term = (valid & ~feet_set[con["collision"]]).any(axis=1) | (r["flags"] & 2).astype(bool)
                q[term], u[term] = gc0[term], gv0[term]
                warm[term] = 0.0
            state["q"], state["u"] = q, u
            state["cs"] += 1
            steps += n * workload.SUBSTEPS
        return steps / spent, spent, steps

    hw = orc.max_threads()
    for _ in range(40):  # untimed: bring the population into the steady contact regime before probing thread counts
        pt = workload.anymal_targets(n, state["cs"]).astype(np.float32).astype(np.float64)
        r = orc.step_batch(state["q"], state["u"], workload.SUBSTEPS, kp, kd, pt, dtg, nthreads=0, want_contacts=reset,
                           lam_warm=warm)
        q, u = r["q"], r["u"]
        if reset:
            con, ncs = r["contacts"], r["n_contacts"]
            valid = np.arange(con.shape[1])[None, :] < ncs[:, None]
            term = (valid & ~feet_set[con["collision"]]).any(axis=1) | (r["flags"] & 2).astype(bool)
            q[term], u[term] = gc0[term], gv0[term]
            warm[term] = 0.0
        state["q"], state["u"] = q, u
        state["cs"] += 1
    try:
        hw = min(hw, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    cands = sorted({1, min(8, hw), min(16, hw), min(32, hw), min(64, hw), hw})
    probe = {t: run(t, 1.0)[0] for t in cands}
    best = max(probe, key=probe.get)
    rate, spent, steps = run(best, max(budget_s - len(cands), 2.0))
    return {"value": rate, "unit": "env-steps/s", "cores": int(best), "kind": "port",
            "single_thread": probe[1],
            "sample": f"{n} envs, {steps} env-steps of the same workload in {spent:.1f} s, fp64 oracle, OpenMP "
                      f"schedule(static) over envs; thread-count probe {{threads: env-steps/s}} = "
                      + json.dumps({str(k): round(v) for k, v in probe.items()})}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from raisimlib_amd import BatchedWorld, Model, rsc_path, workload

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    coll = world_size > 1 or args.force_collective     # the obs all-gather is part of the step
    if coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if coll:
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)

    N = args.envs_per_gpu
    model = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
    feet = model.collision_indices("_foot")
    world = BatchedWorld(model, N, device=local_rank)
    stream = torch.cuda.Stream(device=dev)       # everything below (kernels, copies, events, collectives) is ordered on it
    torch.cuda.set_stream(stream)
    world.set_stream(stream.cuda_stream)
    if args.max_iter > 0:
        world.set_contact_solver_param(1.0, 1.0, 1.0, args.max_iter, 1e-5)
    if args.lanes_per_env:
        world.set_lanes_per_env(args.lanes_per_env)
    if args.early_termination:
        world.set_early_termination(True)
    if args.stall_window >= 0:
        world.set_solver_stagnation_exit(args.stall_window, 0.5)
    if args.freeze_after >= 0 or args.settle_tol >= 0:
        world.set_solver_friction_lag(args.freeze_after if args.freeze_after >= 0 else 5, True,
                                      args.settle_tol if args.settle_tol >= 0 else 1e-4)
    world.set_time_step(workload.DT)
    kp, kd = workload.anymal_gains()
    world.set_pd_gains(kp, kd)

    # per-rank env shard: seeds offset by rank
    gc0, gv0 = workload.anymal_initial_state(N, env_offset=rank * N)
    gc0_d = torch.from_numpy(gc0.astype(np.float32)).to(dev)
    gv0_d = torch.from_numpy(gv0.astype(np.float32)).to(dev)
    world.set_state(gc0, gv0)
    world.set_pd_target(None, np.zeros((N, model.nv), np.float32))
    bank = [torch.from_numpy(workload.anymal_targets(N, k, env_offset=rank * N, amplitude=args.target_amplitude).astype(np.float32)).to(dev)
            for k in range(TARGET_BANK)]
    obs_dim = world.obs_dim(len(feet))
    # obs block of this rank and the gathered block of all ranks; with --overlap-collective double-buffered, so that the
    # all-gather of control step k (RCCL, its own stream) overlaps the kernel of step k+1
    nbuf = 2 if (coll and args.overlap_collective) else 1
    obs_b = [torch.empty((N, obs_dim), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    all_obs_b = [torch.empty((world_size * N, obs_dim), dtype=torch.float32, device=dev) for _ in range(nbuf)] if coll else obs_b
    feet_idx = np.asarray(feet, np.int32)
    reset = not args.no_reset

    # one foreign call per control step (rsb_control_step); the step kernel's launches are bracketed by HIP events
    # inside the library (ring of event pairs on the launch stream, read back after the timed region)
    # the library brackets every EVENT_STRIDE-th launch with a HIP event pair (an event pair costs ~7 us of stream time,
    # 5 % of a step: bracketing every launch would lower the very number being measured)
    n_sampled = max(args.steps // EVENT_STRIDE, 1)
    if not args.no_kernel_events:
        world.enable_timing(max(n_sampled, 2))
        world.set_timing_stride(EVENT_STRIDE if args.steps >= 2 * EVENT_STRIDE else 1)
        if args.steps < 2 * EVENT_STRIDE:
            n_sampled = args.steps
    step_fns = [world.control_step_plan(workload.SUBSTEPS, o.data_ptr(), feet_idx, feet_idx if reset else None,
                                        gc0_d.data_ptr() if reset else 0, gv0_d.data_ptr() if reset else 0, N) for o in obs_b]
    bank_ptr = [b.data_ptr() for b in bank]
    pending = [None] * nbuf

    def control_step(k):
        b = k % nbuf
        if pending[b] is not None:          # the gather that still reads this buffer (stream-side wait, the host runs on)
            pending[b].wait()
            pending[b] = None
        step_fns[b](bank_ptr[k % TARGET_BANK])
        if coll:
            if nbuf == 2:
                pending[b] = dist.all_gather_into_tensor(all_obs_b[b], obs_b[b], async_op=True)
            else:
                dist.all_gather_into_tensor(all_obs_b[b], obs_b[b])

    def drain():
        for b in range(nbuf):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    for k in range(args.warmup):
        control_step(k)
    drain()
    if world_size > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        control_step(args.warmup + k)
    drain()
    t_enqueued = time.perf_counter() - t0       # host side done; the GPU may still be working
    if world_size > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world_size > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    kernel_ms = (world.read_kernel_ms(n_sampled).astype(np.float64) if not args.no_kernel_events   # launches sampled from the timed region
                 else np.full(args.steps, elapsed / args.steps * 1e3))
    env_steps_per_step = N * workload.SUBSTEPS
    total_env_steps = world_size * env_steps_per_step * args.steps
    value = total_env_steps / elapsed
    iters = world.get_solver_iterations()
    counts, _ = world.get_contacts()
    q_end, _ = world.get_state()

    if rank == 0:
        kmean = float(kernel_ms.mean()) * 1e-3
        achieved = BYTES_PER_ENV_STEP * env_steps_per_step / kmean / 1e9
        traffic, traffic_src, pmc = recorded_traffic(N, workload.SUBSTEPS) if reset and not args.max_iter else (None, None, None)
        # what actually bounds the kernel: VALU issue slots of the one wave each SIMD holds (recorded SQ_INSTS_VALU x 4
        # cycles over the measured launch time at the 2.4 GHz shader clock), reported next to the HBM roofline
        valu = None
        if pmc and pmc.get("counters", {}).get("SQ_INSTS_VALU"):
            waves = -(-N * world.lanes_per_env() // 64)
            valu = {"valu_inst_per_wave_per_launch": pmc["counters"]["SQ_INSTS_VALU"] / waves,
                    "issue_slot_frac": pmc["counters"]["SQ_INSTS_VALU"] / waves * 4.0 / (kmean * 2.4e9),
                    "note": "one wave per SIMD: (VALU instructions x 4 cycles) / launch cycles; recorded PMC pass, measured launch time"}
        out = {
            "metric": "env-steps/sec, 4096 ANYmal-C envs flat terrain dt=0.0025",
            "value": value, "unit": "env-steps/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "configs[1]: 4096 ANYmal-C-like (synthetic stand-in URDF) envs per GPU, flat ground, "
                            "dt=0.0025, 4 sub-steps per control step fused in one launch, PD kp=50 kd=0.2, targets = "
                            f"nominal + U(-{args.target_amplitude:g},{args.target_amplitude:g}) rad per control step, per-env seed 1234+i"
                            + (", non-foot contact -> reset (rsg_anymal rule)" if reset else ", no resets")
                            + (", EARLY TERMINATION at the sub-step of the first non-foot contact (not upstream's rule)" if args.early_termination else "")
                            + ", obs (q,u,foot force) gathered each control step",
                "envs_per_gpu": N, "substeps_per_step": workload.SUBSTEPS,
                "contact_solver": {"max_iter": args.max_iter or 150, "threshold_rel": 1e-5, "alpha": [1.0, 1.0, 1.0]},
                "lanes_per_env": world.lanes_per_env(), "parallelism": f"env-shard x{world_size}",
                "obs_all_gather": ("overlapped with the next control step (double-buffered)" if nbuf == 2 else "in line") if coll else "none (1 rank)",
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "rsb_step_kernel", "kernel_ms_mean": float(kernel_ms.mean()),
                         "kernel_ms_p50": float(np.median(kernel_ms)), "kernel_launches_timed": int(len(kernel_ms)),
                         "algorithmic_bytes_per_launch": BYTES_PER_ENV_STEP * env_steps_per_step, "valu_issue": valu},
            "host_enqueue_ms_per_step": t_enqueued / args.steps * 1e3,
            "state_at_end": {"solver_iters_mean": float(iters.mean()), "solver_iters_max": int(iters.max()),
                             "contacts_per_env": float(counts.mean()), "base_height_mean": float(q_end[:, 2].mean())},
        }
        if world_size == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(model, feet, args.max_iter, reset, args.cpu_seconds)
    world.close()
    if coll:
        dist.destroy_process_group()
    if rank == 0:
        try:    # RCCL writes its version banner through C stdio; flush it so that the JSON line is the last line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
