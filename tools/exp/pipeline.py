"""Experiment (round 4): rsb_set_step_pipelining on the benchmark workloads: throughput and bit-identity of the final state.
Usage: python tools/exp/pipeline.py [--config 2 3 5] [--steps 300]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(config, pipe, steps, warmup, N=4096, timing=False):
    import torch
    from raisimlib_amd import BatchedWorld, workload
    dev = torch.device("cuda:0")
    recipe = bench.Recipe(config, -1.0)
    model, feet = recipe.model, np.asarray(recipe.feet, np.int32)
    w = BatchedWorld(model, N, device=0)
    recipe.setup_world(w, N, 0)
    gc0, gv0 = recipe.initial_state(N, 0)
    gc0_d = torch.from_numpy(gc0.astype(np.float32)).to(dev); gv0_d = torch.from_numpy(gv0.astype(np.float32)).to(dev)
    w.set_state(gc0, gv0)
    w.set_pd_target(None, np.zeros((N, model.nv), np.float32))
    bank = [torch.from_numpy(recipe.targets(N, k, 0).astype(np.float32)).to(dev) for k in range(bench.TARGET_BANK)]
    obs = torch.zeros((N, w.obs_dim(len(feet))), dtype=torch.float32, device=dev)
    done = torch.zeros(N, dtype=torch.uint8, device=dev)
    w.set_done_output(done.data_ptr())
    fn = w.control_step_plan(workload.SUBSTEPS, obs.data_ptr(), feet, feet, gc0_d.data_ptr(), gv0_d.data_ptr(), N)
    bp = [b.data_ptr() for b in bank]
    w.set_step_pipelining(pipe)
    k = 0
    for _ in range(warmup):
        fn(bp[k % len(bp)]); k += 1
    w.synchronize()
    torch.cuda.synchronize()
    if timing:
        w.enable_timing(64); w.set_timing_stride(max(1, steps // 64))
    t0 = time.perf_counter()
    for _ in range(steps):
        fn(bp[k % len(bp)]); k += 1
    t_host = time.perf_counter() - t0
    w.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = float(np.mean(w.read_kernel_ms(min(64, steps)))) if timing else float("nan")
    q, u = w.get_state()
    cnt, _ = w.get_contacts()
    st = w.step_pipelining_stats()
    o = obs.cpu().numpy()
    w.close()
    return N * workload.SUBSTEPS * steps / dt, dt / steps * 1e3, t_host / steps * 1e3, kms, q, u, cnt, o, st


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, nargs="+", default=[2, 3, 5])
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, nargs="+", default=[4096])
    a = ap.parse_args()
    for c, n_envs in [(c, n) for c in a.config for n in a.envs]:
        ref = None
        for pipe in (False, True, False, True):
            v, ms, host_ms, kms, q, u, cnt, o, st = run(c, pipe, a.steps, a.warmup, N=n_envs, timing=True)
            same = "" if ref is None else f", state / contacts / obs bit-identical to the first run: {np.array_equal(q, ref[0]) and np.array_equal(u, ref[1]) and np.array_equal(cnt, ref[2]) and np.array_equal(o, ref[3])}"
            if ref is None:
                ref = (q, u, cnt, o)
            print(f"config {c} N {n_envs} pipelining {int(pipe)}: {v / 1e6:8.2f} M env-steps/s, {ms:.4f} ms per control step (host enqueue {host_ms:.4f} ms, launch start->end {kms:.4f} ms), "
                  f"pipelined launches / joins {st}{same}", flush=True)
