"""Experiment (round 4): the benchmark's 4096 envs as S independent worlds of 4096 / S envs, each on its own HIP stream, control steps
enqueued round-robin.  A launch ends with its slowest wave; with S streams the tail of shard i's step k overlaps other shards' step k + 1.
Usage: python tools/exp/shards.py [--config 2] [--steps 200] [--shards 1 2 4 8 16]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(config, S, steps, warmup, N=4096):
    import torch
    from raisimlib_amd import BatchedWorld, workload
    dev = torch.device("cuda:0")
    recipe = bench.Recipe(config, -1.0)
    model, feet = recipe.model, np.asarray(recipe.feet, np.int32)
    n = N // S
    shards = []
    for i in range(S):
        w = BatchedWorld(model, n, device=0)
        recipe.setup_world(w, n, i * n)
        gc0, gv0 = recipe.initial_state(n, i * n)
        gc0_d = torch.from_numpy(gc0.astype(np.float32)).to(dev); gv0_d = torch.from_numpy(gv0.astype(np.float32)).to(dev)
        w.set_state(gc0, gv0)
        w.set_pd_target(None, np.zeros((n, model.nv), np.float32))
        bank = [torch.from_numpy(recipe.targets(n, k, i * n).astype(np.float32)).to(dev) for k in range(bench.TARGET_BANK)]
        obs = torch.zeros((n, w.obs_dim(len(feet))), dtype=torch.float32, device=dev)
        done = torch.zeros(n, dtype=torch.uint8, device=dev)
        w.set_done_output(done.data_ptr())
        fn = w.control_step_plan(workload.SUBSTEPS, obs.data_ptr(), feet, feet, gc0_d.data_ptr(), gv0_d.data_ptr(), n)
        shards.append((w, fn, [b.data_ptr() for b in bank], (gc0_d, gv0_d, bank, obs, done)))
    k = 0
    for _ in range(warmup):
        for w, fn, bp, _ in shards:
            fn(bp[k % len(bp)])
        k += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for w, fn, bp, _ in shards:
            fn(bp[k % len(bp)])
        k += 1
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    q = np.concatenate([w.get_state()[0] for w, *_ in shards])
    for w, *_ in shards:
        w.close()
    return N * workload.SUBSTEPS * steps / dt, dt / steps * 1e3, t_host / steps * 1e3, float(np.abs(q).sum())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--shards", type=int, nargs="+", default=[1, 2, 4, 8, 16])
    a = ap.parse_args()
    for S in a.shards:
        v, ms, host_ms, chk = run(a.config, S, a.steps, a.warmup)
        print(f"config {a.config} shards {S:3d}: {v / 1e6:8.2f} M env-steps/s, {ms:.4f} ms per control step (host enqueue {host_ms:.4f} ms), state checksum {chk:.6e}", flush=True)
