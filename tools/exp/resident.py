"""Experiment (GPU): the resident launch against the pipelined and the lock-step control steps, open loop, same population.
usage: python tools/exp/resident.py [config] [N]   -> env-steps/s for K = 5 .. 300 control steps per timed region (median of 7 regions each)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from raisimlib_amd import BatchedWorld, workload

config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
PERIOD = 128
dev = torch.device("cuda:0")
r = bench.Recipe(config, -1.0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)


def make(mode):
    w = BatchedWorld(r.model, N)
    w.set_stream(stream.cuda_stream)
    r.setup_world(w, N, 0)
    gc0, gv0 = r.initial_state(N, 0)
    w.set_state(gc0, gv0)
    w.set_pd_target(None, np.zeros((N, r.model.nv), np.float32))
    g0 = torch.from_numpy(gc0.astype(np.float32)).to(dev); v0 = torch.from_numpy(gv0.astype(np.float32)).to(dev)
    feet = np.asarray(r.feet, np.int32)
    od = w.obs_dim(len(feet))
    obs = torch.zeros((N, od), device=dev)
    done = torch.zeros(N, dtype=torch.uint8, device=dev)
    if mode == "resident":
        w.set_step_residency(True)
        assert w.residency_status(0)
    elif mode == "pipelined":
        assert w.set_step_pipelining(True)
    fn = w.control_steps_plan(workload.SUBSTEPS, bank.data_ptr(), PERIOD, obs.data_ptr(), 0, feet, feet, g0.data_ptr(), v0.data_ptr(), N, done.data_ptr(), 0)
    return w, fn, (g0, v0, obs, done)


bank = torch.from_numpy(np.stack([r.targets(N, k, 0).astype(np.float32) for k in range(PERIOD)])).to(dev)
for mode in ("resident", "pipelined", "lockstep"):
    w, fn, keep = make(mode)
    k = 0
    fn(200, k); k += 200
    w.synchronize()
    line = []
    for K in (5, 20, 50, 100, 300):
        vals = []
        for rep in range(7):
            fn(5, k); k += 5
            w.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(K, k); k += K
            w.synchronize(); torch.cuda.synchronize()
            vals.append(N * 4 * K / (time.perf_counter() - t0))
        line.append(f"K={K}: {np.median(vals) / 1e6:.1f} M ({min(vals) / 1e6:.1f}-{max(vals) / 1e6:.1f})")
    print(f"config {config} N {N} {mode:10s} " + " | ".join(line), flush=True)
    w.close()
