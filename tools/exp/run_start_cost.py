#!/usr/bin/env python3
"""Round-5 experiment: the FIXED cost of a timed region (what `--steps 20` pays and `--steps 300` amortises).
For K in (5, 10, 20, 40, 80, 160) control steps bracketed the way bench.py brackets them (synchronise, K steps, join + synchronise): the median
time of 7 repetitions; a straight-line fit gives the cost per step (slope) and the fixed cost of a region (intercept), for
  open loop pipelined / lock-step    and    closed loop (linear stage) pipelined / lock-step.
usage: python tools/exp/run_start_cost.py"""
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import bench
from raisimlib_amd import Model, rsc_path, workload

N = 4096
KS = (5, 10, 20, 40, 80, 160)
dev = torch.device("cuda:0")


def region(fn, join, steps):
    join(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    join(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


def fit(name, fn, join):
    med = []
    for k in KS:
        ts = sorted(region(fn, join, k) for _ in range(7))
        med.append(ts[3])
    slope, icpt = np.polyfit(np.asarray(KS, float), np.asarray(med), 1)
    print(f"{name:34s} per step {slope:7.2f} us   fixed {icpt:7.1f} us   medians " + " ".join(f"{k}:{m:.0f}" for k, m in zip(KS, med)) +
          f"   -> rate at 20 steps {N * 4 * 20 / med[2]:.1f} M, asymptote {N * 4 / slope:.1f} M", flush=True)


from test_gpu_pipeline import Rig
recipe = bench.Recipe(2, -1.0)
for pipe in (True, False):
    r = Rig(recipe, N, pipe)
    r.step(300); r.w.synchronize()
    fit(f"open loop {'pipelined' if pipe else 'lock-step'}", r.step, r.w.step_pipeline_join)
    r.close()

model = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
env = workload.closed_loop_env(model, N)
W = torch.from_numpy(workload.closed_loop_policy(env.num_obs, env.num_acts, workload.CLOSED_LOOP_W_SCALE)).to(dev)
noise = torch.from_numpy(workload.closed_loop_noise(N, 128)).to(dev)
for pipe in (True, False):
    env.world.set_step_pipelining(pipe)
    env.reset()
    env.rollout_linear(300, W, noise=noise)
    env.world.step_pipeline_join()
    fit(f"closed loop {'pipelined' if pipe else 'lock-step'}", lambda k: env.rollout_linear(k, W, noise=noise), env.world.step_pipeline_join)
env.close()
