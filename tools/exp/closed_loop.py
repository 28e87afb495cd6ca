#!/usr/bin/env python3
"""Round-5 experiment: what the closed loop's hand-over costs.  One process, N = 4096, config 2:
  open loop  (rsb_control_step, targets from a bank)      pipelined / lock-step rate, mean wait of a step workgroup for its block
  closed loop (rsb_closed_loop_run_linear)                 pipelined / lock-step rate, mean wait
RSB_PIPE_STATS=1 is set here (the wait statistics cost one s_memrealtime pair and one atomic per workgroup and launch).
usage: python tools/exp/closed_loop.py [steps] [stage-grid ...]"""
import os
import sys
import time

os.environ.setdefault("RSB_PIPE_STATS", "1")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import bench
from raisimlib_amd import Model, rsc_path, workload

K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
grids = [int(x) for x in sys.argv[2:]] or [0]
N = 4096
dev = torch.device("cuda:0")


def rate(fn, join, steps):
    join(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    join(); torch.cuda.synchronize()
    return N * 4 * steps / (time.perf_counter() - t0) / 1e6


# ---- open loop
from test_gpu_pipeline import Rig
recipe = bench.Recipe(2, -1.0)
for pipe in (True, False):
    r = Rig(recipe, N, pipe)
    r.step(300); r.w.synchronize()
    if pipe:
        r.w.debug_pipeline_wait_stats()
    v = rate(r.step, r.w.step_pipeline_join, K)
    st = r.w.debug_pipeline_wait_stats() if pipe else (0, 0)
    print(f"open loop   {'pipelined' if pipe else 'lock-step'}: {v:7.1f} M env-steps/s   wait {st[0]:6.2f} us/workgroup, waited {st[1]:.2f}", flush=True)
    r.close()

# ---- closed loop
model = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
for g in grids:
    for scale in (workload.CLOSED_LOOP_W_SCALE,):
        env = workload.closed_loop_env(model, N)
        if g:
            env.set_stage_grid(g)
        W = torch.from_numpy(workload.closed_loop_policy(env.num_obs, env.num_acts, scale)).to(dev)
        noise = torch.from_numpy(workload.closed_loop_noise(N, 128)).to(dev)
        for pipe in (True, False):
            env.world.set_step_pipelining(pipe)
            env.reset()
            env.rollout_linear(300, W, noise=noise)
            env.world.step_pipeline_join()
            if pipe:
                env.world.debug_pipeline_wait_stats()
            v = rate(lambda k: env.rollout_linear(k, W, noise=noise, clip=0.0), env.world.step_pipeline_join, K)
            st = env.world.debug_pipeline_wait_stats() if pipe else (0, 0)
            print(f"closed loop {'pipelined' if pipe else 'lock-step'} (stage grid {g or 'default'}): {v:7.1f} M env-steps/s   wait {st[0]:6.2f} us/workgroup, waited {st[1]:.2f}   faults {env.world.step_pipeline_fault()}", flush=True)
        env.close()
