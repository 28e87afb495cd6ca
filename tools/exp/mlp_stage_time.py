#!/usr/bin/env python3
"""Round-5 experiment: what ONE pass of the MLP stage costs when nothing competes with it.  Lock-step closed-loop runs (pass, step, pass, ...) of K
steps at N = 4 .. 4096 envs with the linear stage and with the MLP stage (34-128-128-12): the difference of the times per step is the difference of
the two stages' pass times (actions clipped to ~0 in both, so that the two runs step the same states).  16 blocks or 1024: the same + 20 us - one
wave's time for one block, not a shared resource.  Without its weight loads (experiment) the MLP pass still costs + 14-16 us: tools/ubench/mlp_inner.hip."""
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from raisimlib_amd import Model, rsc_path, workload

dev = torch.device("cuda:0")
model = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
K = 300
for n in (64, 256, 1024, 4096):
    env = workload.closed_loop_env(model, n)
    env.world.set_step_pipelining(False)
    W = torch.from_numpy(workload.closed_loop_policy(env.num_obs, env.num_acts)).to(dev)
    mlp = [(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)) for a, b in workload.closed_loop_mlp(env.num_obs, env.num_acts)]
    noise = torch.from_numpy(workload.closed_loop_noise(n, 16)).to(dev)
    t = {}
    for kind, run in (("linear", lambda k: env.rollout_linear(k, W, noise=noise, clip=1e-30)), ("mlp", lambda k: env.rollout_mlp(k, mlp, noise=noise, clip=1e-30))):
        env.reset(); run(50); env.world.synchronize()
        t0 = time.perf_counter(); run(K); env.world.synchronize()
        t[kind] = (time.perf_counter() - t0) / K * 1e6
    print(f"N = {n:5d} ({(n + 3) // 4:4d} blocks): lock-step control step {t['linear']:7.1f} us with the linear stage, {t['mlp']:7.1f} us with the MLP stage: + {t['mlp'] - t['linear']:5.1f} us", flush=True)
    env.close()
