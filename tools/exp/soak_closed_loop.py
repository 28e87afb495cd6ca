#!/usr/bin/env python3
"""Soak of the closed loop (round 5): 20 000 control steps with the linear stage and 10 000 with the MLP stage (34-128-128-12), N = 4096, in runs of 500,
pipelined and in lock-step side by side: state, contact lists and the env task's buffers compared bit for bit after every run, rates, pipeline faults."""
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from raisimlib_amd import Model, rsc_path, workload

dev = torch.device("cuda:0")
model = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
N, RUN = 4096, 500
W = torch.from_numpy(workload.closed_loop_policy(34, 12)).to(dev)
mlp = [(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)) for a, b in workload.closed_loop_mlp(34, 12)]
noise = torch.from_numpy(workload.closed_loop_noise(N, 128)).to(dev)
for kind, total in (("linear stage", 20000), ("MLP stage 34-128-128-12", 10000)):
    envs = {}
    for pipe in (True, False):
        e = workload.closed_loop_env(model, N)
        assert e.world.set_step_pipelining(pipe) == pipe
        envs[pipe] = e
    t = {True: 0.0, False: 0.0}
    resets = 0
    for r in range(total // RUN):
        done = {}
        for pipe in (True, False):
            e = envs[pipe]
            ro = {"done": torch.zeros((RUN, N), dtype=torch.uint8, device=dev)}
            e.world.synchronize()
            t0 = time.perf_counter()
            if kind.startswith("linear"):
                e.rollout_linear(RUN, W, noise=noise, rollout=ro)
            else:
                e.rollout_mlp(RUN, mlp, noise=noise, rollout=ro)
            e.world.step_pipeline_join(); e.world.synchronize()
            t[pipe] += time.perf_counter() - t0
            done[pipe] = ro["done"]
        assert torch.equal(done[True], done[False]), (kind, r)
        resets += int(done[True].sum().item())
        qa, ua = envs[True].world.get_state(); qb, ub = envs[False].world.get_state()
        ca, la = envs[True].world.get_contacts(); cb, lb = envs[False].world.get_contacts()
        assert np.array_equal(qa, qb) and np.array_equal(ua, ub) and np.array_equal(ca, cb) and la.tobytes() == lb.tobytes(), (kind, r)
        assert np.isfinite(qa).all()
    f = envs[True].world.step_pipeline_fault()
    print(f"{kind}: {total} control steps x {N} envs in runs of {RUN}: pipelined {N * 4 * total / t[True] / 1e6:.1f} M env-steps/s, lock-step {N * 4 * total / t[False] / 1e6:.1f} M; "
          f"state, contact lists and done flags equal after every run; {resets} resets ({resets / total:.1f} per control step); pipeline faults {f}", flush=True)
    for e in envs.values():
        e.close()
