"""Workload for a PC-sampling pass (GPU): R resident launches of K control steps each, config 2 (4096 ANYmal-like envs), nothing else on the device.
usage: rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1 -d out -- python tools/exp/pcsample_run.py [config] [K] [R] [mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from raisimlib_amd import BatchedWorld, workload

config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
R = int(sys.argv[3]) if len(sys.argv) > 3 else 20
mode = sys.argv[4] if len(sys.argv) > 4 else "resident"
N = 4096
PERIOD = 128
dev = torch.device("cuda:0")
r = bench.Recipe(config, -1.0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
w = BatchedWorld(r.model, N)
w.set_stream(stream.cuda_stream)
r.setup_world(w, N, 0)
if os.environ.get("RSB_EXP_NO_SELF"): w.set_self_collision(False)      # (what the self-collision sweep costs: the benchmark's population has next to no self-collisions)
gc0, gv0 = r.initial_state(N, 0)
w.set_state(gc0, gv0)
w.set_pd_target(None, np.zeros((N, r.model.nv), np.float32))
g0 = torch.from_numpy(gc0.astype(np.float32)).to(dev); v0 = torch.from_numpy(gv0.astype(np.float32)).to(dev)
feet = np.asarray(r.feet, np.int32)
obs = torch.zeros((N, w.obs_dim(len(feet))), device=dev)
done = torch.zeros(N, dtype=torch.uint8, device=dev)
bank = torch.from_numpy(np.stack([r.targets(N, k, 0).astype(np.float32) for k in range(PERIOD)])).to(dev)
if mode == "resident":
    w.set_step_residency(True)
    assert w.residency_status(0)
fn = w.control_steps_plan(workload.SUBSTEPS, bank.data_ptr(), PERIOD, obs.data_ptr(), 0, feet, feet, g0.data_ptr(), v0.data_ptr(), N, done.data_ptr(), 0)
k = 0
import time
fn(200, k); k += 200
w.synchronize()
t0 = time.perf_counter()
for _ in range(R):
    fn(K, k); k += K
    w.synchronize()
dt = time.perf_counter() - t0
import hashlib
q_end, u_end = w.get_state()
digest = hashlib.sha1(q_end.tobytes() + u_end.tobytes() + obs.cpu().numpy().tobytes()).hexdigest()[:12]      # (same kernel semantics => same digest: an A/B of kernel variants that must not change results)
print(f"config {config} {mode} K={K} R={R}: {N * 4 * K * R / dt / 1e6:.1f} M env-steps/s  state digest {digest} sweeps mean {w.get_solver_iterations().mean():.3f} flags&16: {int(((w.get_flags() & 16) != 0).sum())} contacts/env {w.get_contacts()[0].mean():.3f} envs with > 4: {int((w.get_contacts()[0] > 4).sum())}", flush=True)
w.close()
