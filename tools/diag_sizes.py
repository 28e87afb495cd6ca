"""Diagnostic (GPU): unusual sizes - 32768 envs on one GPU, kmax 16 on the quadruped, a single env."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raisimlib_amd import Model, BatchedWorld, rsc_path, workload
m = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
kp, kd = workload.anymal_gains()
for N, kmax in ((32768, 8), (4096, 16), (1, 8), (5, 8)):
    w = BatchedWorld(m, N); w.set_max_contacts(kmax)
    gc, gv = workload.anymal_initial_state(N)
    w.set_pd_gains(kp, kd); w.set_state(gc, gv)
    dtg = np.zeros((N, 18), np.float32)
    for cs in range(60):
        w.set_pd_target(workload.anymal_targets(N, cs), dtg); w.integrate(4)
    w.synchronize(); t0 = time.perf_counter()
    for cs in range(60, 80):
        w.set_pd_target(workload.anymal_targets(N, cs), dtg); w.integrate(4)
    w.synchronize(); el = time.perf_counter() - t0
    q, u = w.get_state(); cnt, _ = w.get_contacts()
    print(f"N={N} kmax={kmax} lanes/env {w.lanes_per_env()}: finite {bool(np.isfinite(q).all() and np.isfinite(u).all())}, contacts/env {cnt.mean():.2f}, "
          f"max contacts {cnt.max()}, base height {q[:, 2].mean():.3f}, {N * 4 * 20 / el / 1e6:.1f}M env-steps/s (host targets every step)")
    w.close()
