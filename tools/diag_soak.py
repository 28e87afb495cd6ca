"""Diagnostic (GPU): long run of the reset workload counting non-finite states, overflowed contact sets and
non-converged solves (flags are read before each reset)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raisimlib_amd import Model, BatchedWorld, rsc_path, workload
N, STEPS = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 3000
m = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
w = BatchedWorld(m, N)
gc, gv = workload.anymal_initial_state(N); kp, kd = workload.anymal_gains()
w.set_pd_gains(kp, kd); w.set_state(gc, gv)
feet = m.collision_indices("_foot"); g0, v0 = gc.astype(np.float32), gv.astype(np.float32)
dtg = np.zeros((N, 18), np.float32)
bank = [workload.anymal_targets(N, k).astype(np.float32) for k in range(64)]
nonfinite = overflow = notconv = resets = 0; umax = 0.0; itmax = 0
for cs in range(STEPS):
    w.set_pd_target(bank[cs % 64], dtg); w.integrate(4)
    fl = w.get_flags()
    nonfinite += int(((fl & 2) != 0).sum()); overflow += int(((fl & 1) != 0).sum()); notconv += int(((fl & 4) != 0).sum())
    if cs % 50 == 0:
        q, u = w.get_state(); umax = max(umax, float(np.abs(u[np.isfinite(u).all(1)]).max())); itmax = max(itmax, int(w.get_solver_iterations().max()))
    resets += int(w.reset_terminated(feet, g0, v0).sum())
print(f"{STEPS} control steps x {N} envs = {STEPS * N * 4 / 1e6:.0f}M env-steps: non-finite states {nonfinite}, contact overflows {overflow}, "
      f"last-sub-step solves not converged {notconv} ({100.0 * notconv / (STEPS * N):.3f} %), resets {resets}, max |u| sampled {umax:.1f}, max sweeps sampled {itmax}")
