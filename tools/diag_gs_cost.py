"""Diagnostic (GPU): Gauss-Seidel cycles per sweep for all-stick vs all-slip contact sets (4 feet on the ground)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raisimlib_amd import Model, BatchedWorld, rsc_path, workload
N = 4096
m = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
w = BatchedWorld(m, N)
kp = np.zeros(18, np.float32); kd = np.zeros(18, np.float32); kp[6:] = 400; kd[6:] = 10
gc, gv = workload.anymal_initial_state(N, height=0.57)
gc[:, 3] = 1; gc[:, 4:7] = 0
w.set_pd_gains(kp, kd); w.set_pd_target(gc, np.zeros((N, 18), np.float32)); w.set_state(gc, gv)
for _ in range(300): w.integrate(4)
q, u = w.get_state()
print("settled: z", q[:, 2].mean(), "|u|max", np.abs(u).max(), "contacts/env", w.get_contacts()[0].mean())
w.debug_phase_cycles(True, False)
for name, vx in (("stick", 0.0), ("slip", 2.0)):
    u2 = u.copy(); u2[:, 0] += vx
    w.set_state(q, u2)
    w.integrate(1)
    p = w.debug_wave_profile()
    pc = w.debug_phase_cycles(True, True)
    it = w.get_solver_iterations()
    print(f"{name}: iters mean {it.mean():.2f} | wave total cycles median {np.median(p[:,0]):.0f}, GS cycles median {np.median(p[:,1]):.0f}, sweeps(wave max) median {np.median(p[:,2]):.0f} -> GS cycles/sweep {np.median(p[:,1]/np.maximum(p[:,2],1)):.0f}")
    names = ["base+down", "collide", "up+chol", "columns", "delassus", "gs", "final"]
    print("   wg0 phases: " + " ".join(f"{n}={pc[i+1]-pc[i]}" for i, n in enumerate(names)) + f" iters={pc[8]} ncw={pc[9]}")
