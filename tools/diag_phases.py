"""Diagnostic (GPU): shader-cycle stamps at the phase boundaries of workgroup 0 (last sub-step of a launch),
averaged over launches of the steady reset workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raisimlib_amd import Model, BatchedWorld, rsc_path, workload
N = 4096
m = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
w = BatchedWorld(m, N)
gc, gv = workload.anymal_initial_state(N); kp, kd = workload.anymal_gains()
w.set_pd_gains(kp, kd); w.set_state(gc, gv)
feet = m.collision_indices("_foot"); g0, v0 = gc.astype(np.float32), gv.astype(np.float32)
dtg = np.zeros((N, 18), np.float32)
for cs in range(150):
    w.set_pd_target(workload.anymal_targets(N, cs), dtg); w.integrate(4); w.reset_terminated(feet, g0, v0)
w.debug_phase_cycles(True, False)
rows = []
for cs in range(150, 200):
    w.set_pd_target(workload.anymal_targets(N, cs), dtg); w.integrate(4)
    p = w.debug_phase_cycles(True, True)
    w.reset_terminated(feet, g0, v0)
    rows.append(np.r_[np.diff(p[:8]), p[8], p[9], p[10] - p[0], p[11] - p[10], p[1] - p[11], p[12] - p[1], p[13] - p[12], p[2] - p[13], p[14] - p[2], p[3] - p[14]])
R = np.array(rows, dtype=np.float64)
names = ["base + down pass (0->1)", "collision detection (1->2)", "up pass / ABA + base factor (2->3)",
         "contact columns + c (3->4)", "Delassus G (4->5)", "Gauss-Seidel (5->6)", "delta-u + integrate (6->7)"]
print("workgroup 0, last sub-step, median over %d launches (cycles):" % len(R))
for i, n in enumerate(names):
    print(f"  {n:36s} {np.median(R[:, i]):8.0f}")
print("  sweeps (median)", np.median(R[:, 7]), "ncw (median)", np.median(R[:, 8]), "total", np.median(R[:, :7].sum(1)))
sub = ["down: base body + joint transforms", "down: level loop (pose, S, V, A)", "down: rigid inertia, bias force, actuation",
       "collision: terrain", "collision: self-collision sweep", "collision: joint limits + counts",
       "up: level loop (articulated inertias)", "up: base gather + Cholesky + W_b"]
for i, n in enumerate(sub):
    print(f"    {n:44s} {np.median(R[:, 9 + i]):8.0f}")
