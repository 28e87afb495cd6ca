"""Diagnostic (GPU): are the slowest waves of a launch the ones that hold an env about to be terminated?"""
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
for cs in range(150, 190):
    w.set_pd_target(workload.anymal_targets(N, cs), dtg); w.integrate(4)
    p = w.debug_wave_profile()
    cnt = w.get_contacts()[0]
    done = w.reset_terminated(feet, g0, v0)
    t = p[:, 0]
    wave_done = done.reshape(-1, 4).max(1)
    wave_cnt = cnt.reshape(-1, 4).max(1)
    o = np.argsort(-t)
    rows.append((t.max(), np.median(t), t[wave_done == 0].max(), wave_done[o[:10]].mean(), int(done.sum()), int(wave_done.sum()),
                 np.median(t[wave_done == 1]) if wave_done.any() else 0, np.percentile(t[wave_done == 0], 99)))
R = np.array(rows)
print("per launch (medians over %d launches):" % len(R))
print("  slowest wave %.0f cycles, median wave %.0f, slowest wave WITHOUT a terminating env %.0f, p99 of those %.0f" % (np.median(R[:, 0]), np.median(R[:, 1]), np.median(R[:, 2]), np.median(R[:, 7])))
print("  fraction of the 10 slowest waves that hold a terminating env: %.2f; terminating envs per launch %.1f (in %.1f waves); median cycles of such waves %.0f" % (R[:, 3].mean(), R[:, 4].mean(), R[:, 5].mean(), np.median(R[:, 6])))
