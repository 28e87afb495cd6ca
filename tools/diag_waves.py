"""Diagnostic (GPU): per-wave cycle distribution of one launch in the steady reset workload."""
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
tot = []
for cs in range(150, 170):
    w.set_pd_target(workload.anymal_targets(N, cs), dtg); w.integrate(4)
    p = w.debug_wave_profile()
    w.reset_terminated(feet, g0, v0)
    tot.append(p)
    if cs < 155:
        t, g, it, nc = p[:, 0], p[:, 1], p[:, 2], p[:, 3]
        o = np.argsort(-t)[:5]
        print(f"launch {cs}: wave cycles median {np.median(t):.0f} p90 {np.percentile(t,90):.0f} p99 {np.percentile(t,99):.0f} max {t.max()} | gs share median {np.median(g/t):.2f} of slowest {g[o[0]]/t[o[0]]:.2f} | slowest waves: cycles {t[o]} gs {g[o]} sweeps {it[o]} ncw {nc[o]}")
P = np.concatenate(tot)
t, g, it, nc = P[:, 0], P[:, 1], P[:, 2], P[:, 3]
print("all: non-GS cycles by max ncw:", {int(k): int(np.median((t - g)[nc == k])) for k in np.unique(nc)})
sw = it > 0
print("GS cycles per sweep (median) by ncw:", {int(k): int(np.median((g[sw & (nc == k)] / it[sw & (nc == k)]))) for k in np.unique(nc) if (sw & (nc == k)).any()})
print("sweeps per launch-wave: median", np.median(it), "p99", np.percentile(it, 99), "max", it.max())
