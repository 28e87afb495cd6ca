"""Diagnostic (GPU): config 5 (Atlas-like, one wave per env, kmax 16): phase cycles of workgroup 0 and per-wave totals."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raisimlib_amd import Model, BatchedWorld, rsc_path
import bench
regime = sys.argv[1] if len(sys.argv) > 1 else "standing"
N = 4096
recipe = bench.Recipe(5, -1.0, regime)          # the benchmark's config-5 recipe: gains, targets, multi-contact solver settings
m = recipe.model
w = BatchedWorld(m, N)
recipe.setup_world(w, N, 0)
if len(sys.argv) > 2:
    w.set_lanes_per_env(int(sys.argv[2]))
gc, gv = recipe.initial_state(N, 0)
w.set_state(gc, gv)
feet = recipe.feet
g0, v0 = gc.astype(np.float32), gv.astype(np.float32)
dtg = np.zeros((N, m.nv), np.float32)
class _W:   # the old script's target helper, on the recipe
    @staticmethod
    def atlas_targets(n, cs, nq):
        return recipe.targets(n, cs, 0)
workload = _W
print("config 5, regime", regime)
print("lanes per env", w.lanes_per_env(), "nb", m.nb, "ncol", m.ncol)
for cs in range(60):
    w.set_pd_target(workload.atlas_targets(N, cs, m.nq), dtg); w.integrate(4); w.reset_terminated(feet, g0, v0)
w.debug_phase_cycles(True, False)
rows, waves = [], []
for cs in range(60, 80):
    w.set_pd_target(workload.atlas_targets(N, cs, m.nq), dtg); w.integrate(4)
    p = w.debug_phase_cycles(True, True); waves.append(w.debug_wave_profile())
    w.reset_terminated(feet, g0, v0)
    rows.append(np.r_[np.diff(p[:8]), p[8], p[9]])
R = np.array(rows, dtype=np.float64)
names = ["base + down pass", "collision detection", "up pass / ABA + base factor", "contact columns + c", "Delassus G", "solver", "delta-u + integrate"]
print("workgroup 0, last sub-step, median over %d launches (cycles):" % len(R))
for i, n in enumerate(names): print(f"  {n:32s} {np.median(R[:, i]):8.0f}")
print("  sweeps (median)", np.median(R[:, 7]), "ncw (median)", np.median(R[:, 8]), "total", np.median(R[:, :7].sum(1)))
P = np.concatenate(waves); t, g, it, nc, ns, nn, nsol = (P[:, i] for i in range(7))
print(f"waves: total cycles median {np.median(t):.0f} p99 {np.percentile(t,99):.0f} max {t.max()} | solver share median {np.median(g/t):.2f} | sweeps/launch median {np.median(it):.0f} max {it.max()} | passes {np.median(nsol):.0f} newton blocks {np.median(nn):.0f} searches {np.median(ns):.0f} | ncw median {np.median(nc):.0f}")
print(f"solver cycles per pass {np.median(g/np.maximum(nsol,1)):.0f}; per sweep {np.median(g/np.maximum(it,1)):.0f}")
