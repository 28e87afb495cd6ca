"""Diagnostic (GPU): configs 2 and 3 side by side under the benchmark's recipes - phase cycles over ALL workgroups' last sub-step are not
available, so: workgroup 0's stamps (median over launches) and the per-wave totals (all waves)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raisimlib_amd import BatchedWorld
import bench
N = 4096
for config in (2, 3):
    recipe = bench.Recipe(config, -1.0)
    m = recipe.model
    w = BatchedWorld(m, N)
    recipe.setup_world(w, N, 0)
    gc, gv = recipe.initial_state(N, 0)
    w.set_state(gc, gv)
    feet = recipe.feet
    g0, v0 = gc.astype(np.float32), gv.astype(np.float32)
    dtg = np.zeros((N, m.nv), np.float32)
    for cs in range(150):
        w.set_pd_target(recipe.targets(N, cs, 0), dtg); w.integrate(4); w.reset_terminated(feet, g0, v0)
    w.debug_phase_cycles(True, False)
    rows, waves = [], []
    for cs in range(150, 200):
        w.set_pd_target(recipe.targets(N, cs, 0), dtg); w.integrate(4)
        p = w.debug_phase_cycles(True, True); waves.append(w.debug_wave_profile())
        w.reset_terminated(feet, g0, v0)
        rows.append(np.r_[np.diff(p[:8]), p[8], p[9], p[10] - p[0], p[11] - p[10], p[1] - p[11], p[12] - p[1], p[13] - p[12], p[2] - p[13], p[14] - p[2], p[3] - p[14]])
    R = np.array(rows, dtype=np.float64)
    names = ["base + down pass", "collision detection", "up pass / ABA + base factor", "contact columns + c", "Delassus G", "solver", "delta-u + integrate"]
    print("config", config, "- workgroup 0, last sub-step, median over %d launches (cycles):" % len(R))
    for i, n in enumerate(names): print(f"  {n:32s} {np.median(R[:, i]):8.0f}")
    print("  sweeps (median)", np.median(R[:, 7]), "ncw (median)", np.median(R[:, 8]), "total", np.median(R[:, :7].sum(1)))
    for i, n in enumerate(["collision: terrain", "collision: self-collision sweep", "collision: joint limits + counts"]):
        print(f"    {n:40s} {np.median(R[:, 12 + i]):8.0f}")
    P = np.concatenate(waves); t, g, it, nc, ns, nn, nsol = (P[:, i] for i in range(7))
    print(f"  waves: total cycles median {np.median(t):.0f} mean {t.mean():.0f} p99 {np.percentile(t,99):.0f} max {t.max()} | solver share median {np.median(g/t):.2f} | sweeps/launch median {np.median(it):.0f} "
          f"p99 {np.percentile(it,99):.0f} max {it.max()} | passes {np.median(nsol):.0f} newton blocks mean {nn.mean():.1f} searches mean {ns.mean():.2f} | ncw median {np.median(nc):.0f}")
    cnt, _ = w.get_contacts()
    print(f"  contacts per env {cnt.mean():.2f}")
    w.close()
