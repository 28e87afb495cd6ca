#!/usr/bin/env python3
"""Round-6 experiment, CPU only, ORACLE FIRST: contact-set reduction before the solve on config 5's own contact problems.
orc_params::reduce_dist merges the two spheres of one foot edge (0.12 m apart, equal normals on flat ground) into one contact at their depth-weighted
midpoint, solves the reduced problem with the benchmark's solver settings and splits the impulses back (oracle/rsb_oracle.h).  Question: how many sweeps
does that save, and does |du| of one integrate() against the unreduced solve stay inside the humanoid tolerance 5e-3 (1 + |u|)?
Populations: the standing and the collapsing regime of bench.Recipe(5), 128 envs, sampled once stationary (tests/test_oracle_solver_heuristics.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import bench
from test_oracle_solver_heuristics import _atlas_oracle, _atlas_population

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for regime in ("standing", "collapsing"):
    recipe = bench.Recipe(5, -1.0, regime)
    m = recipe.model
    samples, resets = _atlas_population(recipe, N, 70, 30, depth=2)
    kp, kd = recipe.kp.astype(np.float64), recipe.kd.astype(np.float64)
    print(f"\nconfig 5 {regime}: {len(samples)} samples x {N} envs ({resets} resets while sampling)")
    base = None
    for dist in (0.0, 0.13, 0.30):
        o = _atlas_oracle(recipe, multi_depth=2, reduce_dist=dist)
        its, ncs, us, fl, u0 = [], [], [], [], []
        for q, u, pt, warm in samples:
            r = o.step_batch(q, u, 1, kp, kd, pt, np.zeros((q.shape[0], m.nv)), lam_warm=warm.copy(), want_contacts=True)
            its.append(r["iters"]); ncs.append(r["n_contacts"]); us.append(r["u"]); fl.append(r["flags"]); u0.append(u)
        its, ncs, us, fl, u0 = (np.concatenate(x) for x in (its, ncs, us, fl, u0))
        sel = ncs > 0
        if base is None:
            base = us
        du = np.abs(us - base)
        tol = 5e-3 * (1.0 + np.abs(base))
        viol = (du > tol).any(axis=1)[sel]
        worst = (du / tol).max(axis=1)[sel]
        it = its[sel]
        print(f"  reduce_dist {dist:4.2f}: contacts/env {ncs[sel].mean():.2f}  sweeps mean {it.mean():5.2f} p50 {np.median(it):.0f} p90 {np.percentile(it, 90):.0f} p99 {np.percentile(it, 99):.0f} max {it.max()}"
              f"  unconverged {int(((fl[sel] & 4) != 0).sum())}/{len(it)}  |  |du| vs unreduced: p50 {np.median(du.max(axis=1)[sel]):.1e} p99 {np.percentile(du.max(axis=1)[sel], 99):.1e} max {du.max():.1e};"
              f"  solves outside 5e-3 (1 + |u|): {int(viol.sum())} ({100.0 * viol.mean():.1f} %), worst = {worst.max():.1f} x the tolerance", flush=True)
