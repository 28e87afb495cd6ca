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
ns, nn, nsol, tsr = P[:, 4], P[:, 5], P[:, 6], P[:, 7]
print(f"per launch-wave medians: contact solves {np.median(nsol):.0f}, global searches {np.median(ns):.0f}, newton blocks {np.median(nn):.0f}, cycles in searches {np.median(tsr):.0f} ({np.median(tsr/np.maximum(g,1)):.2f} of GS), cycles/search {np.median(tsr/np.maximum(ns,1)):.0f}")
print(f"  (RSB_PROF_FINE=1) GS set-up cycles {np.median(P[:,8]):.0f} per launch-wave; phase (A) direction refresh {np.median(P[:,9]):.0f} = {np.median(P[:,9]/np.maximum(it,1)):.0f} per sweep; phase (C) sweep epilogue {np.median(P[:,10]):.0f} = {np.median(P[:,10]/np.maximum(it,1)):.0f} per sweep; phase (B) = rest = {np.median((g-P[:,8]-P[:,9]-P[:,10])/np.maximum(nsol,1)):.0f} per contact solve (incl. the W^T lam scatter)")
print(f"  (RSB_PROF_FINE=1, chained stamps) per pass: rule {np.median(P[:,11]/np.maximum(nsol,1)):.0f}, magnitude + dl {np.median(P[:,15]/np.maximum(nsol,1)):.0f}, exchange {np.median(P[:,12]/np.maximum(nsol,1)):.0f}; per launch-wave: solver end (W^T lam scatter) {np.median(P[:,13]):.0f}")
o = np.argsort(-t)[:20]
print(f"slowest 20 waves: solves {nsol[o].mean():.0f} searches {ns[o].mean():.0f} newton {nn[o].mean():.0f} search cycles {tsr[o].mean():.0f} of GS {g[o].mean():.0f} of total {t[o].mean():.0f}")
print(f"prologue (kernel entry -> state in LDS, not part of the wave totals above): median {np.median(P[:,14]):.0f} cycles, max {P[:,14].max()}")
print("sweeps per launch-wave: median", np.median(it), "p99", np.percentile(it, 99), "max", it.max())

# marginal costs by least squares over all sampled waves: GS cycles ~ c0 + c1*sweeps + c2*solves + c3*newton + c4*searches
A = np.stack([np.ones_like(g), it, nsol, nn, ns], 1).astype(np.float64)
coef, *_ = np.linalg.lstsq(A, g.astype(np.float64), rcond=None)
print("GS cycles fit: const %.0f + %.0f/sweep + %.0f/contact-solve + %.0f/newton block + %.0f/global search" % tuple(coef),
      "| residual rms %.0f" % np.sqrt(np.mean((A @ coef - g) ** 2)))
A2 = np.stack([np.ones_like(g), (nc >= 1) * 1.0, nc], 1).astype(np.float64)
c2, *_ = np.linalg.lstsq(A2, (t - g).astype(np.float64), rcond=None)
print("non-GS cycles fit: %.0f + %.0f (any contact) + %.0f per wave-max contact" % tuple(c2))
