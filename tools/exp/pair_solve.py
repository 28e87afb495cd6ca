#!/usr/bin/env python3
"""Round-5 experiment, CPU only: the JOINT rule for two contacts on one link (orc_params::pair_inner) on the benchmark's own
contact problems.  Every solve of >= 8 sweeps in the config-2 population holds a knee + foot pair on one shank ("the tail");
the launch waits for the wave that holds the worst of them.  Question: does solving the pair together cut those solves by >= 50 %?
Population A of tests/test_oracle_solver_heuristics.py (reset workload, 512 envs, sampled once stationary).  Prints, per K = inner rounds:
solves with >= 8 sweeps, the sweep distribution's tail, rule evaluations spent inside pairs, and the cost of the worst solve in units of one
pass of the device's sweep loop (a pass = one rule evaluation on all contact lanes + one exchange; an inner round = two rule evaluations +
a two-lane exchange ~ 1.6 passes - profiles/r03_diag_waves.txt: pass 0.9 k cycles of which the rule 0.66 k)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from common import Oracle
from raisimlib_amd import Model, rsc_path, workload
from test_oracle_solver_heuristics import _population

m = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
samples = _population(m, N, 110, 60, reset=True)
kp, kd = (a.astype(np.float64) for a in workload.anymal_gains())
base = None
print(f"population: {len(samples)} samples x {N} envs")
for K in (0, 1, 2, 4, 8, 16):
    o = Oracle(m.blob)
    o.p.pair_inner = K
    cnt = C.c_long.in_dll(o.L, "orc_pair_evals")
    cnt.value = 0
    its, ncs, us, fl = [], [], [], []
    for q, u, pt, warm in samples:
        r = o.step_batch(q, u, 1, kp, kd, pt, np.zeros((q.shape[0], 18)), lam_warm=warm.copy(), nthreads=1)
        its.append(r["iters"]); ncs.append(r["n_contacts"]); us.append(r["u"]); fl.append(r["flags"])
    its, ncs, us, fl = np.concatenate(its), np.concatenate(ncs), np.concatenate(us), np.concatenate(fl)
    sel = ncs > 0
    it = its[sel]
    if base is None:
        base = (us, it)
    du = np.abs(us - base[0]).max(axis=1)[sel]
    hard = int((it >= 8).sum())
    # cost of a solve in passes: sweeps x ~2 passes (pair envs: group depth 2) + inner rounds x 1.6; the inner rounds are not recorded per solve,
    # so the per-solve bound uses K rounds per sweep of a pair env
    print(f"K = {K:2d}: solves {len(it)}, sweeps mean {it.mean():.2f} p99 {np.percentile(it, 99):.0f} p99.9 {np.percentile(it, 99.9):.0f} max {it.max()};  "
          f">= 8 sweeps: {hard} ({hard / base[1].__ge__(8).sum():.2f} x K=0);  >= 12: {int((it >= 12).sum())};  unconverged {int(((fl[sel] & 4) != 0).sum())};  "
          f"pair rule evaluations {cnt.value} ({cnt.value / max(len(it), 1):.2f} per solve);  |du| vs K=0 p99.9 {np.percentile(du, 99.9):.1e} max {du.max():.1e}", flush=True)
