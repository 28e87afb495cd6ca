#!/usr/bin/env python3
"""Freeze oracle outputs as golden vectors (run from the repo root: python tests/golden/make_golden.py).

There are no golden vectors in /root/reference — these fixtures pin the ORACLE ITSELF so that
a later edit of oracle/rsb_oracle.c cannot silently change the numbers every parity test is judged against.
Contents (anymal_golden.npz): 24 seeded ANYmal states -> q+, u+ after one integrate(); their mass matrices and
nonlinearities; one 100-step trajectory of env 0 of the config-2 workload; the Delassus problem of 4 contact states.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from common import Oracle, f32, standing_states  # noqa: E402
from raisimlib_amd import Model, rsc_path, workload  # noqa: E402


def main():
    m = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
    o = Oracle(m.blob)
    gc, gv = standing_states(24, seed=2024)
    gc, gv = f32(gc), f32(gv)
    kp, kd = workload.anymal_gains()
    kp, kd = kp.astype(np.float64), kd.astype(np.float64)
    pt = gc.copy()
    pt[:, 7:] = f32(workload.ANYMAL_NOMINAL_JOINTS + 0.1)
    r = o.step_batch(gc, gv, 1, kp, kd, pt, np.zeros((24, 18)), want_contacts=True)
    M = np.stack([o.mass_matrix(q) for q in gc])
    h = np.stack([o.nonlinearities(q, u) for q, u in zip(gc, gv)])
    g0, v0 = workload.anymal_initial_state(1)
    q, u = f32(g0), v0.copy()
    traj = []
    for cs in range(25):
        ptt = f32(workload.anymal_targets(1, cs))
        rr = o.step_batch(q, u, 4, kp, kd, ptt, np.zeros((1, 18)))
        q, u = rr["q"], rr["u"]
        traj.append(np.r_[q[0], u[0]])
    probs = []
    for e in range(24):
        d = o.step_debug(gc[e], gv[e], kp, kd, pt[e], np.zeros(18))
        if 2 <= len(d["c"]) // 3 <= 4 and len(probs) < 4:
            G = np.zeros((12, 12)); c = np.zeros(12); lam = np.zeros(12)
            n3 = len(d["c"])
            G[:n3, :n3] = d["G"]; c[:n3] = d["c"]; lam[:n3] = d["lam"]
            probs.append((e, n3, G, c, lam))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "anymal_golden.npz"),
                        gc=gc, gv=gv, pt=pt, q1=r["q"], u1=r["u"], n_contacts=r["n_contacts"], iters=r["iters"], flags=r["flags"],
                        M=M, h=h, traj=np.array(traj),
                        prob_env=np.array([p[0] for p in probs]), prob_n3=np.array([p[1] for p in probs]),
                        prob_G=np.array([p[2] for p in probs]), prob_c=np.array([p[3] for p in probs]),
                        prob_lam=np.array([p[4] for p in probs]))
    print("wrote anymal_golden.npz:", r["n_contacts"].sum(), "contacts,", len(probs), "contact problems")


if __name__ == "__main__":
    main()
