"""Diagnostic (GPU): find the first env that goes non-finite in the no-reset workload and replay it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raisimlib_amd import Model, BatchedWorld, rsc_path, workload
from oracle.pyoracle import Oracle
N = 4096
m = Model(urdf_path=rsc_path("anymal_c_like.urdf")); o = Oracle(m.blob)
mi = int(sys.argv[1]) if len(sys.argv) > 1 else 150
w = BatchedWorld(m, N); w.set_contact_solver_param(1, 1, 1, mi, 1e-5); o.p.max_iter = mi
gc, gv = workload.anymal_initial_state(N); kp, kd = workload.anymal_gains()
w.set_pd_gains(kp, kd); w.set_state(gc, gv)
dtg = np.zeros((N, 18), np.float32)
for cs in range(400):
    pt = workload.anymal_targets(N, cs).astype(np.float32)
    w.set_pd_target(pt, dtg)
    q0, u0 = w.get_state()
    for sub in range(4):
        qa, ua = w.get_state()
        w.integrate(1)
        q1, u1 = w.get_state()
        bad = ~(np.isfinite(q1).all(1) & np.isfinite(u1).all(1))
        big = np.abs(u1).max(1) > 1e3
        if bad.any() or big.any():
            e = np.where(bad | big)[0][0]
            print("cs", cs, "sub", sub, "env", e, "bad", bad.sum(), "big", big.sum(), "flags", w.get_flags()[e], "iters", w.get_solver_iterations()[e])
            print(" q before", qa[e]); print(" u before", ua[e]); print(" q after", q1[e]); print(" u after", u1[e])
            cnt, con = w.get_contacts(); print(" contacts", cnt[e], con[e][:cnt[e]]["collision"], con[e][:cnt[e]]["impulse"])
            d = o.step_debug(qa[e].astype(np.float64), ua[e].astype(np.float64), kp.astype(np.float64), kd.astype(np.float64), pt[e].astype(np.float64), dtg[e].astype(np.float64))
            print(" oracle u after", d["u"], "iters", d["iters"], "cols", d["contacts"]["collision"]); print(" oracle lam", d["lam"])
            w.set_state(qa, ua); w.debug_select_env(e); w.integrate(1)
            nc, G, c, lam = w.debug_contact_problem()
            np.set_printoptions(precision=4, linewidth=220)
            print(" gpu nc", nc, "lam", lam); print(" c gpu", c); print(" c ref", d["c"])
            print(" G diff max", np.abs(G - d["G"]).max() if G.shape == d["G"].shape else "shape mismatch", "G diag", np.diag(G))
            sys.exit(0)
print("no failure in 400 control steps; zmean", q1[:, 2].mean())
