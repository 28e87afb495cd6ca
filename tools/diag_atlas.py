import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from raisimlib_amd import Model, BatchedWorld, rsc_path
from oracle.pyoracle import Oracle
atlas = Model(urdf_path=rsc_path("atlas_like.urdf"))
f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
rng = np.random.default_rng(4)
N = 128
g = np.zeros((N, 37)); g[:, 2] = rng.uniform(0.88, 0.95, N); g[:, 3] = 1.0; g[:, 7:] = rng.uniform(-0.1, 0.1, (N, 30))
v = np.zeros((N, 36))
kp = np.zeros(36, np.float32); kd = np.zeros(36, np.float32); kp[6:] = 200.0; kd[6:] = 5.0
for warm in (False, True):
    w = BatchedWorld(atlas, N); w.set_max_contacts(16); w.set_solver_warm_start(warm)
    o = Oracle(atlas.blob); o.p.kmax = 16; o.p.warm_start = int(warm)
    dtg = np.zeros((N, 36))
    w.set_pd_gains(kp, kd); w.set_pd_target(g, dtg); w.set_state(g, v)
    q, u, ws = f32(g), f32(v), o.new_warm_state(N)
    for cs in range(24):
        w.integrate(1)
        r = o.step_batch(q, u, 1, kp.astype(np.float64), kd.astype(np.float64), f32(g), dtg, lam_warm=ws)
        q, u = r["q"], r["u"]
        q1, u1 = w.get_state()
        eq = np.abs(q1 - q).max(1); eu = np.abs(u1 - u).max(1)
        if cs % 3 == 2 or cs < 3:
            print("warm", warm, "sub", cs, "median dq %.2e p90 %.2e max %.2e | umax oracle %.2f dev %.2f | iters o %.1f d %.1f flags4 o %d d %d cnt %d/%d" % (
                np.median(eq), np.percentile(eq, 90), eq.max(), np.abs(u).max(), np.abs(u1).max(), r["iters"].mean(), w.get_solver_iterations().mean(),
                ((r["flags"] & 4) != 0).sum(), ((w.get_flags() & 4) != 0).sum(), r["n_contacts"].sum(), w.get_contacts()[0].sum()))
    w.close()
