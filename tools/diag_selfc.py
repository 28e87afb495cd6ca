"""Debug aid: one self-collision of the three-link folder, device contact problem (G, c, lam) vs the oracle's."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from common import Oracle
from raisimlib_amd import BatchedWorld, Model
from test_oracle_kat import FOLDER
np.set_printoptions(precision=5, suppress=True, linewidth=200)
m = Model(urdf_string=FOLDER)
N = 64
w = BatchedWorld(m, N); o = Oracle(m.blob)
w.set_gravity([0, 0, 0]); o.p.gravity[2] = 0
kp = np.array([0] * 6 + [40.0, 40.0]); kd = np.array([0] * 6 + [2.0, 2.0])
pt = np.array([0, 0, 0, 0, 0, 0, 0, 0.3, 3.1])
q = np.array([0, 0, 1.0, 1, 0, 0, 0, 0.0, 2.0]); u = np.zeros(8)
for k in range(12):
    q0, u0 = q.copy(), u.copy()
    r = o.step_debug(q, u, kp, kd, pt, np.zeros(8))
    q, u = r["q"], r["u"]
    if len(r["contacts"]): break
print("oracle step", k, "contacts", r["contacts"]["collision"], "iters", r["iters"])
print("oracle G\n", r["G"], "\nc", r["c"], "lam", r["lam"])
w.set_pd_gains(kp, kd); w.set_pd_target(np.tile(pt, (N, 1)), np.zeros((N, 8)))
w.set_state(np.tile(q0, (N, 1)), np.tile(u0, (N, 1)))
w.debug_select_env(3)
w.integrate(1)
nc, G, c, lam = w.debug_contact_problem()
print("device nc", nc, "\nG\n", G, "\nc", c, "lam", lam)
qd, ud = w.get_state()
print("u oracle", u, "\nu device", ud[3], "\nmax diff", np.abs(ud[3] - u).max())
cnt, con = w.get_contacts()
print(con[3][:cnt[3]])
print(r["contacts"])
