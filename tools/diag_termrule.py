"""Diagnostic (GPU): how often do the two termination rules differ?  upstream: a non-foot contact in the LAST sub-step of a
control step; early termination: a non-foot contact in ANY sub-step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raisimlib_amd import Model, BatchedWorld, rsc_path, workload
N = 4096
m = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
w = BatchedWorld(m, N)
gc, gv = workload.anymal_initial_state(N); kp, kd = workload.anymal_gains()
w.set_pd_gains(kp, kd); w.set_state(gc, gv)
feet = np.zeros(m.ncol, bool); feet[m.collision_indices("_foot")] = True
g0, v0 = gc.astype(np.float32), gv.astype(np.float32)
dtg = np.zeros((N, 18), np.float32)
tot_last = tot_any = tot_only_early = 0
for cs in range(300):
    w.set_pd_target(workload.anymal_targets(N, cs), dtg)
    anyill = np.zeros(N, bool)
    for sub in range(4):
        w.integrate(1)
        cnt, con = w.get_contacts()
        valid = np.arange(con.shape[1])[None, :] < cnt[:, None]
        ill = (valid & ~(feet[con["collision"] & 0xffff] & (con["collision"] < 0x10000))).any(1)
        anyill |= ill
    if cs >= 100:
        tot_last += int(ill.sum()); tot_any += int(anyill.sum()); tot_only_early += int((anyill & ~ill).sum())
    w.reset_terminated(m.collision_indices("_foot"), g0, v0)
print(f"200 control steps x {N} envs: terminated by upstream's rule {tot_last}, by the early rule {tot_any}, "
      f"only by the early rule {tot_only_early} ({100.0 * tot_only_early / max(tot_any, 1):.2f} % of early-rule terminations)")
