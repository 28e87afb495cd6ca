"""Small fixed workload for PMC collection: 4096 envs, reset workload, 30 control steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raisimlib_amd import Model, BatchedWorld, rsc_path, workload
N = 4096
mi = int(sys.argv[1]) if len(sys.argv) > 1 else 5
m = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
w = BatchedWorld(m, N); w.set_contact_solver_param(1, 1, 1, mi, 1e-5)
gc, gv = workload.anymal_initial_state(N); kp, kd = workload.anymal_gains()
w.set_pd_gains(kp, kd); w.set_state(gc, gv)
feet = m.collision_indices("_foot")
g0, v0 = gc.astype(np.float32), gv.astype(np.float32)
dtg = np.zeros((N, 18), np.float32)
for cs in range(60):
    w.set_pd_target(workload.anymal_targets(N, cs), dtg)
    w.integrate(4)
    w.reset_terminated(feet, g0, v0)
w.synchronize()
