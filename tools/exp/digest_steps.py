"""digest of config 2's state after 1, 2, 5, 20 control steps from the initial state (lock-step launches): two kernel variants that are the same arithmetic print the same lines"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from raisimlib_amd import BatchedWorld, workload
config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = 1024
r = bench.Recipe(config, -1.0)
w = BatchedWorld(r.model, N); r.setup_world(w, N, 0)
gc0, gv0 = r.initial_state(N, 0)
w.set_state(gc0, gv0); w.set_pd_target(None, np.zeros((N, r.model.nv), np.float32))
k = 0
for upto in (1, 2, 5, 20):
    while k < upto:
        w.set_pd_target(r.targets(N, k, 0).astype(np.float32), None); w.integrate(workload.SUBSTEPS); k += 1
    q, u = w.get_state()
    print(f"after {upto:2d} control steps: digest {hashlib.sha1(q.tobytes() + u.tobytes()).hexdigest()[:12]}  -0.0 entries {int((np.signbit(q) & (q == 0)).sum() + (np.signbit(u) & (u == 0)).sum())}  sum|q| {np.abs(q).sum():.9e}", flush=True)
print("specialization", w.specialization_status())
