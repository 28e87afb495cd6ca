import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raisimlib_amd import Model, BatchedWorld, VecEnv, rsc_path, workload
anymal = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
N = 200
feet = anymal.collision_indices("_foot")
gc_init = workload.anymal_initial_state(1)[0][0].astype(np.float32)
kp = np.zeros(18, np.float32); kd = np.zeros(18, np.float32); kp[6:] = 50.0; kd[6:] = 0.2
env = VecEnv(anymal, N, gc_init=gc_init)
twin = BatchedWorld(anymal, N); twin.set_pd_gains(kp, kd); twin.set_state(np.tile(gc_init, (N, 1)), np.zeros((N, 18)))
rng = np.random.default_rng(5)
np.set_printoptions(precision=6, linewidth=200)
for k in range(40):
    act = rng.normal(size=(N, 12)).astype(np.float32) * (3.0 if k % 9 == 8 else 1.0)
    pt = np.zeros((N, 19), np.float32); pt[:, 3] = 1; pt[:, 7:] = gc_init[7:] + np.float32(0.3) * act
    rew, done = env.step(act)
    twin.set_pd_target(pt, np.zeros((N, 18), np.float32)); twin.integrate(4)
    d2 = twin.reset_terminated(feet, gc_init, np.zeros(18, np.float32))
    qa, ua = twin.get_state(); qe, ue = env.world.get_state()
    ob = env.observe()
    dq = np.nan_to_num(np.abs(qa - qe), nan=1e9); du = np.nan_to_num(np.abs(ua - ue), nan=1e9)
    print("k", k, "state diff", dq.max(), du.max(), "obs z-axis env0", ob[0, 1:4], "quat env0", qe[0, 3:7], "twin", qa[0, 3:7])
    if k == 2: break
