"""Diagnostic (GPU): height-map one-step parity - which envs disagree with the oracle and why."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from common import Oracle, f32, standing_states
from raisimlib_amd import Model, rsc_path, workload
from test_gpu_parity import run_one_step
anymal = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
H = workload.smoothed_heightmap(128, 128, amplitude=0.1, seed=7)
hm = (128, 128, workload.HEIGHTMAP_SIZE, workload.HEIGHTMAP_SIZE, 0.0, 0.0, H)
N = 1024
gc, gv = standing_states(N, seed=41, z=(0.45, 0.7))
rng = np.random.default_rng(9)
gc[:, 0:2] = rng.uniform(-6.6, 6.6, (N, 2))
o = Oracle(anymal.blob); o.set_heightmap(*hm)
gc[:, 2] += np.array([o.terrain(x, y)[0] for x, y in gc[:, 0:2]])
kp, kd = workload.anymal_gains()
pt = gc.copy(); pt[:, 7:] = workload.ANYMAL_NOMINAL_JOINTS + rng.uniform(-0.3, 0.3, (N, 12))
dev, ref, _ = run_one_step(anymal, gc, gv, pt, kp, kd, heightmap=hm)
print("contact counts equal:", np.array_equal(dev["cnt"], ref["n_contacts"]), "mismatch envs", (dev["cnt"] != ref["n_contacts"]).sum())
conv = (ref["flags"] & 4) == 0
eq = np.abs(dev["q"] - ref["q"]); bad = (eq > 2e-6 + 1e-6 * np.abs(ref["q"])).any(axis=1) & conv
eu = np.abs(dev["u"] - ref["u"]).max(axis=1)
print("envs beyond the dq tolerance:", bad.sum(), "of", conv.sum(), "| du of those", np.round(eu[bad][:10], 5))
for e in np.nonzero(bad)[0][:6]:
    n = ref["n_contacts"][e]
    print("env", e, "nc", n, "dev nc", dev["cnt"][e])
    for k in range(n):
        rc = ref["contacts"][e][k]; dc = dev["con"][e][k]
        print("   col", rc["collision"], "n_ref", np.round(rc["normal"], 5), "n_dev", np.round(dc["normal"], 5), "depth ref %.6f dev %.6f" % (rc["depth"], dc["depth"]), "pos xy", np.round(rc["position"][:2], 4))
