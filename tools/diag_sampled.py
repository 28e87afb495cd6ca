"""diagnostic: which contacts differ between device and oracle for the sampled-collider ANYmal on the rough map"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from raisimlib_amd import Model, rsc_path, workload
from test_gpu_parity import run_one_step, standing_states
for spacing in (0.0, 0.1):
    m = Model(urdf_path=rsc_path("anymal_c_like.urdf"), sample_spacing=spacing)
    H = workload.smoothed_heightmap(64, 64, amplitude=0.2, seed=11)
    hm = (64, 64, 6.4, 6.4, 0.0, 0.0, H)
    gc, gv = standing_states(512, seed=41, z=(0.2, 0.55))
    kp, kd = workload.anymal_gains()
    dev, ref, o = run_one_step(m, gc, gv, gc, kp, kd, heightmap=hm, kmax=16)
    names = m.collision_names()
    diff = np.nonzero(dev["cnt"] != ref["n_contacts"])[0]
    print("spacing", spacing, "ncol", m.ncol, "envs with different counts:", len(diff), "flags dev", np.bincount(dev["flags"] & 1), "ref", np.bincount(ref["flags"] & 1))
    for e in diff[:8]:
        dset = {int(c) for c in dev["con"][e][:dev["cnt"][e]]["collision"]}
        rset = {int(c) for c in ref["contacts"][e][:ref["n_contacts"][e]]["collision"]}
        print(" env", e, "dev", dev["cnt"][e], "ref", ref["n_contacts"][e], "only dev", [(names[c & 0xffff] if c < 0x10000 else hex(c)) for c in dset - rset],
              "only ref", [(names[c & 0xffff] if c < 0x10000 else hex(c)) for c in rset - dset])
        for c in ref["contacts"][e][:ref["n_contacts"][e]]:
            if int(c["collision"]) in rset - dset: print("    ref-only:", c["collision"], "depth", c["depth"], "pos", c["position"], "n", c["normal"])
        for c in dev["con"][e][:dev["cnt"][e]]:
            if int(c["collision"]) in dset - rset: print("    dev-only:", c["collision"], "depth", c["depth"], "pos", c["position"], "n", c["normal"])
