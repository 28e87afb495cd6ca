"""Debug aid: error distribution of the self-collision parity populations (tests/test_gpu_parity.py)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from raisimlib_amd import Model, rsc_path, workload
import test_gpu_parity as T
m = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
for lpe, z in [(16, (2.0, 2.1)), (32, (2.0, 2.1)), (16, (0.25, 0.5)), (64, (0.25, 0.5))]:
    gc, gv = T._contorted_states(512, 300 + lpe, z)
    kp, kd = workload.anymal_gains()
    dev, ref, o = T.run_one_step(m, gc, gv, gc, kp, kd, lpe=lpe)
    conv = ((ref["flags"] | dev["flags"]) & 4) == 0
    eu = np.abs(dev["u"] - ref["u"]).max(axis=1) / (1 + np.abs(ref["u"]).max(axis=1))
    eq = (np.abs(dev["q"] - ref["q"]) / (2e-6 + 1e-6 * np.abs(ref["q"]))).max(axis=1)
    rc = ref["contacts"]
    imp = np.array([np.abs(dev["con"][e][:ref["n_contacts"][e]]["impulse"] - rc[e][:ref["n_contacts"][e]]["impulse"]).max(initial=0) for e in range(512)])
    nself = np.array([((rc[e][:ref["n_contacts"][e]]["collision"] & 0x10000) != 0).sum() for e in range(512)])
    di = np.abs(dev["iters"] - ref["iters"])
    print(lpe, z, "conv", conv.mean(), "eu pct 50/90/98/99/100", np.percentile(eu[conv], [50, 90, 98, 99, 100]), "eq", np.percentile(eq[conv], [50, 98, 100]),
          "imp", np.percentile(imp[conv], [50, 90, 98, 100]), "di", np.percentile(di[conv], [50, 98, 100]))
    bad = np.nonzero(conv & (eu > 5e-4))[0]
    print("  bad envs", bad[:10], "nself", nself[bad[:10]], "nc", ref["n_contacts"][bad[:10]], "iters ref/dev", ref["iters"][bad[:10]], dev["iters"][bad[:10]])
