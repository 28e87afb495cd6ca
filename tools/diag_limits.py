import os, sys, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from raisimlib_amd import Model, BatchedWorld, rsc_path, workload
from common import standing_states
from test_gpu_parity import run_one_step
txt = open(rsc_path("anymal_c_like.urdf")).read()
tight = re.sub(r'(<joint name="[A-Z]{2}_KFE".*?)lower="-6.28" upper="6.28"', r'\1lower="-1.0" upper="1.0"', txt, flags=re.S)
tight = re.sub(r'(<joint name="[A-Z]{2}_HFE".*?)lower="-6.28" upper="6.28"', r'\1lower="-0.5" upper="0.5"', tight, flags=re.S)
model = Model(urdf_string=tight)
gc, gv = standing_states(384, seed=77, z=(0.45, 0.9), vel=2.0)
gc[:, 7:] += np.random.default_rng(2).uniform(-0.5, 0.5, (384, 12))
kp, kd = workload.anymal_gains()
dev, ref, o = run_one_step(model, gc, gv, gc, kp, kd)
conv = (ref["flags"] & 4) == 0
eq = np.abs(dev["q"] - ref["q"]).max(1); eu = np.abs(dev["u"] - ref["u"]).max(1) / (1 + np.abs(ref["u"]).max(1))
print("conv frac", conv.mean(), "cnt equal", np.array_equal(dev["cnt"], ref["n_contacts"]))
print("eq: median %.2e p99 %.2e max %.2e | eu rel: median %.2e p99 %.2e max %.2e" % (np.median(eq[conv]), np.percentile(eq[conv], 99), eq[conv].max(), np.median(eu[conv]), np.percentile(eu[conv], 99), eu[conv].max()))
lo = np.array([model.blob.q_lower[i] for i in range(1, 13)]); hi = np.array([model.blob.q_upper[i] for i in range(1, 13)])
viol = ((gc[:, 7:] > hi) | (gc[:, 7:] < lo)).sum(1)
for v in range(0, 6):
    m = conv & (viol == v)
    if m.any(): print(" violations", v, "n", m.sum(), "eu max %.2e eq max %.2e" % (eu[m].max(), eq[m].max()), "iters dev/ref", dev["iters"][m].mean(), ref["iters"][m].mean())
w = np.argmax(np.where(conv, eu, 0)); print("worst env", w, "viol", viol[w], "contacts", ref["n_contacts"][w], "iters", dev["iters"][w], ref["iters"][w], "flags", dev["flags"][w], ref["flags"][w])
print(" u dev", dev["u"][w]); print(" u ref", ref["u"][w])
