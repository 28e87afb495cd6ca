"""one-step parity of config 3's population vs the oracle, with the offenders listed (debug aid for tests/test_gpu_parity.py::test_per_env_parity_on_the_height_map...)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from raisimlib_amd import BatchedWorld, workload
from oracle.pyoracle import Oracle
f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
N = 4096
recipe = bench.Recipe(3, -1.0)
model = recipe.model
w = BatchedWorld(model, N); recipe.setup_world(w, N, 0)
gc0, gv0 = recipe.initial_state(N, 0)
w.set_state(gc0, gv0); w.set_pd_target(None, np.zeros((N, model.nv), np.float32))
feet = np.asarray(recipe.feet, np.int32)
for k in range(60):
    w.set_pd_target(recipe.targets(N, k, 0).astype(np.float32), None); w.integrate(workload.SUBSTEPS); w.reset_terminated(feet, gc0, gv0)
q, u = w.get_state(); w.close()
pt = recipe.targets(N, 60, 0); gc, gv = q.astype(np.float64), u.astype(np.float64)
w = BatchedWorld(model, N); recipe.setup_world(w, N, 0)
o = Oracle(model.blob); recipe.setup_oracle(o, N, 0)
dtg = np.zeros((N, model.nv))
w.set_pd_target(pt, dtg); w.set_state(gc, gv); w.integrate(1)
q1, u1 = w.get_state(); cnt, con = w.get_contacts(); flags = w.get_flags(); iters = w.get_solver_iterations()
print("specialization status", w.specialization_status())
w.close()
ref = o.step_batch(f32(gc), f32(gv), 1, np.asarray(recipe.kp, np.float64), np.asarray(recipe.kd, np.float64), f32(pt), dtg, None, want_contacts=True, lam_warm=o.new_warm_state(N))
same = cnt == ref["n_contacts"]
eq = np.abs(q1 - ref["q"]); tol = 2e-6 + 1e-6 * np.abs(ref["q"])
bad = np.nonzero(((eq > tol).any(axis=1)) & same & ((ref["flags"] & 4) == 0))[0]
eu = np.abs(u1 - ref["u"]).max(axis=1) / (1 + np.abs(ref["u"]).max(axis=1))
print("contact lists equal", same.mean(), "offenders", len(bad), "max eq", eq[same].max(), "max rel du", eu[same].max())
for e in bad[:12]:
    j = int(np.argmax(eq[e] - tol[e]))
    print("   ref keys", [k for k in ref.keys()][:20]) if e == bad[0] else None
    for kk in ("iters", "sweeps", "n_iters"):
        if kk in ref: print("   ref", kk, ref[kk][e])
    nce = cnt[e]
    print("   dev impulses", np.array2string(con[e][:nce]["impulse"], precision=6), "\n   ref impulses", np.array2string(ref["contacts"][e][:nce]["impulse"], precision=6))
    print("   du", np.array2string(u1[e] - ref["u"][e], precision=2))
    print(f"env {e}: nc dev {cnt[e]} ref {ref['n_contacts'][e]} coord {j} |dq| {eq[e, j]:.3e} rel|du| {eu[e]:.3e} iters dev {iters[e]} flags dev {flags[e]} ref {ref['flags'][e]} collisions {list(con[e][:cnt[e]]['collision'])}")
