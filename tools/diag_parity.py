"""Diagnostic (GPU): one-step and short-trajectory error of the HIP path vs the fp64 oracle, plus a quick timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raisimlib_amd import Model, BatchedWorld, rsc_path, workload
from oracle.pyoracle import Oracle

def one_step(model_name, N, lpe, seed, kmax=8, z_range=(0.3, 1.2)):
    m = Model(urdf_path=rsc_path(model_name))
    o = Oracle(m.blob); o.p.kmax = kmax
    w = BatchedWorld(m, N)
    w.set_max_contacts(kmax)
    w.set_lanes_per_env(lpe)
    gc, gv = workload.random_state(m.nq, m.nv, N, seed=seed, z_range=z_range)
    kp = np.zeros(m.nv, np.float32); kd = np.zeros(m.nv, np.float32); kp[6:] = 50; kd[6:] = 0.2
    pt = gc.copy(); pt[:, 7:] += np.random.default_rng(seed + 1).uniform(-0.3, 0.3, (N, m.nq - 7))
    dt_ = np.zeros((N, m.nv))
    w.set_pd_gains(kp, kd); w.set_pd_target(pt, dt_); w.set_state(gc, gv)
    w.integrate(1); w.synchronize()
    q1, u1 = w.get_state()
    cnt, con = w.get_contacts()
    ref = o.step_batch(gc.astype(np.float32).astype(np.float64), gv.astype(np.float32).astype(np.float64), 1,
                       kp.astype(np.float64), kd.astype(np.float64), pt.astype(np.float32).astype(np.float64), dt_, want_contacts=True)
    eq = np.abs(q1 - ref['q']).max(axis=1); eu = np.abs(u1 - ref['u']).max(axis=1)
    su = np.abs(ref['u']).max(axis=1)
    print(f"{model_name} lpe={lpe} N={N}: contacts gpu={cnt.sum()} ref={ref['n_contacts'].sum()} mismatch_envs={(cnt != ref['n_contacts']).sum()}"
          f" | max|dq|={eq.max():.3e} max|du|={eu.max():.3e} (rel {np.max(eu / (1 + su)):.3e}) median|du|={np.median(eu):.3e}"
          f" | iters gpu max={w.get_solver_iterations().max()} ref max={ref['iters'].max()} flags={np.unique(w.get_flags())}")
    worst = np.argsort(-eu)[:3]
    its = w.get_solver_iterations()
    for e in worst:
        print("   env", e, "du", eu[e], "nc", cnt[e], ref['n_contacts'][e], "iters", its[e], ref['iters'][e])
    e = worst[0]
    if eu[e] > 1e-3 and cnt[e] > 0:
        w.set_state(gc, gv); w.debug_select_env(e); w.integrate(1)
        nc, G, c, lam = w.debug_contact_problem()
        d = o.step_debug(gc[e].astype(np.float32).astype(np.float64), gv[e].astype(np.float32).astype(np.float64), kp.astype(np.float64), kd.astype(np.float64), pt[e].astype(np.float32).astype(np.float64), dt_[e])
        np.set_printoptions(precision=4, linewidth=200, suppress=False)
        print("   bodies", con[e]['body'][:nc], "cols", con[e]['collision'][:nc], "ref bodies", d['contacts']['body'])
        print("   G gpu\n", G); print("   G ref\n", d['G']); print("   c gpu", c, "\n   c ref", d['c']); print("   lam gpu", lam, "\n   lam ref", d['lam'])
    w.close()

def trajectory(N=64, control_steps=50, lpe=16):
    m = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
    o = Oracle(m.blob)
    w = BatchedWorld(m, N); w.set_lanes_per_env(lpe)
    gc, gv = workload.anymal_initial_state(N)
    kp, kd = workload.anymal_gains()
    w.set_pd_gains(kp, kd); w.set_state(gc, gv)
    q, u = gc.astype(np.float32).astype(np.float64), gv.copy()
    dt_ = np.zeros((N, 18))
    for cs in range(control_steps):
        pt = workload.anymal_targets(N, cs).astype(np.float32)
        w.set_pd_target(pt, dt_)
        w.integrate(4)
        r = o.step_batch(q, u, 4, kp.astype(np.float64), kd.astype(np.float64), pt.astype(np.float64), dt_)
        q, u = r['q'], r['u']
        if cs % 10 == 9 or cs < 3:
            q1, u1 = w.get_state()
            print(f"  control step {cs+1}: max|dq|={np.abs(q1-q).max():.3e} max|du|={np.abs(u1-u).max():.3e} median env |du|={np.median(np.abs(u1-u).max(axis=1)):.3e} contacts={w.get_contacts()[0].sum()} z_mean={q1[:,2].mean():.3f}")
    w.close()

def timing(N=4096, lpe=16, steps=200, max_iter=150, reset=False):
    m = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
    w = BatchedWorld(m, N); w.set_lanes_per_env(lpe)
    w.set_contact_solver_param(1.0, 1.0, 1.0, max_iter, 1e-5)
    feet = m.collision_indices("_foot")
    gc, gv = workload.anymal_initial_state(N)
    kp, kd = workload.anymal_gains()
    w.set_pd_gains(kp, kd); w.set_state(gc, gv)
    w.set_pd_target(workload.anymal_targets(N, 0), np.zeros((N, 18)))
    g0 = gc.astype(np.float32); v0 = gv.astype(np.float32)
    nreset = 0
    for _ in range(100):
        w.integrate(4)
        if reset: nreset += int(w.reset_terminated(feet, g0, v0).sum())
    w.synchronize()
    w.enable_timing(True)
    el = 0.0; kms = []
    for _ in range(steps):
        t = time.time(); w.integrate(4); w.synchronize(); el += time.time() - t
        kms.append(w.last_kernel_ms())
        if reset: nreset += int(w.reset_terminated(feet, g0, v0).sum())
    kms = np.array(kms)
    q1, _ = w.get_state()
    w.debug_phase_cycles(True, False); w.integrate(4); pc = w.debug_phase_cycles(True, True)
    names = ["base+down", "collide", "up+chol", "columns", "delassus", "gs", "final"]
    print("   phase cycles (wg0, last substep): " + " ".join(f"{n}={pc[i+1]-pc[i]}" for i, n in enumerate(names)) + f" total={pc[7]-pc[0]} iters={pc[8]} ncw={pc[9]}")
    print(f"timing lpe={lpe} N={N} max_iter={max_iter} reset={reset}: kernel mean {kms.mean()*1e3:.1f} us p50 {np.median(kms)*1e3:.1f} us per control step (4 substeps) -> {N*4/kms.mean()/1e3:.1f} M env-steps/s; "
          f"iters max {w.get_solver_iterations().max()} mean {w.get_solver_iterations().mean():.2f}; resets {nreset}; zmean {q1[:,2].mean():.3f} contacts/env {w.get_contacts()[0].mean():.2f}")
    w.close()

if __name__ == "__main__":
    for lpe in () if (len(sys.argv) > 1 and sys.argv[1] == "time") else (16,) if len(sys.argv) > 1 else (16, 32, 64):
        one_step("anymal_c_like.urdf", 256, lpe, 0)
    if not (len(sys.argv) > 1 and sys.argv[1] == "time"):
        one_step("anymal_c_like.urdf", 256, 16, 5, z_range=(0.2, 0.5))
        one_step("atlas_like.urdf", 128, 32, 1, kmax=16, z_range=(0.6, 1.3))
    if len(sys.argv) > 1 and sys.argv[1] == "quick": sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "time":
        for mi in (1, 5):
            timing(lpe=16, max_iter=mi, steps=50)
        timing(lpe=16, max_iter=5, reset=True, steps=50)
        timing(lpe=64, max_iter=5, reset=True, steps=50)
        sys.exit(0)
    trajectory()
    for lpe in (16, 32, 64):
        timing(lpe=lpe)
    for mi in (1, 5, 10, 30, 60):
        timing(lpe=16, max_iter=mi)
    for mi in (5, 30, 150):
        timing(lpe=16, max_iter=mi, reset=True)
