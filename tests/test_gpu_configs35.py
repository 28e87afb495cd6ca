"""Configs 3 and 5 pinned the way config 2 is (tests/test_gpu_properties.py, tests/test_gpu_parity.py): the benchmark's own
recipes (bench.Recipe: terrain, gains, targets, solver settings) at N = 4096, device vs fp64 oracle.

  * population statistics over the benchmark's control steps with the reset rule: resets, base-height quantiles, contacts,
    sweeps, share of solves that end without meeting the convergence test;
  * parity from the benchmark's STATIONARY states (sampled on the device after a pre-roll), one integrate() and three control
    steps - not from hand-made standing poses.
Individual trajectories of a contact-rich system diverge (fp32 vs fp64 round differently at every stick / slip decision), so
multi-step comparisons state medians and upper quantiles AND a bound on the worst env."""
import numpy as np
import pytest

import bench
from common import Oracle, f32
from raisimlib_amd import BatchedWorld, workload

pytestmark = pytest.mark.gpu


def _device_world(recipe, n):
    import torch
    w = BatchedWorld(recipe.model, n)
    w.set_stream(torch.cuda.current_stream().cuda_stream)
    recipe.setup_world(w, n, 0)
    return w


def _oracle(recipe, n):
    o = Oracle(recipe.model.blob)
    recipe.setup_oracle(o, n, 0)
    return o


def _terminated(recipe, r):
    feet_set = np.zeros(recipe.model.ncol, bool)
    feet_set[recipe.feet] = True
    con, ncs = r["contacts"], r["n_contacts"]
    valid = np.arange(con.shape[1])[None, :] < ncs[:, None]
    return (valid & ~(feet_set[con["collision"] & 0xffff] & (con["collision"] < 0x10000))).any(axis=1) | (r["flags"] & 2).astype(bool)


def _run_population(recipe, n, steps):
    """device (fused control step with resets) and oracle side by side from identical initial states and targets"""
    import torch
    model, feet = recipe.model, np.asarray(recipe.feet, np.int32)
    nv = model.nv
    gc0, gv0 = recipe.initial_state(n, 0)
    gc0 = f32(gc0)
    dev = torch.device("cuda")
    w = _device_world(recipe, n)
    w.set_state(gc0, gv0); w.set_pd_target(None, np.zeros((n, nv), np.float32))
    done_d = torch.zeros(n, dtype=torch.uint8, device=dev); w.set_done_output(done_d.data_ptr())
    obs = torch.empty((n, w.obs_dim(len(feet))), device=dev)
    g0d = torch.from_numpy(gc0.astype(np.float32)).to(dev); v0d = torch.from_numpy(gv0.astype(np.float32)).to(dev)
    step = w.control_step_plan(workload.SUBSTEPS, obs.data_ptr(), feet, feet, g0d.data_ptr(), v0d.data_ptr(), n)
    o = _oracle(recipe, n)
    kp, kd = recipe.kp.astype(np.float64), recipe.kd.astype(np.float64)
    q, u, warm = gc0.copy(), gv0.copy(), o.new_warm_state(n)
    out = {k: [] for k in ("dev_resets", "orc_resets", "dev_iters", "orc_iters", "dev_unconv", "orc_unconv")}
    for cs in range(steps):
        pt = f32(recipe.targets(n, cs, 0))
        ptd = torch.from_numpy(pt.astype(np.float32)).to(dev)
        step(ptd.data_ptr())
        torch.cuda.synchronize()
        out["dev_resets"].append(int(done_d.sum().item()))
        out["dev_iters"].append(w.get_solver_iterations().mean())
        fl_d = w.get_flags()
        out["dev_unconv"].append(((fl_d & 4) != 0).mean())
        r = o.step_batch(q, u, workload.SUBSTEPS, kp, kd, pt, np.zeros((n, nv)), want_contacts=True, lam_warm=warm)
        q, u = r["q"], r["u"]
        term = _terminated(recipe, r)
        out["orc_resets"].append(int(term.sum())); out["orc_iters"].append(r["iters"].mean())
        out["orc_unconv"].append(((r["flags"] & 4) != 0)[~term].mean())     # (flags of the last sub-step... of all sub-steps OR-ed: an upper bound)
        q[term], u[term], warm[term] = gc0[term], gv0[term], 0.0
    qd, ud = w.get_state()
    cnt_d, _ = w.get_contacts()
    w.close()
    out = {k: np.array(v) for k, v in out.items()}
    out.update(qd=qd, ud=ud, cnt_d=cnt_d, q=q, u=u, cnt_o=np.where(term, 0, r["n_contacts"]))
    return out


def test_population_statistics_config3():
    """4096 ANYmal-like envs spread over the benchmark's 128 x 128 height map (closest-feature narrow phase), 100 control steps"""
    recipe = bench.Recipe(3, -1.0)
    s = _run_population(recipe, 4096, 100)
    print(f"config 3: resets device {s['dev_resets'].sum()} oracle {s['orc_resets'].sum()}; last-20 mean {s['dev_resets'][-20:].mean():.1f} vs "
          f"{s['orc_resets'][-20:].mean():.1f}; sweeps {s['dev_iters'][-20:].mean():.2f} vs {s['orc_iters'][-20:].mean():.2f}; contacts "
          f"{s['cnt_d'].mean():.2f} vs {s['cnt_o'].mean():.2f}")
    assert np.isfinite(s["qd"]).all() and np.isfinite(s["ud"]).all()
    assert np.abs(s["qd"][:, :2]).max() > 5.0                                    # the envs really live all over the map
    assert s["orc_resets"].sum() > 1500                                          # rough terrain + random targets: robots fall
    assert abs(s["dev_resets"].sum() - s["orc_resets"].sum()) <= 0.03 * s["orc_resets"].sum()
    assert np.abs(s["dev_resets"][:20] - s["orc_resets"][:20]).max() <= 4        # same trajectories at first: step by step
    for pct in (5, 25, 50, 75, 95):
        assert abs(np.percentile(s["qd"][:, 2], pct) - np.percentile(s["q"][:, 2], pct)) < 6e-3, pct
    assert abs(s["cnt_d"].mean() - s["cnt_o"].mean()) < 0.08
    assert abs(s["dev_iters"][-20:].mean() - s["orc_iters"][-20:].mean()) < 0.12


def test_population_statistics_config3_one_map_per_env():
    """the variant SURVEY.md 8d names "to stress gathers": every env on its own 128 x 128 map (512 envs = 32 MB of maps), 60 control steps"""
    recipe = bench.Recipe(3, -1.0, per_env_maps=True)
    s = _run_population(recipe, 512, 60)
    print(f"config 3, one map per env: resets device {s['dev_resets'].sum()} oracle {s['orc_resets'].sum()}; contacts {s['cnt_d'].mean():.2f} vs {s['cnt_o'].mean():.2f}")
    assert np.isfinite(s["qd"]).all() and s["orc_resets"].sum() > 150
    assert abs(s["dev_resets"].sum() - s["orc_resets"].sum()) <= 0.08 * s["orc_resets"].sum() + 3
    assert np.abs(s["dev_resets"][:15] - s["orc_resets"][:15]).max() <= 3
    for pct in (25, 50, 75):
        assert abs(np.percentile(s["qd"][:, 2], pct) - np.percentile(s["q"][:, 2], pct)) < 1.5e-2, pct
    assert abs(s["cnt_d"].mean() - s["cnt_o"].mean()) < 0.2


@pytest.mark.parametrize("regime,steps", [("standing", 60), ("collapsing", 60)])
def test_population_statistics_config5(regime, steps):
    """4096 Atlas-like envs under the benchmark's config-5 recipe (kmax 16, self-collision on, multi-contact solver settings)"""
    recipe = bench.Recipe(5, -1.0, regime)
    s = _run_population(recipe, 4096, steps)
    print(f"config 5 {regime}: resets device {s['dev_resets'].sum()} oracle {s['orc_resets'].sum()}; sweeps {s['dev_iters'][-20:].mean():.2f} vs "
          f"{s['orc_iters'][-20:].mean():.2f}; contacts {s['cnt_d'].mean():.2f} vs {s['cnt_o'].mean():.2f}; unconverged (last sub-step / any sub-step) "
          f"{100 * s['dev_unconv'][-20:].mean():.1f} % vs {100 * s['orc_unconv'][-20:].mean():.1f} %; base height p5/p50/p95 "
          f"{np.percentile(s['qd'][:, 2], [5, 50, 95]).round(4)} vs {np.percentile(s['q'][:, 2], [5, 50, 95]).round(4)}")
    assert np.isfinite(s["qd"]).all() and np.isfinite(s["ud"]).all()
    if regime == "standing":
        assert s["orc_resets"].sum() == 0 and s["dev_resets"].sum() == 0          # nobody falls: the regime SURVEY.md 8d names
        assert np.percentile(s["qd"][:, 2], 1) > 0.93 and s["cnt_d"].mean() > 3.0
        for pct in (5, 50, 95):
            assert abs(np.percentile(s["qd"][:, 2], pct) - np.percentile(s["q"][:, 2], pct)) < 1e-3, pct
    else:
        assert s["orc_resets"].sum() > 1500
        assert abs(s["dev_resets"].sum() - s["orc_resets"].sum()) <= 0.05 * s["orc_resets"].sum()
        for pct in (25, 50, 75):
            assert abs(np.percentile(s["qd"][:, 2], pct) - np.percentile(s["q"][:, 2], pct)) < 1.5e-2, pct
    assert abs(s["cnt_d"].mean() - s["cnt_o"].mean()) < 0.25
    assert abs(s["dev_iters"][-20:].mean() - s["orc_iters"][-20:].mean()) < 0.1 * s["orc_iters"][-20:].mean() + 0.3
    assert s["dev_unconv"][-20:].mean() <= s["orc_unconv"][-20:].mean() + 0.03   # the device does not give up on more solves than the oracle


def _stationary_states(recipe, n, preroll):
    """(q, u) of the device population after `preroll` control steps of the benchmark's loop (resets included)"""
    import torch
    feet = np.asarray(recipe.feet, np.int32)
    gc0, gv0 = recipe.initial_state(n, 0)
    dev = torch.device("cuda")
    w = _device_world(recipe, n)
    w.set_state(gc0, gv0); w.set_pd_target(None, np.zeros((n, recipe.model.nv), np.float32))
    obs = torch.empty((n, w.obs_dim(len(feet))), device=dev)
    g0d = torch.from_numpy(gc0.astype(np.float32)).to(dev); v0d = torch.from_numpy(gv0.astype(np.float32)).to(dev)
    step = w.control_step_plan(workload.SUBSTEPS, obs.data_ptr(), feet, feet, g0d.data_ptr(), v0d.data_ptr(), n)
    for cs in range(preroll):
        ptd = torch.from_numpy(recipe.targets(n, cs, 0).astype(np.float32)).to(dev)
        step(ptd.data_ptr())
    torch.cuda.synchronize()
    q, u = w.get_state()
    w.close()
    return q.astype(np.float64), u.astype(np.float64)


@pytest.mark.parametrize("regime", ["standing", "collapsing"])
def test_atlas_parity_from_the_benchmarks_stationary_states(regime):
    """Config 5 parity where the benchmark runs: states sampled after the pre-roll (standing on 2-8 foot spheres under target
    jitter / humanoids in every phase of collapsing), both sides restarted from them with a cold solver state.
      one integrate()      same contact sets; converged envs: |dq| < 5e-5, |du| < 5e-3 (1 + |u|_inf)  [the mass matrix of this model
                           has condition ~4e5: 0.125 kg talus links], median 5e-4
      three control steps  (12 x integrate(), warm state carried): median |dq| < 2e-5, p90 < 2e-3, and the worst env bounded"""
    recipe = bench.Recipe(5, -1.0, regime)
    n = 1024
    q0, u0 = _stationary_states(recipe, n, 40)
    model, nv = recipe.model, recipe.model.nv
    kp, kd = recipe.kp.astype(np.float64), recipe.kd.astype(np.float64)
    pt = f32(recipe.targets(n, 40, 0))
    dtg = np.zeros((n, nv))
    # ---- one integrate()
    w = _device_world(recipe, n)
    w.set_pd_target(pt, dtg); w.set_state(q0, u0)
    w.integrate(1)
    q1, u1 = w.get_state(); cnt, con = w.get_contacts(); fl = w.get_flags(); its = w.get_solver_iterations()
    o = _oracle(recipe, n)
    warm = o.new_warm_state(n)
    r = o.step_batch(q0, u0, 1, kp, kd, pt, dtg, want_contacts=True, lam_warm=warm)
    assert r["n_contacts"].mean() > 3.0
    # contact sets: a standing robot's foot spheres REST at the contact boundary (depth ~ 0), where fp32 and fp64 may round to
    # different sides; the sets must agree for nearly all envs, the others are only required to stay bounded
    same = cnt == r["n_contacts"]
    for e in range(0, n, 7):
        if same[e]:
            same[e] = np.array_equal(con[e][:cnt[e]]["collision"], r["contacts"][e][:cnt[e]]["collision"])
    assert same.mean() > (0.97 if regime == "standing" else 0.995), same.mean()
    conv = same & (((r["flags"] | fl) & 4) == 0)
    eu = np.abs(u1 - r["u"]).max(axis=1) / (1 + np.abs(r["u"]).max(axis=1))
    eq = np.abs(q1 - r["q"]).max(axis=1)
    di = np.abs(its[conv] - r["iters"][conv])
    print(f"config 5 {regime}, one step from stationary states: contacts/env {cnt.mean():.2f}, same contact set {100 * same.mean():.1f} %, converged on both sides {100 * conv.mean():.1f} %, "
          f"|du| rel median {np.median(eu):.1e} p99 {np.percentile(eu[conv], 99):.1e} max {eu[conv].max():.1e}; unconverged max {eu[~conv].max() if (~conv).any() else 0:.1e}; "
          f"sweeps equal +-1 for {100 * (di <= 1).mean():.1f} %")
    assert conv.mean() > 0.8
    assert np.all(eu[conv] < 5e-3) and np.median(eu) < 5e-4 and np.all(eq[conv] < 5e-5)
    assert np.all(eu[~conv] < 0.5) and np.isfinite(q1).all() and np.isfinite(u1).all()
    assert (di <= 1).mean() > 0.9
    # ---- three more control steps, warm state carried on both sides
    q, u = r["q"], r["u"]
    w.integrate(workload.SUBSTEPS - 1)
    q, u = (lambda rr: (rr["q"], rr["u"]))(o.step_batch(q, u, workload.SUBSTEPS - 1, kp, kd, pt, dtg, lam_warm=warm))
    for cs in (41, 42):
        ptk = f32(recipe.targets(n, cs, 0))
        w.set_pd_target(ptk, dtg)
        w.integrate(workload.SUBSTEPS)
        rr = o.step_batch(q, u, workload.SUBSTEPS, kp, kd, ptk, dtg, lam_warm=warm)
        q, u = rr["q"], rr["u"]
    qd, ud = w.get_state()
    w.close()
    eq = np.abs(qd - q).max(axis=1)
    print(f"   after 12 integrate(): |dq| median {np.median(eq):.1e} p90 {np.percentile(eq, 90):.1e} p99 {np.percentile(eq, 99):.1e} max {eq.max():.1e}")
    assert np.isfinite(qd).all() and np.median(eq) < 2e-5 and np.percentile(eq, 90) < 2e-3
    assert eq.max() < (0.05 if regime == "standing" else 0.5)     # worst env: a standing robot cannot drift apart; a collapsing one hits the ground elsewhere


def test_anderson_step_on_the_device_matches_the_oracles_with_it_on_and_off():
    """rsb_set_solver_anderson through the C-ABI on the standing humanoids' stationary states: with the step off device and oracle run the
    plain grouped sweep and agree as before; with it on (the default) both need about half the sweeps, and still agree."""
    recipe = bench.Recipe(5, -1.0, "standing")
    n = 1024
    q0, u0 = _stationary_states(recipe, n, 40)
    kp, kd = recipe.kp.astype(np.float64), recipe.kd.astype(np.float64)
    pt = f32(recipe.targets(n, 40, 0)); dtg = np.zeros((n, recipe.model.nv))
    mean_sweeps = {}
    for first in (0, 2):
        recipe.anderson = (first, 20.0)
        w = _device_world(recipe, n)
        w.set_pd_target(pt, dtg); w.set_state(q0, u0)
        o = _oracle(recipe, n)
        warm = o.new_warm_state(n)
        for k in range(3):                       # three integrate() calls: the second and third start warm
            w.integrate(1)
            r = o.step_batch(q0 if k == 0 else r["q"], u0 if k == 0 else r["u"], 1, kp, kd, pt, dtg, lam_warm=warm)
        q1, u1 = w.get_state(); fl = w.get_flags(); its = w.get_solver_iterations(); cnt, _ = w.get_contacts()
        w.close()
        conv = (cnt == r["n_contacts"]) & (((r["flags"] | fl) & 4) == 0)
        eu = np.abs(u1 - r["u"]).max(axis=1) / (1 + np.abs(r["u"]).max(axis=1))
        mean_sweeps[first] = (its.mean(), r["iters"].mean(), np.percentile(its, 99), ((fl & 4) != 0).mean())
        print(f"anderson first sweep {first}: sweeps device {its.mean():.2f} (p99 {np.percentile(its, 99):.0f}) oracle {r['iters'].mean():.2f}, unconverged device {100 * ((fl & 4) != 0).mean():.1f} % "
              f"oracle {100 * ((r['flags'] & 4) != 0).mean():.1f} %, |du| rel median {np.median(eu[conv]):.1e} p99 {np.percentile(eu[conv], 99):.1e}")
        assert conv.mean() > 0.8 and np.median(eu[conv]) < 1e-3 and np.percentile(eu[conv], 99) < 2e-2
        assert abs(its.mean() - r["iters"].mean()) < 0.15 * r["iters"].mean() + 0.3
    assert mean_sweeps[2][0] < 0.7 * mean_sweeps[0][0] and mean_sweeps[2][2] < 0.7 * mean_sweeps[0][2]
    assert mean_sweeps[2][3] <= mean_sweeps[0][3] + 0.005
