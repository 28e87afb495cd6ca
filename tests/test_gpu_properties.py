"""Size-independent properties at BASELINE.json's full size (N = 4096) and the host-side ops around the hot path."""
import ctypes as C

import numpy as np
import pytest

from common import Oracle, f32, sphere_urdf, standing_states
from raisimlib_amd import BatchedWorld, Model, RsbError, workload

pytestmark = pytest.mark.gpu
G, DT = 9.81, 0.0025


def test_free_fall_closed_form_4096_envs(built_lib):
    m = Model(urdf_string=sphere_urdf())
    N = 4096
    w = BatchedWorld(m, N)
    rng = np.random.default_rng(0)
    gc = np.zeros((N, 7), np.float32); gc[:, 2] = rng.uniform(20, 30, N); gc[:, 3] = 1
    gv = np.zeros((N, 6), np.float32); gv[:, :2] = rng.normal(size=(N, 2))
    w.set_state(gc, gv)
    n = 200
    w.integrate(n)
    q, u = w.get_state()
    w.close()
    assert np.allclose(u[:, 2], -G * n * DT, rtol=2e-5)
    assert np.allclose(q[:, 2], gc[:, 2] - G * DT * DT * n * (n + 1) / 2, atol=2e-4)
    assert np.allclose(q[:, :2], gc[:, :2] + gv[:, :2] * n * DT, atol=1e-4) and np.allclose(u[:, :2], gv[:, :2])


def test_full_size_standing_invariants(anymal):
    """N = 4096, 25 control steps of the config-2 workload: finite state, unit quaternions, feet never below the
    ground by more than a sphere radius, impulses inside the friction cone, weight roughly carried."""
    N = 4096
    w = BatchedWorld(anymal, N)
    gc, gv = workload.anymal_initial_state(N)
    kp, kd = workload.anymal_gains()
    w.set_pd_gains(kp, kd); w.set_state(gc, gv)
    dtg = np.zeros((N, 18), np.float32)
    for cs in range(25):
        w.set_pd_target(workload.anymal_targets(N, cs), dtg)
        w.integrate(4)
    q, u = w.get_state()
    cnt, con = w.get_contacts()
    fl = w.get_flags()
    w.close()
    assert np.isfinite(q).all() and np.isfinite(u).all() and not (fl & 2).any()
    assert np.allclose(np.linalg.norm(q[:, 3:7], axis=1), 1.0, atol=1e-5)
    assert (q[:, 2] > 0.25).all() and (q[:, 2] < 0.7).all()
    valid = np.arange(con.shape[1])[None, :] < cnt[:, None]
    imp = con["impulse"]
    assert (imp[..., 2][valid] >= 0).all()
    assert (np.hypot(imp[..., 0], imp[..., 1])[valid] <= 0.8 * imp[..., 2][valid] * (1 + 1e-4) + 1e-7).all()
    assert (con["depth"][valid] > 0).all() and (con["depth"][valid] < 0.03).all()
    fz = np.where(valid, imp[..., 2], 0).sum(axis=1) / DT
    ratio = np.median(fz[cnt >= 3]) / (anymal.total_mass() * G)   # robots are still bouncing on soft PD legs
    assert 0.3 < ratio < 3.0


def test_determinism(anymal):
    gc, gv = standing_states(512, seed=3)
    kp, kd = workload.anymal_gains()
    outs = []
    for _ in range(2):
        w = BatchedWorld(anymal, 512)
        w.set_pd_gains(kp, kd); w.set_pd_target(gc, np.zeros((512, 18))); w.set_state(gc, gv)
        w.integrate(8)
        outs.append(w.get_state())
        w.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_env_independence_and_ragged_batch(anymal):
    """Envs do not interact: an env's result is independent of its wave-mates; N not a multiple of the wave's
    env count works (tail handling)."""
    gc, gv = standing_states(67, seed=17)
    kp, kd = workload.anymal_gains()
    res = {}
    for n in (67, 1, 5):
        w = BatchedWorld(anymal, n)
        w.set_pd_gains(kp, kd); w.set_pd_target(gc[:n], np.zeros((n, 18))); w.set_state(gc[:n], gv[:n])
        w.integrate(4)
        res[n] = w.get_state()
        w.close()
    assert np.array_equal(res[67][0][:5], res[5][0]) and np.array_equal(res[67][1][:1], res[1][1])


def test_masked_set_state_and_reset_terminated(anymal):
    N = 64
    w = BatchedWorld(anymal, N)
    gc, gv = standing_states(N, seed=1)
    w.set_state(gc, gv)
    mask = np.zeros(N, np.uint8); mask[::3] = 1
    g2 = gc.copy(); g2[:, 2] += 5
    w.set_state(g2, None, mask)
    q, u = w.get_state()
    assert np.allclose(q[::3, 2], g2[::3, 2].astype(np.float32)) and np.allclose(q[1::3, 2], gc[1::3, 2].astype(np.float32))
    # belly-flop half of the envs: non-foot contacts -> reset to the init row
    low = gc.copy(); low[:, 2] = np.where(np.arange(N) % 2 == 0, 0.08, 0.9)
    low[:, 3:7] = [1, 0, 0, 0]
    kp, kd = workload.anymal_gains()
    w.set_pd_gains(kp, kd); w.set_pd_target(low, np.zeros((N, 18))); w.set_state(low, np.zeros((N, 18)))
    w.integrate(1)
    init_q, init_u = workload.anymal_initial_state(1)
    done = w.reset_terminated(anymal.collision_indices("_foot"), init_q[0], init_u[0])
    q, u = w.get_state()
    assert done[::2].all() and not done[1::2].any()
    assert np.allclose(q[::2], init_q[0].astype(np.float32)) and np.allclose(u[::2], 0)
    assert (w.get_contacts()[0][::2] == 0).all()
    w.close()


def test_gather_obs_layout(anymal):
    import torch
    N = 128
    w = BatchedWorld(anymal, N)
    gc, gv = standing_states(N, seed=2, z=(0.45, 0.55))
    kp, kd = workload.anymal_gains()
    w.set_pd_gains(kp, kd); w.set_pd_target(gc, np.zeros((N, 18))); w.set_state(gc, gv)
    w.integrate(1)
    feet = anymal.collision_indices("_foot")
    od = w.obs_dim(len(feet))
    assert od == 19 + 18 + 12
    obs = torch.empty((N, od), dtype=torch.float32, device="cuda")
    w.gather_obs(obs.data_ptr(), feet)
    w.synchronize()
    o = obs.cpu().numpy()
    q, u = w.get_state()
    cnt, con = w.get_contacts()
    assert np.array_equal(o[:, :19], q) and np.array_equal(o[:, 19:37], u)
    for e in range(N):
        for k, f in enumerate(feet):
            hit = [c for c in con[e][:cnt[e]] if c["collision"] == f]
            want = hit[0]["impulse"] / DT if hit else np.zeros(3)
            assert np.allclose(o[e, 37 + 3 * k:40 + 3 * k], want, rtol=1e-6, atol=1e-6)
    w.close()


def test_borrowed_stream_and_device_targets(anymal):
    """The world runs on a caller-provided HIP stream (torch's), and accepts device-resident PD targets."""
    import torch
    N = 256
    gc, gv = standing_states(N, seed=8)
    kp, kd = workload.anymal_gains()
    dtg = np.zeros((N, 18), np.float32)
    w = BatchedWorld(anymal, N)
    w.set_pd_gains(kp, kd); w.set_pd_target(gc, dtg); w.set_state(gc, gv); w.integrate(4)
    ref = w.get_state()
    w.close()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        w = BatchedWorld(anymal, N)
        w.set_stream(stream.cuda_stream)
        pt = torch.from_numpy(gc.astype(np.float32)).cuda()
        w.set_pd_gains(kp, kd); w.set_pd_target(None, dtg); w.set_pd_target_device(pt.data_ptr()); w.set_state(gc, gv)
        w.integrate(4)
        stream.synchronize()
        got = w.get_state()
        w.close()
    assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])


def test_control_step_equals_the_separate_calls(anymal):
    """rsb_control_step (targets, 4 x integrate, obs, reset fused into one launch) gives bit-identical state, obs,
    contact counts and stored PD targets to rsb_set_pd_target + rsb_integrate + rsb_gather_obs + rsb_reset_terminated."""
    import torch
    N = 300          # ragged: not a multiple of the envs per workgroup
    feet = anymal.collision_indices("_foot")
    gc, gv = standing_states(N, seed=21, z=(0.3, 0.6), vel=1.5)
    gc[::7, 2] = 0.1                                    # some envs start belly-down -> terminate
    # ... and some start contorted in the air: self-collisions only, whose contact entries carry RSB_CONTACT_SELF_A / _B in the
    # collision id - terminal in both paths, whatever the low bits of the id are (a foot primitive hitting a thigh included)
    contorted = np.arange(3, N, 11)
    gc[contorted, 7:] = np.random.default_rng(5).uniform(-2.5, 2.5, (len(contorted), 12))
    gc[contorted, 2] = 2.0
    kp, kd = workload.anymal_gains()
    init_q, init_u = workload.anymal_initial_state(N)
    g0 = torch.from_numpy(init_q.astype(np.float32)).cuda(); v0 = torch.from_numpy(init_u.astype(np.float32)).cuda()
    od = 19 + 18 + 3 * len(feet)
    res = []
    for fused in (False, True):
        w = BatchedWorld(anymal, N)
        w.set_pd_gains(kp, kd); w.set_pd_target(gc, np.zeros((N, 18))); w.set_state(gc, gv)
        obs = torch.zeros((N, od), dtype=torch.float32, device="cuda")
        step = w.control_step_plan(4, obs.data_ptr(), feet, feet, g0.data_ptr(), v0.data_ptr(), N)
        hist = []
        for k in range(6):
            pt = torch.from_numpy(workload.anymal_targets(N, k).astype(np.float32)).cuda()
            if fused:
                step(pt.data_ptr())
            else:
                w.set_pd_target_device(pt.data_ptr())
                w.integrate(4)
                w.gather_obs(obs.data_ptr(), feet)
                w.reset_terminated_device(feet, g0.data_ptr(), v0.data_ptr(), N)
            w.synchronize()
            q, u = w.get_state()
            hist.append((q, u, obs.cpu().numpy().copy(), w.get_contacts()[0], w.get_flags()))
        w.integrate(2)                                   # uses the world's own copy of the last PD targets
        hist.append(w.get_state())
        res.append(hist)
        w.close()
    terminated = 0
    for a, b in zip(res[0][:6], res[1][:6]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        terminated += int((a[3] == 0).sum())
    assert np.array_equal(res[0][6][0], res[1][6][0]) and np.array_equal(res[0][6][1], res[1][6][1])   # stored targets refreshed
    assert terminated > 0
    q0f = res[1][0][0]                                   # fused path after control step 0: self-colliding envs were reset
    reset0 = np.all(q0f == init_q.astype(np.float32), axis=1)
    assert reset0[contorted].sum() >= 5 and np.array_equal(reset0, np.all(res[0][0][0] == init_q.astype(np.float32), axis=1))


def test_api_errors(anymal):
    w = BatchedWorld(anymal, 8)
    with pytest.raises(RsbError):
        w.set_lanes_per_env(8)
    with pytest.raises(RsbError):
        w.set_max_contacts(0)
    with pytest.raises(RsbError):
        w.set_time_step(-1.0)
    with pytest.raises(RsbError):
        w.integrate(0)
    with pytest.raises(RsbError):
        w.set_contact_solver_param(1, 1, 1, 0, 1e-5)
    bad = np.full(anymal.ncol, 0.8); bad[3] = np.nan
    with pytest.raises(RsbError):
        w.set_collision_materials(mu=bad)
    with pytest.raises(RsbError):
        w.set_collision_materials(res_threshold=np.full(anymal.ncol, np.inf))
    with pytest.raises(RsbError):
        w.set_self_collision_materials(mu=np.full(len(w.self_collision_pairs()), np.nan))
    for bad_args in ((-1, 20.0), (2, 0.0), (2, float("nan"))):
        with pytest.raises(RsbError):
            w.set_solver_anderson(*bad_args)
    for bad_args in ((0, 25.0), (3, 25.0), (2, 0.0), (2, 90.0)):
        with pytest.raises(RsbError):
            w.set_heightmap_contacts(*bad_args)
    w.set_time_step(0.001)
    assert abs(w.get_time_step() - 0.001) < 1e-15
    w.close()
    # two contacts per primitive against a height map: a kernel class of the floating-base systems - a fixed-base model is refused at the step
    from test_oracle_kat import FIXED_PENDULUM
    from raisimlib_amd import Model
    fixed = Model(urdf_string=FIXED_PENDULUM.format(l=0.5, m=1.0))
    assert fixed.blob.fixed_base
    if True:
        wf = BatchedWorld(fixed, 8)
        wf.add_height_map(5, 5, 4.0, 4.0, 0.0, 0.0, np.zeros((5, 5), np.float32))
        wf.set_heightmap_contacts(2)
        with pytest.raises(RsbError, match="floating-base"):
            wf.integrate(1)
        wf.set_heightmap_contacts(1)
        wf.integrate(1)
        wf.close()


def test_no_use_of_uninitialised_lds():
    """Re-run the parity + invariants tests in a child process with RSB_POISON_LDS=1: the kernel first fills its whole
    LDS allocation with NaNs, so any read of never-written LDS that reaches a result turns into a test failure."""
    import os
    import subprocess
    import sys
    from common import ROOT
    env = dict(os.environ, RSB_POISON_LDS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                        os.path.join(ROOT, "tests", "test_gpu_properties.py"), "-k", "not uninitialised and not borrowed"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:]


def _env_reference(q, u, pt, kp, kd, counts, contacts, flags, feet, cfg, q3=None, u3=None, effort=None):
    """numpy restatement of the rsg_anymal-style reward / termination / observation (tests only)."""
    N = q.shape[0]
    w, x, y, z = q[:, 3], q[:, 4], q[:, 5], q[:, 6]
    R = np.empty((N, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    vb = np.einsum("nji,nj->ni", R, u[:, 0:3]); wb = np.einsum("nji,nj->ni", R, u[:, 3:6])
    obs = np.concatenate([q[:, 2:3], R[:, 2, :], q[:, 7:], vb, wb, u[:, 6:]], axis=1)   # R[:, 2, :]: third ROW of R = rsg_anymal's rot.e().row(2)
    # torque cost: the actuator torque the LAST sub-step applied (implicit PD: position error at q + dt u, clipped to the
    # effort limit), i.e. upstream's getGeneralizedForce() after the last integrate(); (q3, u3) = state that sub-step started from
    q3 = q if q3 is None else q3
    u3 = u if u3 is None else u3
    tau = kp[6:] * (pt[:, 7:] - q3[:, 7:] - workload.DT * u3[:, 6:]) - kd[6:] * u3[:, 6:]
    if effort is not None:
        lim = np.where(effort > 0, effort, np.inf)
        tau = np.clip(tau, -lim, lim)
    r = cfg["fwd"] * np.minimum(cfg["clip"], vb[:, 0]) + cfg["tc"] * (tau ** 2).sum(1)
    term = (flags & 2) != 0
    for e in range(N):
        term[e] |= bool(np.any(~np.isin(contacts[e][:counts[e]]["collision"], feet)))
    return obs, np.where(term, r + cfg["term"], r), term   # upstream perAgentStep: reward += terminalReward


def test_device_vecenv_matches_the_host_restatement(anymal):
    """rsb_env_step / rsb_env_observe (device-resident rsg_anymal task) against the same quantities computed in numpy
    from rsb_get_state / rsb_get_contacts of a twin world driven through the plain API, for torch and numpy I/O."""
    import torch
    from raisimlib_amd import VecEnv
    N = 200
    feet = anymal.collision_indices("_foot")
    gc_init = workload.anymal_initial_state(1)[0][0].astype(np.float32)
    cfg = dict(fwd=0.3, clip=4.0, tc=-4e-5, term=-10.0)
    env = VecEnv(anymal, N, gc_init=gc_init)
    assert (env.num_obs, env.num_acts) == (34, 12)
    twin = BatchedWorld(anymal, N)
    kp = np.zeros(18, np.float32); kd = np.zeros(18, np.float32); kp[6:] = 50.0; kd[6:] = 0.2
    twin.set_pd_gains(kp, kd)
    twin.set_state(np.tile(gc_init, (N, 1)), np.zeros((N, 18)))
    effort = np.array([anymal.blob.effort[b] for b in range(1, anymal.nb)])
    ob0 = env.observe()
    assert np.allclose(ob0[:, 0], gc_init[2]) and np.allclose(ob0[:, 1:4], [0, 0, 1]) and np.allclose(ob0[:, 4:16], gc_init[7:])
    rng = np.random.default_rng(5)
    n_done = 0
    for k in range(40):
        act = rng.normal(size=(N, 12)).astype(np.float32) * (3.0 if k % 9 == 8 else 1.0)   # big kicks make some fall
        ob_same_launch = None
        if k % 2 == 0:
            ob_same_launch = np.zeros((N, 34), np.float32)
            rew, done = env.step(act, ob_next=ob_same_launch) if k % 4 == 0 else env.step(act)
        else:
            ob_t = torch.empty((N, 34), device="cuda")
            r_t, d_t = env.step(torch.from_numpy(act).cuda(), ob_next=ob_t)
            torch.cuda.synchronize()
            rew, done, ob_same_launch = r_t.cpu().numpy(), d_t.cpu().numpy(), ob_t.cpu().numpy()
        pt = np.zeros((N, 19), np.float32); pt[:, 3] = 1; pt[:, 7:] = gc_init[7:] + np.float32(0.3) * act
        twin.set_pd_target(pt, np.zeros((N, 18), np.float32))
        twin.integrate(3)
        q3, u3 = twin.get_state()
        twin.integrate(1)
        q, u = twin.get_state(); cnt, con = twin.get_contacts(); fl = twin.get_flags()
        _, r_ref, term = _env_reference(q.astype(np.float64), u.astype(np.float64), pt.astype(np.float64), kp, kd, cnt, con, fl, feet, cfg,
                                        q3.astype(np.float64), u3.astype(np.float64), effort)
        assert np.array_equal(done.astype(bool), term)
        assert np.allclose(rew, r_ref, rtol=2e-5, atol=2e-5)
        twin.reset_terminated(feet, gc_init, np.zeros(18, np.float32))
        q, u = twin.get_state()
        ob_ref, _, _ = _env_reference(q.astype(np.float64), u.astype(np.float64), pt.astype(np.float64), kp, kd, cnt, con, fl, feet, cfg)
        ob = env.observe() if k % 2 else env.observe(torch.empty((N, 34), device="cuda")).cpu().numpy()
        assert np.allclose(ob, ob_ref, rtol=1e-5, atol=1e-5)
        if ob_same_launch is not None and (k % 2 == 1 or k % 4 == 0):
            assert np.array_equal(ob_same_launch, ob)          # the fused observation is the same kernel code
        n_done += int(term.sum())
    assert n_done > 0
    env.close(); twin.close()


def test_vecenv_running_observation_statistics(anymal):
    """observe_normalized keeps RunningMeanStd-style statistics on the device: after many batches the normalised
    observations have ~zero mean / unit variance, and the statistics match a numpy recomputation over all batches."""
    import torch
    from raisimlib_amd import VecEnv
    N = 256
    gc_init = np.zeros(19, np.float32); gc_init[2] = 0.6; gc_init[3] = 1.0; gc_init[7:] = workload.ANYMAL_NOMINAL_JOINTS
    env = VecEnv(anymal, N, gc_init=gc_init)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    raw, ob = [], torch.empty((N, 34), device="cuda")
    for k in range(30):
        env.step(torch.empty((N, 12), device="cuda").uniform_(-1, 1, generator=gen))
        raw.append(env.observe().copy())
        env.observe_normalized(ob)
    allraw = np.concatenate(raw)
    assert np.allclose(env.ob_mean.cpu().numpy(), allraw.mean(0), rtol=1e-3, atol=1e-4)
    assert np.allclose(env.ob_var.cpu().numpy(), allraw.var(0), rtol=2e-2, atol=1e-5)
    z = (raw[-1] - allraw.mean(0)) / np.sqrt(allraw.var(0) + 1e-8)
    assert np.allclose(ob.cpu().numpy(), np.clip(z, -10, 10), rtol=2e-2, atol=2e-2)
    import tempfile
    with tempfile.TemporaryDirectory() as d:                  # scaling files round trip (upstream's mean<i>.csv / var<i>.csv)
        env.save_scaling(d, 7)
        m0, v0 = env.ob_mean.clone(), env.ob_var.clone()
        env.ob_mean.zero_(); env.load_scaling(d, 7)
        assert torch.allclose(env.ob_mean, m0, rtol=1e-6) and torch.allclose(env.ob_var, v0, rtol=1e-6)
    env.close()


def test_early_termination_freezes_the_env_at_its_first_illegal_contact(anymal):
    """rsb_set_early_termination: envs without a non-foot contact are bit-identical to the default mode; an env whose
    first non-foot contact is detected in sub-step k reports the state it had BEFORE sub-step k as its terminal
    observation, is flagged (bit 3) and reset."""
    import torch
    N = 300
    feet = anymal.collision_indices("_foot")
    foot_mask = np.zeros(anymal.ncol, bool); foot_mask[feet] = True
    gc, gv = standing_states(N, seed=31, z=(0.25, 0.6), vel=2.0)
    gc[::5, 2] = 0.12                                   # belly-down starts: illegal contact in the first sub-steps
    kp, kd = workload.anymal_gains()
    init_q, init_u = workload.anymal_initial_state(N)
    g0 = torch.from_numpy(init_q.astype(np.float32)).cuda(); v0 = torch.from_numpy(init_u.astype(np.float32)).cuda()
    pt = torch.from_numpy(gc.astype(np.float32)).cuda()
    od = 19 + 18 + 3 * len(feet)
    out = {}
    for early in (False, True):
        w = BatchedWorld(anymal, N)
        w.set_early_termination(early)
        w.set_pd_gains(kp, kd); w.set_pd_target(gc, np.zeros((N, 18))); w.set_state(gc, gv)
        obs = torch.zeros((N, od), dtype=torch.float32, device="cuda")
        w.control_step_plan(4, obs.data_ptr(), feet, feet, g0.data_ptr(), v0.data_ptr(), N)(pt.data_ptr())
        w.synchronize()
        out[early] = (obs.cpu().numpy().copy(),) + w.get_state()
        w.close()
    # twin: sub-step by sub-step, to find each env's first illegal contact and the state before it
    w = BatchedWorld(anymal, N)
    w.set_pd_gains(kp, kd); w.set_pd_target(gc, np.zeros((N, 18))); w.set_state(gc, gv)
    first = np.full(N, -1); before = [None] * 4
    for k in range(4):
        before[k] = w.get_state()
        w.integrate(1)
        cnt, con = w.get_contacts()
        valid = np.arange(con.shape[1])[None, :] < cnt[:, None]
        ill = (valid & ~(foot_mask[con["collision"] & 0xffff] & (con["collision"] < 0x10000))).any(1)
        first[(first < 0) & ill] = k
    w.close()
    clean = first < 0
    assert clean.sum() > 50 and (~clean).sum() > 50 and len(set(first[~clean])) > 1
    for a, b in zip(out[False], out[True]):
        assert np.array_equal(a[clean], b[clean])                         # untouched envs: identical to the default mode
    ob_e, q_e, u_e = out[True]
    for e in np.where(~clean)[0]:
        qb, ub = before[first[e]]
        assert np.array_equal(ob_e[e, :19], qb[e]) and np.array_equal(ob_e[e, 19:37], ub[e])   # frozen before that sub-step
        assert np.all(ob_e[e, 37:] == 0)                                  # its contacts carry no impulse
        assert np.array_equal(q_e[e], init_q[e].astype(np.float32))       # and it was reset


def test_lanes_per_env_variants_run_the_full_control_step(anymal):
    """The 32- and 64-lane mappings through the fused control step (warm start, resets, observation) over several
    control steps: same terminations and, for envs that stayed clear of resets, the same trajectories up to fp32."""
    import torch
    N = 256
    feet = anymal.collision_indices("_foot")
    gc, gv = workload.anymal_initial_state(N)
    gc[:, 2] = 0.52
    kp, kd = workload.anymal_gains()
    g0 = torch.from_numpy(gc.astype(np.float32)).cuda(); v0 = torch.from_numpy(gv.astype(np.float32)).cuda()
    outs = {}
    for lpe in (16, 32, 64):
        w = BatchedWorld(anymal, N); w.set_lanes_per_env(lpe)
        w.set_pd_gains(kp, kd); w.set_pd_target(gc, np.zeros((N, 18))); w.set_state(gc, gv)
        obs = torch.zeros((N, 49), device="cuda")
        step = w.control_step_plan(4, obs.data_ptr(), feet, feet, g0.data_ptr(), v0.data_ptr(), N)
        for k in range(12):
            pt = torch.from_numpy(workload.anymal_targets(N, k, amplitude=0.5).astype(np.float32)).cuda()
            step(pt.data_ptr()); w.synchronize()
        outs[lpe] = w.get_state() + (obs.cpu().numpy().copy(),)
        assert w.lanes_per_env() == lpe
        w.close()
    q16, u16, o16 = outs[16]
    for lpe in (32, 64):
        q, u, o = outs[lpe]
        dq = np.abs(q - q16).max(1)
        assert np.isfinite(q).all() and np.median(dq) < 1e-5 and (dq < 1e-3).mean() > 0.9, (lpe, np.median(dq), (dq < 1e-3).mean())


def test_population_statistics_match_the_oracle_over_100_control_steps(anymal):
    """The benchmark's own regime, N = 4096, 100 control steps with the reset rule, device vs fp64 oracle from identical
    initial states and per-env seeded targets.  Individual trajectories diverge (contact dynamics is chaotic and the two
    sides round differently), so the POPULATIONS are compared: resets, base-height distribution, contacts and sweeps."""
    import torch
    N, STEPS = 4096, 100
    feet = anymal.collision_indices("_foot")
    feet_set = np.zeros(anymal.ncol, bool); feet_set[feet] = True
    kp, kd = workload.anymal_gains()
    gc0, gv0 = workload.anymal_initial_state(N)
    gc0 = f32(gc0)
    # device: the fused control step with resets and done flags
    w = BatchedWorld(anymal, N)
    w.set_pd_gains(kp, kd); w.set_state(gc0, gv0); w.set_pd_target(None, np.zeros((N, 18), np.float32))
    dev = torch.device("cuda")
    stream = torch.cuda.current_stream(); w.set_stream(stream.cuda_stream)
    done_d = torch.zeros(N, dtype=torch.uint8, device=dev); w.set_done_output(done_d.data_ptr())
    obs = torch.empty((N, w.obs_dim(4)), device=dev)
    g0d = torch.from_numpy(gc0.astype(np.float32)).to(dev); v0d = torch.from_numpy(gv0.astype(np.float32)).to(dev)
    step = w.control_step_plan(4, obs.data_ptr(), feet, feet, g0d.data_ptr(), v0d.data_ptr(), N)
    # oracle: the same recipe on the host
    o = Oracle(anymal.blob)
    q, u, warm = gc0.copy(), gv0.copy(), o.new_warm_state(N)
    dev_resets, orc_resets, dev_iters, orc_iters = [], [], [], []
    for cs in range(STEPS):
        pt = f32(workload.anymal_targets(N, cs))
        ptd = torch.from_numpy(pt.astype(np.float32)).to(dev)
        step(ptd.data_ptr())
        torch.cuda.synchronize()
        dev_resets.append(int(done_d.sum().item()))
        dev_iters.append(w.get_solver_iterations().mean())
        r = o.step_batch(q, u, 4, kp.astype(np.float64), kd.astype(np.float64), pt, np.zeros((N, 18)), want_contacts=True, lam_warm=warm)
        q, u = r["q"], r["u"]
        con, ncs = r["contacts"], r["n_contacts"]
        valid = np.arange(con.shape[1])[None, :] < ncs[:, None]
        term = (valid & ~(feet_set[con["collision"] & 0xffff] & (con["collision"] < 0x10000))).any(axis=1) | (r["flags"] & 2).astype(bool)
        orc_resets.append(int(term.sum())); orc_iters.append(r["iters"].mean())
        q[term], u[term], warm[term] = gc0[term], gv0[term], 0.0
    qd, ud = w.get_state()
    cnt_d, _ = w.get_contacts()
    w.close()
    dev_resets, orc_resets = np.array(dev_resets), np.array(orc_resets)
    print(f"resets: device {dev_resets.sum()} oracle {orc_resets.sum()}; last-20-step mean {dev_resets[-20:].mean():.1f} vs {orc_resets[-20:].mean():.1f}; "
          f"sweeps {np.mean(dev_iters[-20:]):.2f} vs {np.mean(orc_iters[-20:]):.2f}")
    assert np.isfinite(qd).all() and np.isfinite(ud).all()
    assert orc_resets.sum() > 500                                               # the regime does contain falling robots
    assert abs(dev_resets.sum() - orc_resets.sum()) <= 0.03 * orc_resets.sum()  # cumulative resets within 3 %
    # while the two populations are still the same trajectories (first control steps) the counts agree step by step
    assert np.abs(dev_resets[:30] - orc_resets[:30]).max() <= 3
    # the stationary populations have the same shape: base height quantiles within 5 mm, mean contacts within 0.05
    for pct in (5, 25, 50, 75, 95):
        assert abs(np.percentile(qd[:, 2], pct) - np.percentile(q[:, 2], pct)) < 5e-3, pct
    assert abs(cnt_d.mean() - r["n_contacts"][~term].sum() / N) < 0.08
    assert abs(np.mean(dev_iters[-20:]) - np.mean(orc_iters[-20:])) < 0.1      # sweeps of the last sub-step, population mean


def test_masked_integrate_and_done_output(anymal):
    """rsb_integrate_masked advances exactly the masked replicas (bit-identical to an unmasked launch for them, untouched
    rows for the others); rsb_set_done_output reports the envs a fused control step reset."""
    import torch
    N = 96
    gc, gv = standing_states(N, seed=3)
    kp, kd = workload.anymal_gains()
    a, b = BatchedWorld(anymal, N), BatchedWorld(anymal, N)
    for w in (a, b):
        w.set_pd_gains(kp, kd); w.set_pd_target(gc, np.zeros((N, 18))); w.set_state(gc, gv)
        w.integrate(2)                                   # contacts + warm state exist before the masked launch
    qa0, ua0 = a.get_state(); ca0, cona0 = a.get_contacts()
    mask = (np.arange(N) % 3 == 0).astype(np.uint8)
    a.integrate_masked(mask, 1)
    b.integrate(1)
    qa, ua = a.get_state(); qb, ub = b.get_state(); ca, cona = a.get_contacts(); cb, conb = b.get_contacts()
    on = mask.astype(bool)
    assert np.array_equal(qa[on], qb[on]) and np.array_equal(ua[on], ub[on]) and np.array_equal(ca[on], cb[on])
    assert np.array_equal(qa[~on], qa0[~on]) and np.array_equal(ua[~on], ua0[~on]) and np.array_equal(ca[~on], ca0[~on])
    assert cona[~on].tobytes() == cona0[~on].tobytes()
    a.integrate_masked(1 - mask, 1)                      # now the others catch up: both worlds are equal again ...
    a.integrate_masked(mask, 1); b.integrate(1)          # ... after one more step of the first group
    qa, ua = a.get_state(); qb, ub = b.get_state()
    assert np.array_equal(qa[on], qb[on])
    # done flags of the fused control step
    dev = torch.device("cuda")
    done = torch.full((N,), 7, dtype=torch.uint8, device=dev)
    b.set_stream(torch.cuda.current_stream().cuda_stream)
    b.set_done_output(done.data_ptr())
    feet = anymal.collision_indices("_foot")
    g0 = torch.from_numpy(gc[0].astype(np.float32)).to(dev); v0 = torch.zeros(18, device=dev)
    low = gc.copy(); low[: N // 2, 2] = 0.25             # half of the robots start with the belly in the ground
    b.set_state(low, gv)
    ptd = torch.from_numpy(gc.astype(np.float32)).to(dev)
    obs = torch.empty((N, b.obs_dim(4)), device=dev)
    b.control_step_plan(4, obs.data_ptr(), feet, feet, g0.data_ptr(), v0.data_ptr(), 1)(ptd.data_ptr())
    torch.cuda.synchronize()
    d = done.cpu().numpy()
    assert set(np.unique(d)) <= {0, 1} and d[: N // 2].all() and d.sum() < N
    q1, _ = b.get_state()
    assert np.array_equal(q1[d.astype(bool)], np.tile(gc[0].astype(np.float32), (int(d.sum()), 1)))
    a.close(); b.close()


def test_device_shards_equal_the_unsharded_world_bit_for_bit(anymal):
    """Two worlds owning global envs [0, n) and [n, 2n) (what two ranks hold) produce exactly the rows of one world with 2n
    envs: state, contacts and the obs block, over several control steps with resets (no cross-env coupling)."""
    import torch
    n, steps = 300, 12          # 300 is not a multiple of the 4 envs per wave: env -> wave packing differs between the layouts
    feet = anymal.collision_indices("_foot")
    kp, kd = workload.anymal_gains()
    dev = torch.device("cuda")

    def run(lo, hi):
        N = hi - lo
        gc0, gv0 = workload.anymal_initial_state(N, env_offset=lo, height=0.56)
        w = BatchedWorld(anymal, N)
        w.set_stream(torch.cuda.current_stream().cuda_stream)
        w.set_pd_gains(kp, kd); w.set_state(gc0, gv0); w.set_pd_target(None, np.zeros((N, 18), np.float32))
        g0 = torch.from_numpy(gc0.astype(np.float32)).to(dev); v0 = torch.from_numpy(gv0.astype(np.float32)).to(dev)
        obs = torch.empty((N, w.obs_dim(4)), device=dev)
        step = w.control_step_plan(4, obs.data_ptr(), feet, feet, g0.data_ptr(), v0.data_ptr(), N)
        for k in range(steps):
            pt = torch.from_numpy(workload.anymal_targets(N, k, env_offset=lo, amplitude=0.6).astype(np.float32)).to(dev)
            step(pt.data_ptr())
        torch.cuda.synchronize()
        q, u = w.get_state(); c, con = w.get_contacts()
        w.close()
        return q, u, c, con, obs.cpu().numpy()

    full = run(0, 2 * n)
    a, b = run(0, n), run(n, 2 * n)
    for k in range(5):
        sharded = np.concatenate([a[k], b[k]], axis=0)
        assert sharded.tobytes() == full[k].tobytes(), k
    assert (full[2] > 0).any()


def test_rccl_allgather_through_the_c_abi_single_rank(anymal):
    """rsb_comm_* / rsb_allgather_obs (the C/C++ host's collective, RCCL loaded at run time) on a one-rank communicator:
    the gathered block equals rsb_gather_obs.  Runs in a child process without torch, under a timeout (a second RCCL
    in a process that already holds torch's copy is not what a C++ host would do)."""
    import os
    import subprocess
    import sys
    from common import ROOT
    script = r"""
import ctypes as C, numpy as np, sys
sys.path.insert(0, %r)
from raisimlib_amd import BatchedWorld, Model, rsc_path, workload
from raisimlib_amd._capi import lib, check, RSB_HOST
m = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
N = 128
w = BatchedWorld(m, N)
gc, gv = workload.anymal_initial_state(N, height=0.55)
kp, kd = workload.anymal_gains()
w.set_pd_gains(kp, kd); w.set_pd_target(gc, np.zeros((N, 18))); w.set_state(gc, gv); w.integrate(4)
L = lib()
ident = C.create_string_buffer(128)
check(L.rsb_comm_get_unique_id(ident), "rsb_comm_get_unique_id")
check(L.rsb_comm_init(w.handle, 1, 0, ident), "rsb_comm_init")
feet = np.asarray(m.collision_indices("_foot"), np.int32)
od = w.obs_dim(4)
out = np.zeros((N, od), np.float32)
check(L.rsb_allgather_obs(w.handle, feet.ctypes.data_as(C.c_void_p), 4, out.ctypes.data_as(C.c_void_p), RSB_HOST), "rsb_allgather_obs")
q, u = w.get_state()
assert np.array_equal(out[:, :19], q) and np.array_equal(out[:, 19:37], u) and np.abs(out[:, 37:]).max() > 0
check(L.rsb_comm_destroy(w.handle), "rsb_comm_destroy")
w.close()
print("RCCL_OK")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_generalized_force_output_is_the_clipped_pd_torque(anymal):
    """rsb_enable_generalized_force_output / ArticulatedSystem::getGeneralizedForce(): the actuator torque of the LAST sub-step -
    implicit PD evaluated at q + dt u of that sub-step's start, + feed-forward, clipped to the joint's effort limit."""
    N = 256
    gc, gv = standing_states(N, seed=21, vel=2.0)
    kp, kd = workload.anymal_gains()
    kp = kp * 8                                   # large errors: some joints clip at their effort limit
    pt = gc.copy(); pt[:, 7:] += np.random.default_rng(3).uniform(-0.6, 0.6, (N, 12))
    tff = np.zeros((N, 18), np.float32); tff[:, 6:] = np.random.default_rng(4).uniform(-5, 5, (N, 12)); tff[:, :6] = 0.25
    w = BatchedWorld(anymal, N)
    with pytest.raises(Exception):
        w.get_generalized_force()                 # off by default: asking for it without enabling fails loudly
    w.enable_generalized_force_output(True)
    w.set_pd_gains(kp, kd); w.set_pd_target(pt, np.zeros((N, 18))); w.set_generalized_force(tff)
    w.set_state(gc, gv)
    w.integrate(2)
    q2, u2 = w.get_state()
    f2 = w.get_generalized_force()
    w.set_state(gc, gv); w.integrate(1)           # the state the second sub-step started from
    q1, u1 = w.get_state()
    dt = 0.0025
    eff = np.array([anymal.blob.effort[b] for b in range(1, anymal.nb)])
    raw = kp[6:] * (np.float32(pt[:, 7:]) - q1[:, 7:] - dt * u1[:, 6:]) + kd[6:] * (0.0 - u1[:, 6:]) + tff[:, 6:]
    want = np.where(eff > 0, np.clip(raw, -eff, eff), raw)
    assert (np.abs(raw) > eff).sum() > 50 and (np.abs(raw) < eff).sum() > 50
    assert np.allclose(f2[:, 6:], want, rtol=1e-4, atol=2e-3) and np.allclose(f2[:, :6], 0.25)
    w.close()
