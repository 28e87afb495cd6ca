"""The analytic known-answer tests of tests/test_oracle_kat.py run through the C-ABI on the GPU (fp32 tolerances):
with parity to RaiSim unpinned, the HIP path is checked against closed-form physics directly, not only against the
in-repo oracle.  Every KAT runs 64 identical envs (one wave would hide lane-mapping bugs of the others)."""
import numpy as np
import pytest

from common import PENDULUM_URDF, Oracle, sphere_urdf
from raisimlib_amd import BatchedWorld, Model, workload
from test_oracle_kat import SLED

pytestmark = pytest.mark.gpu
G, DT, N = 9.81, 0.0025, 64


def world(urdf, gravity=None, mode=1):
    m = Model(urdf_string=urdf)
    w = BatchedWorld(m, N)
    if gravity is not None:
        w.set_gravity(gravity)
    w.set_control_mode(mode)
    return m, w


def tile(x):
    return np.tile(np.asarray(x, np.float64), (N, 1))


def test_sphere_rest_impulse_is_m_g_dt(built_lib):
    m_, r = 2.0, 0.1
    _, w = world(sphere_urdf(m_, r))
    w.set_state(tile([0, 0, r - 1e-4, 1, 0, 0, 0]), tile(np.zeros(6)))
    w.integrate(5)
    cnt, con = w.get_contacts(); q, u = w.get_state()
    assert (cnt == 1).all()
    imp = np.array([c[0]["impulse"] for c in con])
    assert np.allclose(imp, [0, 0, m_ * G * DT], atol=2e-7) and np.abs(u).max() < 1e-6
    assert np.ptp(imp, axis=0).max() == 0.0           # all 64 envs bit-identical
    w.close()


def test_sliding_friction_decelerates_at_mu_g(built_lib):
    m_, r, mu = 2.0, 0.1, 0.8
    _, w = world(sphere_urdf(m_, r))
    u0 = np.array([3.0, 0, 0, 0, 0, 0])
    w.set_state(tile([0, 0, r - 1e-5, 1, 0, 0, 0]), tile(u0))
    prev = 3.0
    for k in range(10):
        w.integrate(1)
        _, u = w.get_state(); cnt, con = w.get_contacts()
        assert np.allclose(u[:, 0] - prev, -mu * G * DT, atol=3e-6), k
        lam = np.array([c[0]["impulse"] for c in con])
        assert np.allclose(np.hypot(lam[:, 0], lam[:, 1]), mu * lam[:, 2], rtol=1e-5)
        assert (lam[:, 0] < 0).all() and np.abs(lam[:, 1]).max() < 1e-6
        prev = u[0, 0]
    w.close()


@pytest.mark.parametrize("angle_deg,sticks", [(30.0, True), (36.0, True), (42.0, False), (50.0, False)])
def test_stick_slip_threshold_on_incline(built_lib, angle_deg, sticks):
    th = np.radians(angle_deg)
    _, w = world(SLED, gravity=[G * np.sin(th), 0.0, -G * np.cos(th)])
    w.set_state(tile([0, 0, 0.05 - 1e-5, 1, 0, 0, 0]), tile(np.zeros(6)))
    n = 200
    w.integrate(n)
    _, u = w.get_state(); cnt, _ = w.get_contacts()
    assert (cnt == 4).all()
    if sticks:
        assert np.abs(u).max() < 2e-4
    else:
        a = G * (np.sin(th) - 0.8 * np.cos(th))
        assert np.allclose(u[:, 0], a * n * DT, rtol=2e-2)
        mu_eff = (np.sin(th) - u[:, 0] / (n * DT) / G) / np.cos(th)
        assert np.abs(mu_eff - 0.8).max() < 2e-3
        assert np.abs(u[:, 2]).max() < 1e-4 and np.abs(u[:, 3:]).max() < 1e-3


def test_pendulum_spring_period(built_lib):
    l, m_ = 0.5, 1.0
    _, w = world(PENDULUM_URDF.format(l=l, m=m_), gravity=[0, 0, 0])
    kp = np.zeros(7, np.float32); kd = np.zeros(7, np.float32); kp[6] = 40.0
    T = 2 * np.pi * np.sqrt((m_ * l * l + 1e-9) / kp[6])
    pt = tile([0, 0, 0, 1, 0, 0, 0, 0.0])
    w.set_pd_gains(kp, kd); w.set_pd_target(pt, tile(np.zeros(7)))
    w.set_state(tile([0, 0, 0, 1, 0, 0, 0, 0.1]), tile(np.zeros(7)))
    th, t = [], []
    for k in range(int(5 * T / DT)):
        w.integrate(1)
        q, _ = w.get_state()
        th.append(q[0, 7]); t.append((k + 1) * DT)
        if k == 0:
            assert np.ptp(q, axis=0).max() == 0.0
    th = np.array(th, np.float64); t = np.array(t)
    idx = np.where((th[:-1] < 0) & (th[1:] >= 0))[0]
    tc = t[idx] + DT * (-th[idx]) / (th[idx + 1] - th[idx])
    assert len(tc) >= 4 and np.allclose(np.diff(tc), T, rtol=3e-3)
    assert abs(np.abs(th).max() - 0.1) < 3e-3
    w.close()


def test_standing_anymal_carries_its_weight(anymal):
    w = BatchedWorld(anymal, N)
    kp = np.zeros(18, np.float32); kd = np.zeros(18, np.float32); kp[6:] = 400.0; kd[6:] = 10.0
    q0 = np.zeros(19); q0[2] = 0.60; q0[3] = 1; q0[7:] = workload.ANYMAL_NOMINAL_JOINTS
    w.set_pd_gains(kp, kd); w.set_pd_target(tile(q0), tile(np.zeros(18))); w.set_state(tile(q0), tile(np.zeros(18)))
    w.integrate(1500)
    cnt, con = w.get_contacts(); _, u = w.get_state()
    assert (cnt == 4).all() and (w.get_flags() == 0).all()
    lam = np.array([[c["impulse"] for c in ce[:4]] for ce in con])
    assert np.allclose(lam[:, :, 2].sum(1), anymal.total_mass() * G * DT, rtol=2e-4)
    assert np.all(np.hypot(lam[:, :, 0], lam[:, :, 1]) <= 0.8 * lam[:, :, 2] * (1 + 1e-5) + 1e-7)
    assert np.abs(u).max() < 2e-3
    assert {int(c) for c in con[0][:4]["collision"]} == set(anymal.collision_indices("_foot"))
    w.close()


def test_joint_limit_stops_the_pendulum_inelastically(built_lib):
    urdf = PENDULUM_URDF.format(l=0.5, m=1.0).replace('lower="-10" upper="10"', 'lower="-0.3" upper="0.2"')
    _, w = world(urdf, gravity=[0, 0, 0], mode=0)
    u0 = np.zeros(7); u0[6] = 1.0
    w.set_state(tile([0, 0, 0, 1, 0, 0, 0, 0.0]), tile(u0))
    w.integrate(70)
    q, u = w.get_state()
    assert np.allclose(u[:, 6], 1.0, atol=1e-5) and np.allclose(q[:, 7], 70 * DT, atol=1e-5)     # still inside the range
    w.integrate(50)
    q, u = w.get_state(); cnt, _ = w.get_contacts()
    assert (cnt == 0).all()                                   # limit rows are not contacts
    assert (q[:, 7] > 0.2).all() and (q[:, 7] < 0.2 + 1.5 * DT).all() and np.abs(u[:, 6]).max() < 1e-5
    tau = np.zeros((N, 7), np.float32); tau[:, 6] = 3.0
    w.set_generalized_force(tau); w.integrate(40)
    q, u = w.get_state()
    assert (q[:, 7] < 0.2 + 1.5 * DT).all() and np.abs(u[:, 6]).max() < 1e-5
    tau[:, 6] = -3.0
    w.set_generalized_force(tau); w.integrate(1)
    _, u = w.get_state()
    assert np.allclose(u[:, 6], -3.0 / 0.25 * DT, rtol=1e-4)
    w.close()


def test_restitution_bounces_the_ball(built_lib):
    _, w = world(sphere_urdf(2.0, 0.1))
    w.set_default_material(0.8, 0.5, 0.2)
    w.set_state(tile([0, 0, 0.6, 1, 0, 0, 0]), tile(np.zeros(6)))
    bounced, uz = 0, 0.0
    for _ in range(1200):
        w.integrate(1)
        _, u = w.get_state(); cnt, _ = w.get_contacts()
        if cnt[0] and uz < -0.2:
            assert np.allclose(u[:, 2], -0.5 * uz, atol=2e-5)
            bounced += 1
        elif cnt[0] and uz <= 0:
            assert np.abs(u[:, 2]).max() < 2e-5
        uz = float(u[0, 2])
    q, u = w.get_state()
    assert bounced >= 3 and np.abs(u[:, 2]).max() < 2e-5 and np.abs(q[:, 2] - 0.1).max() < 2e-3
    w.close()


def test_two_materials_on_one_body_decelerate_at_the_closed_form_rate(built_lib):
    """rsb_set_collision_materials: per-primitive friction against the terrain (the resolved setMaterialPairProp table)."""
    from test_oracle_kat import DUMBBELL, dumbbell_friction_force
    m_, L, r, mu_f, mu_r = 4.0, 0.4, 0.1, 0.9, 0.2
    _, w = world(DUMBBELL)
    w.set_collision_materials(mu=np.array([mu_f, mu_r]))
    w.set_state(tile([0, 0, r - 1e-6, 1, 0, 0, 0]), tile([2.0, 0, 0, 0, 0, 0]))
    F = dumbbell_friction_force(m_, L, r, mu_f, mu_r)
    w.integrate(20)
    _, u = w.get_state(); prev = u[:, 0].copy()
    for k in range(20):
        w.integrate(1)
        _, u = w.get_state(); cnt, con = w.get_contacts()
        assert (cnt == 2).all()
        assert np.allclose(u[:, 0] - prev, -F / m_ * DT, atol=2e-5), (k, (u[:, 0] - prev).mean(), -F / m_ * DT)
        lam = np.array([[c[0]["impulse"], c[1]["impulse"]] for c in con])
        assert np.allclose(np.hypot(lam[:, 0, 0], lam[:, 0, 1]), mu_f * lam[:, 0, 2], rtol=2e-5)
        assert np.allclose(np.hypot(lam[:, 1, 0], lam[:, 1, 1]), mu_r * lam[:, 1, 2], rtol=2e-5)
        prev = u[:, 0].copy()
    # back to the world's default material: both spheres slide at the default mu
    w.set_collision_materials()
    w.set_state(tile([0, 0, r - 1e-6, 1, 0, 0, 0]), tile([2.0, 0, 0, 0, 0, 0]))
    w.integrate(30); _, u0 = w.get_state(); w.integrate(1); _, u1 = w.get_state()
    assert np.allclose(u1[:, 0] - u0[:, 0], -0.8 * G * DT, atol=2e-5)
    w.close()


def test_sphere_beside_a_ridge_touches_the_edge_not_the_flank(built_lib):
    """Closest-feature sphere x height-map narrow phase through the C-ABI (see the oracle KAT of the same name)."""
    from test_oracle_kat import ridge_map
    r = 0.3
    _, w = world(sphere_urdf(2.0, r))
    w.add_height_map(*ridge_map())
    w.set_state(tile([2.1, 2.5, 1.25, 1, 0, 0, 0]), tile(np.zeros(6)))
    w.integrate(1)
    cnt, con = w.get_contacts()
    assert (cnt == 1).all()
    d = np.hypot(0.1, 0.25)
    assert abs(con[0][0]["depth"] - (r - d)) < 2e-6
    assert np.allclose(con[0][0]["normal"], np.array([0.1, 0.0, 0.25]) / d, atol=2e-6)
    h = np.zeros((5, 5), np.float32); h[2, 2] = 0.5
    w.add_height_map(5, 5, 4.0, 4.0, 2.0, 2.0, h)
    w.set_state(tile([2.0, 2.0, 0.7, 1, 0, 0, 0]), tile(np.zeros(6)))
    w.integrate(1)
    cnt, con = w.get_contacts()
    assert (cnt == 1).all() and abs(con[5][0]["depth"] - 0.1) < 2e-6 and np.allclose(con[5][0]["normal"], [0, 0, 1], atol=2e-6)
    w.close()


def test_sphere_just_past_a_sharp_convex_ridge_touches_the_ridge(built_lib):
    """VERDICT r04 #4a on the device (oracle KAT of the same name): the centre is above the surface but below the extended plane of the flank it has
    just left - the closest feature, the ridge line, is the contact (rounds 1-4: the plane of the face under the centre, twice the depth, a normal
    pointing sideways); a centre below the surface still falls back to the face under it."""
    from test_oracle_kat import sharp_ridge_map
    r = 0.15
    _, w = world(sphere_urdf(2.0, r))
    w.add_height_map(*sharp_ridge_map())
    w.set_state(tile([2.05, 2.1, 1.1, 1, 0, 0, 0]), tile(np.zeros(6)))
    w.integrate(1)
    cnt, con = w.get_contacts()
    d = np.hypot(0.05, 0.1)
    assert (cnt == 1).all()
    for e in (0, 17, 63):
        assert abs(con[e][0]["depth"] - (r - d)) < 2e-6 and np.allclose(con[e][0]["normal"], np.array([0.05, 0.0, 0.1]) / d, atol=3e-6)
    nz = 1.0 / np.sqrt(17.0)
    w.set_state(tile([2.05, 2.1, 0.75, 1, 0, 0, 0]), tile(np.zeros(6)))
    w.integrate(1)
    cnt, con = w.get_contacts()
    assert (cnt == 1).all() and abs(con[9][0]["depth"] - (r + 0.05 * nz)) < 3e-6 and np.allclose(con[9][0]["normal"], [4 * nz, 0, nz], atol=3e-6)
    w.close()


def test_fixed_base_pendulum_period(built_lib):
    from test_oracle_kat import FIXED_PENDULUM
    l = 0.5
    mod, w = world(FIXED_PENDULUM.format(l=l, m=1.0), mode=0)
    assert mod.blob.fixed_base == 1
    g0 = np.array([0, 0, 0, 1, 0, 0, 0, 0.05]); u0 = np.zeros(7); u0[:6] = 0.3
    w.set_state(tile(g0), tile(u0))
    zero, prev, t = [], 0.05, 0.0
    for k in range(600):
        w.integrate(4)
        q, u = w.get_state()
        t += 4 * DT
        if prev > 0 >= q[0, 7]: zero.append(t - 4 * DT * q[0, 7] / (q[0, 7] - prev))
        prev = q[0, 7]
    assert np.allclose(q[:, :7], [0, 0, 0, 1, 0, 0, 0], atol=1e-7) and np.abs(u[:, :6]).max() < 1e-7
    assert abs(np.diff(zero).mean() / (2 * np.pi * np.sqrt(l / G)) - 1) < 1e-2
    w.close()


def test_cylinder_rests_on_its_rims(built_lib):
    """Rim primitives of a <cylinder> through the C-ABI (see the oracle KAT): lying at height R on two contacts, tilted: one."""
    from test_oracle_kat import CYLINDER, _quat_y
    R, L, m_ = 0.1, 0.5, 3.0
    _, w = world(CYLINDER)
    w.set_state(tile([0, 0, R - 1e-5] + _quat_y(np.pi / 2)), tile(np.zeros(6)))
    w.integrate(30)
    cnt, con = w.get_contacts(); _, u = w.get_state()
    assert (cnt == 2).all() and np.abs(u).max() < 1e-4
    imp = np.array([[c[0]["impulse"][2], c[1]["impulse"][2]] for c in con])
    assert np.allclose(imp.sum(axis=1), m_ * G * DT, rtol=1e-5)
    a = np.pi / 6
    drop = L / 2 * np.cos(a) + R * np.sin(a)
    w.set_state(tile([0, 0, drop - 2e-3] + _quat_y(a)), tile(np.zeros(6)))
    w.integrate(1)
    cnt, con = w.get_contacts()
    assert (cnt == 1).all() and abs(con[3][0]["depth"] - 2e-3) < 2e-6
    assert np.allclose(con[3][0]["position"][:2], [-(L / 2) * np.sin(a) + R * np.cos(a), 0.0], atol=2e-6)
    w.close()


def test_self_collision_folds_the_hand_onto_the_torso(built_lib):
    """The oracle KAT's three-link chain through the C-ABI: two flagged entries with opposite normals and impulses, the
    overlap never grows, and the trajectory follows the oracle's."""
    from test_oracle_kat import FOLDER
    mod, w = world(FOLDER)
    o = Oracle(mod.blob)
    w.set_gravity([0, 0, 0]); o.p.gravity[2] = 0.0
    kp = np.array([0] * 6 + [40.0, 40.0]); kd = np.array([0] * 6 + [2.0, 2.0])
    pt = np.array([0, 0, 0, 0, 0, 0, 0, 0.3, 3.1])
    w.set_pd_gains(kp, kd); w.set_pd_target(tile(pt), tile(np.zeros(8)))
    q = np.array([0, 0, 1.0, 1, 0, 0, 0, 0.0, 2.0]); u = np.zeros(8)
    w.set_state(tile(q), tile(u))
    touched, first, worst = 0, None, 0.0
    for k in range(400):
        w.integrate(1)
        q, u, con, _, _ = o.step(q, u, kp, kd, pt, np.zeros(8))
        cnt, dc = w.get_contacts()
        assert (cnt == len(con)).all()
        if len(con):
            touched += 1
            d = dc[2][:2]
            assert list(d["collision"]) == [0 | 0x10000, 1 | 0x20000] and list(d["body"]) == [0, 2]
            assert np.allclose(d["normal"][0], -d["normal"][1]) and np.allclose(d["impulse"][0], -d["impulse"][1])
            assert d["depth"][0] == d["depth"][1] and np.allclose(d["position"][0], d["position"][1])
            first = d["depth"][0] if first is None else first
            worst = max(worst, d["depth"][0])
    qd, ud = w.get_state()
    assert touched > 100 and worst <= first + 2e-4       # (+ the creep of the block's 1e-4 compliance, see the oracle KAT)
    assert np.abs(qd[2] - q).max() < 2e-3 and np.abs(ud[2] - u).max() < 2e-2      # 400 steps of a PD-driven sliding contact, fp32 vs fp64
    w.close()


def test_self_collision_with_the_normal_along_world_x(built_lib):
    """The oracle KAT's clapping hands through the C-ABI: a self-collision whose normal is exactly +-x (the axis the contact
    frame projects) must not produce NaNs; normals reported as +-x, hands stop at touching distance."""
    from test_oracle_kat import CLAPPER
    mod, w = world(CLAPPER, gravity=[0, 0, 0])
    kp = np.array([0] * 6 + [200.0, 200.0]); kd = np.array([0] * 6 + [10.0, 10.0])
    pt = np.array([0, 0, 0, 0, 0, 0, 0, 0.3, -0.3])
    w.set_pd_gains(kp, kd); w.set_pd_target(tile(pt), tile(np.zeros(8)))
    w.set_state(tile([0, 0, 1.0, 1, 0, 0, 0, 0.0, 0.0]), tile(np.zeros(8)))
    touched = 0
    for k in range(600):
        w.integrate(1)
        if k % 20 == 19:
            cnt, dc = w.get_contacts()
            if cnt[0]:
                touched += 1
                assert (cnt == 2).all()
                assert np.allclose(dc[5][0]["normal"], [-1, 0, 0], atol=1e-6) and np.allclose(dc[5][1]["normal"], [1, 0, 0], atol=1e-6)
    q, u = w.get_state()
    assert np.isfinite(q).all() and np.isfinite(u).all() and (w.get_flags() == 0).all()
    gap = (0.2 + q[:, 8]) - (-0.2 + q[:, 7])
    assert touched > 15 and (gap > 0.08).all() and (gap < 0.1 + 1e-6).all() and np.abs(u[:, 6:]).max() < 1e-2
    w.close()


@pytest.mark.parametrize("which", ["capsule over a ridge", "box on a bump"])
def test_sampled_colliders_carry_the_body_on_its_middle(built_lib, which):
    """Sampled colliders through the C-ABI (rsb_model_from_urdf_string_sampled; the oracle KAT of the same name): the sampled log / slab
    is carried by the ridge / bump under its middle from the first step on, the unsampled one (end spheres / corners) falls past it."""
    from test_oracle_kat import LOG_URDF, SLAB_URDF, _bump_map, _ridge_map
    if which.startswith("capsule"):
        urdf, hm, z0, mass, fine = LOG_URDF, _ridge_map(), 0.3 + 0.05 - 1e-4, 4.0, 0.1
    else:
        urdf, hm, z0, mass, fine = SLAB_URDF, _bump_map(), 0.2 + 0.05 - 1e-4, 6.0, 0.2
    out = {}
    for spacing in (0.0, fine):
        m = Model(urdf_string=urdf, sample_spacing=spacing)
        w = BatchedWorld(m, N)
        w.add_height_map(65, 65, 3.2, 3.2, 0.0, 0.0, hm)
        w.set_state(tile([0, 0, z0, 1, 0, 0, 0.0]), tile(np.zeros(6)))
        w.integrate(1)
        cnt, con = w.get_contacts()
        first = (cnt.copy(), con.copy())
        w.integrate(39)
        q, _ = w.get_state()
        assert (w.get_flags() == 0).all()
        out[spacing] = (q.copy(), first)
        w.close()
    q_plain, (cnt_plain, _) = out[0.0]
    q_samp, (cnt, con) = out[fine]
    assert (cnt_plain == 0).all() and (q_plain[:, 2] < z0 - 0.04).all()
    assert (cnt > 0).all() and (np.abs(q_samp[:, 2] - z0) < 2e-3).all()
    for e in (0, N - 1):
        c = con[e][:cnt[e]]
        assert np.abs(c["position"][:, :2]).max() < 0.11
        assert abs(c["impulse"][:, 2].sum() - mass * G * DT) < 1e-3 * mass * G * DT


@pytest.mark.parametrize("mu", [0.8, 0.0])
def test_ball_in_a_valley_rests_on_both_flanks(built_lib, mu):
    """rsb_set_heightmap_contacts(2) through the C-ABI (the oracle KAT of the same name): two contacts of one primitive, the second one
    flagged RSB_CONTACT_SECOND, the ball at rest; with one contact per primitive it rattles."""
    from test_oracle_kat import valley_map
    from raisimlib_amd._capi import RSB_CONTACT_SECOND
    r, slope, mass = 0.3, 0.5, 2.0
    al = np.arctan(slope)
    res = {}
    for hc in (1, 2):
        _, w = world(sphere_urdf(mass, r))
        w.add_height_map(33, 33, 12.8, 12.8, 0.0, 0.0, valley_map(slope=slope))
        w.set_heightmap_contacts(hc)
        w.set_collision_materials(np.array([mu]), np.array([0.0]), np.array([0.0]))
        w.set_state(tile([0.0, 0.1, r / np.cos(al) - 1e-4, 1, 0, 0, 0.0]), tile(np.zeros(6)))
        w.integrate(60)
        vmax = 0.0
        for k in range(35):
            w.integrate(4)
            vmax = max(vmax, np.abs(w.get_state()[1]).max())
        q, u = w.get_state(); cnt, con = w.get_contacts()
        assert (w.get_flags() == 0).all()
        res[hc] = (q, cnt, con, vmax)
        w.close()
    q, cnt, con, vmax = res[2]
    assert (cnt == 2).all() and (con[:, 0]["collision"] == 0).all() and (con[:, 1]["collision"] == RSB_CONTACT_SECOND).all()
    assert vmax < 2e-4 and np.abs(q[:, 0]).max() < 1e-4 and np.abs(q[:, 2] - r / np.cos(al)).max() < 3e-4
    c = con[5][:2]
    nrm = c["normal"][np.argsort(c["normal"][:, 0])]
    assert np.allclose(nrm, [[-np.sin(al), 0, np.cos(al)], [np.sin(al), 0, np.cos(al)]], atol=2e-5)
    assert abs(c["impulse"][:, 2].sum() - mass * G * DT) < 2e-4 * mass * G * DT + 1e-6
    if mu == 0.0:
        lam_n = np.einsum("ij,ij->i", c["impulse"], c["normal"])
        assert np.allclose(lam_n, mass * G * DT / (2 * np.cos(al)), rtol=1e-3)
    assert (res[1][1] == 1).all() and res[1][3] > 5e-3


def test_cylinder_lying_across_a_ridge_rests_on_its_barrel(built_lib):
    """rsb_set_capsule_contacts for a CYLINDER (two rim primitives paired in the blob): the barrel carries it across the ridge (oracle KAT of the
    same name), equal to the oracle's contact"""
    from test_oracle_kat import CYL_LOG_URDF, _ridge_map
    from raisimlib_amd._capi import RSB_CONTACT_CAPSULE
    m, w = world(CYL_LOG_URDF)
    o = Oracle(m.blob)
    hm = _ridge_map()
    w.add_height_map(65, 65, 3.2, 3.2, 0.0, 0.0, hm); o.set_heightmap(65, 65, 3.2, 3.2, 0.0, 0.0, hm)
    w.set_capsule_contacts(True); o.p.hm_capsule = 1
    gc = tile([0.0, -0.3, 0.3 + 0.05 - 1e-3, 1, 0, 0, 0.0])
    gc[:, 0] = np.linspace(-0.2, 0.2, N)
    w.set_state(gc, tile(np.zeros(6)))
    w.integrate(1)
    q, u = w.get_state(); cnt, con = w.get_contacts()
    w.close()
    assert (cnt == 1).all() and (con[:, 0]["collision"] == RSB_CONTACT_CAPSULE).all()
    assert np.abs(con[:, 0]["position"][:, 0]).max() < 0.012 and np.abs(con[:, 0]["normal"][:, 2] - 1.0).max() < 5e-3
    for e in (0, N // 2, N - 1):
        qo, uo, co, _, _ = o.step(gc[e].astype(np.float32).astype(np.float64), np.zeros(6))
        assert len(co) == 1 and np.abs(co["position"][0] - con[e, 0]["position"]).max() < 2e-4 and np.abs(uo - u[e]).max() < 2e-5


def test_capsule_lying_across_a_ridge_rests_on_its_cylinder(built_lib):
    """rsb_set_capsule_contacts through the C-ABI (the oracle KAT of the same name): a 0.6 m capsule lying across a 0.3 m ridge rests on
    it with its cylinder - ONE contact flagged RSB_CONTACT_CAPSULE on the first end sphere's id, on the ridge line, wherever along the
    capsule the ridge sits - and is carried there; with the option off (a capsule = its two end spheres) the same log falls through."""
    from test_oracle_kat import LOG_URDF, _ridge_map
    from raisimlib_amd._capi import RSB_CONTACT_CAPSULE
    mass = 4.0
    res = {}
    for on in (False, True):
        _, w = world(LOG_URDF)
        w.add_height_map(65, 65, 3.2, 3.2, 0.0, 0.0, _ridge_map())
        w.set_capsule_contacts(on)
        w.set_collision_materials(np.array([0.8, 0.8]), np.zeros(2), np.zeros(2))
        gc = tile([0.0, 0.4, 0.3 + 0.05 - 1e-3, 1, 0, 0, 0.0])
        gc[:, 0] = np.linspace(-0.22, 0.22, N)                         # the ridge under a different point of every env's capsule
        w.set_state(gc, tile(np.zeros(6)))
        w.integrate(1)
        q, u = w.get_state(); cnt, con = w.get_contacts()
        res[on] = (q, u, cnt, con, gc)
        w.close()
    q0, u0, cnt0, _, _ = res[False]
    assert (cnt0 == 0).all() and np.abs(u0[:, 2] + G * DT).max() < 1e-6  # end spheres only: free fall
    q, u, cnt, con, gc = res[True]
    assert (cnt == 1).all() and (con[:, 0]["collision"] == RSB_CONTACT_CAPSULE).all() and (con[:, 0]["body"] == 0).all()
    assert np.abs(con[:, 0]["position"][:, 0]).max() < 0.012 and np.abs(con[:, 0]["position"][:, 1] - 0.4).max() < 1e-5
    assert np.abs(con[:, 0]["normal"][:, 2] - 1.0).max() < 5e-3 and np.abs(con[:, 0]["depth"] - 1e-3).max() < 2e-4      # (the located point is within 0.65 % of the capsule's length of the ridge line: the normal tilts by up to 0.08 rad)
    # the contact point stops (v_z + (w x r)_z = 0); centred envs carry the full weight, off-centre ones tip towards their heavy side
    lever = con[:, 0]["position"][:, 0] - gc[:, 0]
    assert np.abs(u[:, 2] - u[:, 4] * lever).max() < 2e-5
    mid = np.abs(gc[:, 0]) < 0.004
    assert mid.any() and np.abs(con[mid, 0]["impulse"][:, 2] - mass * G * DT).max() < 2e-3 * mass * G * DT
    off = np.abs(gc[:, 0]) > 0.05
    assert (u[off, 4] * gc[off, 0] > 0).all()


@pytest.mark.parametrize("case", ["slab on a plateau", "slab on a peak", "beam edge across a ridge"])
def test_box_rests_on_a_face_or_an_edge_between_its_corners(built_lib, case):
    """rsb_set_capsule_contacts for BOXES through the C-ABI (the oracle KAT of the same name): a slab lying on a plateau / on one raised
    terrain vertex, a beam balanced on a long edge across a ridge - ONE contact flagged RSB_CONTACT_CAPSULE on the first corner's id, at
    the plateau's centroid / AT the vertex / at the crossing of the two edges, normal up, exact depth; a different offset of the box in
    every env; with the option off the eight corners hang in the air and the box falls."""
    from test_oracle_kat import SLAB_URDF, BEAM_URDF, _bump_map, _peak_map, _ridge_map
    from raisimlib_amd._capi import RSB_CONTACT_CAPSULE
    rng = np.random.default_rng(5)
    if case == "slab on a plateau":
        urdf, hm, half = SLAB_URDF, _bump_map(), 0.05
        gc = tile([0.0, 0.0, 0.2 + half - 1e-3, 1, 0, 0, 0.0]); gc[:, :2] = rng.uniform(-0.1, 0.1, (N, 2)); gc[0, :2] = 0.0
        where, top = (0.0, 0.0), 0.2
    elif case == "slab on a peak":
        urdf, hm, half = SLAB_URDF, _peak_map(), 0.05
        gc = tile([0.0, 0.0, 0.2 + half - 1e-3, 1, 0, 0, 0.0]); gc[:, :2] = rng.uniform(-0.25, 0.25, (N, 2))
        where, top = (0.0, 0.0), 0.2
    else:
        urdf, hm, half = BEAM_URDF, _ridge_map(), 0.03 * np.sqrt(2.0)
        a = np.pi / 4
        gc = tile([0.0, 0.33, 0.3 + half - 1e-3, np.cos(a / 2), np.sin(a / 2), 0, 0.0]); gc[:, 0] = np.linspace(-0.25, 0.25, N)
        where, top = (0.0, 0.33), 0.3
    res = {}
    for on in (False, True):
        m, w = world(urdf)
        w.add_height_map(65, 65, 3.2, 3.2, 0.0, 0.0, hm)
        w.set_capsule_contacts(on)
        w.set_state(gc, tile(np.zeros(6)))
        w.integrate(1)
        res[on] = (w.get_state(), w.get_contacts(), w.get_flags())
        w.close()
    (_, u0), (cnt0, _), _ = res[False]
    assert (cnt0 == 0).all() and np.abs(u0[:, 2] + G * DT).max() < 1e-6
    (q, u), (cnt, con), flags = res[True]
    assert (flags == 0).all() and (cnt == 1).all() and (con[:, 0]["collision"] == RSB_CONTACT_CAPSULE).all() and (con[:, 0]["body"] == 0).all()
    pos, nrm, dep = con[:, 0]["position"], con[:, 0]["normal"], con[:, 0]["depth"]
    assert np.abs(pos[:, 0] - where[0]).max() < 2e-6 and np.abs(pos[:, 1] - where[1]).max() < 2e-6 and np.abs(pos[:, 2] - (top - 1e-3)).max() < 2e-6
    assert np.abs(nrm[:, 2] - 1.0).max() < 1e-6 and np.abs(dep - 1e-3).max() < 2e-6          # exact up to fp32 coordinates
    # the contact point stops; the oracle's step for a few envs
    rc = pos - gc[:, :3]
    assert np.abs((u[:, :3] + np.cross(u[:, 3:], rc))[:, 2]).max() < 2e-5
    o = Oracle(m.blob); o.set_heightmap(65, 65, 3.2, 3.2, 0.0, 0.0, hm); o.p.hm_capsule = 1
    for e in (0, N // 3, N - 1):
        qo, uo, co, _, _ = o.step(gc[e].astype(np.float32).astype(np.float64), np.zeros(6))
        assert len(co) == 1 and np.abs(co["position"][0] - pos[e]).max() < 2e-6 and np.abs(uo - u[e]).max() < 2e-5
    if case == "slab on a plateau":
        assert abs(con[0, 0]["impulse"][2] - 6.0 * G * DT) < 1e-3 * 6.0 * G * DT and np.abs(u[0]).max() < 1e-5     # centred: carried, at rest


@pytest.mark.parametrize("scheme,theta", [("semi_implicit", 1.0), ("euler", 0.0), ("trapezoid", 0.5)])
def test_integration_schemes(anymal, scheme, theta):
    """rsb_set_integration_scheme through the C-ABI: the free-fall closed forms of the oracle KAT, one-step parity of the quadruped on
    the ground against the oracle with the same theta (RUNGE_KUTTA_4: the tests below)."""
    _, w = world(sphere_urdf(2.0, 0.1))
    w.set_integration_scheme(scheme)
    n = 40
    w.set_state(tile([0, 0, 5.0, 1, 0, 0, 0.0]), tile([0.3, 0, 0, 0, 0, 2.0]))
    w.integrate(n)
    q, u = w.get_state()
    k2 = {1.0: n * (n + 1) / 2, 0.0: n * (n - 1) / 2, 0.5: n * n / 2}[theta]
    assert np.abs(q[:, 2] - (5.0 - G * DT * DT * k2)).max() < 2e-6 and np.abs(u[:, 2] + G * DT * n).max() < 2e-6
    ang = 2.0 * DT * n
    assert np.allclose(q[:, 3:], [np.cos(ang / 2), 0, 0, np.sin(ang / 2)], atol=2e-6)
    w.close()
    from test_gpu_parity import standing_states, f32
    gc, gv = standing_states(N, seed=5, z=(0.45, 0.6), vel=0.5)
    kp, kd = workload.anymal_gains()
    wa = BatchedWorld(anymal, N)
    wa.set_integration_scheme(scheme)
    wa.set_pd_gains(kp, kd); wa.set_pd_target(gc, np.zeros((N, 18))); wa.set_state(gc, gv)
    wa.integrate(1)
    q1, u1 = wa.get_state()
    wa.close()
    o = Oracle(anymal.blob)
    o.p.integ_theta = theta
    r = o.step_batch(f32(gc), f32(gv), 1, kp.astype(np.float64), kd.astype(np.float64), f32(gc), np.zeros((N, 18)), None, lam_warm=o.new_warm_state(N))
    assert np.abs(q1 - r["q"]).max() < 5e-6 and np.abs(u1 - r["u"]).max() < 2e-3
    if theta != 1.0:      # ... and the scheme matters: not the semi-implicit positions
        o1 = Oracle(anymal.blob)
        r1 = o1.step_batch(f32(gc), f32(gv), 1, kp.astype(np.float64), kd.astype(np.float64), f32(gc), np.zeros((N, 18)), None, lam_warm=o1.new_warm_state(N))
        assert np.abs(r1["q"] - r["q"]).max() > 1e-4


TOP_URDF = """<?xml version="1.0"?>
<robot name="top">
  <link name="top">
    <inertial><origin xyz="0 0 0"/><mass value="1.5"/>
      <inertia ixx="0.02" ixy="0.003" ixz="-0.002" iyy="0.05" iyz="0.004" izz="0.09"/></inertial>
  </link>
</robot>
"""


def _top_energy(q, u, I_body):
    w_, x, y, z = q[3:7]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w_ * z), 2 * (x * z + w_ * y)],
                  [2 * (x * y + w_ * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w_ * x)],
                  [2 * (x * z - w_ * y), 2 * (y * z + w_ * x), 1 - 2 * (x * x + y * y)]])
    wb = R.T @ u[3:6]
    return 0.5 * 1.5 * u[:3] @ u[:3] + 0.5 * wb @ I_body @ wb


def test_runge_kutta_4_free_spin_of_an_asymmetric_top_conserves_energy_to_fourth_order(built_lib):
    """IntegrationScheme::RUNGE_KUTTA_4 (rsb_rk4.hip): a free asymmetric top (full inertia tensor, no gravity, no contact)
    spinning about an axis close to its unstable one.  Its kinetic energy and its angular momentum are constants of the motion; the drift of both
    over the same time span falls by ~16x when the time step halves (fourth order; the one-evaluation schemes: first order, shown beside it), down
    to fp32's floor."""
    I = np.array([[0.02, 0.003, -0.002], [0.003, 0.05, 0.004], [-0.002, 0.004, 0.09]])
    q0 = np.array([0, 0, 5.0, np.cos(0.3), np.sin(0.3) * 0.6, np.sin(0.3) * 0.8, 0.0])
    u0 = np.array([0.2, -0.1, 0.3, 6.0, 9.0, -4.0])
    drift = {}
    for scheme, dts in (("runge_kutta_4", (0.02, 0.01)), ("semi_implicit", (0.01,))):
        for dt in dts:
            m, w = world(TOP_URDF, gravity=[0, 0, 0])
            w.set_time_step(dt)
            w.set_integration_scheme(scheme)
            w.set_state(tile(q0), tile(u0))
            e0 = _top_energy(q0, u0, I)
            w.integrate(int(round(0.4 / dt)))
            q, u = w.get_state()
            assert np.ptp(q, axis=0).max() == 0.0                       # all 64 envs bit-identical
            drift[(scheme, dt)] = abs(_top_energy(q[0].astype(np.float64), u[0].astype(np.float64), I) - e0) / e0
            assert abs(np.linalg.norm(q[0, 3:7]) - 1.0) < 1e-6 and np.allclose(q[0, :3], q0[:3] + 0.4 * u0[:3], atol=1e-5)
            w.close()
    big, small, semi = drift[("runge_kutta_4", 0.02)], drift[("runge_kutta_4", 0.01)], drift[("semi_implicit", 0.01)]
    assert small < 2e-5 and small < semi / 50, drift             # far below the one-evaluation scheme at the same step
    assert big / max(small, 2e-7) > 6.0, drift                   # ~16 in exact arithmetic; fp32's floor flattens the small one


def test_runge_kutta_4_pendulum_matches_the_exact_period(built_lib):
    """... and with a joint, gravity and the PD spring off: a pendulum released at 1 rad.  After one exact period (elliptic integral) it is back at the
    start to 1e-4 rad with RUNGE_KUTTA_4 at dt = 5 ms, where the semi-implicit scheme is 30x further off."""
    from scipy.special import ellipk
    l, th0 = 0.5, 1.0
    T = 4.0 * np.sqrt(l / G) * ellipk(np.sin(th0 / 2) ** 2)
    dt = T / 300
    off = {}
    for scheme in ("runge_kutta_4", "semi_implicit"):
        m, w = world(PENDULUM_URDF.format(l=l, m=1.0))
        w.set_time_step(dt)
        w.set_integration_scheme(scheme)
        w.set_pd_gains(np.zeros(m.nv, np.float32), np.zeros(m.nv, np.float32))
        gc = np.zeros(m.nq); gc[3] = 1.0; gc[7] = th0
        w.set_state(tile(gc), tile(np.zeros(m.nv)))
        w.integrate(300)
        q, u = w.get_state()
        off[scheme] = max(abs(q[0, 7] - th0), abs(u[0, 6]) * np.sqrt(l / G))
        w.close()
    assert off["runge_kutta_4"] < 2e-4 and off["semi_implicit"] > 20 * off["runge_kutta_4"], off
