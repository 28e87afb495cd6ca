"""Analytic known-answer tests that pin the CPU oracle (parity with RaiSim itself is unpinned: SURVEY.md §8c)."""
import numpy as np
import pytest

from common import PENDULUM_URDF, Oracle, sphere_urdf

G = 9.81
DT = 0.0025


def make(urdf):
    from raisimlib_amd import Model
    m = Model(urdf_string=urdf)
    return m, Oracle(m.blob)


def test_free_fall_matches_discrete_closed_form(built_lib):
    """Semi-implicit Euler under gravity: v_n = -g n dt, z_n = z0 - g dt^2 n(n+1)/2 (exact for this scheme)."""
    _, o = make(sphere_urdf())
    q = np.array([0.3, -0.2, 50.0, 1, 0, 0, 0.0])
    u = np.zeros(6)
    n = 400
    for _ in range(n):
        q, u, con, _, _ = o.step(q, u)
        assert len(con) == 0
    assert abs(u[2] + G * n * DT) < 1e-10
    assert abs(q[2] - (50.0 - G * DT * DT * n * (n + 1) / 2)) < 1e-9
    assert np.allclose(q[:2], [0.3, -0.2]) and np.allclose(q[3:], [1, 0, 0, 0])


def test_sphere_rest_impulse_is_m_g_dt(built_lib):
    m_, r = 2.0, 0.1
    _, o = make(sphere_urdf(m_, r))
    q = np.array([0, 0, r - 1e-4, 1, 0, 0, 0.0])
    u = np.zeros(6)
    for _ in range(5):
        q, u, con, _, _ = o.step(q, u)
    assert len(con) == 1
    assert np.allclose(con["impulse"][0], [0, 0, m_ * G * DT], atol=1e-12)
    assert np.allclose(u, 0, atol=1e-12)
    assert np.allclose(con["normal"][0], [0, 0, 1]) and abs(con["position"][0][2] - (q[2] - r)) < 1e-12


def test_sliding_friction_decelerates_at_mu_g(built_lib):
    """While the contact slips, the centre of mass decelerates at mu*g (first steps of a fast sliding sphere)."""
    m_, r, mu = 2.0, 0.1, 0.8
    _, o = make(sphere_urdf(m_, r))
    q = np.array([0, 0, r - 1e-5, 1, 0, 0, 0.0])
    u = np.array([3.0, 0, 0, 0, 0, 0])
    for k in range(10):
        q, u2, con, _, _ = o.step(q, u)
        assert abs((u2[0] - u[0]) + mu * G * DT) < 1e-9, k
        lam = con["impulse"][0]
        assert abs(np.hypot(lam[0], lam[1]) - mu * lam[2]) < 1e-9          # on the cone boundary
        assert lam[0] < 0 and abs(lam[1]) < 2e-7                           # opposes the sliding direction
        u = u2


SLED = """<robot name="sled"><link name="sled">
 <inertial><origin xyz="0 0 0"/><mass value="5"/><inertia ixx="0.2" ixy="0" ixz="0" iyy="0.3" iyz="0" izz="0.4"/></inertial>
 <collision><origin xyz="0.3 0.2 0"/><geometry><sphere radius="0.05"/></geometry></collision>
 <collision><origin xyz="0.3 -0.2 0"/><geometry><sphere radius="0.05"/></geometry></collision>
 <collision><origin xyz="-0.3 0.2 0"/><geometry><sphere radius="0.05"/></geometry></collision>
 <collision><origin xyz="-0.3 -0.2 0"/><geometry><sphere radius="0.05"/></geometry></collision>
</link></robot>"""


@pytest.mark.parametrize("angle_deg,sticks", [(30.0, True), (36.0, True), (42.0, False), (50.0, False)])
def test_stick_slip_threshold_on_incline(built_lib, angle_deg, sticks):
    """A four-point sled on a plane with tilted gravity sticks iff tan(theta) < mu = 0.8 (theta* = 38.66 deg);
    above it, it accelerates at g (sin(theta) - mu cos(theta)).  The per-contact minimum-energy slip rule is exact
    Coulomb only for contacts without normal/tangential coupling; the sled's contact points sit below its centre of
    mass, which costs ~0.2% of the friction force, i.e. up to ~2% of the (small) net acceleration."""
    _, o = make(SLED)
    th = np.radians(angle_deg)
    o.p.gravity[0], o.p.gravity[1], o.p.gravity[2] = G * np.sin(th), 0.0, -G * np.cos(th)
    q = np.array([0, 0, 0.05 - 1e-5, 1, 0, 0, 0.0])
    u = np.zeros(6)
    n = 200
    for _ in range(n):
        q, u, con, it, _ = o.step(q, u)
        assert len(con) == 4
    if sticks:
        assert np.abs(u).max() < 1e-4
    else:
        a = G * (np.sin(th) - 0.8 * np.cos(th))
        assert abs(u[0] - a * n * DT) < 2e-2 * a * n * DT + 1e-6
        mu_eff = (np.sin(th) - u[0] / (n * DT) / G) / np.cos(th)
        assert abs(mu_eff - 0.8) < 2e-3
        assert abs(u[2]) < 1e-6 and np.abs(u[3:]).max() < 1e-4


def test_pendulum_spring_period(built_lib):
    """Zero gravity, joint spring via the PD controller (kd = 0): harmonic oscillator with T = 2 pi sqrt(I/kp)."""
    l, m_ = 0.5, 1.0
    _, o = make(PENDULUM_URDF.format(l=l, m=m_))
    o.p.gravity[2] = 0.0
    kp = np.zeros(7); kd = np.zeros(7); kp[6] = 40.0
    I = m_ * l * l + 1e-9
    T = 2 * np.pi * np.sqrt(I / kp[6])
    q = np.array([0, 0, 0, 1, 0, 0, 0, 0.1]); u = np.zeros(7)
    pt = np.zeros(8); pt[3] = 1.0
    dtg = np.zeros(7)
    th, t = [], []
    for k in range(int(5 * T / DT)):
        q, u, _, _, _ = o.step(q, u, kp, kd, pt, dtg)
        th.append(q[7]); t.append((k + 1) * DT)
    th = np.array(th); t = np.array(t)
    # upward zero crossings -> period
    idx = np.where((th[:-1] < 0) & (th[1:] >= 0))[0]
    tc = t[idx] + DT * (-th[idx]) / (th[idx + 1] - th[idx])
    periods = np.diff(tc)
    assert len(periods) >= 3
    assert np.allclose(periods, T, rtol=2e-3)
    assert abs(np.abs(th).max() - 0.1) < 2e-3


def test_pendulum_gravity_period_on_heavy_anchor(built_lib):
    """Real gravity: the 1e9 kg anchor free-falls, so relative to it the bob feels no gravity and must not swing;
    with the anchor's fall cancelled by an upward feed-forward force the bob swings with the large-angle period."""
    l, m_ = 0.5, 1.0
    mdl, o = make(PENDULUM_URDF.format(l=l, m=m_))
    th0 = 0.2
    q = np.array([0, 0, 0, 1, 0, 0, 0, th0]); u = np.zeros(7)
    tau = np.zeros(7); tau[2] = (1e9 + m_) * G          # hold the anchor up
    o.p.control_mode = 0
    th, t = [], []
    T_small = 2 * np.pi * np.sqrt(l / G)
    T = T_small * (1 + th0 ** 2 / 16 + 11 * th0 ** 4 / 3072)
    for k in range(int(4.2 * T / DT)):
        q, u, _, _, _ = o.step(q, u, None, None, None, None, tau)
        th.append(q[7]); t.append((k + 1) * DT)
    th = np.array(th); t = np.array(t)
    idx = np.where((th[:-1] < 0) & (th[1:] >= 0))[0]
    tc = t[idx] + DT * (-th[idx]) / (th[idx + 1] - th[idx])
    assert np.allclose(np.diff(tc), T, rtol=3e-3)
    assert abs(q[2]) < 1e-3                              # anchor stayed put


def test_momentum_and_energy_of_torque_free_flight(anymal):
    """No gravity, no actuation, no contact: momentum and energy are conserved up to the integrator's O(dt) error
    (the drift over a fixed time must halve when dt halves, and be small in absolute terms)."""
    o = Oracle(anymal.blob)
    o.p.gravity[2] = 0.0
    o.p.control_mode = 0
    rng = np.random.default_rng(3)
    q0 = np.zeros(19); q0[2] = 10; q0[3] = 1; q0[7:] = rng.uniform(-0.5, 0.5, 12)
    u0 = rng.normal(size=18) * 0.5
    drift = []
    for dt, n in [(0.0025, 400), (0.00125, 800)]:
        o.p.dt = dt
        q, u = q0.copy(), u0.copy()
        P0, L0 = o.momentum(q, u)
        E0 = sum(o.energy(q, u))
        for _ in range(n):
            q, u, con, _, _ = o.step(q, u)
            assert len(con) == 0
        P1, L1 = o.momentum(q, u)
        E1 = sum(o.energy(q, u))
        drift.append((np.abs(P1 - P0).max(), np.abs(L1 - L0).max(), abs(E1 - E0)))
        if dt == 0.0025:
            assert drift[0][0] < 1e-4 * np.abs(P0).max() and drift[0][1] < 2e-3 * np.abs(L0).max() and drift[0][2] < 1e-4 * E0
    ratio = np.array(drift[0]) / np.array(drift[1])
    assert np.all(ratio > 1.8) and np.all(ratio < 2.2)


def test_standing_anymal_carries_its_weight(anymal):
    """After settling on stiff legs the foot impulses sum to m g dt and stay inside the friction cone."""
    from raisimlib_amd import workload
    o = Oracle(anymal.blob)
    kp = np.zeros(18); kd = np.zeros(18); kp[6:] = 400.0; kd[6:] = 10.0
    q = np.zeros(19); q[2] = 0.60; q[3] = 1; q[7:] = workload.ANYMAL_NOMINAL_JOINTS
    u = np.zeros(18); pt = q.copy(); dtg = np.zeros(18)
    for _ in range(1500):
        q, u, con, it, fl = o.step(q, u, kp, kd, pt, dtg)
    assert fl == 0 and len(con) == 4 and set(con["collision"]) == {7, 11, 15, 19}
    lam = con["impulse"]
    assert abs(lam[:, 2].sum() - anymal.total_mass() * G * DT) < 1e-4
    assert np.all(np.hypot(lam[:, 0], lam[:, 1]) <= 0.8 * lam[:, 2] + 1e-9)
    assert np.abs(u).max() < 1e-3


def test_joint_limit_stops_the_pendulum_inelastically(built_lib):
    """Zero gravity, hinge with range [-0.3, 0.2]: a bob swinging at +1 rad/s stops at the upper limit (overshoot below
    one step), is held there against a feed-forward torque, and leaves freely when the torque reverses."""
    urdf = PENDULUM_URDF.format(l=0.5, m=1.0).replace('lower="-10" upper="10"', 'lower="-0.3" upper="0.2"')
    _, o = make(urdf)
    o.p.gravity[2] = 0.0
    o.p.control_mode = 0
    q = np.array([0, 0, 0, 1, 0, 0, 0, 0.0]); u = np.zeros(7); u[6] = 1.0
    for k in range(120):
        q, u, con, _, _ = o.step(q, u)
        assert len(con) == 0                                  # limit rows are not contacts
        if (k + 1) * DT < 0.19:
            assert abs(u[6] - 1.0) < 1e-9
    assert 0.2 < q[7] < 0.2 + 1.5 * DT and abs(u[6]) < 1e-9
    tau = np.zeros(7); tau[6] = 3.0                           # pushes into the limit: nothing moves
    for _ in range(40):
        q, u, _, _, _ = o.step(q, u, None, None, None, None, tau)
    assert 0.2 < q[7] < 0.2 + 1.5 * DT and abs(u[6]) < 1e-9
    tau[6] = -3.0                                             # pulls away: free again, acceleration tau / (m l^2)
    q, u, _, _, _ = o.step(q, u, None, None, None, None, tau)
    assert abs(u[6] + 3.0 / (1.0 * 0.25 + 1e-9) * DT) < 1e-6


def test_restitution_bounces_the_ball(built_lib):
    """Newton restitution e = 0.5: the step in which the dropped sphere touches down returns it with -e x its approach
    speed; below the threshold speed the contact is inelastic."""
    _, o = make(sphere_urdf(2.0, 0.1))
    o.p.restitution, o.p.res_threshold = 0.5, 0.2
    q = np.array([0, 0, 0.6, 1, 0, 0, 0.0]); u = np.zeros(6)
    bounced = 0
    for _ in range(2000):
        uz = u[2]
        q, u, con, _, _ = o.step(q, u)
        if len(con) and uz < -0.2:
            assert abs(u[2] + 0.5 * uz) < 1e-9
            bounced += 1
        elif len(con) and uz <= 0:
            assert abs(u[2]) < 1e-9                      # slow touch-down: inelastic, the ball stays down
    assert bounced >= 3 and abs(u[2]) < 1e-9 and abs(q[2] - 0.1) < 1e-3


# two spheres of DIFFERENT materials on one rigid body, sliding along the line that joins them (x): the friction below the
# centre of mass pitches the body, which shifts load onto the leading sphere.  With N_f + N_r = m g and the pitch balance
# (N_f - N_r) L = F r (no pitch acceleration while both touch), F = mu_f N_f + mu_r N_r has the closed form below.
DUMBBELL = """<robot name="dumbbell"><link name="bar">
 <inertial><origin xyz="0 0 0"/><mass value="4"/><inertia ixx="0.02" ixy="0" ixz="0" iyy="0.5" iyz="0" izz="0.5"/></inertial>
 <collision name="front"><origin xyz="0.4 0 0"/><geometry><sphere radius="0.1"/></geometry><material name="rubber"/></collision>
 <collision name="rear"><origin xyz="-0.4 0 0"/><geometry><sphere radius="0.1"/></geometry><material name="steel"/></collision>
</link></robot>"""


def dumbbell_friction_force(m_, L, r, mu_f, mu_r):
    d = (mu_f + mu_r) * m_ * G * r / (4 * L) / (1 - (mu_f - mu_r) * r / (2 * L))
    return mu_f * (m_ * G / 2 + d) + mu_r * (m_ * G / 2 - d)


def test_material_names_are_read_from_the_urdf(built_lib):
    m, _ = make(DUMBBELL)
    assert m.collision_materials() == ["rubber", "steel"]
    m2, _ = make(sphere_urdf())
    assert m2.collision_materials() == ["default"]


def test_two_materials_on_one_body_decelerate_at_the_closed_form_rate(built_lib):
    """Per-primitive friction (World::setMaterialPairProp resolved against the terrain's material): each contact slides on its
    own cone; the deceleration is the closed-form total friction / m."""
    m_, L, r, mu_f, mu_r = 4.0, 0.4, 0.1, 0.9, 0.2
    _, o = make(DUMBBELL)
    o.set_collision_materials(mu=np.array([mu_f, mu_r]))
    q = np.array([0, 0, r - 1e-6, 1, 0, 0, 0.0])
    u = np.array([2.0, 0, 0, 0, 0, 0])
    F = dumbbell_friction_force(m_, L, r, mu_f, mu_r)
    for k in range(40):
        q, u2, con, _, _ = o.step(q, u)
        assert len(con) == 2
        if k >= 20:      # the load transfer has settled (the pitch rate is zero again)
            assert abs((u2[0] - u[0]) + F / m_ * DT) < 2e-6, (k, u2[0] - u[0], -F / m_ * DT)
            lam = con["impulse"]
            assert abs(np.hypot(lam[0][0], lam[0][1]) - mu_f * lam[0][2]) < 1e-9     # each contact on ITS cone
            assert abs(np.hypot(lam[1][0], lam[1][1]) - mu_r * lam[1][2]) < 1e-9
            assert lam[0][2] > lam[1][2]                                              # load moved to the leading sphere
        u = u2


def test_per_primitive_restitution(built_lib):
    """One sphere with restitution 0.5 above a 0.1 m/s threshold bounces at half its impact speed; the default stays inelastic."""
    _, o = make(sphere_urdf(2.0, 0.1))
    o.set_collision_materials(restitution=np.array([0.5]), res_threshold=np.array([0.1]))
    q = np.array([0, 0, 0.1 + 1e-4, 1, 0, 0, 0.0])
    u = np.array([0, 0, -2.0, 0, 0, 0])
    q, u1, con, _, _ = o.step(q, u)                    # still above the ground after this step? -> first contact next step
    for _ in range(3):
        if len(con): break
        q, u1, con, _, _ = o.step(q, u1)
    assert len(con) == 1 and abs(u1[2] - 0.5 * 2.0) < 0.03     # v+ = -e v- (gravity adds g dt per step on the way)


def ridge_map():
    """5 x 5 samples, 1 m cells, centred on (2, 2): a tent ridge of height 1 along y at x = 2 (both flanks are planes at 45 deg)."""
    h = np.zeros((5, 5), np.float32)
    h[:, 2] = 1.0
    return (5, 5, 4.0, 4.0, 2.0, 2.0, h)


def sharp_ridge_map():
    """17 x 17 samples over 4 m x 4 m (0.25 m cells) centred on (2, 2): a tent ridge of height 1 along y at x = 2 whose flanks have slope 4"""
    xs = np.linspace(0.0, 4.0, 17)
    prof = np.maximum(0.0, 1.0 - 4.0 * np.abs(xs - 2.0))
    return (17, 17, 4.0, 4.0, 2.0, 2.0, np.tile(prof[None, :], (17, 1)).astype(np.float32))


def test_sphere_just_past_a_sharp_convex_ridge_touches_the_ridge(built_lib):
    """VERDICT r04 #4a.  A ridge sharper than the sphere is close (flank slope 4): the centre sits 0.05 m past the crest and 0.1 m above it - above
    the surface, but BELOW the extended plane of the flank it has just left (1 + 4 * 0.05 = 1.2 > 1.1).  The closest feature is the ridge line:
    distance sqrt(0.05^2 + 0.1^2), normal along (0.05, 0, 0.1).  Rounds 1-4 decided "outside the terrain" by the plane of the triangle that holds
    the closest point - the far flank's, scanned first - and fell back to the plane of the face under the centre: twice the depth and a normal
    that points sideways (kept as orc_params::hm_plane_test = 1 to show it).  Since round 5 the height field itself decides, as the capsule search's
    samples always did."""
    r = 0.15
    _, o = make(sphere_urdf(2.0, r))
    o.set_heightmap(*sharp_ridge_map())
    q = np.array([2.05, 2.1, 1.1, 1, 0, 0, 0.0])
    _, _, con, _, _ = o.step(q, np.zeros(6))
    d = np.hypot(0.05, 0.1)
    assert len(con) == 1 and abs(con["depth"][0] - (r - d)) < 1e-7
    assert np.allclose(con["normal"][0], np.array([0.05, 0.0, 0.1]) / d, atol=1e-6)
    o.p.hm_plane_test = 1
    _, _, old, _, _ = o.step(q, np.zeros(6))
    nz = 1.0 / np.sqrt(17.0)
    assert len(old) == 1 and abs(old["depth"][0] - (r - (1.1 - 0.8) * nz)) < 1e-6 and np.allclose(old["normal"][0], [4 * nz, 0, nz], atol=1e-6)
    assert old["depth"][0] > 1.9 * con["depth"][0]
    o.p.hm_plane_test = 0
    # a centre BELOW the surface still falls back to the face under it (a zero-radius box corner, a sphere pushed in by more than its radius)
    q = np.array([2.05, 2.1, 0.75, 1, 0, 0, 0.0])
    _, _, con, _, _ = o.step(q, np.zeros(6))
    assert len(con) == 1 and abs(con["depth"][0] - (r + 0.05 * nz)) < 1e-6 and np.allclose(con["normal"][0], [4 * nz, 0, nz], atol=1e-6)


def test_sphere_beside_a_ridge_touches_the_edge_not_the_flank(built_lib):
    """Closest-feature narrow phase: the centre sits 0.1 m beside the crest and 0.25 m above it; the foot of the perpendicular
    onto the flank under the centre lies beyond the crest, so the closest feature is the ridge EDGE: distance
    sqrt(0.1^2 + 0.25^2), normal along (0.1, 0, 0.25) - not the flank's plane (distance 0.2475, normal (1, 0, 1)/sqrt 2)."""
    r = 0.3
    _, o = make(sphere_urdf(2.0, r))
    o.set_heightmap(*ridge_map())
    q = np.array([2.1, 2.5, 1.25, 1, 0, 0, 0.0])
    _, _, con, _, _ = o.step(q, np.zeros(6))
    assert len(con) == 1
    d = np.hypot(0.1, 0.25)
    assert abs(con["depth"][0] - (r - d)) < 1e-9
    assert np.allclose(con["normal"][0], np.array([0.1, 0.0, 0.25]) / d, atol=1e-9)
    # on the flank proper (1 m from the crest) the face is the closest feature: the plane distance and the plane's normal
    q = np.array([3.0, 2.5, 0.0 + 0.25 * np.sqrt(2.0), 1, 0, 0, 0.0])
    _, _, con, _, _ = o.step(q, np.zeros(6))
    assert len(con) == 1 and abs(con["depth"][0] - (r - 0.25)) < 1e-9
    assert np.allclose(con["normal"][0], np.array([1.0, 0.0, 1.0]) / np.sqrt(2.0), atol=1e-9)
    # above a VERTEX of a pyramid (one raised sample): distance to the apex, normal straight up
    h = np.zeros((5, 5), np.float32); h[2, 2] = 0.5
    o.set_heightmap(5, 5, 4.0, 4.0, 2.0, 2.0, h)
    _, _, con, _, _ = o.step(np.array([2.0, 2.0, 0.5 + 0.2, 1, 0, 0, 0.0]), np.zeros(6))
    assert len(con) == 1 and abs(con["depth"][0] - 0.1) < 1e-9 and np.allclose(con["normal"][0], [0, 0, 1], atol=1e-9)
    # a valley floor: the sphere rests between two flanks -> the closer flank's face
    h = np.ones((5, 5), np.float32); h[:, 2] = 0.0
    o.set_heightmap(5, 5, 4.0, 4.0, 2.0, 2.0, h)
    _, _, con, _, _ = o.step(np.array([2.05, 2.5, 0.36, 1, 0, 0, 0.0]), np.zeros(6))
    assert len(con) == 1 and np.allclose(con["normal"][0], np.array([-1.0, 0.0, 1.0]) / np.sqrt(2.0), atol=1e-9)


FIXED_PENDULUM = """<?xml version="1.0"?>
<robot name="arm">
  <link name="world"/>
  <link name="mount"><inertial><origin xyz="0 0 0"/><mass value="0.5"/><inertia ixx="1e-3" ixy="0" ixz="0" iyy="1e-3" iyz="0" izz="1e-3"/></inertial></link>
  <joint name="bolt" type="fixed"><origin xyz="0 0 2"/><parent link="world"/><child link="mount"/></joint>
  <link name="bob">
    <inertial><origin xyz="0 0 -{l}"/><mass value="{m}"/>
      <inertia ixx="1e-9" ixy="0" ixz="0" iyy="1e-9" iyz="0" izz="1e-9"/></inertial>
  </link>
  <joint name="hinge" type="revolute">
    <origin xyz="0 0 0"/><parent link="mount"/><child link="bob"/><axis xyz="0 1 0"/>
    <limit effort="0" velocity="100" lower="-10" upper="10"/>
  </joint>
</robot>
"""


def test_fixed_base_pendulum_period(built_lib):
    """A URDF whose root link is "world" is a fixed-base system (RaiSim's convention): the base never moves, however light
    it is, and a point-mass pendulum bolted to it swings with the small-angle period 2 pi sqrt(l / g)."""
    l, m_ = 0.5, 1.0
    mod, o = make(FIXED_PENDULUM.format(l=l, m=m_))
    assert mod.blob.fixed_base == 1 and mod.nb == 2
    o.p.control_mode = 0
    q = np.array([0, 0, 0, 1, 0, 0, 0, 0.05]); u = np.zeros(7)
    u[:6] = 0.3                                   # a base velocity in the row is ignored
    zero, prev, t = [], q[7], 0.0
    for k in range(2400):
        q, u, _, _, _ = o.step(q, u)
        t += DT
        if prev > 0 >= q[7]: zero.append(t - DT * q[7] / (q[7] - prev))
        prev = q[7]
    assert np.allclose(q[:7], [0, 0, 0, 1, 0, 0, 0], atol=1e-12) and np.abs(u[:6]).max() < 1e-12
    period = np.diff(zero).mean()
    assert abs(period / (2 * np.pi * np.sqrt(l / G)) - 1) < 5e-3


CYLINDER = """<robot name="log"><link name="log">
 <inertial><origin xyz="0 0 0"/><mass value="3"/><inertia ixx="0.07" ixy="0" ixz="0" iyy="0.07" iyz="0" izz="0.015"/></inertial>
 <collision><origin xyz="0 0 0"/><geometry><cylinder radius="0.1" length="0.5"/></geometry></collision>
</link></robot>"""


def _quat_y(angle):
    return [np.cos(angle / 2), 0.0, np.sin(angle / 2), 0.0]


def test_cylinder_rests_on_its_rims_lying_standing_and_tilted(built_lib):
    """A <cylinder> touches a plane with the lowest point of each end-cap rim (not with an inscribed capsule): lying on its
    side it rests at height R on two contacts sharing m g dt; standing it rests at height L/2; tilted by 30 degrees the lower
    cap's rim point is the only contact, at the closed-form depth."""
    R, L, m_ = 0.1, 0.5, 3.0
    _, o = make(CYLINDER)
    # lying on its side (axis along x): both rim points, 2 x m g dt / 2
    q = np.array([0, 0, R - 1e-5] + _quat_y(np.pi / 2)); u = np.zeros(6)
    for _ in range(30):
        q, u, con, _, _ = o.step(q, u)
    assert len(con) == 2 and abs(con["impulse"][:, 2].sum() - m_ * G * DT) < 1e-9 and np.abs(u).max() < 1e-6
    assert np.allclose(sorted(con["position"][:, 0]), [-L / 2, L / 2], atol=1e-6) and np.allclose(con["position"][:, 2], 0, atol=2e-5)
    # standing on its lower cap: one contact (the cap's centre), full weight; nothing from the upper cap
    q = np.array([0, 0, L / 2 - 1e-5, 1, 0, 0, 0.0]); u = np.zeros(6)
    for _ in range(10):
        q, u, con, _, _ = o.step(q, u)
    assert len(con) == 1 and abs(con["impulse"][0][2] - m_ * G * DT) < 1e-9
    # tilted by 30 degrees about y: the lowest rim point of the lower cap is L/2 cos a + R sin a below the centre
    a = np.pi / 6
    drop = L / 2 * np.cos(a) + R * np.sin(a)
    q = np.array([0, 0, drop - 2e-3] + _quat_y(a)); u = np.zeros(6)
    _, _, con, _, _ = o.step(q, u)
    assert len(con) == 1 and abs(con["depth"][0] - 2e-3) < 1e-9
    assert np.allclose(con["position"][0][:2], [-(L / 2) * np.sin(a) + R * np.cos(a), 0.0], atol=1e-9)


FOLDER = """<robot name="folder">
 <link name="torso"><inertial><origin xyz="0 0 0"/><mass value="5"/><inertia ixx="0.05" ixy="0" ixz="0" iyy="0.05" iyz="0" izz="0.05"/></inertial>
  <collision><origin xyz="0 0 0"/><geometry><sphere radius="0.1"/></geometry></collision></link>
 <link name="upper"><inertial><origin xyz="0 0 -0.15"/><mass value="1"/><inertia ixx="0.01" ixy="0" ixz="0" iyy="0.01" iyz="0" izz="0.002"/></inertial></link>
 <link name="lower"><inertial><origin xyz="0 0 -0.15"/><mass value="1"/><inertia ixx="0.01" ixy="0" ixz="0" iyy="0.01" iyz="0" izz="0.002"/></inertial>
  <collision><origin xyz="0 0 -0.3"/><geometry><sphere radius="0.05"/></geometry></collision></link>
 <joint name="shoulder" type="revolute"><origin xyz="0.15 0 0"/><parent link="torso"/><child link="upper"/><axis xyz="0 1 0"/>
  <limit effort="100" velocity="100" lower="-10" upper="10"/></joint>
 <joint name="elbow" type="revolute"><origin xyz="0 0 -0.3"/><parent link="upper"/><child link="lower"/><axis xyz="0 1 0"/>
  <limit effort="100" velocity="100" lower="-10" upper="10"/></joint>
</robot>"""


def test_self_collision_is_an_internal_force(built_lib):
    """A floating three-link chain folds its hand onto its own torso (no gravity, PD drive).  Torso and forearm are not
    parent and child, so their spheres collide: the contact is listed once per body with opposite normals and impulses,
    it stops the penetration, and - being internal - its impulse leaves the system's linear and angular momentum untouched:
    M(q) (u+ - u+ without the contact) = J^T lam has no resultant."""
    mod, o = make(FOLDER)
    _, off = make(FOLDER)
    off.set_self_collision(False)
    assert [tuple(p) for p in o.self_pairs()] == [(0, 1)]
    o.p.gravity[2] = off.p.gravity[2] = 0.0
    kp = np.array([0] * 6 + [40.0, 40.0]); kd = np.array([0] * 6 + [2.0, 2.0])
    q = np.array([0, 0, 1.0, 1, 0, 0, 0, 0.0, 2.0]); u = np.zeros(8)
    pt = np.array([0, 0, 0, 0, 0, 0, 0, 0.3, 3.1]); dtg = np.zeros(8)
    touched, depth_max, depth_first, dmom, dvel = 0, 0.0, None, 0.0, 0.0
    for k in range(800):
        q0 = q.copy()
        _, u_off, con_off, _, _ = off.step(q, u, kp, kd, pt, dtg)
        q, u, con, _, fl = o.step(q, u, kp, kd, pt, dtg)
        assert len(con_off) == 0 and fl == 0
        if len(con):
            touched += 1
            assert len(con) == 2 and list(con["collision"]) == [0 | 0x10000, 1 | 0x20000] and list(con["body"]) == [0, 2]
            assert np.allclose(con["position"][0], con["position"][1]) and con["depth"][0] == con["depth"][1]
            assert np.allclose(con["normal"][0], -con["normal"][1]) and np.allclose(con["impulse"][0], -con["impulse"][1])
            assert con["impulse"][0] @ con["normal"][0] >= -1e-12          # the torso is pushed away from the hand
            assert abs(np.linalg.norm(con["normal"][0]) - 1) < 1e-12
            depth_max = max(depth_max, con["depth"][0])
            depth_first = con["depth"][0] if depth_first is None else depth_first
            (l1, a1), (l0, a0) = o.momentum(q0, u), o.momentum(q0, u_off)
            dmom = max(dmom, np.abs(l1 - l0).max(), np.abs(a1 - a0).max())
            dvel = max(dvel, np.abs(u - u_off).max())
        else:
            assert np.array_equal(u, u_off)
    assert touched > 300                      # the hand stays pressed on the torso
    assert depth_max <= depth_first + 2e-4 < 0.01   # the overlap stays what the first detection found (erp 0; + the creep of the
                                                    # block's 1e-4 compliance: the two bodies are two joints apart, their block is singular)
    assert dvel > 1.0 and dmom < 1e-12        # the contact changes velocities by > 1 (m/s, rad/s) and the momentum by nothing
    off_q = np.array([0, 0, 1.0, 1, 0, 0, 0, 0.0, 2.0]); off_u = np.zeros(8)   # without self-collision the hand sinks into the torso
    for k in range(800):
        off_q, off_u, con, _, _ = off.step(off_q, off_u, kp, kd, pt, dtg)
    assert abs(off_q[8] - 3.1) < 0.05 and abs(off_q[7] - 0.3) < 0.05


CLAPPER = """<?xml version="1.0"?>
<robot name="clapper">
 <link name="torso"><inertial><mass value="4"/><inertia ixx="0.1" ixy="0" ixz="0" iyy="0.1" iyz="0" izz="0.1"/></inertial></link>
 <link name="left"><inertial><mass value="1"/><inertia ixx="0.01" ixy="0" ixz="0" iyy="0.01" iyz="0" izz="0.01"/></inertial>
  <collision><geometry><sphere radius="0.05"/></geometry></collision></link>
 <link name="right"><inertial><mass value="1"/><inertia ixx="0.01" ixy="0" ixz="0" iyy="0.01" iyz="0" izz="0.01"/></inertial>
  <collision><geometry><sphere radius="0.05"/></geometry></collision></link>
 <joint name="jl" type="prismatic"><origin xyz="-0.2 0 0"/><parent link="torso"/><child link="left"/><axis xyz="1 0 0"/>
  <limit effort="1000" velocity="100" lower="-10" upper="10"/></joint>
 <joint name="jr" type="prismatic"><origin xyz="0.2 0 0"/><parent link="torso"/><child link="right"/><axis xyz="1 0 0"/>
  <limit effort="1000" velocity="100" lower="-10" upper="10"/></joint>
</robot>"""


def test_self_collision_with_the_normal_along_world_x(built_lib):
    """Two hands on a common slide clap at yaw 0: the self-collision's normal is (+-1, 0, 0) EXACTLY, the direction the contact
    frame used to project onto the tangent plane (t1 = 0 / |0| -> NaN).  The frame now takes world y there: the state stays
    finite, the normal is reported as +-x, the hands stop at touching distance and the system's momentum is untouched."""
    mod, o = make(CLAPPER)
    assert [tuple(p) for p in o.self_pairs()] == [(0, 1)]
    o.p.gravity[2] = 0.0
    kp = np.array([0] * 6 + [200.0, 200.0]); kd = np.array([0] * 6 + [10.0, 10.0])
    q = np.array([0, 0, 1.0, 1, 0, 0, 0, 0.0, 0.0]); u = np.zeros(8)
    pt = np.array([0, 0, 0, 0, 0, 0, 0, 0.3, -0.3]); dtg = np.zeros(8)
    touched = 0
    for k in range(600):
        q, u, con, _, fl = o.step(q, u, kp, kd, pt, dtg)
        assert np.isfinite(q).all() and np.isfinite(u).all() and fl == 0
        if len(con):
            touched += 1
            assert len(con) == 2 and np.allclose(con["normal"][0], [-1, 0, 0], atol=1e-12) and np.allclose(con["normal"][1], [1, 0, 0], atol=1e-12)
            assert np.allclose(con["impulse"][0], -con["impulse"][1]) and con["impulse"][0][0] <= 1e-12 and np.abs(con["impulse"][0][1:]).max() < 1e-12
    gap = (0.2 + q[8]) - (-0.2 + q[7])                # distance of the two sphere centres along the slide
    assert touched > 300 and 0.08 < gap < 0.1 + 1e-9   # pressed together at ~r + r (erp 0; the hands are two joints apart, their block is
                                                        # singular and carries the 1e-4 compliance: 40 N of PD push creep ~1 cm in 1.5 s)
    lin, ang = o.momentum(q, u)
    assert np.abs(lin).max() < 1e-10 and np.abs(ang).max() < 1e-10 and np.abs(u[6:]).max() < 1e-4   # (creeping at the compliance rate)


def test_ignore_collision_between_removes_the_pair(built_lib):
    mod, o = make(FOLDER)
    ign = np.zeros((3, 3), bool); ign[0, 2] = True
    o.set_self_collision(True, ignore=ign)
    assert len(o.self_pairs()) == 0


def test_self_collision_candidates_of_the_benchmark_models(anymal):
    """The candidate list (what the device sweeps every sub-step): primitives of two different bodies that are not parent and
    child, never two points; 160 pairs on the ANYmal-like model, 138 on the Atlas-like one."""
    from raisimlib_amd import Model, rsc_path
    for model, want in ((anymal, 160), (Model(urdf_path=rsc_path("atlas_like.urdf")), 138)):
        o = Oracle(model.blob)
        pairs = o.self_pairs()
        b = model.blob
        assert len(pairs) == want and len({tuple(p) for p in pairs}) == want
        for i, j in pairs:
            bi, bj = b.col_body[i], b.col_body[j]
            assert i < j and bi != bj and b.parent[bi] != bj and b.parent[bj] != bi
            assert b.col_radius[i] + b.col_radius[j] > 0 and b.col_rim[i] == 0 and b.col_rim[j] == 0


LOG_URDF = """<robot name="log"><link name="log">
 <inertial><mass value="4"/><inertia ixx="0.01" ixy="0" ixz="0" iyy="0.12" iyz="0" izz="0.12"/></inertial>
 <collision><origin rpy="0 1.5707963267948966 0"/><geometry><capsule radius="0.05" length="0.6"/></geometry></collision>
</link></robot>"""
CYL_LOG_URDF = LOG_URDF.replace("capsule", "cylinder")
SLAB_URDF = """<robot name="slab"><link name="slab">
 <inertial><mass value="6"/><inertia ixx="0.1" ixy="0" ixz="0" iyy="0.1" iyz="0" izz="0.2"/></inertial>
 <collision><geometry><box size="0.6 0.6 0.1"/></geometry></collision>
</link></robot>"""


def _ridge_map(n=65, size=3.2, height=0.3, half_width=0.1):
    """a ridge along y at x = 0 (tent profile over one cell on either side), flat elsewhere; n x n samples, cells of size / (n - 1)"""
    xs = np.linspace(-size / 2, size / 2, n)
    prof = np.maximum(0.0, height * (1.0 - np.abs(xs) / half_width))
    return np.tile(prof[None, :], (n, 1)).astype(np.float32)


def _bump_map(n=65, size=3.2, height=0.2):
    h = np.zeros((n, n), np.float32)
    c = n // 2
    h[c - 3:c + 4, c - 3:c + 4] = height          # a plateau of +-0.15 m around the origin (cells of 0.05 m), sloping to the ground within one cell
    return h


@pytest.mark.parametrize("which", ["capsule over a ridge", "box on a bump"])
def test_sampled_colliders_find_the_contact_under_the_middle(built_lib, which):
    """Sampled colliders (rsb_model_from_urdf_*_sampled): a capsule lying ACROSS a ridge touches it with its middle, a slab lying on a
    bump touches it with the middle of its bottom face - where the unsampled primitive sets (two end spheres / eight corners: exact on a
    plane) hang in the air.  The sampled body is carried (a contact under its middle with the weight's impulse m g dt), the unsampled one
    falls through until its ends / corners reach the ground."""
    from raisimlib_amd import Model
    if which.startswith("capsule"):
        urdf, hm, z0, mass = LOG_URDF, _ridge_map(), 0.3 + 0.05 - 1e-4, 4.0
    else:
        urdf, hm, z0, mass = SLAB_URDF, _bump_map(), 0.2 + 0.05 - 1e-4, 6.0
    res = {}
    fine = 0.1 if which.startswith("capsule") else 0.2
    for spacing in (0.0, fine):
        m = Model(urdf_string=urdf, sample_spacing=spacing)
        o = Oracle(m.blob)
        o.set_heightmap(65, 65, 3.2, 3.2, 0.0, 0.0, hm)
        q = np.array([0, 0, z0, 1, 0, 0, 0.0]); u = np.zeros(6)
        first = None
        for k in range(40):
            q, u, con, _, fl = o.step(q, u)
            if first is None and len(con):
                first = (k, con.copy())
        res[spacing] = (m.ncol, q.copy(), first)
    n0, q_plain, first_plain = res[0.0]
    n1, q_samp, first_samp = res[fine]
    assert n1 > n0 and n0 == (2 if which.startswith("capsule") else 8)
    assert first_plain is None and q_plain[2] < z0 - 0.04            # 40 steps of free fall: 0.5 g t^2 = 4.9 cm, nothing touched
    k, con = first_samp
    assert k == 0 and abs(q_samp[2] - z0) < 2e-3                     # carried from the first step on
    assert np.abs(con["position"][:, :2]).max() < 0.11               # under the middle, not at the ends / corners
    assert abs(con["impulse"][:, 2].sum() - mass * 9.81 * 0.0025) < 1e-3 * mass * 9.81 * 0.0025


def test_cylinder_lying_across_a_ridge_rests_on_its_barrel(built_lib):
    """The same search between a CYLINDER's two cap centres (its ends are rim primitives): the barrel of a 0.6 m cylinder lying across the
    ridge carries it - one flagged contact on the ridge line, normal up, the weight's impulse - where its two rims hang in the air."""
    from raisimlib_amd import Model
    m = Model(urdf_string=CYL_LOG_URDF)
    assert m.ncol == 2 and list(m.blob.col_capsule[:2]) == [2, 0] and m.blob.col_rim[0] == 0.05 and m.blob.col_radius[0] == 0.0
    hm, z0, mass, dt = _ridge_map(), 0.3 + 0.05 - 1e-3, 4.0, 0.0025
    for on in (0, 1):
        o = Oracle(m.blob)
        o.set_heightmap(65, 65, 3.2, 3.2, 0.0, 0.0, hm)
        o.p.hm_capsule = on
        q, u, con, _, _ = o.step(np.array([0.05, -0.3, z0, 1, 0, 0, 0.0]), np.zeros(6))
        if not on:
            assert len(con) == 0 and abs(u[2] + 9.81 * dt) < 1e-9
            continue
        assert len(con) == 1 and con["collision"][0] == (0 | 0x80000)
        assert abs(con["position"][0, 0]) < 0.012 and abs(con["position"][0, 1] + 0.3) < 1e-6 and abs(con["normal"][0, 2] - 1.0) < 5e-3
        assert abs(con["depth"][0] - 1e-3) < 2e-4 and 0 < con["impulse"][0, 2] <= mass * 9.81 * dt * (1 + 1e-9)
    # a cylinder too short for an interior sample (length < 2 radius) has no barrel contact; flat ground: the rims alone
    short = Model(urdf_string=CYL_LOG_URDF.replace('length="0.6"', 'length="0.08"'))
    o = Oracle(short.blob)
    o.set_heightmap(65, 65, 3.2, 3.2, 0.0, 0.0, hm); o.p.hm_capsule = 1
    _, _, con, _, _ = o.step(np.array([0.0, 0.0, z0, 1, 0, 0, 0.0]), np.zeros(6))
    assert not (con["collision"] & 0x80000).any()


@pytest.mark.parametrize("shift", [0.0, 0.13, -0.21])
def test_capsule_lying_across_a_ridge_rests_on_its_cylinder(built_lib, shift):
    """Exact capsule x height map (orc_params::hm_capsule, the device's rsb_set_capsule_contacts): a 0.6 m capsule lying ACROSS a 0.3 m
    ridge touches it with its cylinder - wherever along its length the ridge happens to sit - while its two end spheres hang in the
    air.  ONE contact, flagged ORC_CAPSULE on the first end sphere's id, at the ridge line, carrying the weight; off centre the log also
    starts to tip about the ridge (torque m g * shift).  Without the option the same log falls through the ridge.  On flat ground the
    option changes nothing: the two end spheres hold the capsule, no third contact appears between them."""
    from raisimlib_amd import Model
    m = Model(urdf_string=LOG_URDF)
    assert m.ncol == 2 and list(m.blob.col_capsule[:2]) == [2, 0]          # the loader pairs the two end spheres
    hm, z0, mass, dt = _ridge_map(), 0.3 + 0.05 - 1e-3, 4.0, 0.0025      # 1 mm into the ridge (the cylinder's contact needs 0.1 mm more depth than the ends have)
    res = {}
    for on in (0, 1):
        o = Oracle(m.blob)
        o.set_heightmap(65, 65, 3.2, 3.2, 0.0, 0.0, hm)
        o.p.hm_capsule = on
        q = np.array([shift, 0.4, z0, 1, 0, 0, 0.0]); u = np.zeros(6)
        q, u, con, _, fl = o.step(q, u)
        res[on] = (q, u, con)
    q0, u0, con0 = res[0]
    q1, u1, con1 = res[1]
    assert len(con0) == 0 and abs(u0[2] + 9.81 * dt) < 1e-9                # two end spheres: free fall through the ridge
    assert len(con1) == 1 and con1["collision"][0] == (0 | 0x80000) and con1["body"][0] == 0
    assert abs(con1["position"][0, 0]) < 0.012 and abs(con1["position"][0, 1] - 0.4) < 1e-6    # on the ridge line (sampling resolution 1.3 % of 0.6 m), under the axis
    assert abs(con1["normal"][0, 2] - 1.0) < 5e-3 and abs(con1["depth"][0] - 1e-3) < 2e-4
    # the contact stops the point above the ridge (normal velocity 0 after the step); the impulse is the weight's when the ridge is under the centre
    lever = con1["position"][0, 0] - q1[0] + 0.0
    v_point = u1[2] - u1[4] * (con1["position"][0, 0] - shift)             # v_z + (w x r)_z with w about y
    assert abs(v_point) < 1e-6
    if shift == 0.0:
        assert abs(con1["impulse"][0, 2] - mass * 9.81 * dt) < 1e-6 and abs(u1[2]) < 1e-6 and np.abs(u1[3:]).max() < 1e-4
    else:
        assert 0 < con1["impulse"][0, 2] < mass * 9.81 * dt and u1[4] * shift > 0    # tips about the ridge, towards its heavy side
    del lever
    # flat ground: nothing changes
    for on in (0, 1):
        o = Oracle(m.blob)
        o.set_heightmap(65, 65, 3.2, 3.2, 0.0, 0.0, np.zeros((65, 65), np.float32))
        o.p.hm_capsule = on
        q, u, con, _, _ = o.step(np.array([0.8, 0.4, 0.05 - 1e-4, 1, 0, 0, 0.0]), np.zeros(6))
        assert len(con) == 2 and set(con["collision"]) == {0, 1}
        res[("flat", on)] = (q, u)
    assert np.array_equal(res[("flat", 0)][0], res[("flat", 1)][0]) and np.array_equal(res[("flat", 0)][1], res[("flat", 1)][1])


BEAM_URDF = """<robot name="beam"><link name="beam">
 <inertial><mass value="3"/><inertia ixx="0.01" ixy="0" ixz="0" iyy="0.1" iyz="0" izz="0.1"/></inertial>
 <collision><geometry><box size="0.6 0.06 0.06"/></geometry></collision>
</link></robot>"""


def _peak_map(n=65, height=0.2):
    h = np.zeros((n, n), np.float32)
    h[n // 2, n // 2] = height                    # ONE raised vertex: a pyramid one cell (0.05 m) wide, flanks of slope 4
    return h


@pytest.mark.parametrize("case", ["slab on a plateau", "slab on a plateau, off centre", "slab on a peak", "beam edge across a ridge"])
def test_box_rests_on_a_face_or_an_edge_between_its_corners(built_lib, case):
    """Exact box x height map (the same switch as the capsules': orc_params::hm_capsule / rsb_set_capsule_contacts).  The eight corners are
    the exact contact set of a box on a PLANE; on a height map the deepest point can be where a terrain vertex meets a face or where a box
    edge crosses a terrain edge.  Closed forms: (1) a slab lying flat on a plateau is carried by ONE contact at the centroid of the plateau
    vertices under it (all equally deep), normal = the face's, impulse = its weight when the centroid is under the centre of mass, a
    tipping torque otherwise; (2) on a single raised vertex (flanks of slope 4: a point sample 1 mm away from it would miss it) the contact is
    AT the vertex with the exact depth; (3) a beam balanced on one of its long edges across a ridge touches where its edge crosses the
    ridge line, normal = edge x ridge = up.  Without the option all three fall through; on flat ground the option adds nothing."""
    from raisimlib_amd import Model
    dt = 0.0025
    if case.startswith("slab on a plateau"):
        urdf, hm, mass = SLAB_URDF, _bump_map(), 6.0
        xy = (0.0, 0.0) if case == "slab on a plateau" else (0.02, -0.03)
        q0 = np.array([xy[0], xy[1], 0.2 + 0.05 - 1e-3, 1, 0, 0, 0.0])
        where, top = (0.0, 0.0), float(np.float32(0.2))
    elif case == "slab on a peak":
        urdf, hm, mass = SLAB_URDF, _peak_map(), 6.0
        q0 = np.array([0.11, -0.07, 0.2 + 0.05 - 1e-3, 1, 0, 0, 0.0])
        where, top = (0.0, 0.0), float(np.float32(0.2))
    else:
        urdf, hm, mass = BEAM_URDF, _ridge_map(), 3.0
        a = np.pi / 4                                # rolled 45 deg about its long axis (x): one long edge down, at 0.03 sqrt(2) under the axis
        q0 = np.array([0.07, 0.33, 0.3 + 0.03 * np.sqrt(2.0) - 1e-3, np.cos(a / 2), np.sin(a / 2), 0, 0.0])
        where, top = (0.0, 0.33), float(np.float32(0.3))
    m = Model(urdf_string=urdf)
    assert m.ncol == 8 and list(m.blob.col_capsule[:8]) == [-1, 0, 0, 0, 0, 0, 0, 0]       # the loader marks the first corner of the box
    res = {}
    for on in (0, 1):
        o = Oracle(m.blob)
        o.set_heightmap(65, 65, 3.2, 3.2, 0.0, 0.0, hm)
        o.p.hm_capsule = on
        res[on] = o.step(q0.copy(), np.zeros(6))
    _, u0, con0, _, _ = res[0]
    q1, u1, con1, _, fl = res[1]
    assert len(con0) == 0 and abs(u0[2] + 9.81 * dt) < 1e-9                # eight corners in the air: free fall
    assert fl == 0 and len(con1) == 1 and con1["collision"][0] == (0 | 0x80000) and con1["body"][0] == 0
    pos, nrm, dep = con1["position"][0], con1["normal"][0], con1["depth"][0]
    assert abs(pos[0] - where[0]) < 1e-9 and abs(pos[1] - where[1]) < 1e-9 and abs(pos[2] - (q0[2] - (0.05 if urdf is SLAB_URDF else 0.03 * np.sqrt(2.0)))) < 1e-9      # exact: a vertex / a crossing, not a sample
    assert abs(nrm[2] - 1.0) < 1e-12 and abs(dep - (top - pos[2])) < 1e-12 and abs(dep - 1e-3) < 2e-8
    centred = case == "slab on a plateau"
    if centred:
        assert abs(con1["impulse"][0, 2] - mass * 9.81 * dt) < 1e-9 and np.abs(u1).max() < 1e-9           # carried, at rest
    else:
        assert 0 < con1["impulse"][0, 2] < mass * 9.81 * dt          # carried at the contact point, tipping about it towards the heavy side
        r = pos[:2] - q0[:2]
        tip = np.array([-r[1], r[0]]) * -1.0                           # gravity's torque about the contact: (r_com - r_c) x (-m g z) ~ (-(dy), +(dx)) with d = com - contact
        assert np.dot(u1[3:5], tip) > 0
        # the contact point itself stops: v + w x r = 0 along the normal
        rc = np.array([pos[0] - q0[0], pos[1] - q0[1], pos[2] - q0[2]])
        assert abs((u1[:3] + np.cross(u1[3:], rc))[2]) < 1e-7
    # flat ground, lying flat and tilted onto a corner: the corners hold the box, nothing is added
    for quat in ([1, 0, 0, 0.0], [np.cos(0.2), np.sin(0.2) * 0.6, np.sin(0.2) * 0.8, 0.0]):
        outs = []
        for on in (0, 1):
            o = Oracle(m.blob)
            o.set_heightmap(65, 65, 3.2, 3.2, 0.0, 0.0, np.zeros((65, 65), np.float32))
            o.p.hm_capsule = on
            w_, x_, y_, z_ = quat                      # third row of the rotation matrix: the corners' heights
            row = np.array([2 * (x_ * z_ - w_ * y_), 2 * (y_ * z_ + w_ * x_), 1 - 2 * (x_ * x_ + y_ * y_)])
            half = np.array([0.3, 0.3, 0.05]) if urdf is SLAB_URDF else np.array([0.3, 0.03, 0.03])
            zc = float(np.abs(row) @ half)
            q, u, con, _, _ = o.step(np.array([0.8, 0.4, zc - 5e-4] + list(quat)), np.zeros(6))
            assert len(con) >= 1 and not (con["collision"] & 0x80000).any()
            outs.append((q, u))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def valley_map(n=33, size=12.8, slope=0.5):
    """a V-shaped valley along y at x = 0, flanks of the given slope; cells of size / (n - 1) = 0.4 m"""
    xs = np.linspace(-size / 2, size / 2, n)
    return np.tile((slope * np.abs(xs))[None, :], (n, 1)).astype(np.float32)


@pytest.mark.parametrize("mu", [0.8, 0.0])
def test_ball_in_a_valley_rests_on_both_flanks_with_two_contacts_per_primitive(built_lib, mu):
    """orc_params::hm_contacts = 2 (rsb_set_heightmap_contacts): a ball lowered into a V-shaped valley touches BOTH flanks - two contacts
    of one primitive, normals (-+sin a, 0, cos a), the second one flagged ORC_SECOND - and rests: zero velocity, the two impulses carry
    m g dt between them (frictionless: each normal impulse m g dt / (2 cos a)).  With one contact per primitive (the default) the same
    ball rattles from flank to flank."""
    from raisimlib_amd import Model
    r, slope, mass = 0.3, 0.5, 2.0
    al = np.arctan(slope)
    m = Model(urdf_string=sphere_urdf(mass, r))
    out = {}
    for hc in (1, 2):
        o = Oracle(m.blob)
        o.p.hm_contacts, o.p.mu = hc, mu
        o.set_heightmap(33, 33, 12.8, 12.8, 0.0, 0.0, valley_map(slope=slope))
        q = np.array([0.0, 0.1, r / np.cos(al) - 1e-4, 1, 0, 0, 0.0]); u = np.zeros(6)
        vmax = 0.0
        for k in range(200):
            q, u, con, it, fl = o.step(q, u)
            vmax = max(vmax, np.abs(u).max()) if k > 50 else vmax
        out[hc] = (q, u, con, vmax)
    q, u, con, vmax = out[2]
    assert len(con) == 2 and list(con["collision"]) == [0, 0x40000]
    assert vmax < 1e-5 and abs(q[0]) < 1e-6 and abs(q[2] - r / np.cos(al)) < 2e-4
    nrm = con["normal"][np.argsort(con["normal"][:, 0])]
    assert np.allclose(nrm, [[-np.sin(al), 0, np.cos(al)], [np.sin(al), 0, np.cos(al)]], atol=1e-6)
    assert abs(con["impulse"][:, 2].sum() - mass * 9.81 * 0.0025) < 1e-6 and abs(con["impulse"][:, 0].sum()) < 1e-6
    if mu == 0.0:
        lam_n = np.einsum("ij,ij->i", con["impulse"], con["normal"])
        assert np.allclose(lam_n, mass * 9.81 * 0.0025 / (2 * np.cos(al)), rtol=1e-6)
    assert len(out[1][2]) == 1 and out[1][3] > 5e-3            # one contact: the ball keeps rattling between the flanks


@pytest.mark.parametrize("scheme,theta", [("semi_implicit", 1.0), ("euler", 0.0), ("trapezoid", 0.5)])
def test_integration_schemes_in_free_fall_and_free_spin(built_lib, scheme, theta):
    """orc_params::integ_theta (rsb_set_integration_scheme): the velocity update is the same for every scheme, the positions move with
    theta u+ + (1 - theta) u.  A ball in free fall after n steps: z = z0 - g dt^2 n (n + 1) / 2 (semi-implicit), n (n - 1) / 2 (explicit
    Euler), n^2 / 2 (trapezoid: exact); a torque-free spin about z turns the quaternion by w dt per step whatever the scheme."""
    from raisimlib_amd import Model
    m = Model(urdf_string=sphere_urdf(2.0, 0.1))
    o = Oracle(m.blob)
    o.p.integ_theta = theta
    n, dt, g = 40, 0.0025, 9.81
    q = np.array([0, 0, 5.0, 1, 0, 0, 0.0]); u = np.array([0.3, 0, 0, 0, 0, 2.0])
    for k in range(n):
        q, u, con, it, fl = o.step(q, u)
    k2 = {1.0: n * (n + 1) / 2, 0.0: n * (n - 1) / 2, 0.5: n * n / 2}[theta]
    assert abs(q[2] - (5.0 - g * dt * dt * k2)) < 1e-12 and abs(u[2] + g * dt * n) < 1e-12
    assert abs(q[0] - 0.3 * dt * n) < 1e-12
    ang = 2.0 * dt * n
    assert np.allclose(q[3:], [np.cos(ang / 2), 0, 0, np.sin(ang / 2)], atol=1e-12)
