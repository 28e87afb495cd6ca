"""URDF subset loader (host logic, CPU): topology, fixed-joint merging, capsules, error behaviour."""
import numpy as np
import pytest

from common import sphere_urdf


def test_anymal_topology(anymal):
    b = anymal.blob
    assert (b.nb, b.nq, b.nv, b.depth) == (13, 19, 18, 4)
    assert list(b.parent[:13]) == [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11]
    assert list(b.level[:13]) == [0, 1, 2, 3] + [1, 2, 3] * 3
    assert anymal.body_names()[:4] == ["base", "LF_HIP", "LF_THIGH", "LF_SHANK"]
    assert anymal.joint_index("RH_KFE") == 12 and anymal.body_index("RF_THIGH") == 5
    # 4 base spheres + 4 x (thigh capsule = 2 spheres, knee, foot); the box on the base is ignored
    assert b.ncol == 20
    assert anymal.collision_indices("_foot") == [7, 11, 15, 19]


def test_fixed_joint_merging_conserves_mass_and_com(anymal):
    b = anymal.blob
    assert abs(anymal.total_mass() - (19.2 + 2.4 + 4 * (2.781 + 3.071 + 0.58 + 0.25))) < 1e-9
    assert abs(b.mass[0] - 21.6) < 1e-12          # base + rigidly attached top_shell
    assert abs(b.mass[3] - 0.83) < 1e-12          # shank + foot
    # shank+foot com = mass-weighted mean of the two link coms (foot frame offset by the fixed joint)
    com = (0.58 * np.array([0.05, 0.007, -0.12]) + 0.25 * (np.array([0.08795, 0.01305, -0.33797]) + [0, 0, 0.01])) / 0.83
    assert np.allclose(b.com[3][:], com, atol=1e-12)
    # merged inertia must be symmetric positive definite
    for i in range(b.nb):
        I = b.inertia[i]
        M = np.array([[I[0], I[1], I[2]], [I[1], I[3], I[4]], [I[2], I[4], I[5]]])
        assert np.all(np.linalg.eigvalsh(M) > 0)


def test_rotated_fixed_frame_inertia(built_lib):
    """A fixed child with a rotated joint/inertial frame: merged inertia equals the hand-computed tensor."""
    from raisimlib_amd import Model
    urdf = """<robot name="t"><link name="a"><inertial><origin xyz="0 0 0"/><mass value="1"/>
      <inertia ixx="1" ixy="0" ixz="0" iyy="1" iyz="0" izz="1"/></inertial></link>
      <link name="b"><inertial><origin xyz="0 0 0" rpy="0 0 0"/><mass value="2"/>
      <inertia ixx="1" ixy="0" ixz="0" iyy="2" iyz="0" izz="3"/></inertial></link>
      <joint name="j" type="fixed"><origin xyz="1 0 0" rpy="0 0 1.5707963267948966"/><parent link="a"/><child link="b"/></joint></robot>"""
    m = Model(urdf_string=urdf)
    b = m.blob
    assert b.nb == 1 and abs(b.mass[0] - 3) < 1e-12
    com = np.array([2.0 / 3.0, 0, 0])
    assert np.allclose(b.com[0][:], com)
    Ia = np.eye(3) + 1 * (com @ com * np.eye(3) - np.outer(com, com))
    d = np.array([1.0, 0, 0]) - com
    Ib = np.diag([2.0, 1.0, 3.0]) + 2 * (d @ d * np.eye(3) - np.outer(d, d))   # yaw 90deg swaps xx/yy
    I = Ia + Ib
    got = b.inertia[0]
    assert np.allclose([got[0], got[3], got[5], got[1], got[2], got[4]], [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]], atol=1e-12)


def test_capsule_becomes_two_end_spheres(anymal):
    b = anymal.blob
    names = anymal.collision_names()
    i = names.index("LF_thigh_capsule/top")
    assert names[i + 1] == "LF_thigh_capsule/bottom"
    assert b.col_body[i] == b.col_body[i + 1] == 2
    top, bot = np.array(b.col_pos[i][:]), np.array(b.col_pos[i + 1][:])
    assert np.allclose(top - bot, [0, 0, 0.16]) and np.allclose((top + bot) / 2, [0, 0.06, -0.14])
    assert b.col_radius[i] == b.col_radius[i + 1] == 0.045


def test_atlas_topology(atlas):
    b = atlas.blob
    assert (b.nb, b.nq, b.nv) == (31, 37, 36)
    assert b.depth == 11
    assert len(atlas.collision_indices("_foot_0")) == 2


def test_single_body_model(built_lib):
    from raisimlib_amd import Model
    m = Model(urdf_string=sphere_urdf())
    assert (m.nb, m.nq, m.nv, m.ncol) == (1, 7, 6, 1)


@pytest.mark.parametrize("urdf,msg", [
    ("<robot><link name='a'></robot>", "closes <link>"),
    ("<robot><link name='a'>", "missing </link>"),
    ("<notrobot/>", "expected <robot>"),
    ("<robot name='r'><link name='a'/><link name='a'/></robot>", "duplicate link"),
    ("<robot name='r'><link name='a'/><joint name='j' type='revolute'><parent link='a'/><child link='zz'/></joint></robot>", "unknown link"),
    ("<robot name='r'><link name='a'/><link name='b'/></robot>", "more than one root"),
    ("<robot name='r'><link name='a'/><link name='b'/><joint name='j' type='planar'><parent link='a'/><child link='b'/></joint></robot>", "unsupported joint type"),
    ("<robot name='r'><link name='world'/><link name='b'/><joint name='j' type='fixed'><parent link='world'/><child link='b'/></joint></robot>", "fixed-base"),
    ("<robot name='r'><link name='a'><inertial><mass value='1'/><inertia ixx='1' iyy='1' izz='1'/></inertial></link><link name='b'/>"
     "<joint name='j' type='revolute'><parent link='a'/><child link='b'/><axis xyz='0 0 1'/></joint></robot>", "no mass"),
])
def test_loader_errors_are_reported_not_crashed(built_lib, urdf, msg):
    from raisimlib_amd import Model, RsbError
    with pytest.raises(RsbError, match=msg):
        Model(urdf_string=urdf)


def test_missing_file(built_lib):
    from raisimlib_amd import Model, RsbError
    with pytest.raises(RsbError, match="cannot open"):
        Model(urdf_path="/nonexistent/robot.urdf")


BOX_URDF = """<robot name="crate"><link name="crate">
 <inertial><origin xyz="0 0 0"/><mass value="4"/><inertia ixx="0.1" ixy="0" ixz="0" iyy="0.15" iyz="0" izz="0.2"/></inertial>
 <collision><origin xyz="0.1 0 0" rpy="0 0 0.5"/><geometry><box size="0.6 0.4 0.2"/></geometry></collision>
 <collision name="pipe"><origin xyz="0 0 0.3"/><geometry><cylinder radius="0.05" length="0.5"/></geometry></collision>
 <collision><geometry><mesh filename="x.stl"/></geometry></collision>
</link></robot>"""


def test_box_becomes_its_corners_and_cylinder_its_inscribed_capsule(built_lib):
    from raisimlib_amd import Model
    m = Model(urdf_string=BOX_URDF)
    b = m.blob
    assert m.ncol == 10
    pos = np.array([list(b.col_pos[i]) for i in range(10)]); rad = np.array([b.col_radius[i] for i in range(10)])
    assert np.all(rad[:8] == 0) and np.allclose(rad[8:], 0.05)
    c, s_ = np.cos(0.5), np.sin(0.5)
    want = {(round(0.1 + c * x - s_ * y, 9), round(s_ * x + c * y, 9), z) for x in (-0.3, 0.3) for y in (-0.2, 0.2) for z in (-0.1, 0.1)}
    got = {(round(p[0], 9), round(p[1], 9), round(p[2], 9)) for p in pos[:8]}
    assert got == want
    assert np.allclose(sorted(pos[8:, 2]), [0.3 - 0.2, 0.3 + 0.2]) and np.allclose(pos[8:, :2], 0)   # segment = length - 2 r
    names = m.collision_names()
    assert names[0].endswith("/c0") and names[7].endswith("/c7") and names[8] == "pipe/top"


def test_crate_rests_on_its_four_bottom_corners(built_lib):
    """Oracle KAT for the box collider: a crate dropped flat settles on its 4 bottom corners, which carry m g dt."""
    from raisimlib_amd import Model
    from common import Oracle
    m = Model(urdf_string=BOX_URDF.replace(' rpy="0 0 0.5"', "").replace('xyz="0.1 0 0"', 'xyz="0 0 0"'))
    o = Oracle(m.blob)
    q = np.array([0, 0, 0.1 - 1e-4, 1, 0, 0, 0.0]); u = np.zeros(6)
    for _ in range(40):
        q, u, con, it, fl = o.step(q, u)
    assert len(con) == 4 and set(con["collision"]) == {0, 1, 2, 3}          # the corners with z = -0.1
    assert abs(con["impulse"][:, 2].sum() - 4 * 9.81 * 0.0025) < 1e-8 and np.abs(u).max() < 1e-7
