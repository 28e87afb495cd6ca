"""URDF subset loader (host logic, CPU): topology, fixed-joint merging, capsules, error behaviour."""
import os

import numpy as np
import pytest

from common import sphere_urdf
from raisimlib_amd import Model


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


def test_box_becomes_its_corners_and_cylinder_its_two_cap_rims(built_lib):
    from raisimlib_amd import Model
    m = Model(urdf_string=BOX_URDF)
    b = m.blob
    assert m.ncol == 10
    pos = np.array([list(b.col_pos[i]) for i in range(10)]); rad = np.array([b.col_radius[i] for i in range(10)])
    assert np.all(rad == 0)                                        # box corners and rim primitives are points
    assert np.allclose([b.col_rim[i] for i in range(10)], [0] * 8 + [0.05] * 2) and np.allclose([list(b.col_axis[i]) for i in (8, 9)], [[0, 0, 1]] * 2)
    c, s_ = np.cos(0.5), np.sin(0.5)
    want = {(round(0.1 + c * x - s_ * y, 9), round(s_ * x + c * y, 9), z) for x in (-0.3, 0.3) for y in (-0.2, 0.2) for z in (-0.1, 0.1)}
    got = {(round(p[0], 9), round(p[1], 9), round(p[2], 9)) for p in pos[:8]}
    assert got == want
    assert np.allclose(sorted(pos[8:, 2]), [0.3 - 0.25, 0.3 + 0.25]) and np.allclose(pos[8:, :2], 0)   # the centres of the two end caps
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


MESH_URDF = """<robot name="crate"><link name="crate">
 <inertial><origin xyz="0 0 0"/><mass value="3"/><inertia ixx="0.1" ixy="0" ixz="0" iyy="0.1" iyz="0" izz="0.1"/></inertial>
 <collision name="hull"><origin xyz="0 0 0.1"/><geometry><mesh filename="package://crate_description/meshes/{fn}" scale="0.5 0.5 0.5"/></geometry>
   <material name="wood"/></collision>
</link></robot>"""


def _write_box_mesh(dirpath, fn, half=(0.4, 0.3, 0.2), extra_interior=True):
    """A box as OBJ or binary STL (12 triangles), plus a few interior / face points that the thinning must not prefer."""
    import itertools, struct
    hx, hy, hz = half
    V = [(sx * hx, sy * hy, sz * hz) for sx, sy, sz in itertools.product((-1, 1), repeat=3)]
    F = [(0, 1, 3), (0, 3, 2), (4, 6, 7), (4, 7, 5), (0, 4, 5), (0, 5, 1), (2, 3, 7), (2, 7, 6), (0, 2, 6), (0, 6, 4), (1, 5, 7), (1, 7, 3)]
    os.makedirs(dirpath, exist_ok=True)
    path = os.path.join(dirpath, fn)
    if fn.endswith(".obj"):
        with open(path, "w") as f:
            f.write("# box\n")
            for v in V + ([(0, 0, hz), (0.1, 0.0, 0.0)] if extra_interior else []):
                f.write("v %.6f %.6f %.6f\n" % v)
            for a, b, c in F:
                f.write("f %d %d %d\n" % (a + 1, b + 1, c + 1))
    elif fn.endswith(".dae"):
        # a minimal Collada document: centimetres, y-up (what many CAD exporters write): the loader scales to metres and turns to z-up
        zup = [(x, y, z) for x, y, z in V]
        yup = [(x * 100, z * 100, -y * 100) for x, y, z in zup]          # (x, y, z)_zup = (x, -z_yup, y_yup)
        floats = " ".join("%.6f" % c for v in yup for c in v)
        tris = " ".join(str(i) for tri in F for i in tri)
        with open(path, "w") as f:
            f.write(f"""<?xml version="1.0" encoding="utf-8"?>
<COLLADA xmlns="http://www.collada.org/2005/11/COLLADASchema" version="1.4.1">
  <asset><unit name="centimeter" meter="0.01"/><up_axis>Y_UP</up_axis></asset>
  <library_geometries><geometry id="box-mesh" name="box"><mesh>
    <source id="box-mesh-normals"><float_array id="box-mesh-normals-array" count="3">0 0 1</float_array></source>
    <source id="box-mesh-positions"><float_array id="box-mesh-positions-array" count="{3 * len(yup)}">{floats}</float_array>
      <technique_common><accessor source="#box-mesh-positions-array" count="{len(yup)}" stride="3"><param name="X" type="float"/><param name="Y" type="float"/><param name="Z" type="float"/></accessor></technique_common></source>
    <vertices id="box-mesh-vertices"><input semantic="POSITION" source="#box-mesh-positions"/></vertices>
    <triangles count="{len(F)}"><input semantic="VERTEX" source="#box-mesh-vertices" offset="0"/><p>{tris}</p></triangles>
  </mesh></geometry></library_geometries>
</COLLADA>
""")
    else:
        with open(path, "wb") as f:
            f.write(b"binary stl".ljust(80, b" ") + struct.pack("<I", len(F)))
            for a, b, c in F:
                f.write(struct.pack("<12fH", 0, 0, 0, *V[a], *V[b], *V[c], 0))
    return path


@pytest.mark.parametrize("fn", ["crate.obj", "crate.stl", "crate.dae"])
def test_mesh_collision_geometry_becomes_a_point_set(built_lib, tmp_path, fn):
    """<mesh> colliders (OBJ, binary STL, Collada with its <unit> and <up_axis>): package:// URI resolved below the URDF's directory, scale applied, the vertex cloud
    thinned to 8 points - for a box exactly its corners - each a zero-radius primitive carrying the collision's material."""
    pkg = tmp_path / "crate_description"
    _write_box_mesh(str(pkg / "meshes"), fn)
    urdf = pkg / "urdf" / "crate.urdf"
    os.makedirs(urdf.parent)
    urdf.write_text(MESH_URDF.format(fn=fn))
    m = Model(urdf_path=str(urdf))
    assert m.skipped_collisions == 0 and m.ncol == 8
    P = np.array([[m.blob.col_pos[i][k] for k in range(3)] for i in range(8)])
    want = np.array([[sx * 0.2, sy * 0.15, sz * 0.1 + 0.1] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
    order = lambda A: A[np.lexsort(np.round(A, 4).T[::-1])]
    assert np.allclose(order(P), order(want), atol=1e-7)            # (binary STL stores float32 vertices)
    assert all(m.blob.col_radius[i] == 0.0 for i in range(8))
    assert m.collision_names()[0].startswith("hull/m") and set(m.collision_materials()) == {"wood"}
    # a URDF given as a string has no directory to resolve the mesh against: the collider is skipped and counted, not fatal
    m2 = Model(urdf_string=MESH_URDF.format(fn=fn))
    assert m2.ncol == 0 and m2.skipped_collisions == 1


def test_mesh_crate_rests_on_its_four_lowest_vertices(built_lib, tmp_path):
    from common import Oracle
    pkg = tmp_path / "crate_description"
    _write_box_mesh(str(pkg / "meshes"), "crate.obj")
    urdf = pkg / "crate.urdf"
    urdf.write_text(MESH_URDF.format(fn="crate.obj"))
    m = Model(urdf_path=str(urdf))
    o = Oracle(m.blob)
    q = np.array([0, 0, -1e-4, 1, 0, 0, 0.0]); u = np.zeros(6)     # the lowest vertices (z = 0 in the body frame) just below the ground
    for _ in range(20):
        q, u, con, _, _ = o.step(q, u)
    assert len(con) == 4 and abs(con["impulse"][:, 2].sum() - 3 * 9.81 * 0.0025) < 1e-9 and np.abs(u).max() < 1e-6   # (four redundant contacts: solved to the 1e-5 relative threshold)


def test_world_root_link_makes_a_fixed_base_model(built_lib):
    """RaiSim's convention: a root link named "world" = fixed base; a massless root is fine (its inertia is never used)."""
    m = Model(urdf_string="<robot name='r'><link name='world'/><link name='b'><inertial><mass value='2'/><inertia ixx='1' iyy='1' izz='1'/></inertial></link>"
                          "<joint name='j' type='revolute'><parent link='world'/><child link='b'/><axis xyz='0 0 1'/></joint></robot>")
    assert m.blob.fixed_base == 1 and (m.nb, m.nq, m.nv) == (2, 8, 7)
    assert Model(urdf_string=sphere_urdf()).blob.fixed_base == 0


def test_blob_validation_rejects_malformed_collision_fields(built_lib):
    """rsb_model_from_blob is the entry a host language binds: rim / axis / fixed_base / material fields are checked, not trusted"""
    import copy
    from raisimlib_amd._capi import RsbError
    good = Model(urdf_string=sphere_urdf()).blob
    assert Model(blob=good).ncol == good.ncol

    def broken(edit):
        b = copy.deepcopy(good)
        edit(b)
        with pytest.raises(RsbError):
            Model(blob=b)

    broken(lambda b: setattr(b, "fixed_base", 7))
    broken(lambda b: b.col_rim.__setitem__(0, -0.1))
    broken(lambda b: b.col_rim.__setitem__(0, float("nan")))
    broken(lambda b: b.col_radius.__setitem__(0, float("inf")))

    def bad_axis(b):
        b.col_rim[0] = 0.1
        b.col_axis[0][0], b.col_axis[0][1], b.col_axis[0][2] = 0.0, 0.0, 2.0
    broken(bad_axis)

    def unterminated(b):
        b.col_material[0].raw = b"x" * len(b.col_material[0].raw)
    broken(unterminated)


def test_mesh_uri_must_not_climb_out_of_the_probed_directories(built_lib, tmp_path):
    """a <mesh filename> with '..' components is not resolved (the loader probes up to 4 parents of the URDF itself): skipped"""
    (tmp_path / "secret.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\n")
    d = tmp_path / "robot" / "urdf"
    d.mkdir(parents=True)
    urdf = """<robot name="r"><link name="base"><inertial><mass value="1"/><inertia ixx="1" ixy="0" ixz="0" iyy="1" iyz="0" izz="1"/></inertial>
 <collision><geometry><mesh filename="package://pkg/../../../secret.obj"/></geometry></collision>
 <collision><geometry><sphere radius="0.1"/></geometry></collision></link></robot>"""
    (d / "r.urdf").write_text(urdf)
    m = Model(urdf_path=str(d / "r.urdf"))
    assert m.ncol == 1 and m.skipped_collisions == 1


def test_sampled_colliders_fill_capsule_axes_and_box_surfaces(built_lib):
    """rsb_model_from_urdf_*_sampled: with a spacing h a capsule gets spheres of its own radius along its axis, a box zero-radius points
    on the lattice of its edges and faces, no two neighbours further apart than h; spheres, cylinders and meshes are left as they are;
    spacing 0 is the plain loader; too many primitives is reported with the remedy."""
    from raisimlib_amd import Model
    from raisimlib_amd import RsbError, rsc_path
    log = """<robot name="log"><link name="log"><inertial><mass value="1"/><inertia ixx="1" ixy="0" ixz="0" iyy="1" iyz="0" izz="1"/></inertial>
     <collision name="c"><origin xyz="0 0 0.5" rpy="0 1.5707963267948966 0"/><geometry><capsule radius="0.05" length="1.0"/></geometry></collision>
     <collision name="b"><geometry><box size="0.4 0.2 0.1"/></geometry></collision>
     <collision name="s"><geometry><sphere radius="0.1"/></geometry></collision></link></robot>"""
    plain, samp = Model(urdf_string=log), Model(urdf_string=log, sample_spacing=0.25)
    assert plain.ncol == 2 + 8 + 1 and Model(urdf_string=log, sample_spacing=0.0).ncol == plain.ncol
    names = samp.collision_names()
    b = samp.blob
    cap = [i for i, n in enumerate(names) if n.startswith("c/")]
    assert len(cap) == 2 + 3                                            # 1.0 / 0.25 = 4 segments: 3 spheres between the two ends
    xs = sorted(b.col_pos[i][0] for i in cap)
    assert np.allclose(xs, [-0.5, -0.25, 0.0, 0.25, 0.5]) and all(abs(b.col_pos[i][2] - 0.5) < 1e-12 and b.col_radius[i] == 0.05 for i in cap)
    box = [i for i, n in enumerate(names) if n.startswith("b/")]
    pts = np.array([list(b.col_pos[i]) for i in box])
    assert len(box) == 3 * 2 * 2 and all(b.col_radius[i] == 0.0 for i in box)    # lattice 3 x 2 x 2 (0.4 -> 2 cells, 0.2 and 0.1 -> 1): all on the surface
    assert sorted(set(np.round(pts[:, 0], 9))) == [-0.2, 0.0, 0.2]
    assert len({tuple(np.round(p, 9)) for p in pts}) == len(pts)
    fine = Model(urdf_string=log, sample_spacing=0.1)
    pf = np.array([list(fine.blob.col_pos[i]) for i, n in enumerate(fine.collision_names()) if n.startswith("b/")])
    assert len(pf) == 5 * 3 * 2                                         # 4 x 2 x 1 cells, no interior lattice point with one cell in z
    on_surface = (np.abs(np.abs(pf) - [0.2, 0.1, 0.05]) < 1e-12).any(axis=1)
    assert on_surface.all()
    assert [n for n in names if n.startswith("s")] == ["s"]
    an = Model(urdf_path=rsc_path("anymal_c_like.urdf"), sample_spacing=0.1)                  # ANYmal: the leg capsules get mid spheres
    assert an.ncol == 24 and Model(urdf_path=rsc_path("anymal_c_like.urdf")).ncol == 20
    with pytest.raises(RsbError, match="larger spacing"):
        Model(urdf_string=log, sample_spacing=0.01)
    with pytest.raises(RsbError):
        Model(urdf_string=log, sample_spacing=-1.0)


def test_collada_node_transforms_place_the_geometry(built_lib, tmp_path):
    """Collada visual-scene nodes (<translate>, <rotate>, <scale>, nested <matrix>) down to <instance_geometry> are applied to the mesh's
    vertices - a unit cube (z-up metres) instanced under translate(1,0,0) . rotate(z, 90 deg) . scale(2,1,1) and, below it, a child node
    whose <matrix> lifts by 0.5: the collider's points are the transformed corners; a geometry no node instantiates is taken as it is."""
    import itertools
    pkg = tmp_path / "cube_description"
    os.makedirs(pkg / "meshes"); os.makedirs(pkg / "urdf")
    V = [(sx * 0.5, sy * 0.5, sz * 0.5) for sx, sy, sz in itertools.product((-1, 1), repeat=3)]
    floats = " ".join("%.6f" % c for v in V for c in v)
    (pkg / "meshes" / "cube.dae").write_text(f"""<?xml version="1.0" encoding="utf-8"?>
<COLLADA xmlns="http://www.collada.org/2005/11/COLLADASchema" version="1.4.1">
  <asset><unit name="meter" meter="1"/><up_axis>Z_UP</up_axis></asset>
  <library_geometries><geometry id="cube"><mesh>
    <source id="cube-pos"><float_array id="cube-pos-array" count="24">{floats}</float_array></source>
    <vertices id="cube-vtx"><input semantic="POSITION" source="#cube-pos"/></vertices>
  </mesh></geometry></library_geometries>
  <library_visual_scenes><visual_scene id="Scene">
    <node id="outer"><translate>1 0 0</translate><rotate>0 0 1 90</rotate><scale>2 1 1</scale>
      <node id="inner"><matrix>1 0 0 0  0 1 0 0  0 0 1 0.5  0 0 0 1</matrix><instance_geometry url="#cube"/></node>
    </node>
  </visual_scene></library_visual_scenes>
</COLLADA>
""")
    (pkg / "urdf" / "cube.urdf").write_text(MESH_URDF.format(fn="cube.dae").replace('scale="0.5 0.5 0.5"', 'scale="1 1 1"'))
    m = Model(urdf_path=str(pkg / "urdf" / "cube.urdf"))
    assert m.skipped_collisions == 0 and m.ncol == 8
    P = np.array([[m.blob.col_pos[i][k] for k in range(3)] for i in range(8)])
    # v -> T R S M v: lift z by 0.5, scale x by 2, rotate 90 deg about z ((x, y) -> (-y, x)), shift x by 1; then the URDF's collision origin
    base = np.array([[-(y), 2 * x, z + 0.5] for x, y, z in V]) + [1.0, 0.0, 0.0]
    off = P.mean(axis=0) - base.mean(axis=0)                          # the <collision><origin> of MESH_URDF
    order = lambda A: A[np.lexsort(np.round(A, 4).T[::-1])]
    assert np.allclose(order(P - off), order(base), atol=1e-9)
    assert np.allclose(np.ptp(P, axis=0), [1.0, 2.0, 1.0], atol=1e-9)   # the cube became 1 x 2 x 1 (scaled along x, then turned)


def test_mesh_point_budget_keeps_more_hull_vertices(built_lib, tmp_path):
    """rsb_set_mesh_point_budget: a <mesh> collider keeps up to 26 support vertices of its convex hull (default 8).  A 42-vertex geodesic ball: 8 points by
    default, 20 distinct ones with the budget at 20 - every one a vertex of the mesh on its hull (radius 0.3), none an interior point -, 26 at most; the
    budget is refused outside 1 .. 26 and applies to models loaded afterwards."""
    from raisimlib_amd import _capi
    L = _capi.lib()
    # icosahedron subdivided once, projected on the sphere (42 vertices) + interior points the hull must drop
    t = (1.0 + 5 ** 0.5) / 2
    V = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    F = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8),
         (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    V = [np.array(v, float) / np.linalg.norm(v) for v in V]
    mid, F2 = {}, []
    for a, b, c in F:
        m = []
        for i, j in ((a, b), (b, c), (c, a)):
            key = (min(i, j), max(i, j))
            if key not in mid:
                p = V[i] + V[j]
                V.append(p / np.linalg.norm(p)); mid[key] = len(V) - 1
            m.append(mid[key])
        F2 += [(a, m[0], m[2]), (b, m[1], m[0]), (c, m[2], m[1]), (m[0], m[1], m[2])]
    assert len(V) == 42
    P = [0.3 * v for v in V] + [np.array([0.05, 0.02, -0.03]), np.zeros(3)]
    pkg = tmp_path / "ball_description"
    os.makedirs(pkg / "meshes"); os.makedirs(pkg / "urdf")
    with open(pkg / "meshes" / "ball.obj", "w") as f:
        for p in P:
            f.write(f"v {p[0]:.9f} {p[1]:.9f} {p[2]:.9f}\n")
        for a, b, c in F2:
            f.write(f"f {a + 1} {b + 1} {c + 1}\n")
    urdf = pkg / "urdf" / "ball.urdf"
    urdf.write_text("""<robot name="ball"><link name="base"><inertial><mass value="2"/><inertia ixx="0.1" iyy="0.1" izz="0.1" ixy="0" ixz="0" iyz="0"/></inertial>
 <collision name="skin"><geometry><mesh filename="package://ball_description/meshes/ball.obj"/></geometry></collision></link></robot>""")
    try:
        assert Model(urdf_path=str(urdf)).ncol == 8
        assert L.rsb_set_mesh_point_budget(20) == 0
        m = Model(urdf_path=str(urdf))
        assert m.ncol == 20
        Q = np.array([[m.blob.col_pos[i][k] for k in range(3)] for i in range(20)])
        assert np.allclose(np.linalg.norm(Q, axis=1), 0.3, atol=1e-6)                     # hull vertices, not the interior points
        assert len({tuple(np.round(q, 6)) for q in Q}) == 20                              # distinct
        assert all(min(np.linalg.norm(np.array(P[:42]) - q, axis=1)) < 1e-6 for q in Q)   # vertices of the mesh
        assert L.rsb_set_mesh_point_budget(26) == 0 and Model(urdf_path=str(urdf)).ncol <= 26 and Model(urdf_path=str(urdf)).ncol > 20
        assert L.rsb_set_mesh_point_budget(27) != 0 and L.rsb_set_mesh_point_budget(0) != 0
    finally:
        L.rsb_set_mesh_point_budget(8)
    assert Model(urdf_path=str(urdf)).ncol == 8
