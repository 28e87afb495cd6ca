#!/usr/bin/env python3
"""Author the two synthetic stand-in robot descriptions used by the tests and the bench.

The real `rsc/anymal_c/urdf/anymal.urdf` and `rsc/atlas/robot.urdf` of upstream raisimLib are not
in /root/reference (a three-file stub) nor anywhere on this machine, so
these files are SYNTHETIC: same topology and roughly the same dimensions/masses as the public
robots (ANYmal C: floating base + 4 x (HAA, HFE, KFE), spherical feet, ~50 kg; Atlas: 30 actuated
joints, ~150 kg, four corner spheres per foot), authored from scratch.  They exercise the URDF
subset the loader supports, including fixed-joint merging and capsule collision geometry.

Run:  python raisimlib_amd/rsc/make_urdfs.py   (rewrites the two .urdf files next to this script)
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def box_inertia(m, sx, sy, sz):
    return (m / 12.0 * (sy * sy + sz * sz), m / 12.0 * (sx * sx + sz * sz), m / 12.0 * (sx * sx + sy * sy))


def link(name, mass, com, inertia_diag, collisions=(), inertia_rpy=(0, 0, 0), off_diag=(0, 0, 0)):
    ixx, iyy, izz = inertia_diag
    ixy, ixz, iyz = off_diag
    s = [f'  <link name="{name}">']
    s.append("    <inertial>")
    s.append(f'      <origin xyz="{com[0]:.6g} {com[1]:.6g} {com[2]:.6g}" rpy="{inertia_rpy[0]:.6g} {inertia_rpy[1]:.6g} {inertia_rpy[2]:.6g}"/>')
    s.append(f'      <mass value="{mass:.6g}"/>')
    s.append(f'      <inertia ixx="{ixx:.6g}" ixy="{ixy:.6g}" ixz="{ixz:.6g}" iyy="{iyy:.6g}" iyz="{iyz:.6g}" izz="{izz:.6g}"/>')
    s.append("    </inertial>")
    for c in collisions:
        s.append(f'    <collision name="{c["name"]}">')
        rpy = c.get("rpy", (0, 0, 0))
        s.append(f'      <origin xyz="{c["xyz"][0]:.6g} {c["xyz"][1]:.6g} {c["xyz"][2]:.6g}" rpy="{rpy[0]:.6g} {rpy[1]:.6g} {rpy[2]:.6g}"/>')
        if c["type"] == "sphere":
            s.append(f'      <geometry><sphere radius="{c["radius"]:.6g}"/></geometry>')
        elif c["type"] == "capsule":
            s.append(f'      <geometry><capsule radius="{c["radius"]:.6g}" length="{c["length"]:.6g}"/></geometry>')
        elif c["type"] == "box":
            s.append(f'      <geometry><box size="{c["size"][0]:.6g} {c["size"][1]:.6g} {c["size"][2]:.6g}"/></geometry>')
        elif c["type"] == "mesh":   # the loader ignores (and counts) mesh colliders: keeps the stand-ins' sphere sets fixed
            s.append(f'      <geometry><mesh filename="{c["name"]}.stl"/></geometry>')
        s.append("    </collision>")
    s.append("  </link>")
    return "\n".join(s)


def joint(name, jtype, parent, child, xyz, axis=None, rpy=(0, 0, 0), effort=80.0, lower=-6.28, upper=6.28,
          damping=0.0, rotor_inertia=0.0):
    s = [f'  <joint name="{name}" type="{jtype}">']
    s.append(f'    <origin xyz="{xyz[0]:.6g} {xyz[1]:.6g} {xyz[2]:.6g}" rpy="{rpy[0]:.6g} {rpy[1]:.6g} {rpy[2]:.6g}"/>')
    s.append(f'    <parent link="{parent}"/>')
    s.append(f'    <child link="{child}"/>')
    if jtype != "fixed":
        s.append(f'    <axis xyz="{axis[0]:.6g} {axis[1]:.6g} {axis[2]:.6g}"/>')
        s.append(f'    <limit effort="{effort:.6g}" velocity="7.5" lower="{lower:.6g}" upper="{upper:.6g}"/>')
        s.append(f'    <dynamics damping="{damping:.6g}" rotor_inertia="{rotor_inertia:.6g}"/>')
    s.append("  </joint>")
    return "\n".join(s)


def anymal_c_like():
    out = ['<?xml version="1.0"?>',
           "<!-- SYNTHETIC ANYmal-C-like stand-in (see make_urdfs.py); not the upstream anymal.urdf -->",
           '<robot name="anymal_c_like">']
    base_cols = [dict(name=f"base_{i}", type="sphere", xyz=(sx * 0.30, sy * 0.10, 0.0), radius=0.10)
                 for i, (sx, sy) in enumerate([(1, 1), (1, -1), (-1, 1), (-1, -1)])]
    base_cols.append(dict(name="base_mesh_ignored", type="mesh", xyz=(0, 0, 0)))
    out.append(link("base", 19.2, (0.0, 0.0, 0.01), box_inertia(19.2, 0.53, 0.27, 0.24), base_cols,
                    off_diag=(0.001, -0.002, 0.0005)))
    # a payload rigidly attached with a rotated frame: exercises fixed-joint merging
    out.append(link("top_shell", 2.4, (0.02, 0.0, 0.03), box_inertia(2.4, 0.3, 0.2, 0.06), inertia_rpy=(0.0, 0.1, 0.3)))
    out.append(joint("base_to_top_shell", "fixed", "base", "top_shell", (0.05, 0.0, 0.09), rpy=(0.0, 0.0, 0.2)))
    for leg, fx, sy in [("LF", 1, 1), ("RF", 1, -1), ("LH", -1, 1), ("RH", -1, -1)]:
        out.append(link(f"{leg}_HIP", 2.781, (fx * 0.05, sy * 0.01, 0.0), (0.0036, 0.0056, 0.0047),
                        off_diag=(sy * fx * 0.0001, 0.0, 0.0)))
        out.append(joint(f"{leg}_HAA", "revolute", "base", f"{leg}_HIP", (fx * 0.2999, sy * 0.104, 0.0), (1, 0, 0)))
        out.append(link(f"{leg}_THIGH", 3.071, (0.0, sy * 0.018, -0.169), (0.0319, 0.0310, 0.0060),
                        [dict(name=f"{leg}_thigh_capsule", type="capsule", xyz=(0.0, sy * 0.06, -0.14),
                              radius=0.045, length=0.16)],
                        off_diag=(0.0, 0.0, sy * 0.001)))
        out.append(joint(f"{leg}_HFE", "revolute", f"{leg}_HIP", f"{leg}_THIGH", (fx * 0.0599, sy * 0.08381, 0.0), (0, 1, 0)))
        out.append(link(f"{leg}_SHANK", 0.58, (fx * 0.05, sy * 0.007, -0.12), (0.0090, 0.0095, 0.0012),
                        [dict(name=f"{leg}_knee", type="sphere", xyz=(0.0, 0.0, 0.0), radius=0.06)]))
        out.append(joint(f"{leg}_KFE", "revolute", f"{leg}_THIGH", f"{leg}_SHANK", (0.0, sy * 0.1003, -0.285), (0, 1, 0)))
        out.append(link(f"{leg}_FOOT", 0.25, (0.0, 0.0, 0.01), (0.0002, 0.0002, 0.0002),
                        [dict(name=f"{leg}_foot", type="sphere", xyz=(0.0, 0.0, 0.0), radius=0.03)]))
        out.append(joint(f"{leg}_SHANK_TO_FOOT", "fixed", f"{leg}_SHANK", f"{leg}_FOOT",
                         (fx * 0.08795, sy * 0.01305, -0.33797)))
    out.append("</robot>\n")
    return "\n".join(out)


def atlas_like():
    out = ['<?xml version="1.0"?>',
           "<!-- SYNTHETIC Atlas-like stand-in (30 actuated joints; see make_urdfs.py); not the upstream robot.urdf -->",
           '<robot name="atlas_like">']
    E = 400.0
    out.append(link("pelvis", 17.9, (0.011, 0.0, 0.027), (0.125, 0.086, 0.165),
                    [dict(name="pelvis_s", type="sphere", xyz=(0, 0, 0.0), radius=0.12)]))   # (clear of the torso capsule, three joints up: links of one system collide)
    out.append(link("ltorso", 2.4, (-0.011, 0.0, 0.075), (0.0040, 0.0055, 0.0035)))
    out.append(joint("back_bkz", "revolute", "pelvis", "ltorso", (-0.0125, 0.0, 0.0), (0, 0, 1), effort=E))
    out.append(link("mtorso", 0.69, (-0.008, 0.0, 0.04), (0.0005, 0.0004, 0.0008)))
    out.append(joint("back_bky", "revolute", "ltorso", "mtorso", (0.0, 0.0, 0.162), (0, 1, 0), effort=E))
    out.append(link("utorso", 63.7, (-0.06, 0.0, 0.26), (1.58, 1.30, 0.85),
                    [dict(name="utorso_back", type="capsule", xyz=(-0.12, 0.0, 0.25), radius=0.2, length=0.3),
                     dict(name="utorso_mesh_ignored", type="mesh", xyz=(0, 0, 0.2))]))
    out.append(joint("back_bkx", "revolute", "mtorso", "utorso", (0.0, 0.0, 0.05), (1, 0, 0), effort=E))
    out.append(link("head", 1.42, (-0.075, 0.0, 0.03), (0.0040, 0.0042, 0.0036),
                    [dict(name="head_s", type="sphere", xyz=(0.0, 0.0, 0.05), radius=0.12)]))
    out.append(joint("neck_ry", "revolute", "utorso", "head", (0.22, 0.0, 0.55), (0, 1, 0), effort=25.0))
    for side, sy in [("l", 1), ("r", -1)]:
        arm = [
            ("shz", "clav", "utorso", (0.1406, sy * 0.2256, 0.4776), (0, 0, 1), 4.47, (0.0, sy * 0.048, 0.084), (0.011, 0.009, 0.004)),
            ("shx", "scap", "clav", (0.0, sy * 0.11, 0.245), (1, 0, 0), 3.90, (0.0, sy * 0.02, -0.01), (0.0032, 0.0046, 0.0052)),
            ("ely", "uarm", "scap", (0.0, sy * 0.187, 0.016), (0, 1, 0), 4.42, (0.0, sy * 0.065, 0.0), (0.0026, 0.013, 0.013)),
            ("elx", "larm", "uarm", (0.0, sy * 0.119, 0.0092), (1, 0, 0), 3.39, (0.0, sy * 0.035, 0.0), (0.0057, 0.0028, 0.0056)),
            ("wry", "ufarm", "larm", (0.0, sy * 0.187, -0.0092), (0, 1, 0), 2.51, (0.0, sy * 0.041, 0.0), (0.0077, 0.0024, 0.0084)),
            ("wrx", "lfarm", "ufarm", (0.0, sy * 0.119, 0.0092), (1, 0, 0), 0.96, (0.0, sy * 0.02, 0.0), (0.0010, 0.0008, 0.0010)),
            ("wry2", "hand", "lfarm", (0.0, sy * 0.06, 0.0), (0, 1, 0), 1.11, (0.0, sy * 0.05, 0.0), (0.0013, 0.0006, 0.0013)),
        ]
        for jn, ln, par, xyz, ax, mass, com, inert in arm:
            cols = []
            if ln == "hand":
                cols = [dict(name=f"{side}_hand_s", type="sphere", xyz=(0.0, sy * 0.08, 0.0), radius=0.07)]
            if ln == "larm":
                cols = [dict(name=f"{side}_elbow_s", type="sphere", xyz=(0.0, 0.0, 0.0), radius=0.08)]
            pname = par if par == "utorso" else f"{side}_{par}"
            out.append(link(f"{side}_{ln}", mass, com, inert, cols))
            out.append(joint(f"{side}_arm_{jn}", "revolute", pname, f"{side}_{ln}", xyz, ax, effort=200.0))
        leg = [
            ("hpz", "uglut", "pelvis", (0.0, sy * 0.089, 0.0), (0, 0, 1), 1.959, (0.0053, sy * -0.0034, 0.0313), (0.0008, 0.0010, 0.0009)),
            ("hpx", "lglut", "uglut", (0.0, 0.0, 0.0), (1, 0, 0), 0.898, (0.0133, sy * 0.017, -0.0312), (0.0007, 0.0009, 0.0007)),
            ("hpy", "uleg", "lglut", (0.05, sy * 0.0225, -0.066), (0, 1, 0), 8.204, (0.0, 0.0, -0.21), (0.09, 0.09, 0.02)),
            ("kny", "lleg", "uleg", (-0.05, 0.0, -0.374), (0, 1, 0), 4.515, (0.001, 0.0, -0.187), (0.077, 0.076, 0.010)),
            ("aky", "talus", "lleg", (0.0, 0.0, -0.422), (0, 1, 0), 0.125, (0.0, 0.0, 0.0), (0.00012, 0.00013, 0.00010)),
            ("akx", "foot", "talus", (0.0, 0.0, 0.0), (1, 0, 0), 2.41, (0.027, 0.0, -0.067), (0.002, 0.007, 0.008)),
        ]
        for jn, ln, par, xyz, ax, mass, com, inert in leg:
            cols = []
            if ln == "foot":
                cols = [dict(name=f"{side}_foot_{k}", type="sphere", xyz=(cx, cy, -0.081 + 0.02), radius=0.02)
                        for k, (cx, cy) in enumerate([(0.17, 0.06), (0.17, -0.06), (-0.08, 0.06), (-0.08, -0.06)])]
            if ln == "lleg":
                cols = [dict(name=f"{side}_knee_s", type="sphere", xyz=(0.0, 0.0, 0.0), radius=0.08)]
            pname = par if par == "pelvis" else f"{side}_{par}"
            out.append(link(f"{side}_{ln}", mass, com, inert, cols))
            out.append(joint(f"{side}_leg_{jn}", "revolute", pname, f"{side}_{ln}", xyz, ax, effort=E,
                             rotor_inertia=0.02 if ln in ("talus", "foot") else 0.0))
    out.append("</robot>\n")
    return "\n".join(out)


def main():
    with open(os.path.join(HERE, "anymal_c_like.urdf"), "w") as f:
        f.write(anymal_c_like())
    with open(os.path.join(HERE, "atlas_like.urdf"), "w") as f:
        f.write(atlas_like())


if __name__ == "__main__":
    main()
