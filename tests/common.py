"""Shared helpers for the test-suite (tests/ is one of the three places allowed to use oracle/)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle.pyoracle import Oracle  # noqa: E402
from raisimlib_amd import workload  # noqa: E402


def f32(a):
    """Round to float32 and back: the exact inputs the device sees, in fp64 for the oracle."""
    return np.asarray(a, np.float32).astype(np.float64)


SPHERE_URDF = """<?xml version="1.0"?>
<robot name="ball">
  <link name="ball">
    <inertial><origin xyz="0 0 0"/><mass value="{m}"/>
      <inertia ixx="{i}" ixy="0" ixz="0" iyy="{i}" iyz="0" izz="{i}"/></inertial>
    <collision><origin xyz="0 0 0"/><geometry><sphere radius="{r}"/></geometry></collision>
  </link>
</robot>
"""


def sphere_urdf(m=2.0, r=0.1):
    return SPHERE_URDF.format(m=m, r=r, i=0.4 * m * r * r)


PENDULUM_URDF = """<?xml version="1.0"?>
<robot name="pendulum">
  <link name="anchor">
    <inertial><origin xyz="0 0 0"/><mass value="1e9"/>
      <inertia ixx="1e9" ixy="0" ixz="0" iyy="1e9" iyz="0" izz="1e9"/></inertial>
  </link>
  <link name="bob">
    <inertial><origin xyz="0 0 -{l}"/><mass value="{m}"/>
      <inertia ixx="1e-9" ixy="0" ixz="0" iyy="1e-9" iyz="0" izz="1e-9"/></inertial>
  </link>
  <joint name="hinge" type="revolute">
    <origin xyz="0 0 0"/><parent link="anchor"/><child link="bob"/><axis xyz="0 1 0"/>
    <limit effort="0" velocity="100" lower="-10" upper="10"/>
  </joint>
</robot>
"""


def standing_states(n, seed=0, z=(0.46, 0.62), joint_noise=0.25, vel=0.5):
    """Physically plausible ANYmal states near the ground (upright +-0.2 rad, feet touching or about to)."""
    rng = np.random.default_rng(seed)
    gc = np.zeros((n, 19))
    gc[:, 0:2] = rng.uniform(-1, 1, (n, 2))
    gc[:, 2] = rng.uniform(z[0], z[1], n)
    rpy = np.c_[rng.uniform(-0.2, 0.2, n), rng.uniform(-0.2, 0.2, n), rng.uniform(-np.pi, np.pi, n)]
    cr, sr, cp, sp, cy, sy = (np.cos(rpy[:, 0] / 2), np.sin(rpy[:, 0] / 2), np.cos(rpy[:, 1] / 2), np.sin(rpy[:, 1] / 2),
                              np.cos(rpy[:, 2] / 2), np.sin(rpy[:, 2] / 2))
    gc[:, 3] = cr * cp * cy + sr * sp * sy
    gc[:, 4] = sr * cp * cy - cr * sp * sy
    gc[:, 5] = cr * sp * cy + sr * cp * sy
    gc[:, 6] = cr * cp * sy - sr * sp * cy
    gc[:, 7:] = workload.ANYMAL_NOMINAL_JOINTS + rng.uniform(-joint_noise, joint_noise, (n, 12))
    gv = rng.normal(size=(n, 18)) * vel
    return gc, gv
