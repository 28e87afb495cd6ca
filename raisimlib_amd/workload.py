"""Synthetic workload definitions shared by tests and bench.py (BASELINE.json configs).

Config 1/2: ANYmal-C-like stand-in on flat ground, dt = 0.0025, 4 sub-steps per control step,
PD kp=50 / kd=0.2 on the 12 joints, targets = nominal + U(-0.3, 0.3) rad resampled per control step,
per-env seed 1234+i, base xy jitter U(-0.1, 0.1) m, yaw U(-pi, pi).
"""
import numpy as np

ANYMAL_NOMINAL_JOINTS = np.array([0.03, 0.4, -0.8, -0.03, 0.4, -0.8, 0.03, -0.4, 0.8, -0.03, -0.4, 0.8])
ANYMAL_INIT_HEIGHT = 0.60   # feet just above the ground with the stand-in's leg lengths
DT = 0.0025
SUBSTEPS = 4
KP, KD = 50.0, 0.2


def anymal_gains(nv=18):
    kp = np.zeros(nv, np.float32)
    kd = np.zeros(nv, np.float32)
    kp[6:] = KP
    kd[6:] = KD
    return kp, kd


def anymal_initial_state(n_envs, seed0=1234, env_offset=0, height=ANYMAL_INIT_HEIGHT):
    """Per-env seeded initial state (gc [N,19], gv [N,18]) in float64."""
    gc = np.zeros((n_envs, 19))
    gv = np.zeros((n_envs, 18))
    for i in range(n_envs):
        rng = np.random.default_rng(seed0 + env_offset + i)
        xy = rng.uniform(-0.1, 0.1, 2)
        yaw = rng.uniform(-np.pi, np.pi)
        gc[i, 0:2] = xy
        gc[i, 2] = height
        gc[i, 3] = np.cos(0.5 * yaw)
        gc[i, 6] = np.sin(0.5 * yaw)
        gc[i, 7:] = ANYMAL_NOMINAL_JOINTS
    return gc, gv


def anymal_targets(n_envs, control_step, seed0=1234, env_offset=0, amplitude=0.3):
    """PD position targets [N,19] for one control step (base entries unused)."""
    pt = np.zeros((n_envs, 19))
    rng = np.random.default_rng([seed0 + env_offset, control_step])
    pt[:, 7:] = ANYMAL_NOMINAL_JOINTS + rng.uniform(-amplitude, amplitude, (n_envs, 12))
    pt[:, 3] = 1.0
    return pt


def random_state(model_nq, model_nv, n_envs, seed=0, joint_range=0.6, vel_scale=1.0, z_range=(0.3, 1.2)):
    """Generic random states for one-step parity tests (any model)."""
    rng = np.random.default_rng(seed)
    gc = np.zeros((n_envs, model_nq))
    gc[:, 0:2] = rng.uniform(-2, 2, (n_envs, 2))
    gc[:, 2] = rng.uniform(z_range[0], z_range[1], n_envs)
    qq = rng.normal(size=(n_envs, 4))
    gc[:, 3:7] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
    gc[:, 7:] = rng.uniform(-joint_range, joint_range, (n_envs, model_nq - 7))
    gv = rng.normal(size=(n_envs, model_nv)) * vel_scale
    return gc, gv


def smoothed_heightmap(xs=128, ys=128, amplitude=0.1, seed=7, passes=3):
    """Config 3 terrain: smoothed uniform noise, amplitude `amplitude` (m), [ys, xs] float32."""
    rng = np.random.default_rng(seed)
    h = rng.uniform(-1.0, 1.0, (ys, xs))
    for _ in range(passes):
        h = (h + np.roll(h, 1, 0) + np.roll(h, -1, 0) + np.roll(h, 1, 1) + np.roll(h, -1, 1)) / 5.0
    h = h / np.abs(h).max() * amplitude
    return h.astype(np.float32)
