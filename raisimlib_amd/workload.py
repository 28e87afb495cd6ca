"""Synthetic workload definitions shared by tests and bench.py (BASELINE.json configs).

Config 1/2: ANYmal-C-like stand-in on flat ground, dt = 0.0025, 4 sub-steps per control step,
PD kp=50 / kd=0.2 on the 12 joints, targets = nominal + U(-0.3, 0.3) rad resampled per control step,
per-env seed 1234+i, base xy jitter U(-0.1, 0.1) m, yaw U(-pi, pi).
Config 3: the same robots on a shared 128 x 128 height map over 12.8 m x 12.8 m (smoothed noise, +-0.1 m, seed 7), base xy
spread over +-6 m of it (per-env seeded) so that the envs of a batch gather from all over the map; variant: one map per env.
Config 5: Atlas-like humanoid, kmax 16.  "standing" regime (default): legs and back kp 3000 / kd 60, arms and neck kp 300 / kd 10,
targets = zero pose + U(-0.03, 0.03) rad (legs, back) / U(-0.1, 0.1) rad (arms, neck) per control step: the humanoid stays on
its eight foot spheres indefinitely (oracle: 256 envs x 2000 control steps, no fall).  "collapsing" regime (rounds 1-2): kp 200 /
kd 5 on every joint with U(-0.1, 0.1) rad noise - too weak for the 164 kg model, every env falls within 0.6 s and is reset.

Every random number is a pure function of (GLOBAL env index, control step, entry): env g draws from the
counter-based stream keyed by seed0 + g, so a rank that owns envs [lo, hi) produces exactly rows lo..hi of the
unsharded arrays ("per-env seed 1234+i", shard-invariant results).
"""
import numpy as np

ANYMAL_NOMINAL_JOINTS = np.array([0.03, 0.4, -0.8, -0.03, 0.4, -0.8, 0.03, -0.4, 0.8, -0.03, -0.4, 0.8])
ANYMAL_INIT_HEIGHT = 0.60   # feet just above the ground with the stand-in's leg lengths
DT = 0.0025
SUBSTEPS = 4
KP, KD = 50.0, 0.2
ATLAS_KP, ATLAS_KD = 200.0, 5.0
ATLAS_INIT_HEIGHT = 0.95


def _mix64(x):
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def env_uniform(seeds, stream, count):
    """U[0,1) numbers [len(seeds), count] (float64, 53 bits): entry (i, j) depends only on (seeds[i], stream, j).

    Counter-based (splitmix64 of a key built from the three integers), so it can be evaluated for any subset of
    envs, in any order, on any rank, with identical results."""
    with np.errstate(over="ignore"):
        s = np.asarray(seeds, np.uint64)[:, None]
        j = np.arange(count, dtype=np.uint64)[None, :]
        key = _mix64(s * np.uint64(0x9E3779B97F4A7C15) + np.uint64(stream) * np.uint64(0xD1B54A32D192ED03) + np.uint64(0x2545F4914F6CDD1D))
        x = _mix64(key + (j + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15))
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


_STREAM_INIT = 0xFFFFFFFF      # stream id of the initial-state draws; control step k uses stream k


def anymal_gains(nv=18):
    kp = np.zeros(nv, np.float32)
    kd = np.zeros(nv, np.float32)
    kp[6:] = KP
    kd[6:] = KD
    return kp, kd


def anymal_initial_state(n_envs, seed0=1234, env_offset=0, height=ANYMAL_INIT_HEIGHT):
    """Per-env seeded initial state (gc [N,19], gv [N,18]) in float64: env i of the call is global env env_offset + i."""
    gc = np.zeros((n_envs, 19))
    gv = np.zeros((n_envs, 18))
    r = env_uniform(seed0 + env_offset + np.arange(n_envs), _STREAM_INIT, 3)
    yaw = (2.0 * r[:, 2] - 1.0) * np.pi
    gc[:, 0:2] = 0.2 * r[:, 0:2] - 0.1
    gc[:, 2] = height
    gc[:, 3] = np.cos(0.5 * yaw)
    gc[:, 6] = np.sin(0.5 * yaw)
    gc[:, 7:] = ANYMAL_NOMINAL_JOINTS
    return gc, gv


def env_heightmaps(n_envs, xs=128, ys=128, amplitude=0.1, seed0=7, env_offset=0):
    """Config 3, per-env-map variant: one smoothed-noise map per env, [n_envs, ys, xs]; env g's map is seeded seed0 + g
    (g = 0 is the shared map of the default variant), so a rank that owns envs [lo, hi) builds exactly those maps."""
    return np.stack([smoothed_heightmap(xs, ys, amplitude, seed0 + env_offset + i) for i in range(n_envs)])


def anymal_initial_state_on_maps(n_envs, maps, env_map, size, seed0=1234, env_offset=0, spread=6.0, clearance=0.02):
    """Config 3: per-env seeded initial states spread over the map - base xy ~ U(-spread, spread)^2 - and dropped from just above
    the terrain: base height = nominal + the highest sample within 0.5 m of the base + clearance.  maps [n_maps, ys, xs]
    centred on the origin, `size` metres wide; env i stands on maps[env_map[i]]."""
    from scipy.ndimage import maximum_filter
    gc, gv = anymal_initial_state(n_envs, seed0, env_offset)
    r = env_uniform(seed0 + env_offset + np.arange(n_envs), _STREAM_INIT, 5)   # entries 0-2 are anymal_initial_state's
    gc[:, 0:2] = spread * (2.0 * r[:, 3:5] - 1.0)
    ys, xs = maps.shape[1:]
    cell = size / (xs - 1)
    ix = np.clip(np.rint((gc[:, 0] + 0.5 * size) / cell).astype(int), 0, xs - 1)
    iy = np.clip(np.rint((gc[:, 1] + 0.5 * size) / cell).astype(int), 0, ys - 1)
    k = 2 * int(round(0.5 / cell)) + 1
    env_map = np.asarray(env_map)
    top = np.zeros(n_envs)
    for mi in np.unique(env_map):
        sel = env_map == mi
        top[sel] = maximum_filter(maps[mi], size=k, mode="nearest")[iy[sel], ix[sel]]
    gc[:, 2] = ANYMAL_INIT_HEIGHT + top + clearance
    return gc, gv


def anymal_targets(n_envs, control_step, seed0=1234, env_offset=0, amplitude=0.3):
    """PD position targets [N,19] for one control step (base entries unused)."""
    pt = np.zeros((n_envs, 19))
    r = env_uniform(seed0 + env_offset + np.arange(n_envs), control_step, 12)
    pt[:, 7:] = ANYMAL_NOMINAL_JOINTS + amplitude * (2.0 * r - 1.0)
    pt[:, 3] = 1.0
    return pt


def atlas_gains(nv=36):
    """the "collapsing" regime of rounds 1-2: kp 200 / kd 5 on every joint (cannot hold the 164 kg model up)"""
    kp = np.zeros(nv, np.float32)
    kd = np.zeros(nv, np.float32)
    kp[6:] = ATLAS_KP
    kd[6:] = ATLAS_KD
    return kp, kd


ATLAS_STAND_KP, ATLAS_STAND_KD = (3000.0, 300.0), (60.0, 10.0)     # (legs + back, arms + neck)
ATLAS_STAND_AMP = (0.03, 0.1)                                        # target noise (rad), same classes


def atlas_stiff_joints(joint_names):
    """bool [nv - 6]: the joints that carry the body (legs, back) - stiff gains, small target noise - vs arms and neck"""
    return np.array([("_leg_" in n) or n.startswith("back_") for n in joint_names[1:]])


def atlas_standing_gains(joint_names):
    """Config 5, "standing" regime: gains that hold the Atlas-like stand-in up (gravity stiffness about the ankles is
    m g h ~ 1600 N m / rad, about the back ~ 200: kp 200 everywhere cannot)."""
    stiff = atlas_stiff_joints(joint_names)
    nv = 6 + len(stiff)
    kp = np.zeros(nv, np.float32)
    kd = np.zeros(nv, np.float32)
    kp[6:] = np.where(stiff, ATLAS_STAND_KP[0], ATLAS_STAND_KP[1])
    kd[6:] = np.where(stiff, ATLAS_STAND_KD[0], ATLAS_STAND_KD[1])
    return kp, kd


def atlas_standing_targets(n_envs, control_step, joint_names, seed0=77, env_offset=0, scale=1.0):
    """Config 5, "standing" regime: zero pose + per-joint-class uniform noise, per-env seeded (same streams as atlas_targets)."""
    stiff = atlas_stiff_joints(joint_names)
    nq = 7 + len(stiff)
    pt = np.zeros((n_envs, nq))
    r = env_uniform(seed0 + env_offset + np.arange(n_envs), control_step, nq - 7)
    pt[:, 7:] = scale * np.where(stiff, ATLAS_STAND_AMP[0], ATLAS_STAND_AMP[1]) * (2.0 * r - 1.0)
    pt[:, 3] = 1.0
    return pt


def atlas_initial_state(n_envs, nq=37, nv=36, height=ATLAS_INIT_HEIGHT):
    """Config 5: every env starts upright at the nominal pose ("standing PD")."""
    gc = np.zeros((n_envs, nq))
    gc[:, 2] = height
    gc[:, 3] = 1.0
    return gc, np.zeros((n_envs, nv))


def atlas_targets(n_envs, control_step, nq=37, seed0=77, env_offset=0, amplitude=0.1):
    """Config 5 PD targets: zero pose + U(-amplitude, amplitude) rad per joint and control step, per-env seeded."""
    pt = np.zeros((n_envs, nq))
    r = env_uniform(seed0 + env_offset + np.arange(n_envs), control_step, nq - 7)
    pt[:, 7:] = amplitude * (2.0 * r - 1.0)
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


HEIGHTMAP_SIZE = 12.8          # config 3: 128 x 128 samples over 12.8 m x 12.8 m (0.1 m cells), centred on the origin
HEIGHTMAP_CLEARANCE = 0.12     # initial base height is raised by this much so that no foot starts inside the +-0.1 m terrain


# ---- the config-2 workload with a policy IN the loop (include/rsb_pipeline.h; bench.py `closed_loop`, tests/test_gpu_closed_loop.py) ----
# The env task (rsb_env_*) turns an action into PD targets  nominal + action_std * action;  with action_std = the config's noise amplitude (0.3 rad)
# and the noise bank below, a zero policy reproduces config 2's targets exactly (the same counter-based draws).  The reference stage adds the
# feedback term W ob on top: a fixed linear policy whose output depends on every step's observation.
CLOSED_LOOP_W_SCALE = 0.02


def closed_loop_noise(n_envs, period, seed0=1234, env_offset=0, n_act=12):
    """[period, N, n_act] float32 in [-1, 1): slice k is the draw config 2's targets use for control step k ((target - nominal) / amplitude)"""
    out = np.empty((period, n_envs, n_act), np.float32)
    for k in range(period):
        out[k] = 2.0 * env_uniform(seed0 + env_offset + np.arange(n_envs), k, n_act) - 1.0
    return out


def closed_loop_policy(n_obs, n_act, scale=CLOSED_LOOP_W_SCALE, seed=5):
    """the fixed linear policy of the closed-loop benchmark: W [n_act, n_obs] ~ U(-scale, scale), seeded"""
    return np.random.default_rng(seed).uniform(-scale, scale, (n_act, n_obs)).astype(np.float32)


def closed_loop_mlp(n_obs, n_act, hidden=(128, 128), out_scale=CLOSED_LOOP_W_SCALE, seed=6):
    """the actor of the closed-loop benchmark's MLP leg: raisimGymTorch's default architecture [RECALL: MLP ob -> 128 -> 128 -> act, LeakyReLU],
    random weights - hidden layers U(+-1/sqrt(fan_in)) as torch.nn.Linear initialises them, the output layer U(+-out_scale) so that the actions stay
    in the regime of the linear leg (the noise does the exploring) -, zero biases.  Returns [(W [out, in], b [out]), ...] float32."""
    rng = np.random.default_rng(seed)
    dims = [n_obs, *hidden, n_act]
    layers = []
    for i in range(len(dims) - 1):
        bound = out_scale if i + 2 == len(dims) else 1.0 / np.sqrt(dims[i])
        layers.append((rng.uniform(-bound, bound, (dims[i + 1], dims[i])).astype(np.float32), np.zeros(dims[i + 1], np.float32)))
    return layers


def closed_loop_env(model, n_envs, device=0, env_offset=0, stream=None):
    """config 2 as a device-resident vectorised env: ANYmal-like robots on flat ground, dt 0.0025 x 4, PD kp 50 / kd 0.2, action_std 0.3 around
    the nominal joints, non-foot contact -> reset to the env's OWN initial state (per-env base xy / yaw as in the open-loop benchmark)"""
    from .vecenv import VecEnv
    gc_init = np.zeros(model.nq, np.float32)
    gc_init[2], gc_init[3] = ANYMAL_INIT_HEIGHT, 1.0
    gc_init[7:] = ANYMAL_NOMINAL_JOINTS
    env = VecEnv(model, n_envs, device=device, simulation_dt=DT, control_dt=DT * SUBSTEPS, action_std=0.3, p_gain=KP, d_gain=KD,
                 gc_init=gc_init, stream=stream)
    gc0, gv0 = anymal_initial_state(n_envs, env_offset=env_offset)
    env.set_reset_states(gc0, gv0)
    env.reset()
    return env
