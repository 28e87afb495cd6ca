"""env-steps/s of the raisimGymTorch-shaped Python boundary at N = 4096 (not the headline: what an UNMODIFIED Environment.hpp
gets through RaisimGymVecEnv -> pybind11 -> VectorizedEnvironment<ENVIRONMENT> -> the control step's integrate() calls fused into one launch), next to the
device-resident env (DeviceRaisimGymEnv: one fused launch per control step, host buffers) on the same box."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raisimlib_amd.gym import RaisimGymVecEnv, build_env_module, load_env_module  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 1000   # (sustained: several 100-ms cgroup periods; bursts of 40 steps measured up to 1.7 x more on a box with a CPU quota)
RSC = os.path.join(ROOT, "raisimlib_amd", "rsc")
THREADS = int(sys.argv[3]) if len(sys.argv) > 3 else 16
CFG = (f"num_envs: {N}\nnum_threads: {THREADS}\nsimulation_dt: 0.0025\ncontrol_dt: 0.01\nrender: false\naction_std: 0.3\n"
       "reward:\n  forwardVel:\n    coeff: 0.3\n  torque:\n    coeff: -4e-5\n")
build_env_module(os.path.join(ROOT, "tests", "cpp", "anymal_env"), name="rsg_anymal")
mod = load_env_module("rsg_anymal")
rng = np.random.default_rng(0)
acts = [rng.uniform(-1, 1, (N, 12)).astype(np.float32) for _ in range(8)]
out = {"num_envs": N, "control_steps_timed": STEPS, "substeps_per_control_step": 4, "num_threads_cfg": THREADS}

t0 = time.perf_counter()
env = RaisimGymVecEnv(mod.RaisimGymEnv(RSC, CFG, False), normalize_ob=False)
out["template_construct_s"] = time.perf_counter() - t0
env.reset()
for k in range(5):
    env.step(acts[k % 8]); env.observe(False)
l0 = env.wrapper.viewLaunches()
t0 = time.perf_counter()
for k in range(STEPS):
    env.step(acts[k % 8]); env.observe(False)
dt = time.perf_counter() - t0
out["template_path"] = {"env_steps_per_s": N * 4 * STEPS / dt, "ms_per_control_step": dt / STEPS * 1e3, "launches": env.wrapper.viewLaunches() - l0,
                        "what": "RaisimGymVecEnv.step + observe on numpy buffers; N unmodified Environment.hpp objects as fibers, the control step's 4 integrate() calls = ONE fused launch + one rsb_view_exchange"}

cfg = mod.VecEnvConfig(); cfg.num_envs = N
cfg.gc_init = [0, 0, 0.57, 1.0, 0.0, 0.0, 0.0, 0.03, 0.4, -0.8, -0.03, 0.4, -0.8, 0.03, -0.4, 0.8, -0.03, -0.4, 0.8]
dev = mod.DeviceRaisimGymEnv(os.path.join(RSC, "anymal_c_like.urdf"), cfg); dev.init()
r, d, o = np.zeros(N, np.float32), np.zeros(N, bool), np.zeros((N, 34), np.float32)
for k in range(5):
    dev.step(acts[k % 8], r, d); dev.observe(o)
t0 = time.perf_counter()
for k in range(STEPS * 5):
    dev.step(acts[k % 8], r, d); dev.observe(o)
dt = time.perf_counter() - t0
out["device_env_host_buffers"] = {"env_steps_per_s": N * 4 * STEPS * 5 / dt, "ms_per_control_step": dt / (STEPS * 5) * 1e3,
                                  "what": "DeviceRaisimGymEnv.step + observe on numpy buffers (task on the GPU, one fused launch per control step, PCIe both ways)"}
print(json.dumps(out))
