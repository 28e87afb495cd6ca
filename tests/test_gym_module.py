"""The Python boundary: a raisim_gym-style pybind11 module built over an UNMODIFIED rsg_anymal-style
Environment.hpp (tests/cpp/anymal_env), driven through the RaisimGymVecEnv class raisimGymTorch's runners use."""
import os

import numpy as np
import pytest

from common import ROOT

ENV_DIR = os.path.join(ROOT, "tests", "cpp", "anymal_env")
RSC = os.path.join(ROOT, "raisimlib_amd", "rsc")
CFG = ("num_envs: {n}\nnum_threads: 8   # ignored\nsimulation_dt: 0.0025\ncontrol_dt: 0.01\nrender: false\naction_std: 0.3\n"
       "reward:\n  forwardVel:\n    coeff: 0.3\n  torque:\n    coeff: -4e-5\n")


@pytest.fixture(scope="module")
def gym_module(built_lib):
    from raisimlib_amd.gym import build_env_module, load_env_module
    build_env_module(ENV_DIR, name="rsg_anymal")
    return load_env_module("rsg_anymal")


def test_module_builds_imports_and_fails_loudly_without_a_gpu(gym_module, built_lib):
    for cls in ("RaisimGymEnv", "DeviceRaisimGymEnv", "VecEnvConfig"):
        assert hasattr(gym_module, cls)
    for meth in ("init", "reset", "observe", "step", "setSeed", "close", "isTerminalState", "setSimulationTimeStep", "setControlTimeStep",
                 "getObDim", "getActionDim", "getNumOfEnvs", "turnOnVisualization", "turnOffVisualization", "curriculumUpdate",
                 "getObStatistics", "setObStatistics"):
        assert hasattr(gym_module.RaisimGymEnv, meth), meth          # upstream's raisim_gym.cpp surface
    if built_lib.rsb_device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu test")
    with pytest.raises(Exception) as ei:                             # no CPU fallback: constructing the batch needs a HIP device
        gym_module.RaisimGymEnv(RSC, CFG.format(n=4))
    assert "device" in str(ei.value).lower()


@pytest.mark.gpu
def test_raisim_gym_vec_env_steps_unmodified_environments_on_the_gpu(gym_module):
    from raisimlib_amd.gym import RaisimGymVecEnv
    n = 64
    env = RaisimGymVecEnv(gym_module.RaisimGymEnv(RSC, CFG.format(n=n), False), normalize_ob=False)
    assert (env.num_envs, env.num_obs, env.num_acts) == (n, 34, 12)
    cfg = gym_module.VecEnvConfig()
    cfg.num_envs = n
    cfg.gc_init = [0, 0, 0.57, 1.0, 0.0, 0.0, 0.0, 0.03, 0.4, -0.8, -0.03, 0.4, -0.8, 0.03, -0.4, 0.8, -0.03, -0.4, 0.8]
    dev = gym_module.DeviceRaisimGymEnv(os.path.join(RSC, "anymal_c_like.urdf"), cfg)
    dev.init()
    env.reset()
    ob = env.observe(False)
    assert ob.shape == (n, 34) and ob is env._observation and abs(ob[0, 0] - 0.57) < 1e-6      # written in place into the wrapper's buffer
    rng = np.random.default_rng(0)
    r2, d2, o2 = np.zeros(n, np.float32), np.zeros(n, bool), np.zeros((n, 34), np.float32)
    l0, resets = env.wrapper.viewLaunches(), 0
    for it in range(25):
        a = (rng.uniform(-1, 1, (n, 12)) * (4.0 if it % 6 == 5 else 1.0)).astype(np.float32)
        r1, d1 = env.step(a)
        dev.step(a, r2, d2)
        dev.observe(o2)
        o1 = env.observe(False)
        assert np.array_equal(d1, d2) and np.abs(r1 - r2).max() < 1e-4 and np.abs(o1 - o2).max() < 1e-4      # == the device-resident env
        resets += int(d1.sum())
    assert env.wrapper.viewLaunches() - l0 == 25            # the 4 integrate() calls of a control step are ONE fused launch for all 64 envs
    assert resets > 0
    with pytest.raises(Exception):
        env.wrapper.step(np.zeros((n, 11), np.float32), env._reward, env._done)      # shape errors are Python exceptions, not crashes
    env.close()
