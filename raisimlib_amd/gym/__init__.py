"""The Python-facing boundary of the drop-in: a raisim_gym-style pybind11 module per environment folder
(build_gym.build_env_module) and the RaisimGymVecEnv wrapper raisimGymTorch's runners use (vec_env.RaisimGymVecEnv)."""
from .build_gym import build_env_module, load_env_module  # noqa: F401
from .vec_env import RaisimGymVecEnv  # noqa: F401
