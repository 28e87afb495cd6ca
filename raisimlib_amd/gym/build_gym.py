"""Builds the raisim_gym-style pybind11 module for one environment folder, in-tree.

Upstream compiles raisimGymTorch/env/raisim_gym.cpp once per folder under env/envs/ (its Environment.hpp is #included by the
module source; CMake names the module after the folder) [RECALL; absent from /root/reference].  Same here, without CMake:

    build_env_module("/path/to/envs/rsg_anymal")          ->  raisimlib_amd/lib/rsg_anymal.<abi>.so
    mod = load_env_module("rsg_anymal");  env = mod.RaisimGymEnv(resource_dir, cfg_text)

Plain g++ (host code only: the module talks to the GPU through librsb.so's C-ABI), pybind11 from the image.
"""
import importlib.util
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(os.path.dirname(HERE), "lib")
SRC = os.path.join(HERE, "raisim_gym.cpp")


def module_path(name):
    return os.path.join(LIB, name + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_env_module(env_dir, name=None, force=False, verbose=False):
    """env_dir holds the user's Environment.hpp (class raisim::ENVIRONMENT : public RaisimGymEnv). Returns the .so path."""
    import pybind11
    from .. import build as libbuild
    libbuild.build(verbose=verbose)                      # librsb.so (the module links against it)
    env_dir = os.path.abspath(env_dir)
    header = os.path.join(env_dir, "Environment.hpp")
    if not os.path.exists(header):
        raise FileNotFoundError(f"{header}: an environment folder holds an Environment.hpp")
    name = name or os.path.basename(env_dir.rstrip("/"))
    out = module_path(name)
    deps = [SRC, header] + [os.path.join(ROOT, "include", "raisim", f) for f in os.listdir(os.path.join(ROOT, "include", "raisim"))] + \
        [os.path.join(ROOT, "include", "rsb.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-Wall", "-pthread",
           f"-DRSG_ENVIRONMENT_HEADER=\"{header}\"", f"-DRSG_MODULE_NAME={name}",
           "-I", pybind11.get_include(), "-I", sysconfig.get_paths()["include"], "-I", os.path.join(ROOT, "include"), "-I", env_dir,
           SRC, "-o", out, "-L", LIB, "-lrsb", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"g++ failed for the {name} gym module:\n{r.stdout}\n{r.stderr}")
    return out


def load_env_module(name):
    """import the built module by name (fails loudly if it was not built: there is no Python fallback)"""
    path = module_path(name)
    if not os.path.exists(path):
        raise ImportError(f"{path} does not exist: build it with raisimlib_amd.gym.build_env_module(<environment folder>)")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    for d in sys.argv[1:]:
        print(build_env_module(d, verbose=True))
