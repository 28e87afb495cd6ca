"""The C++ host side (include/raisim/*.hpp) compiles with plain g++ against the C-ABI; on a GPU it runs."""
import os
import subprocess

import pytest

from common import ROOT

BIN = os.path.join(ROOT, "tests", "cpp", "_build", "facade_test")
URDF = os.path.join(ROOT, "raisimlib_amd", "rsc", "anymal_c_like.urdf")


def compile_facade():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    lib = os.path.join(ROOT, "raisimlib_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "cpp"), "-o", BIN,
                    os.path.join(ROOT, "tests", "cpp", "facade_test.cpp"), "-L", lib, "-lrsb", f"-Wl,-rpath,{lib}"],
                   check=True)


def test_facade_compiles_with_gxx_and_fails_loudly_without_gpu(built_lib):
    compile_facade()
    if built_lib.rsb_device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu test")
    r = subprocess.run([BIN, URDF], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stdout


@pytest.mark.gpu
def test_facade_runs_on_gpu(built_lib):
    compile_facade()
    r = subprocess.run([BIN, URDF], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "facade_test OK" in r.stdout


def config1_on_the_oracle():
    """BASELINE.json configs[0] as SURVEY.md 8d writes it: 1 env, flat ground, dt 0.0025, 4000 integrate() from gc_init (base at 0.50 m, nominal joints),
    gv = 0, PD kp 50 / kd 0.2 on the twelve joints with the initial pose as target, mu 0.8 - on the fp64 oracle, one thread"""
    import time
    import numpy as np
    from common import Oracle
    from raisimlib_amd import Model
    m = Model(urdf_path=URDF)
    o = Oracle(m.blob)
    gc = np.array([[0, 0, 0.50, 1, 0, 0, 0, 0.03, 0.4, -0.8, -0.03, 0.4, -0.8, 0.03, -0.4, 0.8, -0.03, -0.4, 0.8]], np.float64)
    kp, kd = np.r_[np.zeros(6), np.full(12, 50.0)], np.r_[np.zeros(6), np.full(12, 0.2)]
    t0 = time.perf_counter()
    r = o.step_batch(gc, np.zeros((1, 18)), 4000, kp, kd, gc.copy(), np.zeros((1, 18)), nthreads=1, want_contacts=True, lam_warm=o.new_warm_state(1))
    rate = 4000 / (time.perf_counter() - t0)
    n = int(r["n_contacts"][0])
    return {"z": float(r["q"][0, 2]), "qw": float(r["q"][0, 3]), "vmax": float(np.abs(r["u"]).max()), "contacts": sorted(int(c) & 0xffff for c in r["contacts"]["collision"][0, :n]),
            "steps_per_s": rate, "feet": sorted(m.collision_indices("_foot"))}


def test_config1_as_written_on_the_oracle(built_lib):
    """the robot dropped from 0.50 m settles on its four feet and stands for the 10 s of the run"""
    c = config1_on_the_oracle()
    print(f"config 1 on the oracle: base height {c['z']:.4f} m after 4000 steps, {c['steps_per_s']:.0f} env-steps/s on one thread")
    assert 0.40 < c["z"] < 0.50 and c["qw"] > 0.999 and c["vmax"] < 0.1
    assert c["contacts"] == c["feet"] and len(c["feet"]) == 4


@pytest.mark.gpu
def test_config1_as_written_through_raisim_world_on_the_device(built_lib):
    """... and the same 4000 integrate() calls through raisim::World (the facade over an N = 1 batch on the device): both stand, base heights within 2 cm
    of each other, the same four feet in contact at the end"""
    compile_facade()
    r = subprocess.run([BIN, URDF, "config1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("config1 ")][-1]
    f = dict(kv.split("=") for kv in line.split()[1:])
    c = config1_on_the_oracle()
    print(line, "| oracle:", c)
    assert abs(float(f["z"]) - c["z"]) < 0.02 and float(f["qw"]) > 0.999 and float(f["vmax"]) < 0.1 and abs(float(f["t"]) - 10.0) < 1e-6
    assert sorted(int(x) for x in f["contacts"].split(",") if x) == c["feet"] == c["contacts"]


EIGEN_BIN = os.path.join(ROOT, "tests", "cpp", "_build", "facade_eigen_test")


def compile_eigen_facade(src="facade_eigen_test.cpp", out=EIGEN_BIN):
    """the facade with RAISIM_HAS_EIGEN defined: -I tests/cpp/eigen_stub supplies <Eigen/Core> (test infrastructure standing in for Eigen3, the
    reference's one declared dependency, /root/reference/.travis.yml:7 - not installed on any box of this build)"""
    os.makedirs(os.path.dirname(out), exist_ok=True)
    lib = os.path.join(ROOT, "raisimlib_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "cpp"),
                    "-I", os.path.join(ROOT, "tests", "cpp", "eigen_stub"), "-o", out, os.path.join(ROOT, "tests", "cpp", src), "-L", lib, "-lrsb", f"-Wl,-rpath,{lib}"],
                   check=True)


def test_eigen_typed_boundary_compiles_and_its_host_part_runs(built_lib):
    """the facade's Eigen branch (Vec / Mat / VecDyn .e(), Eigen::VectorXd arguments, Eigen::Ref<EigenVec> in observe / step) and an
    environment written the way upstream's rsg_anymal is (Eigen expressions, footIndices_ + getlocalBodyIndex()) meet a compiler; the part that needs
    no GPU (the stand-in's arithmetic, quatToRotMat, rowOf, the stream-style RSFATAL_IF / RSINFO / RSWARN macros) runs here.  The existing
    span-typed environment and facade_test.cpp compile under the Eigen branch too (EigenVecRef then names Eigen::Ref<EigenVec>)."""
    compile_eigen_facade()
    r = subprocess.run([EIGEN_BIN], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "eigen stand-in + facade math OK" in r.stdout, r.stdout + r.stderr
    compile_eigen_facade("facade_test.cpp", os.path.join(os.path.dirname(EIGEN_BIN), "facade_test_eigen_branch"))


@pytest.mark.gpu
def test_eigen_typed_environment_runs_on_gpu(built_lib):
    """... and on the GPU: VectorizedEnvironment<ENVIRONMENT> over the Eigen-typed environment == the device-resident env with the same
    termination rule (by body: feet and knees sit on the shanks), 64 envs x 80 control steps with resets, one launch per control step"""
    compile_eigen_facade()
    r = subprocess.run([EIGEN_BIN, URDF], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "facade_eigen_test OK" in r.stdout and "equal to the device-resident env" in r.stdout, r.stdout + r.stderr


def test_fiber_scheduler_and_yaml_parser_on_cpu():
    """The host-side machinery of VectorizedEnvironment<ENV> (fibers parking in integrate(), cfg.yaml subset) needs no GPU."""
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    exe = os.path.join(os.path.dirname(BIN), "fiber_yaml_test")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-I", os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "fiber_yaml_test.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "fiber_yaml_test OK" in r.stdout, r.stdout + r.stderr
    # ... and on the portable ucontext path that a CET shadow-stack build selects (the hand-written switch returns on a foreign stack)
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-D__SHSTK__", "-I", os.path.join(ROOT, "include"), "-o", exe + "_ucontext",
                    os.path.join(ROOT, "tests", "cpp", "fiber_yaml_test.cpp")], check=True)
    r = subprocess.run([exe + "_ucontext"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "fiber_yaml_test OK" in r.stdout, r.stdout + r.stderr


def test_facade_host_side_against_the_c_abi_double():
    """The per-env World views, their fused flush, lazy downloads and the threaded scheduler under VectorizedEnvironment<ENVIRONMENT>
    (unmodified tests/cpp/anymal_env/Environment.hpp) against tests/cpp/rsb_host_double.cpp - a TEST DOUBLE of the rsb_world entry
    points with a toy update rule (test infrastructure: linked into this binary only, never into librsb.so).  Pins: 4 integrate()
    calls = ONE launch of 4 sub-steps and ONE rsb_view_exchange per control step, bit-identical to a flush per integrate(), on 1 and
    on 5 threads; bodies of different shapes (masked launches by count, writes between integrate() calls, integrate1() inside
    step()); no staged write is lost when other envs force uploads in the same round."""
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    exe = os.path.join(os.path.dirname(BIN), "facade_host_test")
    csrc = os.path.join(ROOT, "raisimlib_amd", "csrc")
    subprocess.run(["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", csrc, "-I", os.path.join(ROOT, "tests", "cpp"), "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "facade_host_test.cpp"), os.path.join(ROOT, "tests", "cpp", "rsb_host_double.cpp"),
                    os.path.join(csrc, "urdf_model.cpp"), os.path.join(csrc, "terrain_io.cpp"), "-lz"], check=True)
    r = subprocess.run([exe, os.path.join(ROOT, "raisimlib_amd", "rsc")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "facade_host_test OK" in r.stdout, r.stdout + r.stderr


LAUNCHER = os.path.join(ROOT, "tests", "cpp", "_build", "comm_launcher")


def compile_launcher():
    os.makedirs(os.path.dirname(LAUNCHER), exist_ok=True)
    lib = os.path.join(ROOT, "raisimlib_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", LAUNCHER,
                    os.path.join(ROOT, "tests", "cpp", "comm_launcher.cpp"), "-L", lib, "-lrsb", f"-Wl,-rpath,{lib}"], check=True)


def test_comm_launcher_compiles_and_reports_no_device(built_lib):
    """the C-ABI multi-process launcher (fork + pipe of the RCCL unique id) builds with plain g++; without a GPU it says so (77)"""
    compile_launcher()
    if built_lib.rsb_device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu test")
    r = subprocess.run([LAUNCHER, URDF], capture_output=True, text=True, timeout=60)
    assert r.returncode == 77 and "no HIP device" in r.stdout


@pytest.mark.gpu
def test_comm_launcher_runs_rsb_allgather_obs_across_processes(built_lib):
    """one process per GPU through rsb_comm_get_unique_id / rsb_comm_init / rsb_allgather_obs: two ranks on a box with >= 2
    GPUs, one rank (same fork / pipe / communicator path) on a 1-GPU box"""
    compile_launcher()
    # NCCL_DEBUG=INFO: RCCL prints its own banner ("RCCL version x.y.z ...") - the run binds the library by dlopen, so its identity is part of the evidence
    r = subprocess.run([LAUNCHER, URDF], capture_output=True, text=True, timeout=300, env=dict(os.environ, NCCL_DEBUG="INFO"))
    assert r.returncode == 0, r.stdout + r.stderr
    want = 2 if built_lib.rsb_device_count() >= 2 else 1
    assert f"comm_launcher OK ranks={want}" in r.stdout
    import re
    m = re.search(r"RCCL runtime version code (\d+), header version code (\d+)", r.stdout)
    assert m and int(m.group(1)) > 0 and int(m.group(2)) > 0, r.stdout      # (the build host has <rccl/rccl.h>: the constants were static_assert-ed against it)
    assert re.search(r"(RCCL|NCCL) version", r.stdout + r.stderr), (r.stdout + r.stderr)[-1500:]
