"""The peer-mapped obs exchange (rsb_obs_peer_*, include/rsb.h): the step kernel's epilogue stores each env's obs row into every
rank's gathered buffer (write-through stores) and the last wave of the launch writes the step number into every rank's flag word; no
collective, no copy kernel, no cache flush.

A 1-GPU box can check the mechanism, not the xGMI path: one rank mapped onto itself, two worlds of one process as two ranks
(plain pointers), and two PROCESSES sharing the GPU through hipIpc handles (tests/cpp/peer_launcher.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from common import ROOT
from raisimlib_amd import BatchedWorld, workload

pytestmark = pytest.mark.gpu
_hip = None


def read_device(ptr, shape):
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
    out = np.empty(shape, np.float32)
    assert _hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(out.nbytes), C.c_int(2)) == 0
    return out


def _shard(anymal, n, lo):
    w = BatchedWorld(anymal, n)
    gc, gv = workload.anymal_initial_state(n, env_offset=lo, height=0.56)
    kp, kd = workload.anymal_gains()
    w.set_pd_gains(kp, kd); w.set_state(gc, gv); w.set_pd_target(gc, np.zeros((n, 18)))
    return w


def test_one_rank_mapped_onto_itself(anymal):
    import torch
    n, feet = 300, np.asarray(anymal.collision_indices("_foot"), np.int32)          # ragged: not a multiple of the envs per workgroup
    od = 19 + 18 + 3 * len(feet)
    w = _shard(anymal, n, 0)
    ref = _shard(anymal, n, 0)
    handle = w.obs_peer_create(1, 0, feet)
    assert len(handle) == 64
    w.obs_peer_connect(handle)
    own = torch.zeros((n, od), dtype=torch.float32, device="cuda")
    g0 = torch.from_numpy(workload.anymal_initial_state(n)[0].astype(np.float32)).cuda(); v0 = torch.zeros((n, 18), device="cuda")
    step = w.control_step_plan(4, own.data_ptr(), feet, feet, g0.data_ptr(), v0.data_ptr(), n)
    step_none = w.control_step_plan(4, 0, feet, feet, g0.data_ptr(), v0.data_ptr(), n)                 # no obs block of the caller's
    step_ref = ref.control_step_plan(4, own.data_ptr(), feet, feet, g0.data_ptr(), v0.data_ptr(), n)   # the plain class of the kernel
    ptrs = set()
    for k in range(6):
        pt = torch.from_numpy(workload.anymal_targets(n, k).astype(np.float32)).cuda()
        (step if k % 2 == 0 else step_none)(pt.data_ptr())
        gathered = w.obs_peer_wait()
        w.synchronize()
        ptrs.add(gathered)
        got = read_device(gathered, (n, od))
        if k % 2 == 0:
            assert np.array_equal(got, own.cpu().numpy())                 # the same rows the caller's own block received
        step_ref(pt.data_ptr()); ref.synchronize()
        assert np.array_equal(got, own.cpu().numpy())                     # ... and what a world without the exchange computes
        assert np.array_equal(w.get_state()[0], ref.get_state()[0])
    assert len(ptrs) == 2 and np.abs(got[:, 37:]).max() > 0               # double-buffered by step parity; feet are pressing
    w.obs_peer_destroy(); w.close(); ref.close()


def test_two_worlds_of_one_process_as_two_ranks(anymal):
    import torch
    n, feet = 192, np.asarray(anymal.collision_indices("_foot"), np.int32)
    od = 19 + 18 + 3 * len(feet)
    ws = [_shard(anymal, n, r * n) for r in range(2)]
    for r, w in enumerate(ws):
        w.obs_peer_create(2, r, feet)
    bases = [w.obs_peer_base() for w in ws]
    for w in ws:
        w.obs_peer_connect_ptrs(bases)
    full = _shard(anymal, 2 * n, 0)                                        # the unsharded world
    blk = torch.zeros((2 * n, od), dtype=torch.float32, device="cuda")
    for k in range(4):
        pts = workload.anymal_targets(2 * n, k).astype(np.float32)
        keep = []
        for r, w in enumerate(ws):
            pt = torch.from_numpy(pts[r * n:(r + 1) * n]).cuda(); keep.append(pt)
            w.control_step_plan(4, 0, feet, None, 0, 0, n)(pt.data_ptr())
        gathered = [w.obs_peer_wait() for w in ws]                        # (after BOTH launches are enqueued: the waits are stream-side)
        for w in ws:
            w.synchronize()
        ptf = torch.from_numpy(pts).cuda()
        full.control_step_plan(4, blk.data_ptr(), feet, None, 0, 0, 2 * n)(ptf.data_ptr()); full.synchronize()
        want = blk.cpu().numpy()
        for r in range(2):
            assert np.array_equal(read_device(gathered[r], (2 * n, od)), want), (k, r)     # rank-major, bit-identical to the unsharded world
    for w in ws:
        w.obs_peer_destroy(); w.close()
    full.close()


def test_two_processes_share_the_gpu_through_ipc_handles(built_lib):
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "peer_launcher")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    lib = os.path.join(ROOT, "raisimlib_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "peer_launcher.cpp"), "-L", lib, "-lrsb", "-L", "/opt/rocm/lib", "-lamdhip64",
                    f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([exe, os.path.join(ROOT, "raisimlib_amd", "rsc", "anymal_c_like.urdf"), "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "peer_launcher OK ranks=2" in r.stdout
