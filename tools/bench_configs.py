"""Secondary configurations of SURVEY.md 8d on one GPU (not bench lines; parity-test cases timed for DESIGN.md):
config 3 = ANYmal-like on a shared 128x128 height map (12.8 m x 12.8 m), config 5 = Atlas-like standing PD, kmax 16.
Prints env-steps/s and the step kernel's mean launch time (library event ring)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raisimlib_amd import Model, BatchedWorld, rsc_path, workload

N, STEPS, WARM = 4096, 200, 100


def run(name, world, make_targets, feet, g0, v0, nq, allowed=None):
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream); world.set_stream(stream.cuda_stream)
    bank = [torch.from_numpy(make_targets(k).astype(np.float32)).to(dev) for k in range(16)]
    od = world.obs_dim(len(feet))
    obs = torch.empty((N, od), dtype=torch.float32, device=dev)
    g0d = torch.from_numpy(g0.astype(np.float32)).to(dev); v0d = torch.from_numpy(v0.astype(np.float32)).to(dev)
    world.enable_timing(STEPS)
    step = world.control_step_plan(4, obs.data_ptr(), feet, allowed if allowed is not None else feet, g0d.data_ptr(), v0d.data_ptr(), N)
    for k in range(WARM):
        step(bank[k % 16].data_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(STEPS):
        step(bank[(WARM + k) % 16].data_ptr())
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    kms = world.read_kernel_ms(STEPS)
    it = world.get_solver_iterations(); cnt, _ = world.get_contacts(); q, _ = world.get_state()
    print(f"{name}: {N * 4 * STEPS / el / 1e6:.1f}M env-steps/s, {el / STEPS * 1e3:.4f} ms/control step, kernel mean {kms.mean() * 1e3:.1f} us, "
          f"lanes/env {world.lanes_per_env()}, contacts/env {cnt.mean():.2f}, sweeps mean {it.mean():.2f} max {it.max()}, "
          f"base height {q[:, 2].mean():.3f}, finite {bool(np.isfinite(q).all())}")
    world.close()


def config3():
    m = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
    w = BatchedWorld(m, N)
    H = workload.smoothed_heightmap(128, 128, amplitude=0.1, seed=7)
    w.add_height_map(128, 128, 12.8, 12.8, 0.0, 0.0, H)
    gc, gv = workload.anymal_initial_state(N); kp, kd = workload.anymal_gains()
    gc[:, 2] += 0.12                                  # clear the terrain's +-0.1 m
    w.set_pd_gains(kp, kd); w.set_state(gc, gv); w.set_pd_target(None, np.zeros((N, 18), np.float32))
    run("config 3 (ANYmal-like, 128x128 height map)", w, lambda k: workload.anymal_targets(N, k), m.collision_indices("_foot"), gc, gv, 19)


def config5():
    m = Model(urdf_path=rsc_path("atlas_like.urdf"))
    w = BatchedWorld(m, N); w.set_max_contacts(16)
    rng = np.random.default_rng(3)
    gc = np.zeros((N, 37)); gc[:, 2] = 0.95; gc[:, 3] = 1.0
    gv = np.zeros((N, 36))
    kp = np.zeros(36, np.float32); kd = np.zeros(36, np.float32); kp[6:] = 200.0; kd[6:] = 5.0
    w.set_pd_gains(kp, kd); w.set_state(gc, gv); w.set_pd_target(None, np.zeros((N, 36), np.float32))

    def targets(k):
        pt = np.zeros((N, 37)); pt[:, 3] = 1.0
        pt[:, 7:] = np.random.default_rng([77, k]).uniform(-0.1, 0.1, (N, 30))
        return pt
    every = list(range(m.blob.ncol))                  # every primitive may touch: only non-finite states reset
    run("config 5 (Atlas-like, kmax 16, standing PD + U(-0.1,0.1) rad)", w, targets, every[:8], gc, gv, 37, allowed=every)


if __name__ == "__main__":
    which = sys.argv[1:] or ["3", "5"]
    if "3" in which: config3()
    if "5" in which: config5()
