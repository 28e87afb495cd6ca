"""End-to-end rate of the device-resident vectorised env (rsb_env_step + rsb_env_observe with torch CUDA tensors):
action tensor in, reward / done / observation tensors out, nothing on the host per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raisimlib_amd import Model, VecEnv, rsc_path, workload

N, STEPS, WARM = 4096, 300, 100
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
gc_init = np.zeros(19, np.float32); gc_init[2] = 0.6; gc_init[3] = 1.0; gc_init[7:] = workload.ANYMAL_NOMINAL_JOINTS
EARLY = "--early-termination" in sys.argv
env = VecEnv(Model(urdf_path=rsc_path("anymal_c_like.urdf")), N, gc_init=gc_init, stream=stream.cuda_stream, early_termination=EARLY)
gen = torch.Generator(device=dev); gen.manual_seed(0)
acts = [torch.empty((N, env.num_acts), device=dev).uniform_(-1, 1, generator=gen) for _ in range(16)]   # U(-1,1) * 0.3 rad
ob = torch.empty((N, env.num_obs), device=dev); rew = torch.empty(N, device=dev); done = torch.empty(N, dtype=torch.uint8, device=dev)
for k in range(WARM):
    env.step(acts[k % 16], rew, done, ob)
env.world.enable_timing(STEPS)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(STEPS):
    env.step(acts[k % 16], rew, done, ob)
t_host = time.perf_counter() - t0
torch.cuda.synchronize(); el = time.perf_counter() - t0
kms = env.world.read_kernel_ms(STEPS)
it = env.world.get_solver_iterations(); cnt = env.world.get_contacts()[0]
print(f"VecEnv (device tensors{', early termination' if EARLY else ''}): {N * 4 * STEPS / el / 1e6:.1f}M env-steps/s = {N * STEPS / el / 1e6:.2f}M control steps/s, "
      f"{el / STEPS * 1e3:.4f} ms per vectorised step, resets in the last step {int(done.to(torch.int32).sum().item())}, mean reward {rew.mean().item():.3f}, "
      f"mean height {ob[:, 0].mean().item():.3f} | step kernel mean {kms.mean() * 1e3:.1f} us, host enqueue {t_host / STEPS * 1e3:.4f} ms/step, "
      f"sweeps mean {it.mean():.2f} max {it.max()}, contacts/env {cnt.mean():.2f}")
env.close()
