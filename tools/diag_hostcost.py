import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raisimlib_amd import Model, VecEnv, rsc_path, workload
N = 4096
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
gc_init = np.zeros(19, np.float32); gc_init[2] = 0.6; gc_init[3] = 1.0; gc_init[7:] = workload.ANYMAL_NOMINAL_JOINTS
env = VecEnv(Model(urdf_path=rsc_path("anymal_c_like.urdf")), N, gc_init=gc_init, stream=stream.cuda_stream)
a = torch.zeros((N, 12), device=dev); ob = torch.empty((N, 34), device=dev); rew = torch.empty(N, device=dev); done = torch.empty(N, dtype=torch.uint8, device=dev)
nd = torch.zeros((), device=dev)
def t(name, f, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    th = time.perf_counter() - t0; torch.cuda.synchronize(); tt = time.perf_counter() - t0
    print(f"{name}: host {th / n * 1e6:.1f} us/call, total {tt / n * 1e6:.1f} us/call")
for _ in range(50): env.step(a, rew, done)
t("env.step", lambda: env.step(a, rew, done))
t("env.observe", lambda: env.observe(ob))
def s():
    global nd
    nd += done.sum()
t("torch done.sum accumulate", s)
w = env.world
import ctypes as C
L, h = w.L, w.handle
pa, pr, pd = C.c_void_p(a.data_ptr()), C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr())
t("raw rsb_env_step", lambda: L.rsb_env_step(h, pa, pr, pd, 1))
t("raw rsb_integrate(4)", lambda: L.rsb_integrate(h, 4))
