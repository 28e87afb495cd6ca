"""Diagnostic (GPU): can a sequence of rsb_control_step calls be captured into a HIP graph (torch.cuda.CUDAGraph on the
borrowed stream) and replayed, and what does it buy?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raisimlib_amd import Model, BatchedWorld, rsc_path, workload
N = 4096
dev = torch.device("cuda:0")
m = Model(urdf_path=rsc_path("anymal_c_like.urdf")); feet = np.asarray(m.collision_indices("_foot"), np.int32)
def make():
    w = BatchedWorld(m, N)
    gc, gv = workload.anymal_initial_state(N); kp, kd = workload.anymal_gains()
    w.set_pd_gains(kp, kd); w.set_state(gc, gv); w.set_pd_target(None, np.zeros((N, 18), np.float32))
    return w, torch.from_numpy(gc.astype(np.float32)).to(dev), torch.from_numpy(gv.astype(np.float32)).to(dev)
bank = [torch.from_numpy(workload.anymal_targets(N, k).astype(np.float32)).to(dev) for k in range(16)]
obs = torch.empty((N, 49), device=dev)
# reference: plain launches
w, g0, v0 = make()
s = torch.cuda.Stream(); torch.cuda.set_stream(s); w.set_stream(s.cuda_stream)
step = w.control_step_plan(4, obs.data_ptr(), feet, feet, g0.data_ptr(), v0.data_ptr(), N)
for k in range(112): step(bank[k % 16].data_ptr())
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(320): step(bank[k % 16].data_ptr())
torch.cuda.synchronize(); t_plain = (time.perf_counter() - t0) / 320
q_plain = w.get_state()[0]; w.close()
# graph: capture 16 control steps, replay
w, g0, v0 = make()
w.set_stream(s.cuda_stream)
step = w.control_step_plan(4, obs.data_ptr(), feet, feet, g0.data_ptr(), v0.data_ptr(), N)
for k in range(16): step(bank[k % 16].data_ptr())          # warm-up outside the graph (lazy init, first-launch attributes)
torch.cuda.synchronize()
try:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for k in range(16): step(bank[k].data_ptr())
    for _ in range(6): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / 320
    q_graph = w.get_state()[0]
    print(f"plain {t_plain * 1e3:.4f} ms/step, graph of 16 steps {t_graph * 1e3:.4f} ms/step, same state: {bool(np.array_equal(q_plain, q_graph))}")
except Exception as e:
    print("capture failed:", repr(e)[:300])
