"""Host profile of the TEMPLATE PATH: RaisimGymVecEnv.step + observe over N unmodified Environment.hpp objects - where the
0.85 ms per control step go.  Counters are clock stamps compiled into the facade (VectorizedEnvironment::stepProfile, BatchedWorld::flushPrepNs /
flushExchangeNs) and into rsb_view_exchange (rsb_debug_view_profile); no sampling profiler is installed on the boxes.
usage: python tools/prof_template_path.py [N] [steps] [threads ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from raisimlib_amd.gym import RaisimGymVecEnv, build_env_module, load_env_module

ROOT = bench.ROOT
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
threads_list = [int(x) for x in sys.argv[3:]] or [16]
rsc = os.path.join(ROOT, "raisimlib_amd", "rsc")
build_env_module(os.path.join(ROOT, "tests", "cpp", "anymal_env"), name="rsg_anymal")
mod = load_env_module("rsg_anymal")
rng = np.random.default_rng(0)
acts = [rng.uniform(-1, 1, (n, 12)).astype(np.float32) for _ in range(8)]
print(f"template path host profile: N = {n}, {steps} control steps, cpu quota {bench.cpu_quota()}, affinity {len(os.sched_getaffinity(0))} CPUs")
for threads in threads_list:
    cfg = (f"num_envs: {n}\nnum_threads: {threads}\nsimulation_dt: 0.0025\ncontrol_dt: 0.01\nrender: false\naction_std: 0.3\n"
           "reward:\n  forwardVel:\n    coeff: 0.3\n  torque:\n    coeff: -4e-5\n")
    env = RaisimGymVecEnv(mod.RaisimGymEnv(rsc, cfg, False), normalize_ob=False)
    env.reset()
    for k in range(20):
        env.step(acts[k % 8]); env.observe(False)
    p0 = dict(env.wrapper.stepProfile(True))
    t_step = t_obs = 0.0
    t0 = time.perf_counter()
    for k in range(steps):
        a = time.perf_counter()
        env.step(acts[k % 8])
        b = time.perf_counter()
        env.observe(False)
        t_obs += time.perf_counter() - b
        t_step += b - a
    wall = time.perf_counter() - t0
    p = dict(env.wrapper.stepProfile(False))
    us = lambda ns: ns / steps / 1e3
    prep = p["flush_prep_ns"] - p0["flush_prep_ns"]; exch = p["flush_exchange_ns"] - p0["flush_exchange_ns"]
    rounds = p["total_ns"] - p["flush_ns"]
    print(f"\nthreads {threads}: {n * 4 * steps / wall / 1e6:.1f} M env-steps/s, {wall / steps * 1e6:.0f} us per control step (step() {t_step / steps * 1e6:.0f} + observe() {t_obs / steps * 1e6:.0f})")
    print(f"  VectorizedEnvironment::step (C++)                 {us(p['total_ns']):7.0f} us   ({p['flushes'] / steps:.2f} flushes per step; python / pybind around it: {t_step / steps * 1e6 - us(p['total_ns']):.0f} us)")
    print(f"    rounds of the N step() bodies on the fiber threads + scheduler hand-overs   {us(rounds):7.0f} us")
    print(f"    flushes (caller's thread, everybody else waits)                              {us(p['flush_ns']):7.0f} us")
    print(f"      prepare (scan of pending counters, upload lists)                           {us(prep):7.0f} us")
    print(f"      rsb_view_exchange                                                          {us(exch):7.0f} us")
    print(f"        enqueue uploads (+ masked state-row kernels of reset envs)               {us(p['exchange_upload_enqueue_ns']):7.0f} us")
    print(f"        enqueue the fused launch                                                 {us(p['exchange_launch_enqueue_ns']):7.0f} us")
    print(f"        enqueue downloads (gc, gv, counts, contacts, generalized force)          {us(p['exchange_download_enqueue_ns']):7.0f} us")
    print(f"        wait for the stream (copies + the step kernel ~110 us + copies)          {us(p['exchange_wait_ns']):7.0f} us")
    env.close()
