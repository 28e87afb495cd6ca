"""Resident launches (rsb_set_step_residency; kernel classes | 64): K control steps per launch of the step kernel, the env blocks stay in LDS, the
action stage of a closed-loop run is evaluated by the block's own wave.  The contract is bit-identity with the same control steps as separate
launches (lock-step: rsb_control_step K times / pass, step, pass ...): every row of every step's obs block / rollout, the done flags, and what the
world holds afterwards (state, contact lists, flags, sweeps) - over >= 1000 control steps with resets, open loop, linear stage and actor network,
configs 2, 3 and 5."""
import numpy as np
import pytest

import bench
from raisimlib_amd import BatchedWorld, workload
from test_gpu_closed_loop import Loop, equal

pytestmark = pytest.mark.gpu

PERIOD = 16


class Open:
    """one BASELINE config as the open-loop control-step caller: a PD-target bank on the device, obs block + done flags per control step, resets"""

    def __init__(self, config, n, resident, pipelined=False, period=PERIOD):
        import torch
        self.torch, self.n = torch, n
        dev = torch.device("cuda:0")
        r = bench.Recipe(5, -1.0, "collapsing") if config == "5c" else bench.Recipe(config, -1.0)      # ("5c": config 5 with the gains that cannot hold the humanoid up - every env falls and is reset)
        self.recipe = r
        w = BatchedWorld(r.model, n)
        w.set_stream(torch.cuda.current_stream().cuda_stream)
        r.setup_world(w, n, 0)
        gc0, gv0 = r.initial_state(n, 0)
        w.set_state(gc0, gv0)
        w.set_pd_target(None, np.zeros((n, r.model.nv), np.float32))
        self.g0 = torch.from_numpy(gc0.astype(np.float32)).to(dev)
        self.v0 = torch.from_numpy(gv0.astype(np.float32)).to(dev)
        self.bank = torch.from_numpy(np.stack([r.targets(n, k, 0).astype(np.float32) for k in range(period)])).to(dev)
        self.feet = np.asarray(r.feet, np.int32)
        self.od = w.obs_dim(len(self.feet))
        self.w, self.period, self.k = w, period, 0
        w.set_step_residency(resident)
        if pipelined:
            assert w.set_step_pipelining(True)

    def run(self, K):
        torch = self.torch
        obs = torch.zeros((K, self.n, self.od), dtype=torch.float32, device="cuda:0")
        done = torch.full((K, self.n), 7, dtype=torch.uint8, device="cuda:0")
        fn = self.w.control_steps_plan(workload.SUBSTEPS, self.bank.data_ptr(), self.period, obs.data_ptr(), self.n * self.od, self.feet, self.feet,
                                       self.g0.data_ptr(), self.v0.data_ptr(), self.n, done.data_ptr(), self.n)
        fn(K, self.k)
        self.k += K
        self.w.synchronize()
        return obs, done

    def final(self, full):
        w = self.w
        q, u = w.get_state()
        cnt, con = w.get_contacts()
        if not full:      # contact slots past an env's count hold whatever an earlier control step left there: a resident launch writes the records of its last step only
            con = con.copy()
            con[np.arange(con.shape[1])[None, :] >= cnt[:, None]] = 0
        return dict(q=q, u=u, cnt=cnt, con=con.tobytes(), flags=w.get_flags(), iters=w.get_solver_iterations(), pt=w.get_pd_target())


@pytest.mark.parametrize("config,n,runs,K,full", [(2, 4096, 20, 50, False), (2, 4096, 3, 30, True), (3, 4096, 10, 50, False), (5, 1024, 8, 25, False), ("5c", 1024, 6, 25, False), (2, 512, 3, 7, False)])
def test_resident_open_loop_equals_lockstep(built_lib, config, n, runs, K, full):
    """1000 control steps of config 2 (20 launches of 50), 500 of config 3, 200 of config 5: every control step's obs block and done flags, and the world
    after every launch, equal the lock-step sequence bit for bit.  `full`: with rsb_debug_resident_full_writes even the contact slots past the count."""
    ref, res = Open(config, n, False), Open(config, n, True)
    assert res.w.residency_status(0), "no resident class for this configuration"
    res.w.debug_resident_full_writes(full)
    resets = 0
    for r in range(runs):
        oa, da = ref.run(K)
        ob, db = res.run(K)
        assert ref.torch.equal(da, db), (r, "done")
        assert ref.torch.equal(oa, ob), (r, "obs", int((oa != ob).any(dim=2).any(dim=1).nonzero()[0]))
        assert equal(ref.final(full), res.final(full)) is None, (r, equal(ref.final(full), res.final(full)))
        resets += int(da.sum().item())
        assert int(da.max().item()) <= 1
    assert resets > 0 or config == 5 or runs * K < 200, "the workload never reset an env: the reset-in-LDS path would not be covered"      # (config 5's standing humanoids do not fall in 200 steps: "5c" covers their resets)
    assert res.w.residency_launches() == runs and ref.w.residency_launches() == 0
    ref.w.close(); res.w.close()


def test_resident_equals_pipelined_and_falls_back_outside_its_classes(built_lib):
    """Residency on top of a pipelined world: the resident launch joins the pipeline and gives the pipelined sequence's results; a world outside the
    resident classes (N not a multiple of the envs per workgroup; the Coulomb slip rule) runs the same call as separate control steps."""
    pip, res = Open(2, 2048, False, pipelined=True), Open(2, 2048, True, pipelined=True)
    for r in range(3):
        oa, da = pip.run(20)
        ob, db = res.run(20)
        assert pip.torch.equal(oa, ob) and pip.torch.equal(da, db)
    assert equal(pip.final(False), res.final(False)) is None
    assert res.w.residency_launches() == 3
    pip.w.close(); res.w.close()
    odd, ref = Open(2, 1023, True), Open(2, 1023, False)
    assert not odd.w.residency_status(0)
    oa, da = ref.run(10)
    ob, db = odd.run(10)
    assert ref.torch.equal(oa, ob) and ref.torch.equal(da, db) and odd.w.residency_launches() == 0
    odd.w.close(); ref.w.close()
    co = Open(2, 1024, True)
    co.w.set_slip_rule("coulomb")
    assert not co.w.residency_status(0)
    co.run(5)
    assert co.w.residency_launches() == 0
    co.w.close()


@pytest.mark.parametrize("stage,n,lpe,runs,K", [("linear", 4096, 0, 10, 100), ("mlp", 4096, 0, 10, 100), ("linear", 1000, 0, 3, 40), ("mlp", 1000, 0, 2, 30)])
def test_resident_closed_loop_equals_lockstep(built_lib, anymal, stage, n, lpe, runs, K):
    """>= 1000 control steps with the policy in the loop (10 launches of 100), linear stage and actor network: every row of every step's rollout
    (observation, action, reward, done) and the final state equal the lock-step run's (pass, step, pass ... as separate launches) bit for bit."""
    ref = Loop(anymal, n, False, lpe=lpe, stage=stage)
    res = Loop(anymal, n, False, lpe=lpe, stage=stage)
    res.env.world.set_step_residency(True)
    assert res.env.world.residency_status(1 if stage == "linear" else 2)
    res.env.world.debug_resident_full_writes(True)      # (Loop.final compares whole contact arrays)
    resets = 0
    for r in range(runs):
        ra, rb = ref.rollout_buffers(K), res.rollout_buffers(K)
        ref.run(K, ra)
        res.run(K, rb)
        ref.env.world.synchronize(); res.env.world.synchronize()
        for key in ("ob", "act", "reward", "done"):
            assert ref.torch.equal(ra[key], rb[key]), (r, key)
        resets += int(ra["done"].sum().item())
    assert equal(ref.final(), res.final()) is None
    assert resets > 0
    assert res.env.world.residency_launches() == runs
    ref.close(); res.close()


def _closed_loop_pair(make_env, run, K, runs, stage_kind):
    """the same closed-loop runs with residency off and on: rollouts of every run and the final world"""
    out = {}
    for resident in (False, True):
        env = make_env()
        env.world.set_step_residency(resident)
        if resident:
            assert env.world.residency_status(stage_kind)
            env.world.debug_resident_full_writes(True)
        ros = []
        for r in range(runs):
            ros.append(run(env, K))
            env.world.synchronize()
        q, u = env.world.get_state()
        cnt, con = env.world.get_contacts()
        out[resident] = (ros, q, u, cnt, con.tobytes(), env.world.residency_launches())
        env.close()
    return out[False], out[True]


def test_resident_closed_loop_on_a_height_map_and_with_the_humanoid(built_lib, anymal):
    """The other two configurations with a policy in the loop: the quadruped on config 3's height map (actor network and linear stage) and the Atlas-like
    humanoid (two envs per block, observation 70, action 30, an MLP 70 -> 96 -> 30): resident == lock-step bit for bit, resets included."""
    import torch
    from raisimlib_amd.vecenv import VecEnv
    dev = torch.device("cuda:0")
    # ---- config 3
    n, K = 2048, 50
    r3 = bench.Recipe(3, -1.0)
    maps, _ = r3.terrain(n, 0)
    gc0, gv0 = r3.initial_state(n, 0)
    mlp = [(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)) for a, b in workload.closed_loop_mlp(34, 12, hidden=(64, 64), out_scale=0.2)]
    W = torch.from_numpy(workload.closed_loop_policy(34, 12)).to(dev)
    noise = torch.from_numpy(workload.closed_loop_noise(n, 16)).to(dev)

    def make3():
        env = workload.closed_loop_env(anymal, n)
        env.world.add_height_map(128, 128, workload.HEIGHTMAP_SIZE, workload.HEIGHTMAP_SIZE, 0.0, 0.0, maps[0])
        env.set_reset_states(gc0, gv0)
        env.reset()
        return env

    def ro3(K):
        return {"ob": torch.zeros((K + 1, n, 34), device=dev), "act": torch.zeros((K, n, 12), device=dev), "reward": torch.zeros((K, n), device=dev),
                "done": torch.zeros((K, n), dtype=torch.uint8, device=dev)}

    def run_mlp3(env, K):
        ro = ro3(K); env.rollout_mlp(K, mlp, activation="tanh", noise=noise, rollout=ro); return ro

    def run_lin3(env, K):
        ro = ro3(K); env.rollout_linear(K, W, noise=noise, rollout=ro); return ro
    for run, kind in ((run_mlp3, 2), (run_lin3, 1)):
        a, b = _closed_loop_pair(make3, run, K, 3, kind)
        for ra, rb in zip(a[0], b[0]):
            for key in ra:
                assert torch.equal(ra[key], rb[key]), (kind, key)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and a[4] == b[4]
        assert sum(int(r["done"].sum().item()) for r in a[0]) > 0 and b[5] == 3 and a[5] == 0
    # ---- config 5's model
    n, K = 512, 30
    r5 = bench.Recipe(5, -1.0)
    gc5, gv5 = r5.initial_state(n, 0)
    g = torch.Generator(device="cpu").manual_seed(11)
    dims = [70, 96, 30]
    mlp5 = [(((torch.rand((dims[i + 1], dims[i]), generator=g) * 2 - 1) * (0.1 if i else 1.0) / np.sqrt(dims[i])).to(dev), torch.zeros(dims[i + 1], device=dev)) for i in range(2)]
    W5 = ((torch.rand((30, 70), generator=g) * 2 - 1) * 0.02).to(dev)
    noise5 = (torch.rand((8, n, 30), generator=g) * 0.1 - 0.05).to(dev)

    def make5():
        env = VecEnv(r5.model, n, gc_init=gc5[0].astype(np.float32), action_std=0.1)
        r5.setup_world(env.world, n, 0)
        env.set_reset_states(gc5, gv5)
        env.reset()
        return env

    def ro5(K):
        return {"ob": torch.zeros((K + 1, n, 70), device=dev), "act": torch.zeros((K, n, 30), device=dev), "done": torch.zeros((K, n), dtype=torch.uint8, device=dev)}

    def run_mlp5(env, K):
        ro = ro5(K); env.rollout_mlp(K, mlp5, activation="tanh", noise=noise5, rollout=ro); return ro

    def run_lin5(env, K):
        ro = ro5(K); env.rollout_linear(K, W5, noise=noise5, rollout=ro); return ro
    for run, kind in ((run_mlp5, 2), (run_lin5, 1)):
        a, b = _closed_loop_pair(make5, run, K, 2, kind)
        for ra, rb in zip(a[0], b[0]):
            for key in ra:
                assert torch.equal(ra[key], rb[key]), (kind, key)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and a[4] == b[4] and b[5] == 2
        assert np.isfinite(b[1]).all()


def test_closed_loop_with_profiling_on_stays_in_lockstep(built_lib, anymal):
    """with rsb_debug_phase_cycles active a step cannot go to the pipeline; the run must notice BEFORE it launches its action stage
    (the stage used to wait 10 s for a step that was never enqueued, the recovery then ran 2K steps) and equal the lock-step run."""
    import time
    ref, pip = Loop(anymal, 512, False), Loop(anymal, 512, True)
    pip.env.world.debug_phase_cycles(True, False)
    ra, rb = ref.rollout_buffers(12), pip.rollout_buffers(12)
    t0 = time.perf_counter()
    ref.run(12, ra); pip.run(12, rb)
    pip.env.world.step_pipeline_join(); ref.env.world.synchronize()
    assert time.perf_counter() - t0 < 5.0
    for key in ra:
        assert ref.torch.equal(ra[key], rb[key]), key
    assert pip.env.world.step_pipeline_fault() == (0, 0)
    assert abs(pip.env.world.get_world_time() - ref.env.world.get_world_time()) < 1e-12
    ref.close(); pip.close()


@pytest.mark.parametrize("n,hidden,act,normalise,K", [(4096, (128, 128), "leaky_relu", True, 60), (1024, (256, 192), "relu", True, 30), (512, (50, 21, 33), "tanh", False, 25), (8, (64,), "tanh", False, 10)])
def test_resident_mlp_stage_matches_torch_and_lockstep(built_lib, anymal, n, hidden, act, normalise, K):
    """The actor network INSIDE the step kernel (classes | 320 widths <= 128, | 448 widths <= 256): the recorded actions equal a torch forward pass over the
    recorded observations (+ noise) to rounding - the comparison that catches a wrong weight ring or permute, which resident == lock-step cannot (ADVICE r05:
    both would be wrong alike) - AND the run equals the lock-step one bit for bit.  Also a batch smaller than a workgroup's share of inputs (n = 8: two env blocks)."""
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(7)
    dims = [34, *hidden, 12]
    layers = []
    for i in range(len(dims) - 1):
        bound = (0.25 if i + 2 == len(dims) else 1.0) / np.sqrt(dims[i])
        layers.append((((torch.rand((dims[i + 1], dims[i]), generator=g) * 2 - 1) * bound).to(dev), ((torch.rand(dims[i + 1], generator=g) * 2 - 1) * 0.1).to(dev)))
    mean = (torch.rand(34, generator=g) * 0.2 - 0.1).to(dev) if normalise else None
    var = (torch.rand(34, generator=g) * 2 + 0.05).to(dev) if normalise else None
    noise = torch.from_numpy(workload.closed_loop_noise(n, 16)).to(dev)
    f = {"tanh": torch.tanh, "relu": torch.relu, "leaky_relu": lambda v: torch.nn.functional.leaky_relu(v, 0.01)}[act]

    def forward(ob):
        x = ob.double()
        if normalise:
            x = torch.clamp((x - mean.double()) * torch.rsqrt(var + 1e-8).double(), -10.0, 10.0)
        for i, (W, b) in enumerate(layers):
            x = x @ W.double().t() + b.double()
            if i + 1 < len(layers):
                x = f(x)
        return x

    ro = {}
    for resident in (False, True):
        env = workload.closed_loop_env(anymal, n)
        env.world.set_step_residency(resident)
        if resident:
            assert env.world.residency_status(2)
        for r in range(2):
            ro[resident] = {"ob": torch.zeros((K + 1, n, 34), device=dev), "act": torch.zeros((K, n, 12), device=dev),
                            "reward": torch.zeros((K, n), device=dev), "done": torch.zeros((K, n), dtype=torch.uint8, device=dev)}
            env.rollout_mlp(K, layers, activation=act, ob_mean=mean, ob_var=var, noise=noise, clip=3.0, rollout=ro[resident])
            env.world.synchronize()
        assert env.world.residency_launches() == (2 if resident else 0)
        env.close()
    for key in ro[False]:
        assert torch.equal(ro[False][key], ro[True][key]), key
    t = torch.arange(K, device=dev)
    nz = noise[(K + t) % noise.shape[0]]
    want = torch.clamp(forward(ro[True]["ob"][:K]) + nz.double(), -3.0, 3.0)
    err = (ro[True]["act"].double() - want).abs().max().item()
    assert err < 2e-5, err
