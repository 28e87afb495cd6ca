"""Pipelined control steps (rsb_set_step_pipelining): consecutive launches overlap on the device at workgroup granularity.  The contract is
bit-identity with the un-pipelined sequence of the same calls - envs are independent and every env's steps still run in order - whatever
the caller interleaves: reads, uploads, plain integrate() calls, its own kernels on the borrowed stream (after rsb_get_stream), a consumer
on another stream (rsb_step_pipeline_publish / _wait_event: the multi-GPU obs gather's pattern)."""
import numpy as np
import pytest

import bench
from raisimlib_amd import BatchedWorld, Model, workload

pytestmark = pytest.mark.gpu


class Rig:
    """one world of a bench.Recipe with device-resident targets, obs block, done flags and the in-kernel reset rule"""

    def __init__(self, recipe, n, pipe, lpe=0, stream=None, nbuf=1):
        import torch
        self.torch = torch
        dev = torch.device("cuda:0")
        self.n, self.recipe = n, recipe
        model, self.feet = recipe.model, np.asarray(recipe.feet, np.int32)
        w = BatchedWorld(model, n)
        if stream is not None:
            w.set_stream(stream.cuda_stream)
        recipe.setup_world(w, n, 0)
        if lpe:
            w.set_lanes_per_env(lpe)
        gc0, gv0 = recipe.initial_state(n, 0)
        self.gc0, self.gv0 = gc0, gv0
        self.gc0_d = torch.from_numpy(gc0.astype(np.float32)).to(dev); self.gv0_d = torch.from_numpy(gv0.astype(np.float32)).to(dev)
        w.set_state(gc0, gv0)
        w.set_pd_target(None, np.zeros((n, model.nv), np.float32))
        self.bank = [torch.from_numpy(recipe.targets(n, k, 0).astype(np.float32)).to(dev) for k in range(16)]
        self.obs = [torch.zeros((n, w.obs_dim(len(self.feet))), dtype=torch.float32, device=dev) for _ in range(nbuf)]
        self.done = torch.zeros(n, dtype=torch.uint8, device=dev)
        w.set_done_output(self.done.data_ptr())
        self.fns = [w.control_step_plan(workload.SUBSTEPS, o.data_ptr(), self.feet, self.feet, self.gc0_d.data_ptr(), self.gv0_d.data_ptr(), n) for o in self.obs]
        w.set_step_pipelining(pipe)
        self.w, self.k = w, 0

    def step(self, count=1):
        for _ in range(count):
            self.fns[self.k % len(self.fns)](self.bank[self.k % 16].data_ptr())
            self.k += 1

    def snapshot(self):
        q, u = self.w.get_state()
        cnt, con = self.w.get_contacts()
        return dict(q=q, u=u, cnt=cnt, con=con.tobytes(), flags=self.w.get_flags(), iters=self.w.get_solver_iterations(),
                    obs=[o.cpu().numpy() for o in self.obs], done=self.done.cpu().numpy())

    def close(self):
        self.w.close()


def same(a, b):
    for k in a:
        if isinstance(a[k], list):
            if not all(np.array_equal(x, y) for x, y in zip(a[k], b[k])):
                return k
        elif isinstance(a[k], bytes):
            if a[k] != b[k]:
                return k
        elif not np.array_equal(a[k], b[k]):
            return k
    return None


@pytest.mark.parametrize("config,n,lpe", [(2, 4096, 0), (2, 1000, 0), (2, 512, 32), (2, 256, 64), (2, 20000, 0), (3, 4096, 0), (5, 4096, 0), (5, 300, 64)])
def test_pipelined_steps_are_bit_identical(built_lib, config, n, lpe):
    """60 control steps of the benchmark recipes (targets from device memory, obs block, in-kernel resets), pipelined and not: state,
    contact lists, flags, sweep counts, obs rows and done flags equal bit for bit; all 60 launches went through the pipeline, joined once
    (by the reads at the end).  Batches that fill the chip exactly (4096 x 16 lanes), partly (grid not a multiple of 8: the plain block
    order, hand-over at agent scope), five times the chip (20 000 envs: a launch is dispatched completely only when most of its predecessor has
    finished), and the 32- / 64-lane mappings (other kernel instances)."""
    recipe = bench.Recipe(config, -1.0)
    out = {}
    for pipe in (False, True):
        r = Rig(recipe, n, pipe, lpe=lpe)
        r.step(60)
        out[pipe] = r.snapshot()
        st = r.w.step_pipelining_stats()
        assert st == ((60, 1) if pipe else (0, 0)), st
        r.close()
    assert same(out[False], out[True]) is None, same(out[False], out[True])
    assert out[True]["done"].dtype == np.uint8 and np.isfinite(out[True]["q"]).all()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_any_other_call_joins_the_pipeline(built_lib, seed):
    """A random program of library calls on a pipelined world and on a plain twin: control steps in bursts, state / contact / flag reads,
    uploads of states for some envs, gain changes, plain and masked integrate() calls, a synchronize.  Every read returns the same bytes on
    both, at every point of the program."""
    recipe = bench.Recipe(2, -1.0)
    n = 2048
    rng = np.random.default_rng(seed)
    prog = []
    for _ in range(40):
        op = rng.choice(["steps", "steps", "steps", "state", "contacts", "set_state", "gains", "integrate", "masked", "sync", "flags"])
        prog.append((op, int(rng.integers(1, 6)), int(rng.integers(0, 1 << 30))))
    logs = {}
    for pipe in (False, True):
        r = Rig(recipe, n, pipe)
        log = []
        for op, cnt, sd in prog:
            g = np.random.default_rng(sd)
            if op == "steps":
                r.step(cnt)
            elif op == "state":
                log.append(np.concatenate([x.ravel() for x in r.w.get_state()]).tobytes())
            elif op == "contacts":
                c, con = r.w.get_contacts(); log.append(c.tobytes() + con.tobytes())
            elif op == "flags":
                log.append(r.w.get_flags().tobytes() + r.w.get_solver_iterations().tobytes())
            elif op == "set_state":
                mask = (g.random(n) < 0.3).astype(np.uint8)
                r.w.set_state(r.gc0, r.gv0, mask=mask)
            elif op == "gains":
                kp, kd = np.array(recipe.kp, np.float32).copy(), np.array(recipe.kd, np.float32).copy()
                kp[6:] *= float(g.uniform(0.8, 1.2))
                r.w.set_pd_gains(kp, kd)
            elif op == "integrate":
                r.w.integrate(cnt)
            elif op == "masked":
                r.w.integrate_masked((g.random(n) < 0.5).astype(np.uint8), 1)
            elif op == "sync":
                r.w.synchronize()
        log.append(b"".join(v if isinstance(v, bytes) else np.ascontiguousarray(v).tobytes() for v in [x for k, x in sorted(r.snapshot().items()) if not isinstance(x, list)]))
        logs[pipe] = (log, r.w.step_pipelining_stats())
        r.close()
    assert len(logs[False][0]) == len(logs[True][0])
    for i, (a, b) in enumerate(zip(logs[False][0], logs[True][0])):
        assert a == b, f"read {i} differs"
    launches, joins = logs[True][1]
    assert launches == sum(c for op, c, _ in prog if op == "steps") and joins >= 1


def test_two_pipelined_worlds_and_a_change_of_the_lane_mapping(built_lib):
    """Two pipelined worlds of one process stepped alternately (their waiting workgroups compete for the same SIMDs: each world's gate only speaks
    for its own launches), and a change of the lanes-per-env mapping in mid-run (another grid: the pipeline's bookkeeping is rebuilt): both
    equal to plain twins."""
    ra, rb = bench.Recipe(2, -1.0), bench.Recipe(3, -1.0)
    out = {}
    for pipe in (False, True):
        a, b = Rig(ra, 4096, pipe), Rig(rb, 4096, pipe)
        for k in range(40):
            a.step(1); b.step(2 if k % 3 == 0 else 1)
        a.w.set_lanes_per_env(32)
        for k in range(10):
            a.step(1); b.step(1)
        a.w.set_lanes_per_env(16)
        a.step(5)
        out[pipe] = (a.snapshot(), b.snapshot(), a.w.step_pipelining_stats(), b.w.step_pipelining_stats())
        a.close(); b.close()
    assert same(out[False][0], out[True][0]) is None and same(out[False][1], out[True][1]) is None
    assert out[True][2][0] == 55 and out[True][3][0] == 64


def test_borrowed_stream_sees_the_steps_after_get_stream(built_lib):
    """The world on a torch stream: after rsb_get_stream (which joins) the caller's own kernels on that stream read completed steps - the
    obs block copied by torch after every burst equals the plain world's."""
    import torch
    recipe = bench.Recipe(2, -1.0)
    n = 4096
    got = {}
    for pipe in (False, True):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            r = Rig(recipe, n, pipe, stream=s)
            copies = []
            for burst in (3, 1, 7, 2):
                r.step(burst)
                assert r.w.get_stream() == s.cuda_stream
                copies.append(r.obs[0].clone())          # torch's copy kernel, on the borrowed stream, right behind the join
            s.synchronize()
            got[pipe] = [c.cpu().numpy() for c in copies]
            r.close()
    for a, b in zip(got[False], got[True]):
        assert np.array_equal(a, b)


def test_consumer_on_another_stream_with_publish_and_wait_event(built_lib):
    """The multi-GPU obs gather's pattern without a second GPU: the control step writes its obs block into one of two buffers; a consumer on
    ITS OWN stream copies the block of step k (rsb_step_pipeline_publish orders it behind step k only) while step k + 1 is already running;
    step k + 2, which overwrites the buffer, waits for the consumer's event (rsb_step_pipeline_wait_event).  All 40 copies equal the plain
    world's obs of the same step, and the pipeline was never joined in between."""
    import torch
    recipe = bench.Recipe(2, -1.0)
    n, K = 4096, 40
    ref = []
    r = Rig(recipe, n, False, nbuf=2)
    for k in range(K):
        r.step(1)
        r.w.synchronize()
        ref.append(r.obs[k % 2].cpu().numpy())
    r.close()
    r = Rig(recipe, n, True, nbuf=2)
    cs = torch.cuda.Stream()
    out = [torch.empty_like(r.obs[0]) for _ in range(K)]
    evs = [None, None]
    for k in range(K):
        b = k % 2
        if evs[b] is not None:
            r.w.step_pipeline_wait_event(evs[b].cuda_event)      # the copy that still reads buffer b
        r.step(1)
        r.w.step_pipeline_publish(cs.cuda_stream)
        with torch.cuda.stream(cs):
            out[k].copy_(r.obs[b], non_blocking=True)
            evs[b] = torch.cuda.Event()
            evs[b].record(cs)
    launches, joins = r.w.step_pipelining_stats()
    assert (launches, joins) == (K, 0)
    torch.cuda.synchronize()
    for k in range(K):
        assert np.array_equal(out[k].cpu().numpy(), ref[k]), k
    r.close()


@pytest.mark.parametrize("which", ["fixed base", "second flank", "trapezoid", "coulomb"])
def test_pipelined_twins_of_the_other_kernel_classes(built_lib, which):
    """Classes 1, 4, 8 and 32 have pipelined twins as well (17, 20, 24, 48): a fixed-base pendulum, the quadruped with two contacts per primitive on
    the benchmark's height map, the trapezoidal scheme, the classical Coulomb slip rule."""
    import torch
    dev = torch.device("cuda:0")
    n, K = 1024, 30
    res = {}
    for pipe in (False, True):
        if which == "fixed base":
            from test_oracle_kat import FIXED_PENDULUM
            model = Model(urdf_string=FIXED_PENDULUM.format(l=0.5, m=1.0))
            w = BatchedWorld(model, n)
            gc = np.zeros((n, model.nq)); gc[:, 3] = 1.0; gc[:, 7:] = np.linspace(-1, 1, n)[:, None]
            w.set_state(gc, np.zeros((n, model.nv)))
            w.set_pd_gains(np.zeros(model.nv, np.float32), np.zeros(model.nv, np.float32))
            feet = np.zeros(0, np.int32)
        else:
            recipe = bench.Recipe(3 if which == "second flank" else 2, -1.0)
            model = recipe.model
            w = BatchedWorld(model, n)
            recipe.setup_world(w, n, 0)
            if which == "second flank":
                w.set_heightmap_contacts(2, 30.0)
            elif which == "coulomb":
                w.set_slip_rule("coulomb")
            else:
                w.set_integration_scheme("trapezoid")
            gc, gv = recipe.initial_state(n, 0)
            w.set_state(gc, gv)
            feet = np.asarray(recipe.feet, np.int32)
        w.set_pd_target(None, np.zeros((n, model.nv), np.float32))
        tgt = torch.from_numpy(np.asarray(gc, np.float32)).to(dev)
        obs = torch.zeros((n, w.obs_dim(len(feet))), dtype=torch.float32, device=dev)
        fn = w.control_step_plan(2, obs.data_ptr(), feet, None, 0, 0, n)
        w.set_step_pipelining(pipe)
        for _ in range(K):
            fn(tgt.data_ptr())
        q, u = w.get_state()
        res[pipe] = (q, u, obs.cpu().numpy(), w.step_pipelining_stats())
        w.close()
    assert res[True][3][0] == K and res[False][3][0] == 0
    assert np.array_equal(res[False][0], res[True][0]) and np.array_equal(res[False][1], res[True][1]) and np.array_equal(res[False][2], res[True][2])
    assert np.isfinite(res[True][0]).all()


def test_profiling_and_the_peer_exchange_fall_back_to_plain_launches(built_lib):
    """What has no pipelined twin runs un-pipelined, silently and correctly: launches with the cycle stamps on (profiling instance) and
    control steps that upload a d_target."""
    recipe = bench.Recipe(2, -1.0)
    r = Rig(recipe, 512, True)
    r.step(3)
    r.w.debug_phase_cycles(True, read=False)
    r.step(2)
    r.w.debug_phase_cycles(False, read=False)
    r.step(2)
    launches, joins = r.w.step_pipelining_stats()
    assert launches == 5 and joins >= 1
    q, _ = r.w.get_state()
    assert np.isfinite(q).all()
    r.close()


def test_the_switch_stays_off_under_a_serialising_profiler(built_lib, monkeypatch):
    """rocprofv3 --pmc (ROCPROF_COUNTER_COLLECTION=1) runs one kernel at a time in an order of its own: pipelined launches would wait for
    predecessors that may not start.  The library keeps the steps in lock-step there (and with RSB_STEP_PIPELINING=0) and says so."""
    recipe = bench.Recipe(2, -1.0)
    for var in ("ROCPROF_COUNTER_COLLECTION", "RSB_STEP_PIPELINING"):
        monkeypatch.setenv(var, "1" if var.startswith("ROCPROF") else "0")
        r = Rig(recipe, 512, False)
        assert r.w.set_step_pipelining(True) is False and not r.w.step_pipelining_enabled()
        r.step(4)
        assert r.w.step_pipelining_stats() == (0, 0)
        r.close()
        monkeypatch.delenv(var)
    r = Rig(recipe, 512, False)
    assert r.w.set_step_pipelining(True) is True and r.w.step_pipelining_enabled()
    r.close()
