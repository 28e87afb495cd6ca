"""Specialised code objects of the step kernel (rsb_set_specialization; raisimlib_amd/csrc/step_spec.h, rsb_spec.hip): the SAME kernel class compiled with the
model's dimensions and the world's switches as compile-time constants.  The contract is that specialisation changes speed, never results: every control
step's obs block and done flags and the world afterwards equal the ahead-of-time class's bit for bit - lock-step, pipelined and resident launches, open
loop and with a policy in the loop, configs 2, 3 and 5 - and that the launch counters say which code ran."""
import os
import subprocess
import sys

import numpy as np
import pytest

from raisimlib_amd import _capi
from test_gpu_closed_loop import Loop, equal
from test_gpu_resident import Open

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("config,n,mode,runs,K", [(2, 4096, "lockstep", 3, 20), (2, 4096, "resident", 6, 50), (2, 2048, "pipelined", 3, 20), (3, 2048, "resident", 3, 30),
                                                  (3, 1024, "lockstep", 2, 10), (5, 512, "lockstep", 2, 10), (5, 512, "resident", 3, 20), ("5c", 512, "resident", 3, 25)])
def test_specialized_equals_generic(built_lib, config, n, mode, runs, K):
    gen = Open(config, n, mode == "resident", pipelined=mode == "pipelined")
    spe = Open(config, n, mode == "resident", pipelined=mode == "pipelined")
    gen.w.set_specialization("off")
    spe.w.set_specialization("compile")
    for r in range(runs):
        oa, da = gen.run(K)
        ob, db = spe.run(K)
        assert gen.torch.equal(da, db), (r, "done")
        assert gen.torch.equal(oa, ob), (r, "obs", int((oa != ob).any(dim=2).any(dim=1).nonzero()[0]))
        assert equal(gen.final(False), spe.final(False)) is None, (r, equal(gen.final(False), spe.final(False)))
    mode_s, ns, ng = spe.w.specialization_status()
    assert mode_s == 2 and ns > 0 and ng == 0, (mode_s, ns, ng)
    mode_g, ns, ng = gen.w.specialization_status()
    assert mode_g == 0 and ns == 0 and ng > 0, (mode_g, ns, ng)
    if mode == "resident":
        assert spe.w.residency_launches() == runs
    gen.w.close(); spe.w.close()


@pytest.mark.parametrize("stage", ["linear", "mlp"])
def test_specialized_closed_loop_equals_generic(built_lib, anymal, stage):
    """the resident classes with the action stage inside (linear policy, actor network) and their lock-step twins (stage kernel + plain class)"""
    n, K = 2048, 40
    for resident in (True, False):
        gen, spe = Loop(anymal, n, False, stage=stage), Loop(anymal, n, False, stage=stage)
        gen.env.world.set_specialization("off")
        spe.env.world.set_specialization("compile")
        for L in (gen, spe):
            L.env.world.set_step_residency(resident)
            L.env.world.debug_resident_full_writes(True)
        for r in range(3):
            ra, rb = gen.rollout_buffers(K), spe.rollout_buffers(K)
            gen.run(K, ra); spe.run(K, rb)
            gen.env.world.synchronize(); spe.env.world.synchronize()
            for key in ("ob", "act", "reward", "done"):
                assert gen.torch.equal(ra[key], rb[key]), (resident, r, key)
        assert equal(gen.final(), spe.final()) is None
        _, ns, ng = spe.env.world.specialization_status()
        assert ns > 0 and ng == 0, (resident, ns, ng)
        gen.close(); spe.close()


def test_a_change_of_the_world_changes_the_key(built_lib):
    """switches that are part of the key (here: self-collision off -> no candidate pairs; another sub-step count) get code objects of their own; results stay
    those of the ahead-of-time class"""
    gen, spe = Open(2, 1024, False), Open(2, 1024, False)
    gen.w.set_specialization("off"); spe.w.set_specialization("compile")
    for w in (gen.w, spe.w):
        w.set_self_collision(False)
    oa, da = gen.run(10)
    ob, db = spe.run(10)
    assert gen.torch.equal(oa, ob) and gen.torch.equal(da, db)
    _, ns, ng = spe.w.specialization_status()
    assert ns == 10 and ng == 0
    gen.w.close(); spe.w.close()


_CHILD = r"""
import os, sys
sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, {root!r})
import numpy as np, torch
from test_gpu_resident import Open
o = Open(2, 256, False)
o.w.set_specialization({mode!r})
obs, done = o.run(5)
print("STATUS", *o.w.specialization_status())
print("SUM", float(obs.double().sum().item()))
"""


def test_cached_mode_never_compiles_and_a_bad_code_object_is_refused(built_lib, tmp_path):
    """$RSB_SPEC_DIR empty + the default mode: the ahead-of-time class runs, the key is appended to $RSB_SPEC_RECORD; a file of that name that is not a code
    object is refused with a message and the ahead-of-time class still runs; compiled from the recorded line (rsb_spec_compile: host only) it is used."""
    d, rec = tmp_path / "spec", tmp_path / "wanted.txt"
    d.mkdir()
    env = dict(os.environ, RSB_SPEC_DIR=str(d), RSB_SPEC_RECORD=str(rec))
    env.pop("RSB_SPECIALIZE", None)

    def child(mode):
        p = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT, mode=mode)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        st = [l for l in p.stdout.splitlines() if l.startswith("STATUS")][0].split()[1:]
        return [int(x) for x in st], [l for l in p.stdout.splitlines() if l.startswith("SUM")][0], p.stderr

    st, s0, _ = child("cached")
    assert st == [1, 0, 5] and not list(d.iterdir())
    lines = rec.read_text().splitlines()
    assert len(lines) == 1 and lines[0].startswith("16 8 0 4 | -DRSB_SPECIALIZED -DRSB_SPEC_NB=13 ")
    L = _capi.lib()
    name = _capi.C.create_string_buffer(256)
    assert L.rsb_spec_file_name(lines[0].encode(), name, 256) == 0
    (d / name.value.decode()).write_bytes(b"not a code object" * 100)
    st, s1, err = child("cached")
    assert st == [1, 0, 5] and "refused" in err and s1 == s0
    (d / name.value.decode()).unlink()
    # the host-only compile entry point writes into rsb_spec_dir() of THIS process: compile in a child with the same $RSB_SPEC_DIR
    p = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from raisimlib_amd import _capi; L = _capi.lib(); rc = L.rsb_spec_compile(%r.encode()); print(rc, L.rsb_last_error()); sys.exit(rc)" % (ROOT, lines[0])],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    st, s2, err = child("cached")
    assert st == [1, 5, 0] and s2 == s0, (st, err)


def test_the_manifest_covers_the_workloads_of_bench_py(built_lib, tmp_path):
    """build() compiles one code object per line of raisimlib_amd/spec_manifest.txt; the default bench line (configs 2, 3, 5, closed loop, template path)
    must not meet a key outside it: with prebuilt objects only, nothing is recorded as missing and no launch of the line's world runs an ahead-of-time class."""
    import json
    rec = tmp_path / "missing.txt"
    env = dict(os.environ, RSB_SPEC_RECORD=str(rec))
    env.pop("RSB_SPECIALIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--specialization", "cached", "--steps", "4", "--warmup", "2", "--no-cpu", "--repeats", "3"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    b = json.loads([l for l in p.stdout.strip().splitlines() if l.startswith("{")][-1])
    missing = rec.read_text().splitlines() if rec.exists() else []
    assert not missing, "keys outside raisimlib_amd/spec_manifest.txt:\n" + "\n".join(missing)
    assert b["specialization"]["mode"] == "cached" and b["specialization"]["step_launches_generic"] == 0
