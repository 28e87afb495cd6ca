"""Static budgets of the step kernel that other parts of the design rest on, read off the compiler's own resource report (hipcc -S, no GPU):
the pipelined classes (| 16) share their SIMD with a wave of an action stage - include/rsb_pipeline.h tells a caller's stage kernel to stay under 96
registers, the in-repo stages are built to that - so they may not take more than 416 of the SIMD's 512.  (Round 6 broke this once without any test
noticing: the pipelined closed loop with the actor network fell from 172 M to 120 M env-steps/s, profiles/r06_spec_log.txt #7.)"""
import os
import re
import subprocess

import pytest

from raisimlib_amd import build as _b

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _resources(lpe, kmax, cl, ml, defs=(), tmp="/tmp"):
    out = os.path.join(tmp, f"budget_{lpe}_{kmax}_{cl}_{ml}_{abs(hash(tuple(defs))) % 10 ** 8}.s")
    cmd = ["/opt/rocm/bin/hipcc", *[f for f in _b.FLAGS if f not in ("-fPIC", "-Wall", "-Wno-unused-function")], "-I", os.path.join(ROOT, "include"), "-I", _b.CSRC,
           f"-DRSB_I_LPE={lpe}", f"-DRSB_I_KMAX={kmax}", f"-DRSB_I_CL={cl}", f"-DRSB_I_ML={ml}", "-DRSB_I_PROF=0", *defs, "--cuda-device-only", "-S", "-o", out,
           os.path.join(_b.CSRC, "step_instance.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    txt = open(out).read()
    os.remove(out)
    return {k: int(re.search(rf"^\s+\.{k}:\s+(\d+)", txt, re.M).group(1)) for k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")}


def _manifest_defs(cls, must=""):
    for line in open(_b.SPEC_MANIFEST):
        if line.startswith(cls + " |") and must in line:
            return line.split("|", 1)[1].split()
    raise AssertionError(f"no manifest line for class {cls}")


@pytest.mark.parametrize("cl", [16, 20, 24, 48])
def test_pipelined_quadruped_classes_leave_96_registers_to_a_stage_wave(cl):
    r = _resources(16, 8, cl, 4)
    assert r["vgpr_count"] <= 416 and r["private_segment_fixed_size"] == 0, r


def test_specialised_pipelined_classes_of_the_benchmark_keep_the_budget_too():
    for must in ("TERRAIN=0", "TERRAIN=1"):
        r = _resources(16, 8, 16, 4, _manifest_defs("16 8 16 4", must))
        assert r["vgpr_count"] <= 416 and r["private_segment_fixed_size"] == 0, (must, r)


def test_benchmark_classes_do_not_spill_to_scratch():
    """the quadruped's plain and resident classes, ahead of time and specialised: registers only (the humanoid's resident class is known to spill: DESIGN.md)"""
    for cl in (0, 64, 192, 320):
        assert _resources(16, 8, cl, 4)["private_segment_fixed_size"] == 0, cl
        assert _resources(16, 8, cl, 4, _manifest_defs(f"16 8 {cl} 4", "TERRAIN=0"))["private_segment_fixed_size"] == 0, cl


def test_lds_layouts_of_the_benchmark_models_keep_their_workgroups_per_cu():
    """A CU holds min(4, 160 KiB / workgroup LDS) single-wave workgroups.  Both benchmark models run FOUR (one wave per SIMD; the Atlas-like one at 32 lanes per
    env, two envs per workgroup): 4 x 40 032 B and 4 x 40 624 B of 163 840 - within 1.3 KB of the boundary, and both have been pushed over it once (an 84-float per-env
    table in round 6's first session: config 2 halved; the model table's conflict-free pitch in the second: config 5 fell from 38.9 M to 26.4 M env-steps/s,
    gpurun r06d - make_layout now takes that pitch only where it costs no workgroup)."""
    import bench
    from raisimlib_amd import Model, rsc_path
    anymal = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
    n = anymal.lds_bytes(kmax=8, self_collision=True, lanes_per_env=16)
    assert 4 * n <= 160 * 1024, n
    atlas = bench.Recipe(5, -1.0).model
    n = atlas.lds_bytes(kmax=16, self_collision=True, lanes_per_env=32)
    assert 4 * n <= 160 * 1024, n
    assert atlas.lds_bytes(kmax=16, self_collision=True, lanes_per_env=0) == n      # (and 32 lanes per env is what the library picks for it)
