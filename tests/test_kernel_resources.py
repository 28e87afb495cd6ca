"""Static resources of the benchmark's step-kernel instance (hipcc cross-compiles without a GPU): no scratch (a spill to private
memory costs HBM traffic and latency on every launch - it once hid in the prologue's table copy), one wave per SIMD by design."""
import os
import re
import shutil
import subprocess

import pytest

from common import ROOT
from raisimlib_amd import build as rb


@pytest.mark.parametrize("lpe,kmax,ml,cl", [(16, 8, 4, 0), (32, 16, 12, 0), (64, 16, 12, 0), (16, 8, 4, 4), (16, 8, 4, 16), (32, 16, 12, 16)])   # the benchmark's ANYmal-like and Atlas-like (two envs / one env per wave) instances; the second-flank class; the benchmark's pipelined twins
def test_step_instances_use_no_scratch(tmp_path, lpe, kmax, ml, cl):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "k.s"
    csrc = os.path.join(ROOT, "raisimlib_amd", "csrc")
    cmd = [hipcc, *rb.FLAGS, "-I", os.path.join(ROOT, "include"), "-I", csrc, f"-DRSB_I_LPE={lpe}", f"-DRSB_I_KMAX={kmax}", f"-DRSB_I_CL={cl}", f"-DRSB_I_ML={ml}",
           "-DRSB_I_PROF=0", "--cuda-device-only", "-S", "-o", str(out), os.path.join(csrc, "step_instance.hip")]
    subprocess.run(cmd, check=True, capture_output=True)
    txt = out.read_text()
    scratch = int(re.search(r"\.private_segment_fixed_size:\s*(\d+)", txt).group(1))
    vgpr = int(re.search(r"\.vgpr_count:\s*(\d+)", txt).group(1))
    spills = int(re.search(r"\.vgpr_spill_count:\s*(\d+)", txt).group(1))
    assert scratch == 0 and spills == 0, (scratch, spills)
    assert "scratch_store" not in txt and "scratch_load" not in txt
    assert vgpr <= 512                       # arch VGPRs + AGPRs of the one wave a SIMD holds
    assert not re.search(r"\bv_mfma", txt)   # documented in DESIGN.md section 4: no MFMA on this path, by measurement
    # round 3's instruction-level steps stay in the ISA: packed fp32 in the contact exchange (2 sites x the register-held coupling blocks x 3 columns: five
    # blocks in the quadruped classes since round 6 - step_phase_solver.inc, NPK -, twelve in the large ones) and in the up pass's sums, and the sweep loop's
    # fetch-window pin
    assert len(re.findall(r"\bv_pk_fma_f32\b", txt)) >= (30 if kmax == 8 else 48) and len(re.findall(r"\bv_pk_add_f32\b", txt)) >= 28
    assert re.search(r"\.p2align\s+5", txt)
    # pipelined control steps live in the class twins (| 16) alone: the plain instances carry neither the hand-over's cache operations nor its
    # spin ("the plain instances carry none of this"); the twins have the XCD-affine hand-over AND the agent-scope fallback
    pipe_ops = [len(re.findall(p, txt)) for p in (r"\bbuffer_wbl2\b", r"\bbuffer_inv\b", r"\bs_sleep\b", r"HW_REG_XCC_ID")]
    if cl & 16:
        assert all(n >= 1 for n in pipe_ops), pipe_ops
    else:
        assert pipe_ops == [0, 0, 0, 0], pipe_ops
