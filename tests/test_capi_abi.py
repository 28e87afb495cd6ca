"""The C-ABI library loads on a CPU-only box and exports every symbol include/rsb.h, include/rsb_ext.h and include/rsb_pipeline.h declare."""
import ctypes as C
import os
import re

import pytest

from common import ROOT


def header_functions():
    """every function the C-ABI declares: include/rsb.h and the closed-loop pipeline's include/rsb_pipeline.h (its C part: the device-side
    serve loop behind `#if defined(__HIPCC__)` is a header-only template of the caller's kernel, not a symbol of the library)"""
    names = set()
    for h in ("rsb.h", "rsb_ext.h", "rsb_pipeline.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = src.split("#if defined(__HIPCC__)")[0]
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(rsb_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_header_declares_the_boundary():
    names = header_functions()
    for must in ["rsb_create", "rsb_destroy", "rsb_integrate", "rsb_integrate1", "rsb_integrate2", "rsb_set_state",
                 "rsb_get_state", "rsb_set_pd_gains", "rsb_set_pd_target", "rsb_set_generalized_force", "rsb_set_ground",
                 "rsb_set_heightmap", "rsb_get_contacts", "rsb_gather_obs", "rsb_model_from_urdf_file"]:
        assert must in names


def test_library_exports_every_declared_symbol(built_lib):
    for name in header_functions():
        assert hasattr(built_lib, name), f"librsb.so does not export {name}"


def test_python_prototypes_match_header(built_lib):
    from raisimlib_amd import _capi
    assert sorted(_capi.PROTOTYPES) == header_functions()


def test_struct_layouts_match(built_lib):
    """ModelBlob / Contact ctypes mirrors have the C sizes (checked through a round trip)."""
    from raisimlib_amd import Model, rsc_path, _capi
    m = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
    m2 = Model(blob=m.blob)      # C side validates nb/nq/nv/parent/level consistency of what it received
    assert m2.nb == m.nb == 13 and m2.nv == 18 and m2.nq == 19
    assert C.sizeof(_capi.Contact) == 48


def test_policy_struct_mirrors_have_the_c_layout(tmp_path):
    """rsb_linear_policy / rsb_mlp_policy are passed by pointer from Python: the ctypes mirrors must have the C compiler's sizes and field offsets."""
    import subprocess
    from raisimlib_amd import _capi
    src = tmp_path / "sizes.c"
    src.write_text(r'''#include <stdio.h>
#include <stddef.h>
#include "rsb.h"
int main(void) {
  printf("%zu %zu %zu %zu\n", sizeof(rsb_linear_policy), offsetof(rsb_linear_policy, clip), offsetof(rsb_linear_policy, rollout_done), sizeof(rsb_contact));
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(rsb_mlp_policy), offsetof(rsb_mlp_policy, Wt), offsetof(rsb_mlp_policy, activation), offsetof(rsb_mlp_policy, ob_mean),
         offsetof(rsb_mlp_policy, noise_period), offsetof(rsb_mlp_policy, rollout_done));
  return 0;
}
''')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    a, b = (list(map(int, l.split())) for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines())
    L, M = _capi.LinearPolicy, _capi.MlpPolicy
    assert a == [C.sizeof(L), L.clip.offset, L.rollout_done.offset, C.sizeof(_capi.Contact)]
    assert b == [C.sizeof(M), M.Wt.offset, M.activation.offset, M.ob_mean.offset, M.noise_period.offset, M.rollout_done.offset]


def test_no_cpu_fallback(built_lib):
    """Without a GPU rsb_create must fail loudly (RSB_E_NO_DEVICE), never fall back to a CPU path."""
    from raisimlib_amd import BatchedWorld, Model, RsbError, rsc_path
    if built_lib.rsb_device_count() > 0:
        pytest.skip("a GPU is visible")
    m = Model(urdf_path=rsc_path("anymal_c_like.urdf"))
    with pytest.raises(RsbError, match="no HIP device"):
        BatchedWorld(m, 4)


def test_product_never_imports_oracle():
    """The product package must not include, import, link or dlopen the oracle (test infrastructure)."""
    bad = re.compile(r'#\s*include\s*[<"][^>"]*oracle|\bimport\s+oracle|\bfrom\s+oracle|pyoracle|librsb_oracle')
    shared_lib = re.compile(r'"([^"]*\.so[.\d]*)"')
    for top in ("raisimlib_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                    txt = open(os.path.join(dirpath, f)).read()
                    assert not bad.search(txt), f"{f} references the oracle"
                    if "dlopen" in txt or "CDLL" in txt:    # the only libraries the product loads at run time: itself and RCCL
                        for name in shared_lib.findall(txt):
                            assert os.path.basename(name).startswith(("librccl.so", "librsb.so")), f"{f} loads {name}"
