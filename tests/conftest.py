import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def built_lib():
    """librsb.so built in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    from raisimlib_amd import build
    build.build(verbose=False)
    build.build_specializations(verbose=False)      # (the manifest's code objects: nothing to do where build() has run; an object of other sources is replaced)
    from raisimlib_amd import _capi
    lib = _capi.lib()
    # provenance: the library under test must have been built from THIS tree's sources (a stale object cache or a binary shipped
    # without rebuilding would otherwise pass every test with yesterday's kernels)
    have, want = lib.rsb_source_hash().decode(), build.source_hash()
    if have != want and not os.environ.get("RSB_LIB_PATH"):
        raise RuntimeError(f"librsb.so was built from other sources than this tree's (library {have}, tree {want}): rebuild with python -m raisimlib_amd.build --force")
    return lib


@pytest.fixture(scope="session")
def anymal(built_lib):
    from raisimlib_amd import Model, rsc_path
    return Model(urdf_path=rsc_path("anymal_c_like.urdf"))


@pytest.fixture(scope="session")
def atlas(built_lib):
    from raisimlib_amd import Model, rsc_path
    return Model(urdf_path=rsc_path("atlas_like.urdf"))
