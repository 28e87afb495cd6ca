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
    from raisimlib_amd import _capi
    return _capi.lib()


@pytest.fixture(scope="session")
def anymal(built_lib):
    from raisimlib_amd import Model, rsc_path
    return Model(urdf_path=rsc_path("anymal_c_like.urdf"))


@pytest.fixture(scope="session")
def atlas(built_lib):
    from raisimlib_amd import Model, rsc_path
    return Model(urdf_path=rsc_path("atlas_like.urdf"))
