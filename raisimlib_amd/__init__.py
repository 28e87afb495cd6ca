"""raisimlib_amd — MI355X-native batched rigid-body simulator behind a RaiSim-shaped API.

The product is `lib/librsb.so` (hand-written HIP kernels for gfx950 + C-ABI, include/rsb.h) and the
C++ facade headers under include/raisim/.  This Python package is the host-side mirror used by the
tests, the bench and the vectorised-env wrapper; it only moves pointers around.
"""
from ._capi import (RSB_DEVICE, RSB_FORCE_AND_TORQUE, RSB_HOST, RSB_MAX_CONTACTS,  # noqa: F401
                    RSB_PD_PLUS_FEEDFORWARD_TORQUE, RsbError)
from .world import CONTACT_DTYPE, BatchedWorld, Model, rsc_path  # noqa: F401
from .vecenv import VecEnv  # noqa: F401
