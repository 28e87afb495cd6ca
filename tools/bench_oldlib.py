"""A/B helper: run bench.py against an OLDER librsb (RSB_LIB_PATH) that lacks entry points added since - the missing setters become no-ops
(only meaningful for configurations that do not depend on them: configs 2 and 3)."""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import raisimlib_amd._capi as capi
L = C.CDLL(capi.LIB_PATH, mode=C.RTLD_GLOBAL)
for name in list(capi.PROTOTYPES):
    if not hasattr(L, name):
        del capi.PROTOTYPES[name]
        sys.stderr.write(f"bench_oldlib: {name} missing in {capi.LIB_PATH}: no-op\n")
_lib = capi.lib
class _Shim:
    def __init__(self, l): self._l = l
    def __getattr__(self, n):
        try: return getattr(self._l, n)
        except AttributeError: return lambda *a: 0
import raisimlib_amd.world as W
_orig = W.BatchedWorld.__init__
def _init(self, *a, **k):
    _orig(self, *a, **k); self.L = _Shim(self.L)
W.BatchedWorld.__init__ = _init
sys.argv = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
