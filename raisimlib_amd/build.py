"""In-tree build of librsb.so (HIP kernels + C-ABI host code) for gfx950 with hipcc.

`python -m raisimlib_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a GPU.
The .so stays in-tree (raisimlib_amd/lib/) so it travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "librsb.so")
SOURCES = ["urdf_model.cpp", "terrain_io.cpp", "rsb_world.hip"]
HEADERS = ["rsb_internal.h", "step_kernel.h", "query_kernel.h", os.path.join(ROOT, "include", "rsb.h")]


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, extra_flags=()):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: raisimlib_amd needs the ROCm toolchain (no CPU fallback exists)")
    if not force and not _stale():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wall", "-Wno-unused-function",
           # measured on the step kernel (profiles/r01_notes.md): SLP packing into v_pk_* costs more v_mov shuffles
           # than it saves and pushes the kernel into scratch; IEEE-exact fp32 div/sqrt sequences are not needed
           # at the stated parity tolerance (2.5 ulp hardware approximations + Newton step instead)
           "-fno-slp-vectorize", "-fno-hip-fp32-correctly-rounded-divide-sqrt",
           *extra_flags, "-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES] + ["-lz"]   # zlib: PNG height maps
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, extra_flags=[a for a in sys.argv[1:] if a.startswith("-") and a != "--force"])
