"""In-tree build of librsb.so (HIP kernels + C-ABI host code) for gfx950 with hipcc.

`python -m raisimlib_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a GPU.
The .so stays in-tree (raisimlib_amd/lib/) so it travels to the GPU box with the repo snapshot.

The fused step kernel has ten (LPE, KMAX, CL, ML) classes x {production, profiling}; each instance is its own object
(step_instance.hip + five -D macros, list in step_launch.h) and the objects are compiled in parallel, so a clean build
takes about a minute on 8 cores instead of several in one translation unit.  Objects are cached under
raisimlib_amd/lib/obj/ and rebuilt when a source they depend on is newer.
"""
import hashlib
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "librsb.so")
OBJ = os.path.join(HERE, "lib", "obj")
RSB_H = os.path.join(ROOT, "include", "rsb.h")
RSB_TYPES_H = os.path.join(ROOT, "include", "rsb_types.h")    # the part of the ABI the kernels compile against
RSB_EXT_H = os.path.join(ROOT, "include", "rsb_ext.h")         # the entry points without an upstream counterpart (solver heuristics, scheduling, timing, debug aids); included by rsb.h
RSB_PIPELINE_H = os.path.join(ROOT, "include", "rsb_pipeline.h")   # the closed-loop pipeline: C declarations + the device-side serve loop of an action stage
_WORLD_DEPS = ["rsb_world.h", "rsb_internal.h", "rsb_spec.h", "step_types.h", "step_spec.h", RSB_H, RSB_EXT_H, RSB_TYPES_H, RSB_PIPELINE_H]
HOST_SOURCES = {   # source -> headers it depends on
    "urdf_model.cpp": ["rsb_internal.h", RSB_H, RSB_EXT_H, RSB_TYPES_H],
    "terrain_io.cpp": ["rsb_internal.h", RSB_H, RSB_EXT_H, RSB_TYPES_H],
    "rsb_world.hip": _WORLD_DEPS + ["step_launch.h", "query_kernel.h", "env_task.h"],
    "rsb_pipeline.hip": _WORLD_DEPS + ["stage_bodies.h"],
    "rsb_comm.hip": _WORLD_DEPS,
    "rsb_rk4.hip": _WORLD_DEPS,
    "rsb_spec.hip": _WORLD_DEPS,       # specialised code objects of the step kernel: key, cache directory, compile, load, launch
}
# the fused step kernel: the template's skeleton (step_kernel.h), its device helpers (step_math / step_terrain / step_slip .h) and its body, one
# fragment per phase (step_phase_*.inc, included inside the kernel: same token stream as the one 2 500-line function of rounds 1-4)
KERNEL_DEPS = ["step_instance.hip", "step_kernel.h", "step_types.h", "step_spec.h", "step_launch.h", "env_task.h", "step_math.h", "step_terrain.h", "step_slip.h", "stage_bodies.h", RSB_PIPELINE_H,
               *sorted(f for f in os.listdir(CSRC) if f.startswith("step_phase_") and f.endswith(".inc")), RSB_TYPES_H]
# measured on the step kernel (profiles/r01_notes.md): SLP packing into v_pk_* costs more v_mov shuffles than it saves and
# pushes the kernel into scratch; IEEE-exact fp32 div/sqrt sequences are not needed at the stated parity tolerance
# (2.5 ulp hardware approximations + Newton step instead)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fno-slp-vectorize", "-fno-hip-fp32-correctly-rounded-divide-sqrt"]


def step_instances():
    """[(lpe, kmax, cl, ml)] parsed from the RSB_STEP_INSTANCES line of step_launch.h (single source of truth)."""
    txt = open(os.path.join(CSRC, "step_launch.h")).read()
    line = re.search(r"RSB_STEP_INSTANCES:(.*)", txt).group(1)
    return [tuple(int(x) for x in tok.split(",")) for tok in line.split()]


def source_hash(extra_flags=()):
    """sha256 over everything librsb.so is compiled from: the files of csrc/, include/rsb.h and the compiler flags.  build() links it
    into the library (rsb_source_hash()); tests/conftest.py compares the two, so a library built from other sources than the tree's -
    a stale object cache, a binary that travelled to the GPU box without its sources - fails the suite instead of passing it."""
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip", ".cpp", ".inc"))) + [RSB_H, RSB_EXT_H, RSB_TYPES_H, RSB_PIPELINE_H]
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        h.update(open(f, "rb").read())
    h.update(" ".join([*FLAGS, *extra_flags]).encode())
    return h.hexdigest()[:32]


def _path(p):
    return p if os.path.isabs(p) else os.path.join(CSRC, p)


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(_path(d)) > t for d in deps)


def build(force=False, verbose=True, extra_flags=(), jobs=None):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: raisimlib_amd needs the ROCm toolchain (no CPU fallback exists)")
    os.makedirs(OBJ, exist_ok=True)
    inc = ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
    tag = "_".join(f.strip("-").replace("=", "") for f in extra_flags)     # objects built with other flags do not mix
    out = OUT if not tag else OUT[:-3] + f".{tag}.so"                      # ... and link into their own library (RSB_LIB_PATH selects it)
    tasks = []   # (object path, command)
    for src, deps in HOST_SOURCES.items():
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + (f".{tag}" if tag else "") + ".o")
        if force or _newer(obj, [src] + deps):
            tasks.append((obj, [hipcc, *FLAGS, *extra_flags, "-x", "hip", *inc, "-c", _path(src), "-o", obj]))
    objs = [os.path.join(OBJ, os.path.splitext(src)[0] + (f".{tag}" if tag else "") + ".o") for src in HOST_SOURCES]
    # RSB_BUILD_ONLY="16,8,0,4 32,16,0,12": kernel experiments rebuild these instances only; the other objects are linked as they are
    # (rsb_source_hash() then no longer describes the library: the test-suite refuses it - a full build() is what ships)
    only = {tuple(int(x) for x in tok.split(",")) for tok in os.environ.get("RSB_BUILD_ONLY", "").split()}
    for lpe, kmax, cl, ml in step_instances():
        for prof in ((0,) if cl & (18 | 64) else (0, 1)):     # (the peer-exchange, the pipelined and the resident classes have no profiling twin: rsb_world.hip, launch_step)
            obj = os.path.join(OBJ, f"step_{lpe}_{kmax}_{cl}_{ml}_{prof}" + (f".{tag}" if tag else "") + ".o")
            objs.append(obj)
            if only and (lpe, kmax, cl, ml) not in only and os.path.exists(obj):
                continue
            if force or _newer(obj, KERNEL_DEPS):
                tasks.append((obj, [hipcc, *FLAGS, *extra_flags, *inc, f"-DRSB_I_LPE={lpe}", f"-DRSB_I_KMAX={kmax}",
                                    f"-DRSB_I_CL={cl}", f"-DRSB_I_ML={ml}", f"-DRSB_I_PROF={prof}", "-c",
                                    _path("step_instance.hip"), "-o", obj]))
    # build provenance: the hash of the sources this library is built from, as a translation unit of its own (rsb_source_hash())
    shash = source_hash(extra_flags)
    if only:
        shash += "-partial"     # RSB_BUILD_ONLY leaves stale kernel objects in the library: it must not pass for a build of this tree (tests/conftest.py refuses it)
    # A library that says it was built from exactly these sources and flags IS up to date, whatever the object cache looks like: the GPU box
    # receives librsb.so without lib/obj/ (.gpurunignore) and used to recompile all ~95 objects in every test session.  The sidecar file only
    # short-cuts the build; tests/conftest.py asks the loaded library itself (rsb_source_hash()).
    side = out + ".hash"
    if not force and os.path.exists(out) and os.path.exists(side) and open(side).read().strip() == shash:
        return out
    stamp_src = os.path.join(OBJ, "build_stamp" + (f".{tag}" if tag else "") + ".cpp")
    stamp_obj = stamp_src[:-4] + ".o"
    stamp_txt = f'extern "C" const char* rsb_source_hash(void) {{ return "{shash}"; }}\n'
    if not os.path.exists(stamp_src) or open(stamp_src).read() != stamp_txt or not os.path.exists(stamp_obj):
        open(stamp_src, "w").write(stamp_txt)
        tasks.append((stamp_obj, ["g++", "-O1", "-fPIC", "-c", stamp_src, "-o", stamp_obj]))
    objs.append(stamp_obj)
    if not tasks and not only and os.path.exists(out) and all(os.path.getmtime(o) <= os.path.getmtime(out) for o in objs):
        open(side, "w").write(shash + "\n")     # (the library was linked from these objects, the stamp among them)
        return out

    def run(task):
        obj, cmd = task
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {os.path.basename(obj)}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip() and verbose:
            print(r.stderr, file=sys.stderr)
        return obj

    jobs = jobs or int(os.environ.get("RSB_BUILD_JOBS", "0")) or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        list(pool.map(run, tasks))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, "-lz"]   # zlib: PNG height maps
    if verbose:
        print(" ".join(link[:6]), f"... ({len(objs)} objects) -lz", file=sys.stderr)
    subprocess.run(link, check=True)
    if only:
        if os.path.exists(side):
            os.remove(side)      # (no short-cut for the next build(): it has to look at the objects)
    else:
        open(side, "w").write(shash + "\n")
    return out


SPEC_MANIFEST = os.path.join(HERE, "spec_manifest.txt")


def build_specializations(verbose=True, jobs=None):
    """Compile the specialised code objects of the shipped workloads (csrc/step_spec.h): one per line of spec_manifest.txt - "<lpe> <kmax> <cl> <ml> | -D..." as
    a miss appends it to $RSB_SPEC_RECORD - into lib/spec/ through the library's own host-only entry point (rsb_spec_compile: hipcc --genco, ~3 s each, no GPU
    needed).  File names carry the library's source hash, so objects of older sources are never loaded; they are removed here.  A world whose key is not in the
    manifest runs its ahead-of-time class (or compiles on demand: rsb_set_specialization(RSB_SPEC_COMPILE))."""
    import ctypes
    from . import _capi
    lib = _capi.lib()
    lines = [l.strip() for l in open(SPEC_MANIFEST) if l.strip() and not l.startswith("#")] if os.path.exists(SPEC_MANIFEST) else []
    spec_dir = lib.rsb_spec_dir().decode()
    os.makedirs(spec_dir, exist_ok=True)
    want = set()
    for l in lines:
        buf = ctypes.create_string_buffer(256)
        if lib.rsb_spec_file_name(l.encode(), buf, 256) != 0:
            raise RuntimeError(f"spec_manifest.txt: bad line: {l}")
        want.add(buf.value.decode())
    for f in os.listdir(spec_dir):
        if f not in want:
            os.remove(os.path.join(spec_dir, f))
    todo = [l for l in lines]

    def one(l):
        rc = lib.rsb_spec_compile(l.encode())
        if rc != 0:
            raise RuntimeError(f"rsb_spec_compile failed ({rc}) for: {l}\n{(lib.rsb_last_error() or b'').decode()}")
    jobs = jobs or int(os.environ.get("RSB_BUILD_JOBS", "0")) or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        list(pool.map(one, todo))
    if verbose:
        print(f"specialised code objects: {len(want)} in {spec_dir}", file=sys.stderr)
    return sorted(want)


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a.startswith("-") and a not in ("--force", "--spec")]
    build(force="--force" in sys.argv, extra_flags=flags)
    if not flags:
        build_specializations()
