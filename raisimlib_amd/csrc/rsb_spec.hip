// rsb_spec.hip — specialised code objects of the step kernel (step_spec.h): key, cache directory, compilation, loading, launch.
//
// A launch of kernel class (LPE, KMAX, CL, ML) with StepArgs `a` has the key  "<lpe> <kmax> <cl> <ml> | -DRSB_SPEC_NB=13 -DRSB_SPEC_NQ=19 ..."
// (RSB_SPEC_FIELDS evaluated over `a`).  Its code object is  <dir>/step_<lpe>_<kmax>_<cl>_<ml>_<fnv1a64(source hash, key)>.hsaco  where <dir> is
// $RSB_SPEC_DIR or spec/ next to librsb.so, and the source hash is this library's (rsb_source_hash()): a code object never outlives the sources it
// was compiled from.  Modes (rsb_set_specialization):
//   RSB_SPEC_OFF      the ahead-of-time classes only
//   RSB_SPEC_CACHED   (default) a code object found in the directory is loaded and used; a miss runs the ahead-of-time class and, when
//                     $RSB_SPEC_RECORD names a file, appends the key to it (how raisimlib_amd/spec_manifest.txt was made: build() compiles its lines)
//   RSB_SPEC_COMPILE  a miss compiles the code object first (hipcc --genco of step_instance.hip, ~25 s, once per key and source hash)
// A code object that does not load, lacks the kernel symbol or was compiled against another StepArgs layout (rsb_spec_abi) is refused with a message
// on stderr once, and the ahead-of-time class runs: specialisation changes speed, never results (tests/test_gpu_spec.py: bit-identical).
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <array>
#include <cstddef>
#include <cstdio>
#include <cctype>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "rsb_world.h"
#include "rsb_spec.h"
#include "step_spec.h"

namespace {

std::mutex g_mu;
struct Loaded { hipModule_t mod = nullptr; hipFunction_t fn = nullptr; bool tried = false; };
std::map<std::string, Loaded> g_loaded;      // "<device>:<file name>" -> module (one per process and device)

uint64_t fnv1a64(const std::string& s) {
  uint64_t h = 1469598103934665603ull;
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  return h;
}

std::string lib_dir() {
  Dl_info info;
  if (dladdr(reinterpret_cast<const void*>(&rsb_spec_dir), &info) && info.dli_fname) {
    char real[4096];
    std::string p = ::realpath(info.dli_fname, real) ? real : info.dli_fname;
    const size_t k = p.rfind('/');
    return k == std::string::npos ? std::string(".") : p.substr(0, k);
  }
  return ".";
}

std::string env_or(const char* name, const std::string& dflt) {
  const char* v = std::getenv(name);
  return (v && *v) ? std::string(v) : dflt;
}

bool file_exists(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0 && st.st_size > 0; }

// "-DRSB_SPEC_NB=13 ..." of a launch
std::string defs_of(const StepArgs& a) {
  std::string d = "-DRSB_SPECIALIZED";
#define RSB_SPEC_DEF(NAME, expr) d += std::string(" -DRSB_SPEC_" #NAME "=") + std::to_string((int)(expr));
  RSB_SPEC_FIELDS(RSB_SPEC_DEF)
#undef RSB_SPEC_DEF
  // kernel experiments (tools/exp): extra -DRSB_X_... flags become part of the key, so an A/B of kernel variants is two runs on one box, each compiling its own code object
  static const std::string extra = [] {
    std::string e = env_or("RSB_SPEC_EXTRA_DEFS", "");
    for (char ch : e) if (!(std::isalnum((unsigned char)ch) || ch == '_' || ch == '=' || ch == '-' || ch == ' ')) { std::fprintf(stderr, "librsb: RSB_SPEC_EXTRA_DEFS ignored (only -DNAME[=value] tokens)\n"); return std::string(); }
    return e;
  }();
  if (!extra.empty()) d += " " + extra;
  return d;
}

// compiler-flag experiments (tools/exp/ab_defs.sh): $RSB_SPEC_EXTRA_FLAGS is appended to the hipcc command line of a code object and is part of its name
const std::string& extra_flags() {
  static const std::string flags = [] {
    std::string e = env_or("RSB_SPEC_EXTRA_FLAGS", "");
    for (char ch : e) if (!(std::isalnum((unsigned char)ch) || ch == '_' || ch == '=' || ch == '-' || ch == ' ' || ch == '.' || ch == ',')) { std::fprintf(stderr, "librsb: RSB_SPEC_EXTRA_FLAGS ignored (unexpected character)\n"); return std::string(); }
    return e;
  }();
  return flags;
}

std::string key_line(const rsbw::SpecClass& c, const std::string& defs) {
  return std::to_string(c.lpe) + " " + std::to_string(c.kmax) + " " + std::to_string(c.cl) + " " + std::to_string(c.ml) + (c.prof ? " p" : "") + " | " + defs;
}

std::string file_of(const rsbw::SpecClass& c, const std::string& defs) {
  char buf[160];
  std::snprintf(buf, sizeof buf, "step_%d_%d_%d_%d%s_%016llx.hsaco", c.lpe, c.kmax, c.cl, c.ml, c.prof ? "p" : "",
                (unsigned long long)fnv1a64(std::string(rsb_source_hash()) + " " + key_line(c, defs) + (extra_flags().empty() ? "" : " ## " + extra_flags())));
  return buf;
}

// Itanium mangling of rsbk::rsb_step_kernel<LPE, KMAX, CL, ML, false>(rsbk::StepArgs)
std::string symbol_of(const rsbw::SpecClass& c) {
  char buf[160];
  std::snprintf(buf, sizeof buf, "_ZN4rsbk15rsb_step_kernelILi%dELi%dELi%dELi%dELb%dEEEvNS_8StepArgsE", c.lpe, c.kmax, c.cl, c.ml, c.prof ? 1 : 0);
  return buf;
}

bool parse_line(const char* line, rsbw::SpecClass& c, std::string& defs) {
  int n = 0;
  c.prof = 0;
  if (!line || std::sscanf(line, "%d %d %d %d %n", &c.lpe, &c.kmax, &c.cl, &c.ml, &n) != 4 || n == 0) return false;
  if (line[n] == 'p') { c.prof = 1; ++n; while (line[n] == ' ') ++n; }
  if (line[n] != '|') return false;
  ++n; while (line[n] == ' ') ++n;
  defs = line + n;
  while (!defs.empty() && (defs.back() == '\n' || defs.back() == '\r' || defs.back() == ' ')) defs.pop_back();
  // the flags go onto a compiler command line: nothing but -DRSB_SPEC... tokens of [A-Z_0-9=] (and a leading -DRSB_SPECIALIZED)
  if (defs.rfind("-DRSB_SPECIALIZED", 0) != 0) return false;
  for (char ch : defs) if (!(std::isalnum((unsigned char)ch) || ch == '_' || ch == '=' || ch == '-' || ch == ' ')) return false;
  return true;
}

int compile(const rsbw::SpecClass& c, const std::string& defs) {
  const std::string dir = rsb_spec_dir(), out = dir + "/" + file_of(c, defs);
  if (file_exists(out)) return RSB_OK;
  ::mkdir(dir.c_str(), 0777);
  const std::string lib = lib_dir();
  const std::string src = env_or("RSB_SRC_DIR", lib + "/../csrc"), inc = env_or("RSB_INCLUDE_DIR", lib + "/../../include");
  for (const std::string* pth : {&dir, &src, &inc})      // (the paths go onto a shell command line, single-quoted)
    if (pth->find('\'') != std::string::npos) { rsb::set_error("rsb specialisation: a path with a single quote in it: " + *pth); return RSB_E_INVALID; }
  if (!file_exists(src + "/step_instance.hip")) { rsb::set_error("rsb specialisation: kernel sources not found in " + src + " (RSB_SRC_DIR)"); return RSB_E_UNSUPPORTED; }
  const std::string hipcc = env_or("HIPCC", file_exists("/opt/rocm/bin/hipcc") ? "/opt/rocm/bin/hipcc" : "hipcc");
  const std::string tmp = out + ".tmp" + std::to_string((long)::getpid());
  // (the flags of raisimlib_amd/build.py FLAGS; --genco: device code object only)
  const std::string cmd = hipcc + " --genco --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-hip-fp32-correctly-rounded-divide-sqrt -I '" + inc + "' -I '" + src + "'" +
                          " -DRSB_I_LPE=" + std::to_string(c.lpe) + " -DRSB_I_KMAX=" + std::to_string(c.kmax) + " -DRSB_I_CL=" + std::to_string(c.cl) +
                          " -DRSB_I_ML=" + std::to_string(c.ml) + " -DRSB_I_PROF=" + std::to_string(c.prof ? 1 : 0) + " " + defs + " " + extra_flags() + " -o '" + tmp + "' '" + src + "/step_instance.hip' > '" + tmp + ".log' 2>&1";
  const int rc = std::system(cmd.c_str());
  if (rc != 0 || !file_exists(tmp)) {
    rsb::set_error("rsb specialisation: hipcc failed (log: " + tmp + ".log) for " + key_line(c, defs));
    ::unlink(tmp.c_str());
    return RSB_E_UNSUPPORTED;
  }
  ::unlink((tmp + ".log").c_str());
  if (::rename(tmp.c_str(), out.c_str()) != 0) { ::unlink(tmp.c_str()); if (!file_exists(out)) { rsb::set_error("rsb specialisation: cannot write " + out); return RSB_E_UNSUPPORTED; } }
  return RSB_OK;
}

void record_miss(const std::string& line) {
  static const char* path = std::getenv("RSB_SPEC_RECORD");
  if (!path || !*path) return;
  static std::map<std::string, bool> seen;
  if (seen[line]) return;
  seen[line] = true;
  if (FILE* f = std::fopen(path, "a")) { std::fprintf(f, "%s\n", line.c_str()); std::fclose(f); }
}

// loads <dir>/<file> on the current device; nullptr when it is absent or refused (said once on stderr)
hipFunction_t load(int device, const rsbw::SpecClass& c, const std::string& defs) {
  const std::string file = file_of(c, defs), id = std::to_string(device) + ":" + file;
  Loaded& L = g_loaded[id];
  if (L.tried) return L.fn;
  const std::string path = std::string(rsb_spec_dir()) + "/" + file;
  if (!file_exists(path)) return nullptr;          // (not `tried`: a later compile may put it there)
  L.tried = true;
  auto refuse = [&](const std::string& why) { std::fprintf(stderr, "librsb: specialised code object %s refused (%s): the ahead-of-time kernel class runs instead\n", path.c_str(), why.c_str()); if (L.mod) { (void)hipModuleUnload(L.mod); L.mod = nullptr; } L.fn = nullptr; (void)hipGetLastError();   /* (the failed call's error must not surface at the next launch's hipGetLastError) */ return (hipFunction_t) nullptr; };
  hipError_t e = hipModuleLoad(&L.mod, path.c_str());
  if (e != hipSuccess) { L.mod = nullptr; return refuse(std::string("hipModuleLoad: ") + hipGetErrorString(e)); }
  hipDeviceptr_t abi_ptr = nullptr; size_t abi_bytes = 0;
  unsigned abi[4] = {0, 0, 0, 0};
  e = hipModuleGetGlobal(&abi_ptr, &abi_bytes, L.mod, "rsb_spec_abi");
  if (e != hipSuccess || abi_bytes != sizeof abi) return refuse("no rsb_spec_abi record");
  e = hipMemcpyDtoH(abi, abi_ptr, sizeof abi);
  if (e != hipSuccess) return refuse(std::string("reading rsb_spec_abi: ") + hipGetErrorString(e));
  const unsigned want[4] = {(unsigned)sizeof(StepArgs), (unsigned)sizeof(LdsLayout), (unsigned)offsetof(StepArgs, L), (unsigned)rsbk::kSpecFields};
  if (std::memcmp(abi, want, sizeof abi) != 0) return refuse("compiled against another StepArgs layout");
  hipFunction_t fn = nullptr;
  e = hipModuleGetFunction(&fn, L.mod, symbol_of(c).c_str());
  if (e != hipSuccess || !fn) return refuse("kernel symbol " + symbol_of(c) + " not found");
  L.fn = fn;
  return fn;
}

}  // namespace

namespace rsbw {

int spec_default_mode() {
  const char* v = std::getenv("RSB_SPECIALIZE");
  if (!v || !*v) return RSB_SPEC_CACHED;
  if (!std::strcmp(v, "0") || !std::strcmp(v, "off")) return RSB_SPEC_OFF;
  if (!std::strcmp(v, "compile")) return RSB_SPEC_COMPILE;
  return RSB_SPEC_CACHED;
}

hipFunction_t spec_find(rsb_world* w, const SpecClass& c, const StepArgs& a) {
  if (w->spec_mode == RSB_SPEC_OFF) return nullptr;
  // the world's own memo: class + field values -> function (or nullptr), looked up at every launch
  std::array<int, 4 + rsbk::kSpecFields> k{};
  int n = 0;
  k[n++] = c.lpe; k[n++] = c.kmax; k[n++] = c.cl | (c.prof ? (1 << 20) : 0); k[n++] = c.ml;
#define RSB_SPEC_VAL(NAME, expr) k[n++] = (int)(expr);
  RSB_SPEC_FIELDS(RSB_SPEC_VAL)
#undef RSB_SPEC_VAL
  std::vector<int> kv(k.begin(), k.end());
  kv.push_back(w->spec_mode);
  auto it = w->spec_memo.find(kv);
  if (it != w->spec_memo.end()) return static_cast<hipFunction_t>(it->second);
  std::lock_guard<std::mutex> lock(g_mu);
  const std::string defs = defs_of(a);
  hipFunction_t fn = load(w->device, c, defs);
  if (!fn && w->spec_mode == RSB_SPEC_COMPILE) {
    if (compile(c, defs) == RSB_OK) fn = load(w->device, c, defs);
    else std::fprintf(stderr, "librsb: %s\n", rsb_last_error());
  }
  if (!fn) record_miss(key_line(c, defs));
  w->spec_memo[kv] = fn;
  return fn;
}

int spec_launch(hipFunction_t fn, const StepArgs& a, int blocks, size_t lds_bytes, hipStream_t stream) {
  StepArgs args = a;
  size_t size = sizeof args;
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  HIP_TRY(hipModuleLaunchKernel(fn, (unsigned)blocks, 1, 1, 64, 1, 1, (unsigned)lds_bytes, stream, nullptr, extra));
  return RSB_OK;
}

}  // namespace rsbw

extern "C" {

const char* rsb_spec_dir(void) {
  static const std::string dir = env_or("RSB_SPEC_DIR", lib_dir() + "/spec");
  return dir.c_str();
}

int rsb_spec_compile(const char* manifest_line) {
  rsbw::SpecClass c; std::string defs;
  if (!parse_line(manifest_line, c, defs)) { rsb::set_error("rsb_spec_compile: expected '<lpe> <kmax> <cl> <ml> | -DRSB_SPECIALIZED -DRSB_SPEC_...=...'"); return RSB_E_INVALID; }
  return compile(c, defs);
}

int rsb_spec_file_name(const char* manifest_line, char* out, int capacity) {
  rsbw::SpecClass c; std::string defs;
  if (!parse_line(manifest_line, c, defs) || !out || capacity <= 0) { rsb::set_error("rsb_spec_file_name: bad manifest line or buffer"); return RSB_E_INVALID; }
  std::snprintf(out, (size_t)capacity, "%s", file_of(c, defs).c_str());
  return RSB_OK;
}

int rsb_set_specialization(rsb_world* w, int mode) {
  if (!w || mode < RSB_SPEC_OFF || mode > RSB_SPEC_COMPILE) { rsb::set_error("rsb_set_specialization: mode is RSB_SPEC_OFF, RSB_SPEC_CACHED or RSB_SPEC_COMPILE"); return RSB_E_INVALID; }
  w->spec_mode = mode;
  return RSB_OK;
}

int rsb_specialization_status(const rsb_world* w, long long* specialized_launches, long long* generic_launches) {
  if (!w) { rsb::set_error("null world"); return RSB_E_INVALID; }
  if (specialized_launches) *specialized_launches = w->spec_launches;
  if (generic_launches) *generic_launches = w->generic_launches;
  return w->spec_mode;
}

}  // extern "C"
