// rsb_pipeline.hip — pipelined control steps (rsb_set_step_pipelining, rsb.h) and the closed-loop run with an action stage in the loop
// (include/rsb_pipeline.h): private streams and their probes, the gates in front of the launches, the join with its fault recovery
// (no device trap anywhere: an error word, a snapshot of the last joined state, a replay in lock-step), the in-repo linear-policy stage.
//
// Upstream counterpart: none (RaiSim steps its worlds one after the other on CPU threads; VectorizedEnvironment::step is a host loop
// [RECALL raisimGymTorch/env/VectorizedEnvironment.hpp, absent from /root/reference]).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "rsb_world.h"
#include "stage_bodies.h"

#define RSB_PRAGMA_UNROLL _Pragma("unroll")

namespace rsbw {

namespace {
// Layout of the control block d_pipe_started points to.  Every word that somebody polls or hammers with atomics sits on a 256-byte line of its
// own: the first layout packed them into 256 bytes, and the idle polls of the action stage's waves on the error word then queued - agent-scope
// accesses are served at the memory side, one line at a time - in front of the `started` and ticket atomics every step workgroup issues before it
// can pick its env block (measured: -25 % with 256 stage waves resident, profiles/r05_closed_loop_log.txt).
//   +0     u64  workgroups of pipelined step launches that have started since the block was cleared (the gates wait on it)
//   +256   u32  workgroups of action stages that have started
//   +512   i32  the pipeline's error word (RSB_PIPE_ERR_*)
//   +768   2 x i32  the stream probe's flags
//   +1024  2 x u64  wait statistics (RSB_PIPE_STATS)
//   +2048  16 x 256 B: ticket counter of the step workgroups of XCD x at +256 x (monotonic)
//   +6144  16 x 256 B: arrival counter of the action stage's waves of XCD x at +256 x (monotonic)
constexpr size_t kCtlBytes = 10240;

inline char* ctl(rsb_world* w) { return reinterpret_cast<char*>(w->d_pipe_started); }
inline uint32_t* stage_started_ptr(rsb_world* w) { return reinterpret_cast<uint32_t*>(ctl(w) + 256); }
inline int* err_ptr(rsb_world* w) { return reinterpret_cast<int*>(ctl(w) + 512); }
inline int* probe_flags_ptr(rsb_world* w) { return reinterpret_cast<int*>(ctl(w) + 768); }
inline unsigned long long* stats_ptr(rsb_world* w) { return reinterpret_cast<unsigned long long*>(ctl(w) + 1024); }
inline unsigned* step_ticket_ptr(rsb_world* w) { return reinterpret_cast<unsigned*>(ctl(w) + 2048); }
inline uint32_t* stage_ticket_ptr(rsb_world* w) { return reinterpret_cast<uint32_t*>(ctl(w) + 6144); }

constexpr int kStageGridDefault = 256;      // workgroups of the action stage (one per CU)
constexpr int kStageGridMlp = 512;          // ... of the MLP stage: a block's network takes ~25 us of one wave, so fewer blocks queue behind one wave (two waves per CU; with one on
                                            // every SIMD - 1024 - the dispatcher once stopped dealing the step workgroups round-robin over the XCDs: profiles/r05_closed_loop_log.txt)

long long timeout_ticks() {      // RSB_PIPE_TIMEOUT_MS (default 10 s), in ticks of the 100 MHz wall clock; read at every launch
  const char* e = std::getenv("RSB_PIPE_TIMEOUT_MS");
  const double ms = e && std::atof(e) > 0 ? std::atof(e) : 10000.0;
  return (long long)(ms * 1e5);
}

// Do kernels on streams a and b run CONCURRENTLY?  HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4) and two
// streams on one queue run in order: the pipeline would be correct but gain nothing (measured: 107 M instead of 160 M env-steps/s when the
// second world of a process drew an aliased pair, profiles/r04_ab_log.txt).  Probe: a kernel on a waits (<= ~2 ms) for a flag that a kernel on b sets.
__global__ void pipe_probe_wait_kernel(int* flag) {
  const long long t0 = wall_clock64();
  int seen = 0;
  while (!(seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) && wall_clock64() - t0 < 200000) __builtin_amdgcn_s_sleep(32);
  flag[1] = seen ? 1 : 2;
}
__global__ void pipe_probe_set_kernel(int* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
int streams_run_concurrently(hipStream_t a, hipStream_t b, int* d_flag, bool* yes) {
  HIP_TRY(hipMemset(d_flag, 0, 2 * sizeof(int)));
  hipLaunchKernelGGL(pipe_probe_wait_kernel, dim3(1), dim3(1), 0, a, d_flag);
  hipLaunchKernelGGL(pipe_probe_set_kernel, dim3(1), dim3(1), 0, b, d_flag);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(a));
  HIP_TRY(hipStreamSynchronize(b));
  int h[2] = {0, 0};
  HIP_TRY(hipMemcpy(h, d_flag, sizeof h, hipMemcpyDeviceToHost));
  *yes = h[1] == 1;
  return RSB_OK;
}
// One more private stream whose kernels overlap with those of every stream in `with` (up to 8 candidates).  *out stays nullptr when none of the
// candidates does; *fallback then receives a stream that is correct but runs in order with one of them.
int make_concurrent_stream(rsb_world* w, const std::vector<hipStream_t>& with, hipStream_t* out, hipStream_t* fallback, int* rejected_n) {
  *out = nullptr; *fallback = nullptr;
  std::vector<hipStream_t> rejected;
  int st = RSB_OK;
  for (int attempt = 0; attempt < 8 && st == RSB_OK && !*out; ++attempt) {
    hipStream_t c = nullptr;
    if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) break;
    bool yes = true;
    for (size_t i = 0; i < with.size() && st == RSB_OK && yes; ++i)
      st = streams_run_concurrently(with[i], c, probe_flags_ptr(w), &yes);
    if (st == RSB_OK && yes) *out = c; else rejected.push_back(c);
  }
  if (rejected_n) *rejected_n = (int)rejected.size();
  if (st == RSB_OK && !*out && !rejected.empty()) { *fallback = rejected.back(); rejected.pop_back(); }
  for (hipStream_t c : rejected) (void)hipStreamDestroy(c);      // (after the search: a destroyed stream's queue slot would be handed out again)
  return st;
}
// the two private streams of the step launches: a pair that the probe has seen overlap.  Nothing of the world changes unless both exist.
int pipe_make_streams(rsb_world* w) {
  if (w->pipe_stream[0] && w->pipe_stream[1]) return RSB_OK;
  hipStream_t s0 = nullptr, s1 = nullptr, fb = nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  const int st = make_concurrent_stream(w, {s0}, &s1, &fb, &w->pipe_probe_rejected);
  if (st == RSB_OK && !s1 && fb) { s1 = fb; fb = nullptr; w->pipe_overlap = false; }   // correct, but in order
  if (fb) (void)hipStreamDestroy(fb);
  if (st != RSB_OK || !s1) {
    (void)hipStreamDestroy(s0);
    if (s1) (void)hipStreamDestroy(s1);
    if (st == RSB_OK) rsb::set_error("rsb_set_step_pipelining: no second stream could be created");
    return st == RSB_OK ? RSB_E_HIP : st;
  }
  w->pipe_stream[0] = s0; w->pipe_stream[1] = s1;
  return RSB_OK;
}
// the action stage's stream: overlaps with both step streams
int pipe_make_stage_stream(rsb_world* w) {
  if (w->pipe_stage_stream) return RSB_OK;
  hipStream_t s = nullptr, fb = nullptr;
  int rej = 0;
  const int st = make_concurrent_stream(w, {w->pipe_stream[0], w->pipe_stream[1]}, &s, &fb, &rej);
  if (st != RSB_OK) { if (fb) (void)hipStreamDestroy(fb); return st; }
  if (!s && fb) { s = fb; fb = nullptr; w->pipe_stage_overlap = false; }
  if (fb) (void)hipStreamDestroy(fb);
  if (!s) { rsb::set_error("rsb_closed_loop_run: no stream for the action stage could be created"); return RSB_E_HIP; }
  w->pipe_stage_stream = s;
  return RSB_OK;
}
// How many XCDs does the dispatcher deal this device's workgroups to, and is it a plain round-robin?  (MI355X in SPX mode: 8, and it is - but the
// XCD of workgroup 0 differs from launch to launch, profiles/r04_ubench_xcc_map.txt.)  Returns 0 when the pattern is anything else.
__global__ void pipe_xcc_probe_kernel(int* out) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(x & 15u);
}
int pipe_probe_xcds(rsb_world* w, int* n_xcds) {
  *n_xcds = 0;
  static const bool off = std::getenv("RSB_PIPE_XCD") && std::atoi(std::getenv("RSB_PIPE_XCD")) == 0;   // A/B switch: agent-scope hand-over everywhere
  if (off) return RSB_OK;
  constexpr int G = 256;
  int* d = nullptr;
  HIP_TRY(hipMalloc(&d, G * sizeof(int)));
  int h[G];
  bool ok = true;
  int nx = 0;
  for (int rep = 0; rep < 2 && ok; ++rep) {
    hipLaunchKernelGGL(pipe_xcc_probe_kernel, dim3(G), dim3(64), 0, w->pipe_stream[rep], d);
    if (hipStreamSynchronize(w->pipe_stream[rep]) != hipSuccess || hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
    int mx = 0;
    for (int b = 0; b < G; ++b) mx = std::max(mx, h[b]);
    const int n = mx + 1;
    ok = n >= 1 && n <= 16 && G % n == 0 && (rep == 0 || n == nx);
    for (int b = 0; ok && b < G; ++b) ok = h[b] == (h[0] + b) % n;
    nx = n;
  }
  (void)hipFree(d);
  if (ok) *n_xcds = nx;
  return RSB_OK;
}
// The gate in front of a pipelined launch: one thread that spins until the launch before it (other stream) has been dispatched completely
// (`started` has reached `target`), so that a waiting workgroup never holds a slot its predecessor needs.  No trap: past the time-out it
// stores the error word and lets the launch behind it run into it (every workgroup of that launch then leaves at its first look at the word).
__device__ inline void raise_error(int* err, int* err_host, int code) {      // the first code stays; the host reads its own copy
  if (atomicCAS(err, 0, code) == 0) __hip_atomic_store(err_host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <class T>
__global__ void pipe_gate_kernel(const T* started, T target, int* err, int* err_host, long long timeout) {
  int spins = 0;
  long long t0 = 0;
  while (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    __builtin_amdgcn_s_sleep(32);
    if ((++spins & 63) == 0) {
      const long long now = wall_clock64();
      if (t0 == 0) t0 = now;
      else if (now - t0 > timeout) { raise_error(err, err_host, RSB_PIPE_ERR_TIMEOUT_GATE); return; }
    }
  }
}
__global__ void fill_i32_kernel(int* a, int n, int v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}
__global__ void set_word_kernel(int* err, int* err_host, int v) { raise_error(err, err_host, v); }
// the state the steps since the last join started from: gc | gv | warm records, one buffer (restored by pipe_recover)
__global__ void snapshot_kernel(float* dst, const float* gc, size_t n0, const float* gv, size_t n1, const float* warm, size_t n2, int restore,
                                float* gc_w, float* gv_w, float* warm_w) {
  const size_t total = n0 + n1 + n2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    if (!restore) dst[i] = i < n0 ? gc[i] : i < n0 + n1 ? gv[i - n0] : warm[i - n0 - n1];
    else if (i < n0) gc_w[i] = dst[i];
    else if (i < n0 + n1) gv_w[i - n0] = dst[i];
    else warm_w[i - n0 - n1] = dst[i];
  }
}
int snapshot_on(rsb_world* w, bool restore, hipStream_t stream);
int snapshot(rsb_world* w, bool restore) { return snapshot_on(w, restore, w->stream); }
int snapshot_on(rsb_world* w, bool restore, hipStream_t stream) {
  const size_t n0 = (size_t)w->N * w->blob.nq, n1 = (size_t)w->N * w->blob.nv, n2 = (size_t)w->N * rsbk::kWarmRow;
  if (!restore && w->snap_cap < n0 + n1 + n2) {
    if (w->d_snap) HIP_TRY(hipFree(w->d_snap));
    w->d_snap = nullptr; w->snap_cap = 0;
    HIP_TRY(hipMalloc(&w->d_snap, (n0 + n1 + n2) * sizeof(float)));
    w->snap_cap = n0 + n1 + n2;
  }
  hipLaunchKernelGGL(snapshot_kernel, dim3(1024), dim3(256), 0, stream, w->d_snap, w->d_gc, n0, w->d_gv, n1, w->d_warm, n2, restore ? 1 : 0,
                     w->d_gc, w->d_gv, w->d_warm);
  HIP_TRY(hipGetLastError());
  return RSB_OK;
}

// (re)builds the pipeline's bookkeeping for a grid of `blocks` workgroups; everything in flight has been joined
int pipe_prepare(rsb_world* w, int blocks) {
  if (blocks == w->pipe_blocks && w->pipe_stream[0]) return RSB_OK;
  (void)stream_of(w);
  HIP_TRY(hipStreamSynchronize(w->stream));
  if (w->d_pipe_prog) HIP_TRY(hipFree(w->d_pipe_prog));
  w->d_pipe_prog = nullptr; w->pipe_blocks = 0;
  if (const char* e = std::getenv("RSB_PIPE_WORD_STRIDE")) w->pipe_stride = std::min(std::max(std::atoi(e), 1), 4096);
  HIP_TRY(hipMalloc(&w->d_pipe_prog, (size_t)2 * blocks * w->pipe_stride * sizeof(int)));   // step_prog | act_prog
  if (!w->d_pipe_started) HIP_TRY(hipMalloc(&w->d_pipe_started, kCtlBytes));
  if (!w->h_pipe_err) {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&w->h_pipe_err), 64, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&w->d_pipe_err_host), w->h_pipe_err, 0));
    *w->h_pipe_err = 0;
  }
  HIP_TRY(hipMemset(w->d_pipe_prog, 0, (size_t)2 * blocks * w->pipe_stride * sizeof(int)));
  HIP_TRY(hipMemset(w->d_pipe_started, 0, kCtlBytes));
  w->pipe_wg_total = 0; w->pipe_stats_wg0 = 0; w->pipe_seq = 0; w->pipe_xcc_uses = 0; w->stage_started_total = 0; w->stage_ticket_total = 0;
  if (!w->pipe_stream[0]) {
    const int ps = pipe_make_streams(w);
    if (ps != RSB_OK) return ps;
    const int px = pipe_probe_xcds(w, &w->pipe_xcds);
    if (px != RSB_OK) return px;
    HIP_TRY(hipMemset(w->d_pipe_started, 0, kCtlBytes));
  }
  for (int i = 0; i < 4; ++i) if (!w->pipe_ev[i]) HIP_TRY(hipEventCreateWithFlags(&w->pipe_ev[i], hipEventDisableTiming));
  w->pipe_blocks = blocks;     // (last: a failure above leaves the world un-pipelined, not half set up)
  return RSB_OK;
}

// fork (open loop): nothing is in flight.  The snapshot a fault is replayed from goes to the stream of the FIRST launch, in front of it; the second
// launch (other stream) is gated on the first having started, so it needs no event of its own.  The world's stream is waited for only when it is
// busy: an event wait between streams costs ~50 us even when there is nothing to wait for (2.5 % of a 20-step run).
int pipe_fork(rsb_world* w) {
  hipStream_t first = w->pipe_stream[w->pipe_next];
  if (hipStreamQuery(w->stream) != hipSuccess) {
    (void)hipGetLastError();
    HIP_TRY(hipEventRecord(w->pipe_ev[2], w->stream));
    HIP_TRY(hipStreamWaitEvent(first, w->pipe_ev[2], 0));
  }
  const int st = snapshot_on(w, false, first);
  if (st != RSB_OK) return st;
  w->pipe_n = 0;
  w->pipe_log.clear();
  w->pipe_time_logged = 0.0;
  return RSB_OK;
}

int closed_loop_lockstep(rsb_world* w, int K, rsb_stage_launch_fn launch, void* user, long long pass_global0);
int launch_mlp_stage(void* user, const rsb_stage_ctx* c);

// A fault on the device: every pipelined workgroup since has left without touching its envs, the envs are at different steps.  Back to the
// state of the last join, pipelining off, the logged steps once more in lock-step.
int pipe_recover(rsb_world* w, int code) {
  ++w->pipe_faults; w->pipe_last_code = code; w->pipe_fault_pending = true;
  *w->h_pipe_err = 0;
  std::fprintf(stderr, "raisimlib_amd: pipelined control steps faulted on the device (code %d: %s); restoring the last joined state and replaying %zu call(s) in lock-step, pipelining off\n",
               code, code == RSB_PIPE_ERR_TICKET ? "env-block ticket outside its XCD's range" : code == RSB_PIPE_ERR_TIMEOUT ? "a step workgroup's wait ran past the time-out"
               : code == RSB_PIPE_ERR_TIMEOUT_GATE ? "a gate's wait ran past the time-out" : code == RSB_PIPE_ERR_TIMEOUT_STAGE ? "an action-stage wave's wait ran past the time-out"
               : code == RSB_PIPE_ERR_STAGE ? "action stage geometry" : "injected", w->pipe_log.size());
  {   // where everybody stood (the words are about to be cleared)
    unsigned long long cb[1] = {};
    unsigned stg[1] = {};
    std::vector<int> all((size_t)2 * w->pipe_blocks * w->pipe_stride, 0), words((size_t)2 * w->pipe_blocks, 0);
    if (hipMemcpy(cb, w->d_pipe_started, sizeof cb, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(stg, stage_started_ptr(w), sizeof stg, hipMemcpyDeviceToHost) == hipSuccess &&
        hipMemcpy(all.data(), w->d_pipe_prog, all.size() * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess && w->pipe_blocks > 0) {
      for (size_t i = 0; i < words.size(); ++i) words[i] = all[i * w->pipe_stride];
      auto mm = [&](int k) { auto b = words.begin() + (size_t)k * w->pipe_blocks; auto r = std::minmax_element(b, b + w->pipe_blocks); return std::make_pair(*r.first, *r.second); };
      const auto sp = mm(0), ap = mm(1);
      std::fprintf(stderr, "raisimlib_amd:   step workgroups started %llu of %llu launched, stage workgroups started %u of %llu, sequence %u, per block: step_prog [%d, %d] act_prog [%d, %d], "
                   "XCDs %d, streams overlap: steps %d stage %d\n", cb[0], w->pipe_wg_total, stg[0], w->stage_started_total, w->pipe_seq, sp.first, sp.second, ap.first, ap.second,
                   w->pipe_xcds, (int)w->pipe_overlap, (int)w->pipe_stage_overlap);
    }
  }
  HIP_TRY(hipMemset(w->d_pipe_started, 0, kCtlBytes));
  w->pipe_wg_total = 0; w->pipe_stats_wg0 = 0; w->pipe_xcc_uses = 0; w->stage_started_total = 0; w->stage_ticket_total = 0;
  { const int nw = 2 * w->pipe_blocks * w->pipe_stride;
    hipLaunchKernelGGL(fill_i32_kernel, dim3((nw + 255) / 256), dim3(256), 0, w->stream, w->d_pipe_prog, nw, (int)w->pipe_seq); }
  HIP_TRY(hipGetLastError());
  if (code == RSB_PIPE_ERR_TICKET) w->pipe_xcds = 0;       // should pipelining be switched on again: hand-over at agent scope, blocks by workgroup index
  w->pipe_on = false;
  int st = snapshot(w, true);
  if (st != RSB_OK) return st;
  w->world_time -= w->pipe_time_logged;
  std::vector<rsb_world::PipeLog> log;
  log.swap(w->pipe_log);
  const uint8_t* keep_done = w->d_done_out;
  for (auto& e : log) {
    if (e.closed) {
      st = closed_loop_lockstep(w, e.K, e.is_linear ? nullptr : e.is_mlp ? launch_mlp_stage : e.launch, e.is_linear ? (void*)&e.lin : e.is_mlp ? (void*)&e.mlp : e.user, e.pass_global0);
    } else {
      w->fuse = e.f; w->fuse.pipeline = false;
      w->d_done_out = e.done_out;
      st = do_integrate(w, e.nsub);
    }
    if (st != RSB_OK) break;
  }
  w->d_done_out = const_cast<uint8_t*>(keep_done);
  w->integrate1_valid = false;
  return st;
}

}  // namespace

// Joins the pipelined control steps (if any are in flight): the host waits for the private streams, reads the error word and - after a fault -
// recovers.  Whatever is enqueued on the world's stream next runs after the steps.
int pipe_join(rsb_world* w) {
  if (!w->pipe_active) return RSB_OK;
  w->pipe_active = false;
  ++w->pipe_joins;
  hipError_t e = hipSuccess;
  for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipStreamSynchronize(w->pipe_stream[i]);
  if (e == hipSuccess && w->pipe_stage_stream) e = hipStreamSynchronize(w->pipe_stage_stream);
  const int code = e == hipSuccess ? *static_cast<volatile int*>(w->h_pipe_err) : 0;      // (written by whoever raised the error, visible once its kernel has completed)
  if (e != hipSuccess) { rsb::set_error(std::string("joining the pipelined control steps: ") + hipGetErrorString(e)); w->pipe_log.clear(); return RSB_E_HIP; }
  if (code != 0) return pipe_recover(w, code);
  w->pipe_log.clear();
  return RSB_OK;
}
// the world's stream for any use other than a pipelined step launch
hipStream_t stream_of(rsb_world* w) {
  if (w->pipe_active) (void)pipe_join(w);
  return w->stream;
}
// RSB_E_PIPELINE once after a fault (the state has been recovered by then), else RSB_OK
int fault_status(rsb_world* w) {
  if (!w->pipe_fault_pending) return RSB_OK;
  w->pipe_fault_pending = false;
  rsb::set_error("pipelined control steps faulted on the device (rsb_step_pipeline_fault has the code); the state of the last join was restored and the steps were replayed in lock-step, pipelining is off");
  return RSB_E_PIPELINE;
}

// A pipelined step launch (do_integrate): bookkeeping, fork, sequence numbers, the gate.  *ls receives the private stream of this launch.
int pipe_begin_launch(rsb_world* w, StepArgs& a, int blocks, bool closed_loop, hipStream_t* ls) {
  int st = pipe_prepare(w, blocks);
  if (st != RSB_OK) return st;
  a.pipe_prog = w->d_pipe_prog; a.pipe_started = w->d_pipe_started;
  a.pipe_err = err_ptr(w); a.pipe_err_host = w->d_pipe_err_host; a.pipe_timeout = timeout_ticks();
  if (!w->pipe_active) {
    // (sequence numbers and the started count carry on: every earlier pipelined launch has completed.  A closed-loop run forks itself.)
    st = pipe_fork(w);
    if (st != RSB_OK) return st;
  }
  *ls = w->pipe_stream[w->pipe_next];
  a.pipe_stride = w->pipe_stride;
  { static const bool stats = std::getenv("RSB_PIPE_STATS") != nullptr; a.pipe_stats = stats ? stats_ptr(w) : nullptr; }
  a.pipe_wait_ptr = closed_loop ? w->d_pipe_prog + (size_t)blocks * w->pipe_stride : w->d_pipe_prog;
  a.pipe_wait_on = (closed_loop || w->pipe_n > 0) ? 1 : 0;
  a.pipe_wait = (int)w->pipe_seq; a.pipe_seq = (int)(w->pipe_seq + 1u);
  a.pipe_xcds = (w->pipe_xcds > 0 && blocks % w->pipe_xcds == 0) ? w->pipe_xcds : 0;
  a.pipe_xcc_ctr = step_ticket_ptr(w);
  a.pipe_xcc_base = a.pipe_xcds > 0 ? w->pipe_xcc_uses * (unsigned)(blocks / a.pipe_xcds) : 0u;
  if (w->debug_fault) {      // rsb_debug_pipeline_fault: this launch fails on the device
    if (w->debug_fault == 1 && a.pipe_xcds > 0) a.pipe_xcc_base -= 1u;
    else if (w->debug_fault == 2) { a.pipe_wait_on = 1; a.pipe_wait += 1 << 20; }
    else hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, *ls, err_ptr(w), w->d_pipe_err_host, RSB_PIPE_ERR_INJECTED);
    w->debug_fault = 0;
  }
  // (the gate also keeps the per-XCD tickets of consecutive launches apart)
  if (w->pipe_n > 0)
    hipLaunchKernelGGL(pipe_gate_kernel<unsigned long long>, dim3(1), dim3(1), 0, *ls, (const unsigned long long*)w->d_pipe_started, w->pipe_wg_total, err_ptr(w), w->d_pipe_err_host, timeout_ticks());
  else if (closed_loop)    // first step of a closed-loop run: the action stage is on the chip (it never has to compete with waiting step workgroups for a slot)
    hipLaunchKernelGGL(pipe_gate_kernel<uint32_t>, dim3(1), dim3(1), 0, *ls, (const uint32_t*)stage_started_ptr(w), (uint32_t)w->stage_started_total, err_ptr(w), w->d_pipe_err_host, timeout_ticks());
  if (w->pipe_dep) { HIP_TRY(hipStreamWaitEvent(*ls, w->pipe_dep, 0)); w->pipe_dep = nullptr; }   // rsb_step_pipeline_wait_event
  HIP_TRY(hipGetLastError());
  return RSB_OK;
}
// ... and once it is on its way (only a launch that is on its way counts: the gate of the next one waits for this one's workgroups)
void pipe_end_launch(rsb_world* w, const StepArgs& a, hipStream_t ls) {
  w->pipe_active = true;
  ++w->pipe_n; w->pipe_next ^= 1; w->pipe_seq = (unsigned)a.pipe_seq;
  if (a.pipe_xcds > 0) ++w->pipe_xcc_uses;
  w->pipe_wg_total += (unsigned long long)w->pipe_blocks;
  w->pipe_last = ls;
  ++w->pipe_launches;
}

void pipe_destroy(rsb_world* w) {
  for (int i = 0; i < 2; ++i) if (w->pipe_stream[i]) (void)hipStreamDestroy(w->pipe_stream[i]);
  if (w->pipe_stage_stream) (void)hipStreamDestroy(w->pipe_stage_stream);
  for (int i = 0; i < 4; ++i) if (w->pipe_ev[i]) (void)hipEventDestroy(w->pipe_ev[i]);
  if (w->pipe_pub) (void)hipEventDestroy(w->pipe_pub);
  if (w->d_pipe_prog) (void)hipFree(w->d_pipe_prog);
  if (w->d_pipe_started) (void)hipFree(w->d_pipe_started);
  if (w->d_snap) (void)hipFree(w->d_snap);
  if (w->h_pipe_err) (void)hipHostFree(w->h_pipe_err);
}

namespace {

// A profiler that SERIALISES dispatches (rocprofv3 --pmc / counter collection, thread trace with serialize-all) runs one kernel at a time in an
// order of its own: a pipelined launch then waits for a predecessor that is not allowed to start (until round 5 the kernels trapped after
// ~10 s and rocprofv3 hung in its signal handler: profiles/r04_ab_log.txt, call S; now they time out into the error word and the steps are
// replayed).  Under such a tool - or with RSB_STEP_PIPELINING=0 - the switch stays off: counters are collected on the plain kernel classes.
bool pipelining_forbidden() {
  auto set = [](const char* n) { const char* v = std::getenv(n); return v && *v && std::strcmp(v, "0") != 0 && std::strcmp(v, "false") != 0 && std::strcmp(v, "False") != 0; };
  const char* force = std::getenv("RSB_STEP_PIPELINING");
  if (force && std::strcmp(force, "0") == 0) return true;
  return set("ROCPROF_COUNTER_COLLECTION") || set("ROCPROF_ATT_PARAM_SERIALIZE_ALL") || set("ROCPROFILER_COUNTER_COLLECTION");
}

// ---- the in-repo stages: a fixed linear policy (rsb_linear_policy) and an actor network on the matrix cores (rsb_mlp_policy).  Their per-env-block bodies
// live in stage_bodies.h: the resident step classes evaluate the same code inside the step kernel
__global__ void __launch_bounds__(64) linear_stage_kernel(const rsb_stage_ctx c, const rsb_linear_policy p) {
  rsb_stage::serve(c, [&](int, int env0, int n_env, int pass, bool final) { rsb_stage_body::linear_block(c, p, env0, n_env, pass, final); });
}
int launch_linear_stage(void* user, const rsb_stage_ctx* c) {
  hipLaunchKernelGGL(linear_stage_kernel, dim3(c->grid), dim3(64), 0, (hipStream_t)c->stream, *c, *static_cast<const rsb_linear_policy*>(user));
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int NS>
__global__ void __launch_bounds__(64) mlp_stage_kernel(const rsb_stage_ctx c, const rsb_mlp_policy p) {
  rsb_stage::serve(c, [&](int, int env0, int n_env, int pass, bool final) __attribute__((always_inline)) { rsb_stage_body::mlp_block<NS>(c, p, env0, n_env, pass, final); });
}
int mlp_width(const rsb_mlp_policy& p) { int m = 0; for (int l = 0; l <= p.n_layers; ++l) m = std::max(m, (int)p.dims[l]); return m; }
int launch_mlp_stage(void* user, const rsb_stage_ctx* c) {
  const rsb_mlp_policy& p = *static_cast<const rsb_mlp_policy*>(user);
  const int wd = mlp_width(p);
  if (wd <= 128) hipLaunchKernelGGL(mlp_stage_kernel<2>, dim3(c->grid), dim3(64), 0, (hipStream_t)c->stream, *c, p);
  else hipLaunchKernelGGL(mlp_stage_kernel<4>, dim3(c->grid), dim3(64), 0, (hipStream_t)c->stream, *c, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int cl_check(rsb_world* w, int K, const char* who) {
  if (!w || K < 1) { rsb::set_error(std::string(who) + ": bad argument"); return RSB_E_INVALID; }
  if (!w->env_ready) { rsb::set_error(std::string(who) + ": call rsb_env_configure first (the closed loop runs the device-resident env task)"); return RSB_E_STATE; }
  HIP_TRY(hipSetDevice(w->device));
  return RSB_OK;
}
void cl_fill_ctx(rsb_world* w, rsb_stage_ctx* c, int K, long long pass_global0) {
  std::memset(c, 0, sizeof *c);
  const int lpe = effective_lpe(w), epb = 64 / lpe;
  c->blocks = (w->N + epb - 1) / epb; c->envs_per_block = epb; c->n_envs = w->N;
  c->n_steps = K; c->pass_global0 = pass_global0; c->timeout_ticks = timeout_ticks();
  c->ob = w->d_env_ob; c->act = w->d_env_act; c->reward = w->d_env_reward; c->done = w->d_env_done;
  c->ob_dim = 10 + 2 * (w->blob.nv - 6); c->act_dim = w->blob.nv - 6;
}
// one closed-loop step of the env task: actions from the world's own buffer, reward / done / next observation into the world's own buffers
void cl_fuse(rsb_world* w, bool pipelined) {
  rsb_world::Fuse f;
  f.act = w->d_env_act;
  f.have_allowed = 1; f.allowed = w->env_allowed;
  f.do_reset = 1; f.gc0 = w->d_env_gc0; f.gv0 = w->d_env_gv0; f.rows = 1;
  if (w->d_env_gc0_rows) { f.gc0 = w->d_env_gc0_rows; f.gv0 = w->d_env_gv0_rows; f.rows = w->N; }      // rsb_env_set_reset_states
  f.env_task = true; f.env_reward = w->d_env_reward; f.env_ob = w->d_env_ob; f.env_done = w->d_env_done;
  f.closed_loop = pipelined; f.pipeline = pipelined;
  w->fuse = f;
}

// pass 0, step 1, pass 1, ..., step K, pass K on the world's stream; launch == nullptr: the linear stage with *user = rsb_linear_policy
int closed_loop_lockstep(rsb_world* w, int K, rsb_stage_launch_fn launch, void* user, long long pass_global0) {
  hipStream_t s = stream_of(w);
  int st = launch_env_obs(w, w->d_env_ob, s);
  if (st != RSB_OK) return st;
  rsb_stage_ctx c;
  cl_fill_ctx(w, &c, K, pass_global0);
  c.lockstep = 1; c.stream = s; c.grid = std::min(c.blocks, 1024);
  if (!launch) launch = launch_linear_stage;
  for (int t = 0; t <= K; ++t) {
    c.pass_first = c.pass_last = t;
    if (launch(user, &c) != 0) { rsb::set_error("rsb_closed_loop_run: the action stage's launch function failed"); return RSB_E_HIP; }
    if (t == K) break;
    cl_fuse(w, false);
    st = do_integrate(w, w->env_cfg.n_substeps);
    if (st != RSB_OK) return st;
  }
  w->integrate1_valid = false;
  return RSB_OK;
}

// ONE resident launch of the step kernel for the whole run (rsb_set_step_residency): pass 0, step 1, pass 1, ..., step K, pass K by the env block's own wave
int closed_loop_resident(rsb_world* w, int K, const rsb_linear_policy* lin, const rsb_mlp_policy* mlp, long long pass_global0) {
  hipStream_t s = stream_of(w);      // joins
  int st = RSB_OK;
  if (!w->env_ob_valid) { st = launch_env_obs(w, w->d_env_ob, s); if (st != RSB_OK) return st; }
  cl_fuse(w, false);
  w->fuse.res_steps = K; w->fuse.res_stage = lin ? 1 : 2; w->fuse.res_pass_global0 = pass_global0;
  if (lin) w->fuse.res_lin = *lin;
  if (mlp) w->fuse.res_mlp = *mlp;
  st = do_integrate(w, w->env_cfg.n_substeps);
  w->integrate1_valid = false;
  return st;
}

// the conditions under which do_integrate launches a step of a closed-loop run on the pipeline (else it takes the world's stream - and joins, with the
// action stage in flight: ADVICE r05).  Checked BEFORE the stage is launched; a run that fails the test stays in lock-step.
bool closed_loop_can_pipeline(const rsb_world* w) {
  static const bool poison = std::getenv("RSB_POISON_LDS") != nullptr;
  return w->pipe_on && !w->d_prof && w->dbg_env < 0 && !poison && !w->peer.connected && !w->launch_mask && !(w->integ_rk4 && !w->rk4_inner);
}

int closed_loop_run_inner(rsb_world* w, int K, rsb_stage_launch_fn launch, void* user, const rsb_linear_policy* lin, const rsb_mlp_policy* mlp, long long pg0);
int closed_loop_run(rsb_world* w, int K, rsb_stage_launch_fn launch, void* user, const rsb_linear_policy* lin, const rsb_mlp_policy* mlp = nullptr) {
  // the world's global pass index (the noise slice of the in-repo stages) advances only when the run has been enqueued (a refused run
  // used to shift the slices of every later one).  Pass K of this run sees what pass 0 of the next one sees: the index counts steps.
  const long long pg0 = w->cl_passes;
  const int st = closed_loop_run_inner(w, K, launch, user, lin, mlp, pg0);
  if (st == RSB_OK) w->cl_passes = pg0 + K;
  return st;
}
int closed_loop_run_inner(rsb_world* w, int K, rsb_stage_launch_fn launch, void* user, const rsb_linear_policy* lin, const rsb_mlp_policy* mlp, const long long pg0) {
  rsb_linear_policy lin_copy{};
  rsb_mlp_policy mlp_copy{};
  if (lin) { lin_copy = *lin; user = &lin_copy; launch = launch_linear_stage; }
  if (mlp) { mlp_copy = *mlp; user = &mlp_copy; launch = launch_mlp_stage; }
  if (w->res_on && (lin || mlp)) {     // (a caller-supplied stage cannot be compiled into the step kernel: it keeps the paths below)
    if (resident_class(w, lin ? 1 : 2, mlp ? mlp_width(*mlp) : 0) >= 0) return closed_loop_resident(w, K, lin, mlp, pg0);
  }
  if (!closed_loop_can_pipeline(w)) return closed_loop_lockstep(w, K, launch, user, pg0);
  // ---- pipelined: ONE launch of the stage for passes 0 .. K on its own stream, K step launches alternating between the two step streams
  hipStream_t s = stream_of(w);        // joins: a run starts from a quiet world (its snapshot is what a fault is replayed from)
  if (!closed_loop_can_pipeline(w)) return closed_loop_lockstep(w, K, launch, user, pg0);      // (that join found a fault: pipelining is off now)
  rsb_stage_ctx c;
  cl_fill_ctx(w, &c, K, pg0);
  int st = check_lpe(w, effective_lpe(w));
  if (st != RSB_OK) return st;
  st = upload_image(w);          // (do_integrate would do it - and JOIN for it, with the stage already in flight)
  if (st != RSB_OK) return st;
  st = pipe_prepare(w, c.blocks);
  if (st != RSB_OK) return st;
  st = pipe_make_stage_stream(w);
  if (st != RSB_OK) return st;
  if (!w->pipe_overlap || !w->pipe_stage_overlap) {
    // an action stage that shares a hardware queue with a step stream would not be slow but STUCK (it stays resident until the steps behind it in
    // that queue have run): without three streams that overlap the run stays in lock-step
    static bool said = false;
    if (!said) { std::fprintf(stderr, "raisimlib_amd: no three streams on different hardware queues (GPU_MAX_HW_QUEUES?): closed-loop runs stay in lock-step\n"); said = true; }
    return closed_loop_lockstep(w, K, launch, user, pg0);
  }
  // The run's own sequence numbers start ONE past the last one published (pass 0 publishes act_prog = seq0, the steps seq0 + 1 .. seq0 + K): whatever
  // an earlier run left in the words is smaller than anything this run waits for - no kernel has to reset them.
  w->pipe_seq += 1u;
  const int seq0 = (int)w->pipe_seq;
  int* act_prog = w->d_pipe_prog + (size_t)c.blocks * w->pipe_stride;
  // Everything the run needs before its first step goes to the STAGE's stream, in front of the stage kernel: the observation of the current state
  // (unless the last env-task step left it there), the snapshot a fault is replayed from.  The first step is gated on the stage having started, the
  // second on the first, ...: the step streams need no event of their own.  The world's stream is waited for only when it is busy (an event wait
  // between streams costs ~50 us of a 20-step run's 1.7 ms even when there is nothing to wait for).
  hipStream_t A = w->pipe_stage_stream;
  if (hipStreamQuery(s) != hipSuccess) {
    (void)hipGetLastError();
    HIP_TRY(hipEventRecord(w->pipe_ev[2], s));
    HIP_TRY(hipStreamWaitEvent(A, w->pipe_ev[2], 0));
  }
  if (!w->env_ob_valid) { st = launch_env_obs(w, w->d_env_ob, A); if (st != RSB_OK) return st; }
  st = snapshot_on(w, false, A);
  if (st != RSB_OK) return st;
  w->pipe_n = 0;
  w->pipe_log.clear();
  w->pipe_time_logged = 0.0;
  c.step_prog = w->d_pipe_prog; c.act_prog = act_prog; c.ticket = stage_ticket_ptr(w);
  c.err = err_ptr(w); c.err_host = w->d_pipe_err_host; c.started = stage_started_ptr(w);
  c.word_stride = w->pipe_stride;
  { const char* e = std::getenv("RSB_STAGE_POLL"); c.poll_sleep = e ? std::min(std::max(std::atoi(e), 0), 64) : 2; }
  c.xcds = (w->pipe_xcds > 0 && c.blocks % w->pipe_xcds == 0) ? w->pipe_xcds : 0;
  c.seq0 = seq0; c.pass_first = 0; c.pass_last = K; c.lockstep = 0;
  c.stream = w->pipe_stage_stream;
  // the stage's waves: every wave serves a fixed share of at most 64 blocks of its XCD (rsb_stage::serve)
  const int nx = c.xcds > 0 ? c.xcds : 1, per = c.blocks / nx;
  int T = (w->cl_grid > 0 ? w->cl_grid : mlp ? kStageGridMlp : kStageGridDefault) / nx;
  T = std::max(T, (per + 63) / 64);
  T = std::min(std::max(T, 1), per);
  c.grid = T * nx;
  c.ticket_base = (uint32_t)w->stage_ticket_total;
  w->stage_ticket_total += (unsigned long long)T;
  w->stage_started_total += (unsigned long long)c.grid;
  if (launch(user, &c) != 0) { rsb::set_error("rsb_closed_loop_run: the action stage's launch function failed"); w->stage_started_total -= (unsigned long long)c.grid; w->stage_ticket_total -= (unsigned long long)T; return RSB_E_HIP; }
  w->pipe_active = true;       // the stage is in flight: whatever happens below, the next join waits for it
  rsb_world::PipeLog e;
  e.closed = true; e.K = K; e.launch = launch; e.user = user; e.pass_global0 = pg0;
  e.is_linear = lin != nullptr; if (lin) e.lin = *lin;
  e.is_mlp = mlp != nullptr; if (mlp) e.mlp = *mlp;
  w->pipe_log.push_back(e);
  w->pipe_log_suppress = true;
  for (int t = 0; t < K && st == RSB_OK; ++t) {
    cl_fuse(w, true);
    st = do_integrate(w, w->env_cfg.n_substeps);
    if (st == RSB_OK && !w->pipe_active) {      // the step did not go to the pipeline (it joined): the stage has nobody to wait for - treat it as a failed enqueue
      rsb::set_error("rsb_closed_loop_run: a step of the run left the pipeline (profiling / debug instrumentation switched on mid-run?)");
      st = RSB_E_STATE;
    }
  }
  w->pipe_log_suppress = false;
  if (st != RSB_OK) {
    // a step could not be enqueued: the stage would wait for it until its time-out.  Tell the device now, then join (recovers from the snapshot)
    hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, w->stream, err_ptr(w), w->d_pipe_err_host, RSB_PIPE_ERR_INJECTED);
    const std::string msg = rsb::last_error();
    (void)pipe_join(w);
    rsb::set_error(msg);
    return st;
  }
  w->integrate1_valid = false;
  return RSB_OK;
}

}  // namespace
}  // namespace rsbw

using namespace rsbw;

extern "C" {

int rsb_set_step_pipelining(rsb_world* w, int on) {
  if (!w) { rsb::set_error("rsb_set_step_pipelining: null world"); return RSB_E_INVALID; }
  HIP_TRY(hipSetDevice(w->device));
  (void)stream_of(w);
  if (on && pipelining_forbidden()) {
    static bool said = false;
    if (!said) { std::fprintf(stderr, "raisimlib_amd: control steps stay un-pipelined (RSB_STEP_PIPELINING=0 or a dispatch-serialising profiler in the environment)\n"); said = true; }
    w->pipe_on = false;
    return fault_status(w);
  }
  w->pipe_on = on != 0;
  return fault_status(w);
}
int rsb_step_pipelining_enabled(const rsb_world* w) { return w && w->pipe_on ? 1 : 0; }
int rsb_step_pipeline_publish(rsb_world* w, void* hip_stream) {
  if (!w) { rsb::set_error("rsb_step_pipeline_publish: null world"); return RSB_E_INVALID; }
  HIP_TRY(hipSetDevice(w->device));
  if (!w->pipe_pub) HIP_TRY(hipEventCreateWithFlags(&w->pipe_pub, hipEventDisableTiming));
  const bool in_flight = w->pipe_active && w->pipe_last;
  if (!in_flight && (hipStream_t)hip_stream == w->stream) return RSB_OK;       // nothing in flight and the world's own stream: already ordered
  HIP_TRY(hipEventRecord(w->pipe_pub, in_flight ? w->pipe_last : w->stream));   // (no pipelined step in flight: the last step is on the world's stream)
  HIP_TRY(hipStreamWaitEvent((hipStream_t)hip_stream, w->pipe_pub, 0));
  return RSB_OK;
}
int rsb_step_pipeline_wait_event(rsb_world* w, void* hip_event) {
  if (!w) { rsb::set_error("rsb_step_pipeline_wait_event: null world"); return RSB_E_INVALID; }
  w->pipe_dep = (hipEvent_t)hip_event;
  return RSB_OK;
}
int rsb_step_pipelining_stats(rsb_world* w, long long* launches, long long* joins) {
  if (!w) return RSB_E_INVALID;
  if (launches) *launches = w->pipe_launches;
  if (joins) *joins = w->pipe_joins;
  return (w->pipe_overlap && w->pipe_stage_overlap) ? RSB_OK : 1;
}
int rsb_step_pipeline_join(rsb_world* w) {
  if (!w) { rsb::set_error("rsb_step_pipeline_join: null world"); return RSB_E_INVALID; }
  HIP_TRY(hipSetDevice(w->device));
  if (w->pipe_active) { const int st = pipe_join(w); if (st != RSB_OK) return st; }
  return fault_status(w);
}
// diagnostics (RSB_PIPE_STATS=1 in the environment): mean wait of a pipelined step workgroup for its block, in microseconds, and the share of the
// workgroups that had to wait at all, over the pipelined launches since the last call; joins
int rsb_debug_pipeline_wait_stats(rsb_world* w, double* mean_wait_us, double* waited_frac) {
  if (!w || !w->d_pipe_started) return RSB_E_INVALID;
  HIP_TRY(hipSetDevice(w->device));
  HIP_TRY(hipStreamSynchronize(stream_of(w)));
  unsigned long long st[2] = {0, 0};
  unsigned long long* d = stats_ptr(w);
  HIP_TRY(hipMemcpy(st, d, sizeof st, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(d, 0, sizeof st));
  const double n = (double)(w->pipe_wg_total - w->pipe_stats_wg0);
  w->pipe_stats_wg0 = w->pipe_wg_total;
  if (mean_wait_us) *mean_wait_us = n > 0 ? (double)st[0] / n * 0.01 : 0.0;
  if (waited_frac) *waited_frac = n > 0 ? (double)st[1] / n : 0.0;
  return RSB_OK;
}
int rsb_step_pipeline_fault(const rsb_world* w, int* faults, int* last_code) {
  if (!w) return RSB_E_INVALID;
  if (faults) *faults = w->pipe_faults;
  if (last_code) *last_code = w->pipe_last_code;
  return RSB_OK;
}
int rsb_debug_pipeline_fault(rsb_world* w, int kind) {
  if (!w || (kind != 0 && kind != 1 && kind != 2 && kind != 4)) { rsb::set_error("rsb_debug_pipeline_fault: kind must be 0, 1, 2 or 4"); return RSB_E_INVALID; }
  w->debug_fault = kind;
  return RSB_OK;
}

int rsb_closed_loop_run(rsb_world* w, int n_steps, rsb_stage_launch_fn launch, void* user) {
  int st = cl_check(w, n_steps, "rsb_closed_loop_run"); if (st != RSB_OK) return st;
  if (!launch) { rsb::set_error("rsb_closed_loop_run: no launch function"); return RSB_E_INVALID; }
  return closed_loop_run(w, n_steps, launch, user, nullptr);
}
int rsb_closed_loop_run_linear(rsb_world* w, int n_steps, const rsb_linear_policy* policy) {
  int st = cl_check(w, n_steps, "rsb_closed_loop_run_linear"); if (st != RSB_OK) return st;
  if (!policy || !policy->W || (policy->noise && policy->noise_period < 1)) { rsb::set_error("rsb_closed_loop_run_linear: W is required, noise needs noise_period >= 1"); return RSB_E_INVALID; }
  return closed_loop_run(w, n_steps, nullptr, nullptr, policy);
}
int rsb_closed_loop_run_mlp(rsb_world* w, int n_steps, const rsb_mlp_policy* p) {
  int st = cl_check(w, n_steps, "rsb_closed_loop_run_mlp"); if (st != RSB_OK) return st;
  if (!p || p->n_layers < 1 || p->n_layers > RSB_MLP_MAX_LAYERS) { rsb::set_error("rsb_closed_loop_run_mlp: 1 .. RSB_MLP_MAX_LAYERS layers"); return RSB_E_INVALID; }
  const int od = 10 + 2 * (w->blob.nv - 6), ad = w->blob.nv - 6;
  if (p->dims[0] != od || p->dims[p->n_layers] != ad) { rsb::set_error("rsb_closed_loop_run_mlp: dims[0] must be the env's observation size and dims[n_layers] its action size"); return RSB_E_INVALID; }
  for (int l = 0; l <= p->n_layers; ++l) if (p->dims[l] < 1 || p->dims[l] > 256) { rsb::set_error("rsb_closed_loop_run_mlp: layer widths 1 .. 256"); return RSB_E_INVALID; }
  for (int l = 0; l < p->n_layers; ++l) if (!p->Wt[l]) { rsb::set_error("rsb_closed_loop_run_mlp: a layer's weight pointer is null"); return RSB_E_INVALID; }
  if (p->activation != RSB_ACT_TANH && p->activation != RSB_ACT_RELU && p->activation != RSB_ACT_LEAKY_RELU) { rsb::set_error("rsb_closed_loop_run_mlp: unknown activation"); return RSB_E_INVALID; }
  if (p->noise && p->noise_period < 1) { rsb::set_error("rsb_closed_loop_run_mlp: noise needs noise_period >= 1"); return RSB_E_INVALID; }
  if (effective_lpe(w) < 16) { rsb::set_error("rsb_closed_loop_run_mlp: at most four envs per block"); return RSB_E_UNSUPPORTED; }
  return closed_loop_run(w, n_steps, nullptr, nullptr, nullptr, p);
}
int rsb_closed_loop_buffers(rsb_world* w, float** ob, float** act, float** reward, uint8_t** done) {
  int st = cl_check(w, 1, "rsb_closed_loop_buffers"); if (st != RSB_OK) return st;
  if (ob) *ob = w->d_env_ob;
  if (act) *act = w->d_env_act;
  if (reward) *reward = w->d_env_reward;
  if (done) *done = w->d_env_done;
  return RSB_OK;
}
int rsb_closed_loop_set_stage_grid(rsb_world* w, int workgroups) {
  if (!w || workgroups < 0 || workgroups > 4096) { rsb::set_error("rsb_closed_loop_set_stage_grid: 0 .. 4096 workgroups"); return RSB_E_INVALID; }
  w->cl_grid = workgroups;
  return RSB_OK;
}

}  // extern "C"
