// rsb_world.h — the batched world's host-side state and the helpers its translation units share (not installed):
//   rsb_world.hip     the core of the C-ABI (include/rsb.h): creation, setters, state transfer, the step launch (do_integrate), queries, env task
//   rsb_pipeline.hip  pipelined control steps: private streams, probes, gates, join + fault recovery, the closed-loop run (include/rsb_pipeline.h)
//   rsb_comm.hip      multi-GPU: the RCCL obs all-gather and the peer-mapped obs exchange
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "rsb.h"
#include "rsb_pipeline.h"
#include "rsb_internal.h"
#include "step_types.h"

using rsbk::DevModel;
using rsbk::LdsLayout;
using rsbk::StepArgs;

#define HIP_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      rsb::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                        \
      return RSB_E_HIP;                                                                         \
    }                                                                                           \
  } while (0)

struct rsb_world {
  rsb_model_blob blob;
  int N = 0, device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  DevModel* d_model = nullptr;
  float *d_gc = nullptr, *d_gv = nullptr, *d_pt = nullptr, *d_dt = nullptr, *d_tff = nullptr;
  float *d_kp = nullptr, *d_kd = nullptr, *d_heights = nullptr;
  bool raw_dt = false, raw_tff = false, raw_state = false;   // rsb_device_ptr handed the field's device pointer out: the zero-row shortcut / the cached env observation are off for good
  bool dt_zero = true, tff_zero = true;   // d_dt / d_tff hold nothing but zeros (never written, or written with zeros from the host): the step kernel does not read them
  float *d_tmp_gc = nullptr, *d_tmp_gv = nullptr;
  uint8_t* d_tmp_mask = nullptr;
  float *d_M = nullptr, *d_h = nullptr, *d_Minv = nullptr, *d_Mwork = nullptr;
  int32_t* d_obs_idx = nullptr;
  int32_t* d_hm_index = nullptr;   // [N] height map of each env (rsb_set_heightmaps), NULL: all envs share map 0
  float* d_image = nullptr;           // the step kernel's per-block tables in their LDS layout (StepArgs::lds_image), rebuilt when a setter dirties it
  std::vector<float> h_kp, h_kd;      // host mirror of the PD gains (baked into the image)
  std::vector<double> col_mu, col_rest, col_rthr;   // per-primitive overrides, < 0 = the world's default
  bool image_dirty = true;
  // self-collision (rsb_set_self_collision): candidate primitive pairs i < j in enumeration order, body pairs the caller
  // excluded (rsb_ignore_collision_between), per-pair material overrides (< 0 = the world's default)
  bool self_collision = true;
  std::vector<uint8_t> self_ignore;            // [nb * nb]
  std::vector<int> self_pairs;                 // 2 ints per pair
  std::vector<double> self_mu, self_rest, self_rthr;
  float* d_self_mat = nullptr;
  float* d_genf = nullptr;              // [N, nv] generalized force applied in the last sub-step (rsb_enable_generalized_force_output)
  bool want_genf = false;
  size_t self_mat_cap = 0;
  float* d_warm = nullptr;   // [N, kWarmRow] contact-solver warm state (StepArgs::warm: one record per contact of the last integrate())
  bool warm_start = true;
  uint8_t* d_done_out = nullptr;        // caller-owned device buffer (rsb_set_done_output): done flags of the fused control step
  const uint8_t* launch_mask = nullptr; // env mask of the next launch only (rsb_integrate_masked)
  uint8_t* d_launch_mask = nullptr;     // staging for host masks
  uint8_t* d_view_masks = nullptr;      // [n_launches][N] launch masks of rsb_view_exchange
  size_t view_masks_cap = 0;
  void* comm = nullptr;                 // ncclComm_t (rsb_comm_init)
  int comm_ranks = 0, comm_rank = 0;
  float *d_obs_local = nullptr, *d_obs_all = nullptr;   // staging of rsb_allgather_obs
  size_t obs_local_cap = 0, obs_all_cap = 0;
  bool early_term = false;   // rsb_set_early_termination
  std::vector<int32_t> obs_idx_host;   // what d_obs_idx currently holds (re-uploaded only when the caller's list changes)
  float* d_dbg = nullptr;
  long long* d_prof = nullptr;
  int dbg_env = -1;
  rsb_contact* d_contacts = nullptr;
  int32_t *d_count = nullptr, *d_flags = nullptr, *d_iters = nullptr;
  // parameters
  double dt = 0.0025, gravity[3] = {0, 0, -9.81}, mu = 0.8, erp = 0.0;
  double alpha_init = 1.0, alpha_min = 1.0, alpha_decay = 1.0, threshold = 1e-5;
  int max_iter = 150, section_rounds = 2, stall_window = 4, freeze_after = 6, refine = 1, kmax = 8, control_mode = RSB_PD_PLUS_FEEDFORWARD_TORQUE;
  int multi_depth = 3, multi_light = 0, multi_freeze_after = 0, multi_stall_window = 16;   // rsb_set_solver_multi_contact
  int anderson = 2; double anderson_clip = 20.0;                                           // rsb_set_solver_anderson
  int hm_contacts = 1; double hm_second_cos = 0.70710678118654752;                                        // rsb_set_heightmap_contacts
  bool hm_capsule = false; int32_t* d_cap = nullptr; int n_cap = 0;                                       // rsb_set_capsule_contacts: [n_cap][2] end primitives of the model's capsules / cylinders, (first corner, -1) of its boxes
  int slip_rule = 0;                                                                       // rsb_set_slip_rule (RSB_SLIP_ENERGY / RSB_SLIP_COULOMB)
  bool integ_rk4 = false, rk4_inner = false;                                                // IntegrationScheme::RUNGE_KUTTA_4 (rsb_rk4.hip); rk4_inner: the scheme's own contact step is being launched
  float* d_rk = nullptr;                                                                   // its scratch
  double integ_theta = 1.0;                                                                // rsb_set_integration_scheme
  // peer-mapped obs exchange (rsb_obs_peer_*).  ONE allocation per rank, the same layout on every rank:
  //   [gathered buffer, parity 0 | parity 1]  2 x n_ranks * N * obs_dim floats
  //   [flags, parity 0 | parity 1]            2 x RSB_MAX_RANKS uint32: flags[parity][p] = last control step whose rows rank p delivered
  struct Peer {
    int ranks = 0, rank = 0, slots = 0, od = 0;
    bool connected = false, wait_by_kernel = false;
    void* base = nullptr; size_t bytes = 0;
    void* peer_base[RSB_MAX_RANKS] = {};
    bool imported[RSB_MAX_RANKS] = {};
    uint32_t step = 0;                       // sequence number of the last control step issued with the exchange
    std::vector<int32_t> idx;                // force slots' collision primitives (empty: 0..slots-1)
    int32_t* d_idx = nullptr;
  } peer;
  int terrain_type = 0, hm_xs = 0, hm_ys = 0;
  double ground_z = 0, hm_xsize = 0, hm_ysize = 0, hm_cx = 0, hm_cy = 0;
  float hm_max = 0.f;
  double stall_factor = 0.5, settle_tol = 0.0, restitution = 0.0, res_threshold = 0.0;
  int lpe = 0, max_kid = 0;
  bool chain = false;         // base + consecutively numbered serial chains, <= 16 bodies (StepArgs::chain)
  double world_time = 0;
  bool integrate1_valid = false;
  bool env_ob_valid = false;            // d_env_ob holds the env-task observation of the CURRENT state (left there by the last env-task step; any other state change clears it)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timing = false;
  // epilogue / prologue fused into the next launch by rsb_control_step (consumed by do_integrate)
  struct Fuse { bool peer = false; const float* act = nullptr; const float* ptarget_src = nullptr; float* obs_out = nullptr; const int32_t* obs_idx = nullptr; int obs_slots = 0;
                int do_reset = 0, have_allowed = 0; unsigned long long allowed = 0; const float *gc0 = nullptr, *gv0 = nullptr; int rows = 1;
                float* env_reward = nullptr; float* env_ob = nullptr; uint8_t* env_done = nullptr; bool env_task = false; bool pipeline = false; bool closed_loop = false;
                // resident launch (rsb_set_step_residency): K control steps in this ONE launch; stage 0 = open loop (targets from a bank), 1 = linear policy, 2 = actor network
                int res_steps = 0, res_stage = 0; const float* res_targets = nullptr; int res_period = 0; long long res_first = 0, res_obs_stride = 0, res_done_stride = 0, res_pass_global0 = 0;
                uint8_t* res_done = nullptr; rsb_linear_policy res_lin{}; rsb_mlp_policy res_mlp{}; } fuse;
  // device-resident vectorised env (rsb_env_*)
  bool env_ready = false;
  rsb_env_config env_cfg{};
  unsigned long long env_allowed = 0;
  float *d_env_mean = nullptr, *d_env_gc0 = nullptr, *d_env_gv0 = nullptr, *d_env_io = nullptr, *d_env_ob = nullptr, *d_env_reward = nullptr, *d_env_tau2 = nullptr;
  uint8_t* d_env_done = nullptr;
  std::vector<hipEvent_t> ring0, ring1;   // event pairs around the most recent step-kernel launches (rsb_enable_timing(w, n))
  size_t ring_next = 0, ring_count = 0;
  int timing_stride = 1;       // events bracket every timing_stride-th launch only (an event pair costs ~7 us of stream time)
  long long launch_index = 0;
  float last_ms = -1.f;
  // pipelined control steps (rsb_set_step_pipelining): consecutive rsb_control_step launches alternate between two private streams and
  // overlap on the device (see StepArgs::pipe_prog); any other use of the world's stream joins them first (stream_of)
  bool pipe_on = false, pipe_active = false;
  hipStream_t pipe_stream[2] = {nullptr, nullptr};
  hipEvent_t pipe_ev[4] = {nullptr, nullptr, nullptr, nullptr};   // [2]: fork event of the world's stream
  int pipe_next = 0, pipe_blocks = 0;
  unsigned long long pipe_n = 0;                         // pipelined launches since the fork
  unsigned long long pipe_wg_total = 0;                  // their workgroups (== *d_pipe_started once they have all started)
  unsigned pipe_seq = 0;                                 // sequence number of the last pipelined launch (published by its workgroups)
  unsigned long long* d_pipe_started = nullptr;
  int* d_pipe_prog = nullptr;                            // step_prog [pipe_blocks * pipe_stride] | act_prog [pipe_blocks * pipe_stride]: block b's words at b * pipe_stride
  int pipe_stride = 64;                                  // ints between the words of consecutive blocks (256 B: spread over the memory channels; RSB_PIPE_WORD_STRIDE)
  hipStream_t launch_stream = nullptr;                   // stream of the step launch being enqueued (do_integrate)
  hipStream_t pipe_last = nullptr;                       // private stream of the most recent pipelined launch
  hipEvent_t pipe_dep = nullptr, pipe_pub = nullptr;     // rsb_step_pipeline_wait_event: the next pipelined launch waits for it; event of rsb_step_pipeline_publish
  long long pipe_launches = 0, pipe_joins = 0;
  bool pipe_overlap = true;                              // the probe found two streams whose kernels run concurrently (pipe_make_streams)
  int pipe_probe_rejected = 0;
  int pipe_xcds = 0;                                     // XCDs the dispatcher deals workgroups to round-robin (0: pattern not recognised -> agent-scope hand-over)
  unsigned pipe_xcc_uses = 0;                            // pipelined launches since the counters were cleared
  // ---- round 5: fault recovery instead of device traps, and the closed-loop run (include/rsb_pipeline.h, rsb_pipeline.hip)
  int pipe_faults = 0, pipe_last_code = 0;               // faults so far, device code of the last one (RSB_PIPE_ERR_*)
  bool pipe_fault_pending = false;                       // the next status-returning call that joins reports RSB_E_PIPELINE once
  int debug_fault = 0;                                   // rsb_debug_pipeline_fault: the next pipelined launch fails this way
  int* h_pipe_err = nullptr; int* d_pipe_err_host = nullptr;   // the error word's copy for the host: page-locked host memory and its device address
  float* d_snap = nullptr; size_t snap_cap = 0;          // gc | gv | warm records at the last fork (what a faulted pipeline is replayed from)
  struct PipeLog {                                       // one call since the last fork: a control step (open loop) or a whole closed-loop run
    bool closed = false;
    Fuse f; int nsub = 0; uint8_t* done_out = nullptr;
    int K = 0; rsb_stage_launch_fn launch = nullptr; void* user = nullptr; long long pass_global0 = 0;
    bool is_linear = false; rsb_linear_policy lin{};
    bool is_mlp = false; rsb_mlp_policy mlp{};
  };
  std::vector<PipeLog> pipe_log;
  bool pipe_log_suppress = false;                        // the steps of a closed-loop run are logged as ONE entry
  double pipe_time_logged = 0.0;                         // world time the logged calls advanced (taken back before a replay)
  hipStream_t pipe_stage_stream = nullptr;               // the action stage's stream (overlaps with both step streams)
  bool pipe_stage_overlap = true;
  unsigned long long pipe_stats_wg0 = 0;                 // rsb_debug_pipeline_wait_stats: workgroups counted up to the last call
  unsigned long long stage_ticket_total = 0;             // per-XCD arrival tickets the stages launched so far have drawn
  unsigned long long stage_started_total = 0;            // stage workgroups launched since the control block was cleared (the first step of a run waits for them)
  float *d_env_gc0_rows = nullptr, *d_env_gv0_rows = nullptr;   // optional per-env reset states [N, nq] / [N, nv] (rsb_env_set_reset_states)
  float* d_env_act = nullptr;                            // [N, nv - 6] the env task's action rows (closed loop: written by the stage, read by the step)
  long long cl_passes = 0;                               // closed-loop steps this world has run (index of the next run's pass 0)
  int cl_grid = 0;                                       // workgroups of the action stage (0: default)
  // ---- round 6: resident launches (rsb_set_step_residency): K control steps per launch of the step kernel, the env blocks stay in LDS
  long long view_prof[5] = {0, 0, 0, 0, 0};             // rsb_debug_view_profile: ns the host spent in rsb_view_exchange enqueueing uploads / launches / downloads, waiting; calls
  bool res_on = false, res_full = false;
  long long res_launches = 0;
  // ---- specialised code objects of the step kernel (rsb_set_specialization; rsb_spec.hip)
  int spec_mode = RSB_SPEC_CACHED;
  long long spec_launches = 0, generic_launches = 0;     // step launches that ran a specialised code object / an ahead-of-time class
  std::map<std::vector<int>, void*> spec_memo;           // (class, field values, mode) -> hipFunction_t or nullptr
};

// helpers shared by the translation units (rsb_world.hip unless noted)
namespace rsbw {
int do_integrate(rsb_world* w, int nsub);
int upload_image(rsb_world* w);                                 // the step kernel's per-block tables, when a setter dirtied them (joins)
int effective_lpe(const rsb_world* w);
int check_lpe(const rsb_world* w, int lpe);
int copy_in(rsb_world* w, float* dst, const float* src, size_t n, int space);
int copy_out(rsb_world* w, void* dst, const void* src, size_t bytes, int space);
int launch_env_obs(rsb_world* w, float* dst, hipStream_t s);
int launch_dynamics_query(rsb_world* w, hipStream_t s);           // M, h and M^-1 of the current state into d_M / d_h / d_Minv (the query kernels)
int resident_class(rsb_world* w, int stage, int mlp_width);          // the resident kernel class (CL bits) of this world as configured, or -1 with the reason in the error string
int rk4_integrate(rsb_world* w, int nsub);                        // rsb_rk4.hip     // the stand-alone env-task observation of the current state
// rsb_pipeline.hip
hipStream_t stream_of(rsb_world* w);                              // the world's stream for any use other than a pipelined launch (joins first)
int pipe_join(rsb_world* w);
int fault_status(rsb_world* w);                                   // RSB_E_PIPELINE once after a fault, else RSB_OK
int pipe_begin_launch(rsb_world* w, StepArgs& a, int blocks, bool closed_loop, hipStream_t* ls);
void pipe_end_launch(rsb_world* w, const StepArgs& a, hipStream_t ls);
void pipe_destroy(rsb_world* w);
}  // namespace rsbw
