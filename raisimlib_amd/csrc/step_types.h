// step_types.h — argument / layout structs shared by the step kernel (step_kernel.h, device) and the host code that
// fills them (rsb_world.hip).  Split out so that the host translation unit does not have to compile the kernel body:
// every (LPE, KMAX, CL, ML, PROF) instance of the kernel is its own object file (step_instance.hip, built in parallel).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsb_types.h"
#include "rsb_pipeline.h"   // rsb_linear_policy / rsb_mlp_policy: the resident classes carry the action stage's policy in their arguments

namespace rsbk {

constexpr int kMaxB = RSB_MAX_BODIES;
constexpr int kMaxC = RSB_MAX_COLLISIONS;
#ifndef RSB_X_BODYSLOT            // (layout experiments: slot pitches and the per-env pad decide the LDS bank pattern, not the instruction stream)
#define RSB_X_BODYSLOT 24
#endif
#ifndef RSB_X_UPSLOT
#define RSB_X_UPSLOT 28
#endif
#ifndef RSB_X_ENVPAD
#define RSB_X_ENVPAD 0
#endif
constexpr int kBodySlot = RSB_X_BODYSLOT;    // R9 r3 V6 A6 (A is reused for the delta-velocity of the final pass) [+ pad]
constexpr int kUpSlot = RSB_X_UPSLOT;        // Ia21 Zc6 pad   (one per body)
constexpr int kEnvPad = RSB_X_ENVPAD;        // floats added to an env's LDS region (shifts the banks the wave's envs start on)
constexpr int kFactSlot = 16;    // S6 UD6 rsD invD pad2
constexpr int kConSlot = 16;     // x3 depth | t1 body | t2 col | n pad
constexpr int kColSlot = 12;     // per collision primitive in LDS: centre3 radius | body mu restitution res_threshold | axis3 rim (rim > 0: see rsb_model_blob::col_rim)
constexpr int kModelSlot = 32;   // per-body constants staged in LDS (see DevModel::bodyf)
#ifndef RSB_X_MODEL_PITCH
#define RSB_X_MODEL_PITCH 36
#endif
constexpr int kModelPitch = RSB_X_MODEL_PITCH;   // floats between two bodies' constants IN LDS: with 36 the sixteen lanes of an env (lane = body) start their 16-byte reads on
                                  // sixteen different four-bank groups (36 i mod 64 = 0, 36, 8, 44 ...); with the slot's own 32 they shared two - an 8-way bank conflict
                                  // on each of the eight reads of the down pass
constexpr float kLambdaFloor = 1e-3f;  // N s, floor of the relative convergence test (== ORC_LAMBDA_FLOOR)
constexpr float kDenMin = 1e-6f;       // == ORC_DEN_MIN
constexpr float kDenFreeze = 0.1f;    // == ORC_DEN_FREEZE
constexpr float kDenNewton = 1e-3f;   // == ORC_DEN_NEWTON
constexpr int kPolishSteps = 2;       // == ORC_POLISH_STEPS
constexpr int kSecond = 0x40000;      // collision id flag of a primitive's SECOND contact with a height map (== RSB_CONTACT_SECOND, ORC_SECOND)
constexpr int kCapsule = 0x80000;     // ... of the contact of a capsule's cylinder, carried by its first end sphere's id (== RSB_CONTACT_CAPSULE, ORC_CAPSULE)
constexpr int kExtra = kSecond | kCapsule;   // contacts that belong to a primitive besides its first one (class-4 kernels): the primitive's material, a cold start
constexpr float kCapsuleMargin = 1e-4f;   // == ORC_CAPSULE_MARGIN
constexpr int kCapsuleRounds = 4;         // == ORC_CAPSULE_ROUNDS
constexpr float kBoxTie = 1e-5f;          // == ORC_BOX_TIE
constexpr int kBoxSpan = 32;              // == ORC_BOX_SPAN
constexpr int kSelfA = 0x10000;       // collision id flags of the two entries of a self-collision (== RSB_CONTACT_SELF_A / _B, ORC_SELF_A / _B)
constexpr int kSelfB = 0x20000;
constexpr int kSelfBatch = 5;          // passes per batch of the self-collision sweep
constexpr float kSelfReg = 1e-4f;      // == ORC_SELF_REG: compliance of a self-collision's Delassus block, relative to its mean diagonal
constexpr int kWarmRec = 8;           // floats per warm-state record in HBM: impulse (3), friction direction (2), direction valid, primitive + 1, pad
constexpr int kWarmRow = kWarmRec * RSB_MAX_CONTACTS;   // floats per env row of StepArgs::warm
constexpr int kHmRec = 28;            // floats per slot of the height-map narrow phase: sphere, cell range, 4 x 4 corner heights
constexpr int kHmSlots = 16;          // least number of slots of the height-map narrow phase (LdsLayout::hm_slots: one per primitive of the model)

struct DevModel {
  int nb, nq, nv, ncol, depth, cw;  // cw: compact contact-column width = 6 + depth-1 rounded up to 4
  int max_kid, fixed_base, reserved[2];   // most children of one MOVING body (the base's are counted in kid_count[0]); fixed base: body 0 never moves
  int parent[kMaxB], level[kMaxB], jtype[kMaxB];
  int anc[kMaxB * kMaxB];           // anc[b*depth + l] = ancestor of b at level l (l <= level[b]), else -1
  int kid_start[kMaxB], kid_count[kMaxB], kid_list[kMaxB];  // children of each body: kid_list[kid_start[b] .. + kid_count[b]); the base's lead the list
  // bodyf[b]: 0-2 axis, 3 jtype (int bits), 4-6 ptree, 7 mass, 8-16 rtree, 17-19 com, 20-25 inertia,
  //           26 armature, 27 damping, 28 effort, 29 q_lower, 30 q_upper
  float bodyf[kMaxB][kModelSlot];
  // fields used by the slow-path query kernel
  float axis[kMaxB][4], ptree[kMaxB][4], rtree[kMaxB][12], com[kMaxB][4], inertia[kMaxB][8];
  float mass[kMaxB], armature[kMaxB];
  int col_body[kMaxC];
  float col_pos[kMaxC][4];  // xyz, radius
};

struct LdsLayout {
  // per-block tables (floats from the start of LDS)
  int t_model, t_gain, t_parlv, t_anc, t_dir, t_col, t_kids, t_kidx, t_spair, shared_total;
  // per-env arrays (floats from the env base)
  int q, u, pt, dtg, tf, body, fact, wb, con, wc, cv, g, ginv, lam, warm;   // (the up pass's hand-over slots alias g)
  int tact;         // [nv] actuator torques of the current sub-step
  int cen, selft;   // self-collision: primitive centres [ncol][4] (may alias wc: dead before the contact columns), per-slot pair record [kcap][4]
  int gstride;
  int per_env;
  int model_pitch;  // floats between two bodies' constants in the t_model table: kModelPitch where that costs no workgroup per CU, else kModelSlot (make_layout)
};

struct StepArgs {
  const DevModel* model;
  float* gc;
  float* gv;
  const float* ptarget;
  const float* dtarget;
  const float* tauff;
  const float* kp;
  const float* kd;
  const float* lds_image;      // [L.shared_total] the per-block tables exactly as they sit in LDS (rsb_world.hip: build_lds_image): body constants,
                               // PD gains, parent / level, ancestors, slip-search brackets, collision primitives + their materials, children lists
  rsb_contact* contacts;  // [N, kmax]
  int32_t* contact_count;
  int32_t* flags;
  int32_t* iters;
  const float* heights;        // [n_maps][hm_ys][hm_xs]
  const int32_t* hm_index;     // [N] height map of each env (NULL: every env uses map 0)
  float* warm;                 // [N, kWarmRow] solver warm state, one kWarmRec-float record per CONTACT of the last integrate(): impulse (3,
                               // contact frame), friction direction (2), direction valid, collision primitive + 1 (0 = empty record); the
                               // first kmax records of a row are read and written.  NULL = every solve starts cold
  // fused control-step epilogue / prologue (rsb_control_step); all optional
  float* ptarget_store;        // p_target rows read from `ptarget` are also stored here (the world's own copy)
  const float* act;            // [N, nv-6] actions (rsb_env_step): joint targets = act_mean + act_std * act, NULL = use ptarget
  const float* act_mean;       // [nv-6]
  float act_std;
  float* obs_out;              // [N, nq + nv + 3*obs_slots]: q, u, contact force of obs_idx[slot] (last sub-step)
  const int32_t* obs_idx;      // [obs_slots] collision primitive of each force slot (NULL: slot k = primitive k)
  int obs_slots;
  int early_term;              // an env stops integrating at the sub-step in which a contact outside `allowed` is detected
  int do_reset;                // envs with a non-finite state or a contact outside `allowed` restart from gc0 / gv0
  unsigned long long allowed;  // bit c set: collision primitive c may touch the terrain
  const float* gc0;            // [reset_rows, nq], reset_rows = 1 or N
  const float* gv0;
  int reset_rows;
  float* tau2_out;             // [N] optional: |actuator torque|^2 over the joints in the last sub-step (clipped PD + feed-forward), for the env reward
  uint8_t* done_out;           // [N] optional: 1 for the envs this launch reset (do_reset), else 0 (rsb_set_done_output)
  // rsg_anymal task epilogue (rsb_env_step; all optional): reward of the state the control step ended in (forward velocity in the
  // body frame, clipped, minus the torque cost of the last sub-step; + terminal_reward for a terminated env) and the observation
  // (height, third row of the base rotation, joint angles, body-frame linear / angular velocity, joint velocities) of the state
  // the NEXT step starts from (i.e. after the reset of a terminated env)
  float* env_reward;           // [N]
  float* env_ob;               // [N, 10 + 2 (nv - 6)]
  float env_fwd_coeff, env_fwd_clip, env_torque_coeff, env_terminal_reward;
  // self-collision (rsb_set_self_collision): candidate primitive pairs (ids in the LDS image, LdsLayout::t_spair) and their materials
  int n_self;                  // candidate pairs; 0 = self-collision off
  const float* self_mat;       // [n_self][4] mu, restitution, restitution threshold, - of each pair
  const uint8_t* env_mask;     // [N] optional: envs with 0 are not integrated and none of their rows is written (rsb_integrate_masked)
  long long* prof;  // optional [16] cycle stamps (s_memtime) of block 0's phases in the last sub-step
  float* dbg;       // optional dump of env dbg_env's contact problem (nc, G, c, lam)
  int dbg_env;
  int prof_fine;    // debug: stamp the inner solver blocks too (each stamp costs ~100 cycles)
  int poison_lds;   // debug: fill the whole LDS allocation with NaNs first (catches reads of never-written LDS)
  int lds_floats;
  int N, nsub, kmax, control_mode;
  int nb, nq, nv, ncol, depth, cw, max_kid, fixed_base;   // model dimensions (DevModel's, repeated here: see the kernel's first lines)
  int chain;                   // the tree is the base + serial chains numbered consecutively (every body of level >= 2 has parent = itself - 1) and fits a 16-lane row:
                               // specialised code objects then hand a body's results to its child by a DPP row shift instead of through LDS (step_spec.h)
  float dt, gx, gy, gz, mu, erp;
  float alpha_init, alpha_min, alpha_decay, threshold;
  int max_iter, section_rounds, stall_window, freeze_after, refine;
  float stall_factor, settle_tol, restitution, res_threshold;
  int terrain_type, hm_xs, hm_ys;
  float ground_z, hm_x0, hm_y0, hm_dx, hm_dy, hm_inv_dx, hm_inv_dy, hm_max;
  LdsLayout L;
  float* tau_out;              // [N, nv] optional: the generalized force the actuators applied in the last sub-step (rsb_enable_generalized_force_output)
  // multi-contact envs (>= multi_depth contacts on one limb in this sub-step: redundant sets; rsb_set_solver_multi_contact)
  int multi_depth, multi_light, multi_freeze_after, multi_stall_window;
  // Anderson acceleration of the sweep map in multi-contact envs of the KMAX > 8 classes (rsb_set_solver_anderson): first sweep (0 = off), clip
  int anderson;
  float anderson_clip;
  // height map: contacts per primitive (2: the class-4 kernels also report the closest feature of a second flank) and the cosine of the least angle between the two normals
  int hm_contacts;
  float hm_second_cos;
  // spheres per env the height-map narrow phase can examine in one sub-step (= the model's primitives, at least kHmSlots).  Here, not in
  // LdsLayout: the layout is loaded once and lives in SGPRs for the whole kernel - one more word there costs the plane path 0.8 %
  // (same-box bisect, profiles/r03_ab_log.txt); this one is read inside the height-map branch only
  int hm_slots;
  // integration scheme of the positions (rsb_set_integration_scheme; class-8 kernels): q+ = q (+) dt (u + theta (u+ - u)); 0 = explicit Euler, 0.5 = trapezoid
  float integ_theta;
  // peer-mapped obs exchange (rsb_obs_peer_*): the epilogue stores the env's obs row (the obs_out layout) into the gathered buffer
  // of EVERY rank at row obs_row0 + env - system-scope (write-through) stores through peer-mapped pointers into fine-grained
  // memory, over xGMI for the other GPUs.  Publication without a cache flush: a wave waits for its stores to be acknowledged
  // (s_waitcnt vmcnt(0)) and checks in on a device counter; the LAST wave of the launch then stores this rank's step number into
  // every rank's flag array.  (A system-scope RELEASE per wave writes the whole L2 back: +15 % kernel time, measured; a stream
  // memory-write packet behind the launch costs 8 us, a one-wave flag kernel 3 us: profiles/r03_ab_log.txt.)
#ifdef RSB_X_SMALLARGS   /* experiment: does the kernarg segment's size matter? (peer exchange unusable in this variant) */
  float* obs_peer[1];
  uint32_t* obs_flag[1];
#else
  float* obs_peer[RSB_MAX_RANKS];        // [n_obs_peers] rank p's gathered buffer [n_ranks * N, obs_dim] of this control step's parity
  uint32_t* obs_flag[RSB_MAX_RANKS];     // [n_obs_peers] &flags_of_rank_p[parity][my rank]
#endif
  uint32_t* obs_ctr;                     // this rank's wave-arrival counter (device memory, zero between launches)
  int n_obs_peers, obs_row0;
  uint32_t obs_step;                     // value published in the flags: the control step's sequence number (>= 1)
  // exact capsule x height map (rsb_set_capsule_contacts; class-4 kernels): the cylinder between a capsule's end spheres reports its deepest point.
  // At the END of the struct: every offset the benchmark classes read stays where it was.  hm_cap: [hm_capsule][2] primitive indices of the
  // two ends of every capsule / cylinder of the model (device memory); hm_capsule = 0: off
  int hm_capsule;
  const int32_t* hm_cap;
  // pipelined control steps (rsb_set_step_pipelining; classes | 16): consecutive launches alternate between two streams and overlap on the device;
  // workgroup b of launch k + 1 starts its env block when workgroup b of launch k has published pipe_seq in pipe_prog[b] (release / acquire at
  // agent scope: the dispatcher's round-robin over the XCDs starts somewhere else in every launch, so the two workgroups sit behind different
  // L2s: profiles/r04_ubench_xcc_map.txt).  pipe_prog == nullptr: a plain launch.  (Measured alternative, not kept: a ticket queue that lets the
  // j-th workgroup to arrive continue the j-th block to finish - 1-3 % slower than the fixed assignment, profiles/r04_ab_log.txt.)
  unsigned long long* pipe_started;      // workgroups of pipelined launches that have started since the fork (the gate of the next launch waits for a full launch)
  int* pipe_prog;                        // [blocks] sequence number of the last pipelined launch whose workgroup b has finished
  int pipe_wait_on, pipe_wait, pipe_seq; // wait for pipe_prog[b] >= pipe_wait (unless !pipe_wait_on: first launch after a fork); publish pipe_seq
  int pipe_xcds;                         // > 0: XCD-affine blocks (the device's XCD count; the grid is a multiple of it): block = XCD x (grid / XCDs) + ticket
  unsigned* pipe_xcc_ctr;                // tickets taken per XCD (monotonic): XCD x's counter at [64 x] (a 256-byte line of its own)
  unsigned pipe_xcc_base;                // their value at the start of this launch
  // round 5: no device trap anywhere in the pipeline.  A workgroup that draws a ticket outside its XCD's range, or whose wait runs past
  // pipe_timeout (ticks of the 100 MHz wall clock), stores a code (rsb_pipeline.h: RSB_PIPE_ERR_*) into *pipe_err and returns WITHOUT integrating or
  // publishing; every waiting workgroup (and gate) sees the word in its spin and returns too, so the streams drain and the host's next join
  // finds the word: it restores the state of the last join and replays the steps in lock-step (rsb_world.hip: pipe_recover).
  int* pipe_err;
  int* pipe_err_host;                    // the same code for the host: page-locked host memory (a system-scope store; the join reads it without a copy)
  long long pipe_timeout;
  // closed loop (rsb_closed_loop_run): a step does not wait for its OWN predecessor's word but for the action stage's - workgroup b of step k + 1
  // starts when the stage has published block b's actions computed from step k's observation (act_prog[b] >= pipe_wait).  Open loop: == pipe_prog
  const int* pipe_wait_ptr;
  // the words of consecutive blocks sit pipe_stride ints apart: polled densely packed, all of a launch's waiting workgroups (and the action stage's
  // waves) hammer the one or two memory channels a 4-KB array maps to - agent-scope loads are served at the memory side
  int pipe_stride;
  // diagnostics (RSB_PIPE_STATS=1): [0] += wall-clock ticks (100 MHz) this workgroup spent waiting for its block, [1] += 1 per workgroup that had to wait at all
  unsigned long long* pipe_stats;
  // ---- round 6: RESIDENT launches (classes | 64; rsb_set_step_residency).  ONE launch runs res_steps control steps: an env block's state stays in
  // LDS from the first sub-step to the last, per control step only the obs block / reward / done flags / next observation go to HBM (state rows,
  // warm records and contact records after the LAST control step, or after every one with res_full), a terminated env is reset in LDS.  A wave's time
  // over the launch is a SUM over control steps, so the slowest-wave tail of the lock-step launches averages out without any cross-launch hand-over.
  // Open loop (class | 64): control step j reads its PD-target rows from slice (res_first + j) % res_period of res_targets.
  // Closed loop (| 128 linear policy, | 256 actor network of widths <= 128, | 384 <= 256): the env block's own wave evaluates the action stage between
  // two control steps (stage_bodies.h: the SAME body the stage kernels of rsb_pipeline.hip run, so resident == pipelined == lock-step bit for bit).
  // At the END of the struct: every offset the other classes read stays where it was.
  int res_steps;                         // control steps of this launch (0 / 1 in every other class)
  int res_full;                          // 1: every control step writes what a lock-step launch writes
  const float* res_targets;              // [res_period][N][nq]
  int res_period;
  long long res_first;
  long long res_obs_stride;              // floats between the obs_out blocks of consecutive control steps (0: each step overwrites obs_out)
  long long res_done_stride;             // bytes between their done_out rows (0: each step overwrites done_out)
  long long res_pass_global0;            // closed loop: passes served by earlier runs of this world (rsb_stage_ctx::pass_global0)
  union ResPolicy { rsb_linear_policy lin; rsb_mlp_policy mlp; } res_pol;
#ifdef RSB_X_ARGPAD
  char x_pad[RSB_X_ARGPAD];
#endif
};

}  // namespace rsbk
