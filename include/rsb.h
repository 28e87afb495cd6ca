/*
 * rsb.h — C-ABI of the MI355X-native batched rigid-body simulator ("rsb" = RaiSim-batched).
 *
 * This is the drop-in boundary for the hot path named by BASELINE.json's north_star:
 * thousands of independent raisim::World replicas, each holding one raisim::ArticulatedSystem
 * on a Ground / HeightMap, stepped in lock-step on one GPU.
 *
 * Reference interface replaced.  The mounted reference /root/reference is a
 * three-file stub (.gitignore:1-12, .travis.yml:1-12, README.md:1) that contains NO headers, so
 * every "replaces" note below names the upstream raisimLib symbol from recollection [RECALL] and
 * the file the symbol would live in; the file:line is "absent" for all of them
 * (SURVEY.md §0, §8a).
 *
 *   rsb_create / rsb_destroy            <- raisim::World::World(), ~World(),
 *                                          World::addArticulatedSystem(urdf)      (World.hpp, absent)
 *   rsb_set_ground / rsb_set_heightmap  <- World::addGround(z), World::addHeightMap(...)
 *   rsb_set_timestep / rsb_set_gravity  <- World::setTimeStep, World::setGravity
 *   rsb_set_erp / rsb_set_contact_solver_param / rsb_set_friction
 *                                       <- World::setERP, World::setContactSolverParam,
 *                                          World::setDefaultMaterial               (World.hpp, absent)
 *   rsb_set_state / rsb_get_state       <- ArticulatedSystem::setState/getState    (ArticulatedSystem.hpp, absent)
 *   rsb_set_pd_gains / rsb_set_pd_target / rsb_set_generalized_force / rsb_set_control_mode
 *                                       <- ArticulatedSystem::setPdGains/setPdTarget/
 *                                          setGeneralizedForce/setControlMode
 *   rsb_integrate / rsb_integrate1 / rsb_integrate2
 *                                       <- World::integrate(), integrate1(), integrate2()
 *   rsb_get_contacts                    <- ArticulatedSystem::getContacts() (contact/Contact.hpp, absent)
 *   rsb_get_mass_matrix / rsb_get_nonlinearities
 *                                       <- ArticulatedSystem::getMassMatrix()/getNonlinearities()
 *   rsb_gather_obs                      <- (new) the (q, u, contact-force) observation block that
 *                                          VectorizedEnvironment::observe() is built from
 *
 * Conventions
 *   - No exceptions cross this ABI. Every call returns RSB_OK (0) or a negative rsb_status;
 *     rsb_last_error() returns a thread-local message for the last failure.
 *   - Buffers are caller-owned.  `space` says whether a pointer is host or device memory.
 *   - Batched arrays are row-major [num_envs, dim] float32 (the layout raisimGymTorch's
 *     VectorizedEnvironment uses for observation/action matrices).
 *   - Generalized coordinates follow RaiSim: gc = [x y z  qw qx qy qz  joints...] (nq = 7+nj),
 *     gv = [world linear vel (3), world angular vel (3), joint vels...] (nv = 6+nj).
 *   - One handle owns one HIP stream (or borrows the caller's, rsb_set_stream). A handle is not
 *     re-entrant; distinct handles may be used from distinct threads.
 *   - There is NO CPU fallback: rsb_create fails with RSB_E_NO_DEVICE when no HIP device exists.
 */
#ifndef RSB_H_
#define RSB_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#include "rsb_types.h"   /* constants, enums, rsb_model_blob, rsb_contact, rsb_field: what the device code shares with this header */

typedef struct rsb_model rsb_model;  /* host-side parsed model           */
typedef struct rsb_world rsb_world;  /* batched device world             */

const char* rsb_last_error(void);
const char* rsb_version(void);
/* build provenance: hash of the sources (the files of raisimlib_amd/csrc, this header, compiler flags) the loaded library was built from,
 * as raisimlib_amd/build.py: source_hash() computes it for the tree (the test-suite refuses a library that does not match its tree) */
const char* rsb_source_hash(void);

/* ---- model (host only; cold path) -------------------------------------- */
int rsb_model_from_urdf_file(const char* path, rsb_model** out);
int rsb_model_from_urdf_string(const char* xml, rsb_model** out);
/* Sampled colliders (no upstream counterpart; opt-in).  The loader turns a capsule into its two end spheres and a box into its eight
 * corners: the exact contact sets on a plane.  With sample_spacing h > 0 a capsule also gets spheres of its radius along its axis and
 * a box zero-radius points on the lattice of its edges and faces, no further apart than h, so that a height-field feature under the
 * MIDDLE of a capsule or of a box face, and a capsule touching another link with its middle, are found to within h by the same sphere
 * tests (the narrow phase of the kernel is unchanged).  Costs primitives (at most RSB_MAX_COLLISIONS per model) and adds redundant
 * contacts where the unsampled set was already exact.  h = 0: the functions above. */
int rsb_model_from_urdf_file_sampled(const char* path, double sample_spacing, rsb_model** out);
int rsb_model_from_urdf_string_sampled(const char* xml, double sample_spacing, rsb_model** out);
/* <mesh> colliders (upstream: the mesh itself against the terrain, through ODE's trimesh collider [RECALL; absent]) are loaded as a POINT SET: support
 * vertices of the mesh's convex hull, each a zero-radius primitive - exact against a plane for a convex mesh whose lowest vertices are among them.
 * n = points kept per mesh: default 8 (the body diagonals' support vertices: a box keeps its corners), at most 26 (+ the 6 axes' and the 12 face
 * diagonals'); they count against RSB_MAX_COLLISIONS.  Process-wide, read when a URDF is loaded. */
int rsb_set_mesh_point_budget(int n);
int rsb_model_from_blob(const rsb_model_blob* blob, rsb_model** out);
int rsb_model_destroy(rsb_model* m);
int rsb_model_get_blob(const rsb_model* m, rsb_model_blob* out);
int rsb_model_body_index(const rsb_model* m, const char* link_name);   /* <0 if absent */
int rsb_model_joint_index(const rsb_model* m, const char* joint_name); /* body index driven by joint */
double rsb_model_total_mass(const rsb_model* m);
int rsb_model_skipped_collisions(const rsb_model* m);   /* <collision> elements the loader could not turn into primitives (unreadable meshes, unknown geometry) */
const char* rsb_model_collision_material(const rsb_model* m, int collision);   /* material name of a collision primitive ("default" if the URDF names none) */

/* ---- world ------------------------------------------------------------------------------ */
int rsb_device_count(void);
int rsb_create(const rsb_model* m, int num_envs, int device, rsb_world** out);
int rsb_destroy(rsb_world* w);
int rsb_set_stream(rsb_world* w, void* hip_stream);   /* borrow the caller's hipStream_t (NULL = HIP default stream) */
void* rsb_get_stream(rsb_world* w);
int rsb_synchronize(rsb_world* w);

int rsb_num_envs(const rsb_world* w);
int rsb_dims(const rsb_world* w, int* nb, int* nq, int* nv, int* ncol, int* kmax);

int rsb_set_timestep(rsb_world* w, double dt);
double rsb_get_timestep(const rsb_world* w);
double rsb_get_world_time(const rsb_world* w);
int rsb_set_gravity(rsb_world* w, const double g[3]);
int rsb_set_erp(rsb_world* w, double erp);
int rsb_set_friction(rsb_world* w, double mu);
/* World::setDefaultMaterial(friction, restitution, resThreshold) [RECALL]: one material per world; a contact approaching
 * faster than res_threshold (m/s) leaves with v_n+ = -restitution * v_n- (Newton restitution), slower ones are inelastic */
int rsb_set_material(rsb_world* w, double mu, double restitution, double res_threshold);
/* World::setMaterialPairProp(material1, material2, friction, restitution, resThreshold) [RECALL; upstream Materials.hpp is
 * absent from /root/reference]: every env holds ONE terrain object, so the pair table collapses to one (mu, restitution,
 * res_threshold) triple per collision primitive of the robot = the pair (primitive's material, terrain's material).
 * Arrays of rsb_dims().ncol doubles; a NULL array (or a negative entry) means "the world's default" (rsb_set_material).
 * The C++ facade resolves material NAMES (rsb_model_collision_material, addGround(z, material)) into these arrays. */
int rsb_set_collision_materials(rsb_world* w, const double* mu, const double* restitution, const double* res_threshold);
/* Self-collision: RaiSim collides the links of one articulated system with each other, parent-child pairs excepted [RECALL;
 * upstream ArticulatedSystem.hpp is absent from /root/reference].  Here: sphere x sphere between collision primitives of two
 * bodies that are not parent and child (capsule = its two end spheres, box = its corner points against spheres; rim
 * primitives of cylinders take no part).  On by default; a self-collision takes two of the max_contacts slots.
 * rsb_ignore_collision_between = ArticulatedSystem::ignoreCollisionBetween(bodyIdx1, bodyIdx2) (not undoable).
 * rsb_self_collision_pairs lists the candidate primitive pairs (pairs[2k] < pairs[2k+1]) and returns their count;
 * rsb_set_self_collision_materials takes one (mu, restitution, res_threshold) per candidate pair in that order
 * (NULL array / negative entry = the world's default; reset when the candidate list changes). */
int rsb_set_self_collision(rsb_world* w, int enable);
int rsb_ignore_collision_between(rsb_world* w, int body_a, int body_b);
int rsb_self_collision_pairs(const rsb_world* w, int32_t* pairs, int capacity);
int rsb_set_self_collision_materials(rsb_world* w, const double* mu, const double* restitution, const double* res_threshold);
int rsb_set_contact_solver_param(rsb_world* w, double alpha_init, double alpha_min,
                                 double alpha_decay, int max_iter, double threshold);
/* ArticulatedSystem::setIntegrationScheme [RECALL; upstream file absent].  The velocity update is the same for every scheme (one
 * dynamics evaluation, one contact solve: u+ = u + M^-1 (dt tau + J^T lambda)); the scheme picks the velocity the positions move with:
 *   RSB_INTEGRATION_SEMI_IMPLICIT (default, RaiSim's)  q+ = q (+) dt u+
 *   RSB_INTEGRATION_EULER                               q+ = q (+) dt u
 *   RSB_INTEGRATION_TRAPEZOID                           q+ = q (+) dt (u + u+) / 2     (exact positions under a constant acceleration)
 *   RSB_INTEGRATION_RUNGE_KUTTA_4                       the classical four-stage scheme on the smooth equations of motion (explicit PD at the stage
 *                                                       states, the base orientation advanced on SO(3)), contacts and joint limits on top of it as in the
 *                                                       other schemes: one detection at q, one solve (rsb_rk4.hip states the construction).  Host-driven
 *                                                       over the query kernels - four dynamics evaluations with a dense M^-1 per integrate(): the slow,
 *                                                       accurate path; plain rsb_integrate / rsb_integrate_masked / World-view calls only
 *                                                       (RSB_E_UNSUPPORTED from rsb_control_step / rsb_env_step)
 * (the enum values are raisim::IntegrationScheme's [RECALL]).  EULER and TRAPEZOID run in a kernel class of their own (floating-base systems
 * of tree depth <= 13, no peer-mapped obs exchange, one contact per primitive; RSB_E_UNSUPPORTED from the step otherwise): the default's
 * kernels do not carry the choice. */
#define RSB_INTEGRATION_TRAPEZOID 0
#define RSB_INTEGRATION_SEMI_IMPLICIT 1
#define RSB_INTEGRATION_EULER 2
#define RSB_INTEGRATION_RUNGE_KUTTA_4 3
int rsb_set_integration_scheme(rsb_world* w, int scheme);
int rsb_set_max_contacts(rsb_world* w, int kmax);   /* 1..RSB_MAX_CONTACTS */

int rsb_set_ground(rsb_world* w, double height);
/* heights: host pointer, row-major [y_samples][x_samples] (x fastest), shared by all envs */
int rsb_set_heightmap(rsb_world* w, int x_samples, int y_samples, double x_size, double y_size,
                      double center_x, double center_y, const float* heights);
/* terrain curricula: n_maps height maps of one geometry, heights [n_maps][y_samples][x_samples] (host), and the map
 * each env stands on, env_map [num_envs] (host; may be NULL when n_maps == 1) */
int rsb_set_heightmaps(rsb_world* w, int n_maps, int x_samples, int y_samples, double x_size, double y_size,
                       double center_x, double center_y, const float* heights, const int32_t* env_map);

/* ---- height-map sources (host side, no GPU needed): fill a [y_samples][x_samples] float buffer for rsb_set_heightmap.
 * Upstream counterparts [RECALL, absent]: World::addHeightMap(pngFile, centerX, centerY, xSize, ySize, heightScale,
 * heightOffset), World::addHeightMap(centerX, centerY, TerrainProperties&), World::addHeightMap(textFile, ...).
 * PNG: non-interlaced 8/16-bit grey / grey+alpha / RGB / RGBA (first channel); height = pixel / max_pixel *
 * height_scale + height_offset; image row r, column c -> sample (y = r, x = c).
 * Perlin: fractal sum of improved Perlin noise (this repo's own permutation from `seed`: terrains are reproducible
 * here, not bit-identical to RaiSim's), h = z_scale * sum_o gain^o * noise(f lacunarity^o * (x, y)), optionally
 * rounded to multiples of step_size, + height_offset.
 * Text: "xSamples ySamples xSize ySize" followed by xSamples*ySamples heights (x fastest). */
typedef struct rsb_terrain_properties {   /* field meaning of raisim::TerrainProperties [RECALL] */
  double frequency, z_scale, x_size, y_size;
  int32_t x_samples, y_samples, fractal_octaves;
  uint32_t seed;
  double fractal_lacunarity, fractal_gain, step_size, height_offset;
} rsb_terrain_properties;
int rsb_heightmap_png_size(const char* path, int* x_samples, int* y_samples);
int rsb_heightmap_png_read(const char* path, double height_scale, double height_offset, float* heights, int n);
int rsb_heightmap_perlin(const rsb_terrain_properties* tp, float* heights);
int rsb_heightmap_text_size(const char* path, int* x_samples, int* y_samples, double* x_size, double* y_size);
int rsb_heightmap_text_read(const char* path, float* heights, int n);

/* mask: optional uint8 [num_envs] (same memspace); envs with mask==0 are left untouched */
int rsb_set_state(rsb_world* w, const float* gc, const float* gv, const uint8_t* mask, int space);
int rsb_get_state(rsb_world* w, float* gc, float* gv, int space);

/* one env's row of a state field (slow path behind the per-env raisim::ArticulatedSystem views):
 * field = RSB_F_GC / RSB_F_GV / RSB_F_PTARGET / RSB_F_DTARGET / RSB_F_TAU_FF; data is a host pointer of dim floats */
int rsb_set_env_row(rsb_world* w, int field, int env, const float* data);
int rsb_get_env_row(rsb_world* w, int field, int env, float* data);

/* a whole state field at once: field = RSB_F_GC / RSB_F_GV / RSB_F_PTARGET / RSB_F_DTARGET / RSB_F_TAU_FF, out [N, dim] float32 */
int rsb_get_field(rsb_world* w, int field, float* out, int space);
/* ArticulatedSystem::getGeneralizedForce() [RECALL; upstream ArticulatedSystem.hpp is absent from /root/reference]: the generalized
 * force the actuators applied in the last sub-step of the last integrate() - clipped PD + feed-forward on the joints (without the
 * joints' passive damping), the feed-forward wrench on the base rows.  Off by default (one more [N, nv] row written per launch);
 * once enabled it is read as field RSB_F_GENERALIZED_FORCE with rsb_get_field / rsb_get_env_row. */
int rsb_enable_generalized_force_output(rsb_world* w, int on);

int rsb_set_control_mode(rsb_world* w, int mode);
int rsb_set_pd_gains(rsb_world* w, const float* kp, const float* kd);      /* host, [nv] each  */
int rsb_set_pd_target(rsb_world* w, const float* p_target, const float* d_target, int space); /* [N,nq],[N,nv]; either may be NULL */
int rsb_set_generalized_force(rsb_world* w, const float* tau, int space);  /* [N,nv] feed-forward */

int rsb_integrate(rsb_world* w, int n_substeps);
int rsb_integrate1(rsb_world* w);
int rsb_integrate2(rsb_world* w);
/* World::integrate() of a SUBSET of the replicas: envs whose mask byte is 0 are not integrated and none of their rows
 * (state, contacts, flags, warm state) is touched.  mask: uint8 [num_envs] in `space` (a host mask is staged to the
 * device first).  This is what the per-env raisim::World views (include/raisim/World.hpp) flush through, so that N
 * views calling integrate() cost one launch in which every env advances exactly once. */
int rsb_integrate_masked(rsb_world* w, int n_substeps, const uint8_t* mask, int space);

/* One flush of the per-env raisim::World views (include/raisim/World.hpp) in ONE call and ONE stream synchronisation:
 * staged uploads -> launch -> the downloads the environments read.  What upstream's VectorizedEnvironment<ENV>::step pays per
 * World::integrate() is nothing (its worlds live on the host); here every crossing of PCIe costs a latency, so the facade
 * bundles them: round 3 paid up to five synchronous copies per integrate().  All pointers are HOST pointers (page-locked memory
 * from rsb_host_alloc makes the copies asynchronous up to the final synchronisation); a NULL pointer skips that transfer.
 *   uploads   p_target [N,nq], d_target [N,nv], tau_ff [N,nv]; gc / gv [N,nq] / [N,nv] with state_mask [N] (rows with mask 0 stay,
 *             rows with mask 1 are overwritten and their solver warm state cleared: rsb_set_state's semantics)
 *   launches  n_launches entries: launch_substeps[i] sub-steps for the envs whose launch_masks[i * N + env] != 0
 *             (launch_masks NULL = every env in every launch)
 *   downloads gc_out, gv_out, contact_counts [N], contacts [N,kmax], generalized_force [N,nv] (needs
 *             rsb_enable_generalized_force_output) */
typedef struct rsb_view_io {
  const float* p_target; const float* d_target; const float* tau_ff;
  const float* gc; const float* gv; const uint8_t* state_mask;
  int32_t n_launches;
  const int32_t* launch_substeps;
  const uint8_t* launch_masks;
  float* gc_out; float* gv_out;
  int32_t* contact_counts; rsb_contact* contacts;
  float* generalized_force;
} rsb_view_io;
int rsb_view_exchange(rsb_world* w, const rsb_view_io* io);
/* page-locked host memory for the buffers of rsb_view_exchange (and any other RSB_HOST argument) */
int rsb_host_alloc(size_t bytes, void** out);
int rsb_host_free(void* p);
/* device memory for a C / C++ caller that does not link the HIP runtime itself (a policy's weights for rsb_closed_loop_run_linear, action
 * buffers for RSB_DEVICE arguments): allocation on the world's device, synchronous copies (kind: 0 = host -> device, 1 = device -> host),
 * ordered behind everything the world has enqueued (they join a step pipeline like every other call) */
int rsb_device_alloc(rsb_world* w, size_t bytes, void** out);
int rsb_device_free(rsb_world* w, void* p);
int rsb_device_copy(rsb_world* w, void* dst, const void* src, size_t bytes, int kind);

/* contacts of the last sub-step: counts [N] int32, contacts [N,kmax] rsb_contact */
int rsb_get_contacts(rsb_world* w, int32_t* counts, rsb_contact* contacts, int space);
/* valid after rsb_integrate1: M [N,nv,nv], h [N,nv] */
int rsb_get_mass_matrix(rsb_world* w, float* M, int space);
int rsb_get_nonlinearities(rsb_world* w, float* h, int space);
/* ArticulatedSystem::getInverseMassMatrix() [RECALL]: M(q)^-1 [N, nv, nv] of the state rsb_integrate1 saw (slow path) */
int rsb_get_inverse_mass_matrix(rsb_world* w, float* Minv, int space);
/* per-env status flags of the last launch (bit0: contact overflow, bit1: non-finite state, bit2: the contact
 * solver of the last sub-step stopped without meeting the convergence test: max_iter or stagnation exit) */
int rsb_get_flags(rsb_world* w, int32_t* flags, int space);
/* iterations the contact solver used in the last sub-step, [N] int32 */
int rsb_get_solver_iterations(rsb_world* w, int32_t* iters, int space);

/* obs block [N, nq+nv+3*n_force_slots] = (q, u, contact force on chosen collision primitives).
 * collision_indices: host array of n_force_slots collision-primitive indices (e.g. the feet);
 * NULL = primitives 0..n_force_slots-1.  Force = impulse / dt of the last sub-step (world frame). */
int rsb_obs_dim(const rsb_world* w, int n_force_slots);
int rsb_gather_obs(rsb_world* w, float* out, const int32_t* collision_indices, int n_force_slots, int space);

/* VectorizedEnvironment support (raisimGymTorch's isTerminalState()+reset() fan-out [RECALL]), on device:
 * every env whose last sub-step holds a contact on a collision primitive NOT listed in
 * allowed_collisions (host array; e.g. the feet -> "terminate on any non-foot contact"), or whose state is
 * non-finite, is reset to row e (rows == N) or row 0 (rows == 1) of gc0/gv0.  done (uint8 [N], same
 * memspace as gc0/gv0, may be NULL) receives 1 for the envs that were reset, 0 otherwise. */
int rsb_reset_terminated(rsb_world* w, const int32_t* allowed_collisions, int n_allowed, const float* gc0,
                         const float* gv0, int rows, uint8_t* done, int space);

/* One control step of VectorizedEnvironment::step() [RECALL raisimGymTorch/env/VectorizedEnvironment.hpp] enqueued
 * with a single call on the handle's stream: rsb_set_pd_target (device pointers, either may be NULL) ->
 * rsb_integrate(n_substeps) -> rsb_gather_obs (skipped when obs_out is NULL) -> rsb_reset_terminated (skipped
 * when gc0 or gv0 is NULL).  Index lists are host arrays, everything else lives on the device. */
int rsb_control_step(rsb_world* w, const float* p_target, const float* d_target, int n_substeps, float* obs_out,
                     const int32_t* force_collisions, int n_force_slots, const int32_t* allowed_collisions,
                     int n_allowed, const float* gc0, const float* gv0, int rows);

/* ---- round 6: RESIDENT control steps.  K control steps of a vectorised env in ONE launch of the step kernel: an env block's state (and the solver's
 * warm table, the model tables) stays in LDS from the first sub-step to the last; per control step only the obs block, the done flags and - env task -
 * reward / next observation go to HBM; state rows, warm records and contact records are written after the LAST control step (the world then holds exactly
 * what K separate rsb_control_step calls leave: the tests compare bit for bit).  A terminated env restarts in LDS.  Why: a launch lasts as long as its
 * slowest wave, and a wave's time over K control steps is a SUM - the tail averages out with no hand-over between launches, and the per-launch
 * prologue / epilogue (8 k of a launch's 180 k cycles) is paid once.  Upstream counterpart: the loop over VectorizedEnvironment::step
 * [RECALL raisimGymTorch/env/VectorizedEnvironment.hpp; absent from /root/reference].
 *
 * rsb_control_steps: K control steps of the OPEN loop.  p_targets [period][N][nq] device memory: control step j reads slice (first + j) % period (the
 * benchmark's pre-drawn PD-target bank; an action sequence of sampling-based MPC; a replay).  obs_out (may be NULL): control step j's obs block goes to
 * obs_out + j * obs_step_stride floats (0: every step overwrites the same block; N * rsb_obs_dim: a [K, N, obs_dim] rollout); done_out (may be NULL, device
 * memory): its done flags to done_out + j * done_step_stride bytes.  The other arguments are rsb_control_step's.
 * With residency OFF (the default), or for a world outside the resident kernel classes (see rsb_step_residency_status), the same call runs K
 * rsb_control_step launches - pipelined or in lock-step as rsb_set_step_pipelining says - with the same results. */
int rsb_control_steps(rsb_world* w, int n_steps, const float* p_targets, int period, long long first, int n_substeps, float* obs_out,
                      long long obs_step_stride, const int32_t* force_collisions, int n_force_slots, const int32_t* allowed_collisions, int n_allowed,
                      const float* gc0, const float* gv0, int rows, uint8_t* done_out, long long done_step_stride);
/* ---- multi-GPU without Python: the obs all-gather over RCCL / xGMI.  One process per GPU; envs are sharded
 * contiguously, rank r owns global envs [r*N, (r+1)*N); nothing inside integrate() communicates.  librccl.so.1 is loaded
 * at run time by the first rsb_comm_* call (a host that never calls them needs no RCCL).  Upstream has no counterpart
 * (RaiSim is single-process); this is the C/C++ twin of raisimlib_amd/dist.py.
 *   rsb_comm_get_unique_id : rank 0 creates the id (ncclGetUniqueId), the launcher hands it to the other ranks
 *   rsb_comm_init          : ncclCommInitRank on the world's device (collective: every rank calls it)
 *   rsb_allgather_obs      : the rank's obs block (rsb_gather_obs semantics) -> out [n_ranks*N, obs_dim], rank-major, on the
 *                            handle's stream; out in `space` (DEVICE: gathered in place, nothing synchronises; HOST: staged) */
/* RSB_MAX_RANKS (ranks of one node; peer-mapped obs exchange below): rsb_types.h */
#define RSB_COMM_ID_BYTES 128
/* versions of RCCL as NCCL_VERSION_CODE: of the librccl.so.1 loaded at run time (ncclGetVersion) and of the <rccl/rccl.h> this library's constants were checked
 * against when it was built (0: the header was not installed on the build host).  Either pointer may be NULL. */
int rsb_comm_rccl_version(int* runtime_version, int* header_version);
int rsb_comm_get_unique_id(char id[RSB_COMM_ID_BYTES]);
int rsb_comm_init(rsb_world* w, int n_ranks, int rank, const char id[RSB_COMM_ID_BYTES]);
int rsb_comm_destroy(rsb_world* w);
int rsb_allgather_obs(rsb_world* w, const int32_t* collision_indices, int n_force_slots, float* out, int space);

/* ---- the same exchange WITHOUT a collective or a copy kernel: peer-mapped gathered buffers.  The step kernel fills every CU
 * of the chip, so a collective's copy kernel cannot overlap it and costs a kernel slot per control step (measured: 8 % at one
 * rank).  Here every rank owns a gathered buffer [n_ranks * N, obs_dim] (double-buffered by control-step parity) that the other
 * ranks map - hipIpc handles across processes, plain pointers within one process -, and the epilogue of rsb_control_step's ONE
 * launch stores each env's obs row into the buffer of every rank (write-through stores; the other GPUs' over xGMI); the last wave
 * of the launch to finish writes the step number into every rank's flag array, and rsb_obs_peer_wait makes the stream wait
 * (command-processor poll of the flag words, no kernel) until every rank has delivered the rows of the last control step issued.
 *   rsb_obs_peer_create       allocate this rank's buffer (fine-grained device memory); `handle` (may be NULL) receives its IPC handle
 *   rsb_obs_peer_connect      handles [n_ranks][RSB_OBS_HANDLE_BYTES] of all ranks (own entry ignored), e.g. from an all-gather of the launcher
 *   rsb_obs_peer_connect_ptrs the same within ONE process: base pointers (rsb_obs_peer_base) of the other worlds, peer access enabled by the caller
 *   rsb_obs_peer_wait         see above; *gathered (may be NULL) receives the device pointer of the complete block of that step, rank-major
 * From rsb_obs_peer_connect on, every rsb_control_step of the world runs the exchange (obs_out may be NULL).  Floating-base models
 * of tree depth <= 13.  RCCL (above) stays the default of bench.py until a multi-GPU box has measured both. */
#define RSB_OBS_HANDLE_BYTES 64
int rsb_obs_peer_create(rsb_world* w, int n_ranks, int rank, const int32_t* collision_indices, int n_force_slots, char handle[RSB_OBS_HANDLE_BYTES]);
int rsb_obs_peer_connect(rsb_world* w, const char* handles);
int rsb_obs_peer_connect_ptrs(rsb_world* w, void* const* bases);
void* rsb_obs_peer_base(rsb_world* w);
int rsb_obs_peer_wait(rsb_world* w, float** gathered);
int rsb_obs_peer_destroy(rsb_world* w);

/* done flags of the fused control step: when `done_device` (uint8 [num_envs], DEVICE memory, caller-owned) is set,
 * every following rsb_control_step writes 1 for the envs it reset and 0 for the others (NULL switches it off). */
int rsb_set_done_output(rsb_world* w, uint8_t* done_device);

/* ---- device-resident vectorised env: the per-env observe / step / reward / terminate / reset of
 * VectorizedEnvironment<ENVIRONMENT> with rsg_anymal's task [RECALL raisimGymTorch/env/envs/rsg_anymal/Environment.hpp,
 * absent from /root/reference], computed on the GPU so that a learner whose policy runs on the same device never
 * crosses PCIe:
 *   action [N, nv-6]   -> PD position targets  action_mean + action_std * action  on the actuated joints
 *   observation [N, 10 + 2(nv-6)] = height, third ROW of the base rotation matrix (the world z-axis expressed in the
 *                        body frame: rsg_anymal's rot.e().row(2)), joint angles, body-frame linear velocity (3),
 *                        body-frame angular velocity (3), joint velocities
 *   reward = forward_vel_coeff * min(forward_vel_clip, body-frame v_x) + torque_coeff * |tau|^2, tau = the actuator
 *            torque the last sub-step applied (PD + feed-forward after the effort clip; upstream reads
 *            getGeneralizedForce() after the last integrate())
 *   done   = a contact on a primitive outside foot_collisions, or a non-finite state; such envs get
 *            reward += terminal_reward (upstream perAgentStep) and restart from gc_init / gv_init. */
typedef struct rsb_env_config {
  int32_t n_substeps;            /* control_dt / simulation_dt */
  float action_std;
  float forward_vel_coeff, forward_vel_clip, torque_coeff, terminal_reward;
  int32_t n_foot;
  int32_t foot_collisions[RSB_MAX_COLLISIONS];
} rsb_env_config;
/* action_mean [nv-6], gc_init [nq], gv_init [nv]: host arrays (copied) */
int rsb_env_configure(rsb_world* w, const rsb_env_config* cfg, const float* action_mean, const float* gc_init,
                      const float* gv_init);
int rsb_env_dims(const rsb_world* w, int* ob_dim, int* action_dim);
/* Per-env reset states (optional; gc0 [N, nq], gv0 [N, nv] in `space`, copied): a terminated env - and rsb_env_reset - restarts from ITS row
 * instead of the one gc_init / gv_init of rsb_env_configure (the benchmark's per-env base position and heading; raisimGymTorch environments
 * that randomise their initial state in reset() [RECALL]).  NULL, NULL: back to the single initial state. */
int rsb_env_set_reset_states(rsb_world* w, const float* gc0, const float* gv0, int space);
int rsb_env_reset(rsb_world* w);                                   /* every env to gc_init / gv_init */
int rsb_env_observe(rsb_world* w, float* ob, int space);           /* [N, ob_dim] */
/* action [N, action_dim] in; reward [N] float, done [N] uint8 and ob_next [N, ob_dim] (the observation the next step
 * starts from, i.e. after the resets) out, any of which may be NULL; all in `space`.  Two launches: the step kernel
 * (action -> PD targets in its prologue) and one reward / termination / reset / observation kernel. */
int rsb_env_step(rsb_world* w, const float* action, float* reward, uint8_t* done, float* ob_next, int space);

/* zero-copy access to the resident state (device pointers; row-major [N,dim] float32): see rsb_field */
void* rsb_device_ptr(rsb_world* w, int field);

#ifdef __cplusplus
}
#endif

#include "rsb_ext.h"        /* what has no upstream counterpart: solver heuristics, launch scheduling (pipelining, residency), timing, debug aids */
#include "rsb_pipeline.h"   /* the closed-loop pipeline (C declarations; under hipcc also the device-side serve loop of an action stage) */

#endif /* RSB_H_ */
