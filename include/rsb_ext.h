/*
 * rsb_ext.h — the part of the C-ABI that has NO upstream counterpart: knobs of this implementation, not of raisim::World.
 *   solver heuristics      stagnation exit, lagged friction directions, multi-contact settings, Anderson step, slip rule, warm start, early termination
 *   collider options       second flank / exact capsule and box contacts against a height map (kernel classes of their own)
 *   kernel mapping         lanes per env
 *   launch scheduling      pipelined control steps (rsb_set_step_pipelining ...), resident launches (rsb_set_step_residency ...; rsb_control_steps itself is in rsb.h)
 *   timing and debug aids  HIP-event brackets of the step launches, phase / wave cycle stamps, the contact-problem dump, host profile of rsb_view_exchange
 * Split out of rsb.h in round 6 ("~40 of the 127 entry points are solver / scheduling knobs"); rsb.h includes this file, so
 * `#include "rsb.h"` still declares everything.  What mirrors an upstream symbol stays in rsb.h, with the symbol it replaces.
 */
#ifndef RSB_EXT_H_
#define RSB_EXT_H_

#include "rsb.h"   /* (rsb.h includes this file behind its own declarations: either order of inclusion works) */

#ifdef __cplusplus
extern "C" {
#endif

/* Stagnation exit of the contact solver (not a RaiSim parameter): the Gauss-Seidel loop of an env stops when the
 * best relative error of the last `window` sweeps is not below `factor` x the best of the previous window
 * (defaults 4, 0.5; window = 0 disables it and only max_iter caps non-converging solves). */
int rsb_set_solver_stagnation_exit(rsb_world* w, int window, double factor);
/* Lagged friction directions (not a RaiSim parameter): from sweep `freeze_after` on, a slipping contact keeps the
 * friction direction of its last slip solve and only re-solves the impulse magnitude (default 6; 0 = always
 * re-optimise the direction).  Solves that converge within freeze_after sweeps are unaffected.
 * refine != 0 (default): before that, a contact that already slipped in this solve updates its direction by one
 * guarded Newton step on the curve's energy instead of a new global search (falls back to the search when the
 * step is not a safe descent step).
 * settle_tol (default 0 = never): a refinement that moved the direction by less than this (rad) marks it settled; settled
 * directions are kept like lagged ones for the rest of the solve.  1e-4 saved 19 % of the refinements in round 1 but put
 * the p99.9 deviation from the plain per-contact iteration at 1.9e-4 m/s instead of 7e-6, so it is off by default. */
int rsb_set_solver_friction_lag(rsb_world* w, int freeze_after, int refine, double settle_tol);
/* Redundant contact sets (not a RaiSim parameter).  An env that holds >= `depth` contacts on one limb in the current sub-step
 * (the four spheres of a humanoid's foot, a quadruped on its belly) is a "multi-contact" env: the per-contact iteration
 * converges linearly and slowly there, and the two accelerations above - tuned on the quadruped's usual one or two contacts
 * per limb - cut it short (measured on the humanoid's standing population against the natural-map residual of the returned
 * impulses: lagged directions make 10 % of the "converged" solves wrong by > 5e-3 relative, the 4-sweep stagnation window
 * stops 8 % of them early; tests/test_oracle_solver_heuristics.py).  Such envs run with their own settings:
 *   light_passes  != 0: friction directions of ALL contacts refreshed in the first pass of a sweep only (rounds 1-2;
 *                 a third of the work per pass, 9 % of the standing solves unconverged after 150 sweeps); 0 (default): every
 *                 pass refreshes its members' directions, as in every other env
 *   freeze_after  sweeps before directions lag in such envs (default 0 = never)
 *   stall_window  stagnation window in such envs (default 16; 0 = only max_iter caps the solve)
 * depth: default 3; 2 also covers a foot standing on one of its edges (what the humanoid benchmark uses); 0 = no distinction
 * (every env uses rsb_set_solver_friction_lag / rsb_set_solver_stagnation_exit). */
int rsb_set_solver_multi_contact(rsb_world* w, int depth, int light_passes, int freeze_after, int stall_window);
/* Anderson acceleration of the sweep (not RaiSim behaviour; default ON, first_sweep 2, clip 20) in multi-contact envs of worlds
 * with more than 8 contact slots (rsb_set_max_contacts > 8: the large-model kernel classes; the quadruped's classes do not carry
 * it - their envs converge in 3-4 sweeps).  A sweep is a fixed-point map g of the impulses; on redundant contact sets the
 * iteration crawls along one dominant mode.  From sweep `first_sweep` on, the sweep's result g(x_k) is replaced as the START of
 * the next sweep by the depth-1 Anderson (secant) step  x_k+1 = g(x_k) - gamma (g(x_k) - g(x_k-1)),
 * gamma = <r_k, r_k - r_k-1> / |r_k - r_k-1|^2,  r = g(x) - x,  projected into the friction cones; |gamma| > clip drops the step.
 * The convergence test stays the sweep's own |g(x) - x|: the fixed points are those of the per-contact iteration, and a solve
 * never returns an extrapolated iterate unchecked.  Measured on the Atlas-like standing population (oracle): 18.8 -> 10.6 sweeps,
 * p99 86 -> 41, unconverged 3.9 % -> 0.9 %, natural-map residual p99 2.7e-2 -> 1.1e-5.  first_sweep = 0 switches it off. */
int rsb_set_solver_anderson(rsb_world* w, int first_sweep, double clip);
/* Slip rule of the per-contact iteration (not a RaiSim parameter; default RSB_SLIP_ENERGY).
 *   RSB_SLIP_ENERGY   the published rule (Hwangbo, Lee, Hutter 2018): a slipping contact takes the point of {v_n+ = 0} x {cone boundary} of least
 *                     contact-space kinetic energy;
 *   RSB_SLIP_COULOMB  classical Coulomb friction (Stewart-Trinkle, Anitescu-Potra): the point of the same curve where the post-impulse slip
 *                     velocity is ANTI-PARALLEL to the friction impulse.
 * The two coincide where the normal row of the contact's Delassus block does not couple with the tangential ones (a sphere or a box corner on flat
 * ground); on the foot of a bent leg they differ - the energy rule's friction impulse sits 45 deg (p50) off the opposite of its own slip
 * velocity, 29 % of the impulse (tests/test_oracle_independent.py).  Whether RaiSim's shipped solver is the one or the other
 * cannot be read from /root/reference.  The Coulomb root is located like the energy minimum (16 grid directions, 16-section, Newton) on
 * P = N x d instead of dE/dtheta; a contact problem without a bracketed root (6 % of random strongly coupled blocks) takes the energy rule's point.
 * A kernel class of its own: floating-base systems of tree depth <= 5 with <= 8 contact slots, default integration scheme, one contact per
 * primitive, no peer-mapped obs exchange (RSB_E_UNSUPPORTED from the step otherwise); pipelined twin: yes (open and closed loop). */
#define RSB_SLIP_ENERGY 0
#define RSB_SLIP_COULOMB 1
int rsb_set_slip_rule(rsb_world* w, int rule);

/* Early termination (not RaiSim behaviour; default OFF): in rsb_control_step / rsb_env_step - the calls that know which
 * collision primitives may touch the terrain - an env stops integrating at the sub-step in which any other primitive
 * touches; that sub-step and the rest of the control step are not integrated for it, the detected contacts are
 * reported with zero impulses, flag bit 3 (8) is set and the env is terminated.  Upstream's rsg_anymal looks at the
 * contacts of the LAST sub-step only, so a primitive that touches and lifts off again within one control step ends
 * the episode here but not there.  What it buys: episodes in their last control step (a robot falling onto its knees)
 * are the hardest contact problems of a launch and every launch waits for its slowest env. */
int rsb_set_early_termination(rsb_world* w, int on);
/* Warm start of the contact solver (not a RaiSim parameter; default on): every collision primitive in contact starts
 * the next integrate() from the impulse and friction direction it ended the previous one with.  The state is per
 * env, lives on the device, and is cleared for the envs touched by rsb_set_state / rsb_set_env_row / any reset. */
int rsb_set_solver_warm_start(rsb_world* w, int on);

/* Kernel mapping knob: lanes of a wavefront that cooperate on one env (16, 32 or 64).
 * 64 = the north star's "one wavefront per env"; 0 = pick the measured-fastest default. */
int rsb_set_lanes_per_env(rsb_world* w, int lanes);
int rsb_get_lanes_per_env(const rsb_world* w);

/* Contacts per collision primitive against a height map (not RaiSim's collider; default 1 = the closest feature).  With 2, a
 * sphere that penetrates a SECOND flank - the closest penetrating point of the surface whose direction differs from the first
 * contact's normal by more than min_angle_deg (pass 45: the oracle's default) - reports it as a second contact: a ball in
 * a valley then rests on both sides instead of rattling between them.  The second contact carries RSB_CONTACT_SECOND in
 * rsb_contact::collision, uses its primitive's material, starts cold in every solve, follows all first contacts in the list and
 * counts as its primitive for the termination rule and the foot forces of rsb_control_step.  A kernel class of its own (the
 * default kernels are what they were): floating-base systems of tree depth <= 13, no peer-mapped obs exchange
 * (RSB_E_UNSUPPORTED from the step otherwise). */
int rsb_set_heightmap_contacts(rsb_world* w, int per_primitive, double min_angle_deg);
/* Exact capsule / cylinder x height map (default off: a capsule is its two end spheres, a cylinder the lowest points of its two rims -
 * their exact contact sets on a PLANE).  With on != 0 the barrel between the two ends of every capsule and cylinder of the model
 * (rsb_model_blob::col_capsule; <capsule> and <cylinder> elements of the URDF; a cylinder's samples keep r / L away from its flat caps)
 * also reports its deepest point against a height map when that point penetrates and is deeper than both end spheres by more than
 * 0.1 mm: a shank lying across a ridge rests on the ridge.  The point is located by four rounds of four closest-feature queries along
 * the capsule's axis (resolution 1.3 % of its length; faces, edges and vertices of the triangulated surface alike).  The contact
 * carries RSB_CONTACT_CAPSULE | the FIRST end sphere's index in rsb_contact::collision, uses that primitive's material, starts cold
 * in every solve, follows the first (and second-flank) contacts in the list and counts as that primitive for the termination rule
 * and the foot forces.  Runs in the kernel class of rsb_set_heightmap_contacts (same restrictions); no effect on a plane.
 * Boxes (<box>: eight corner primitives, col_capsule = -1 on the first) get the same treatment: of each pair of opposite faces the one that
 * looks down is searched for its deepest point (three rounds of 4 x 4 point samples, resolution 3.2 % of the face's edge), and the deepest
 * of them is one more contact of the box when it penetrates and is deeper than every corner by more than 0.1 mm (a slab lying on a bump).
 * Upstream counterpart: RaiSim's ODE capsule / box x height-field colliders [RECALL; absent from /root/reference]. */
int rsb_set_capsule_contacts(rsb_world* w, int on);

/* Pipelined control steps (default off).  A launch of the step kernel ends with its slowest wave (a robot that has just fallen: five contacts,
 * three times the sweeps), and a stream runs one launch after the other: every SIMD whose wave has finished idles until the last one has.  With
 * on != 0, consecutive rsb_control_step calls that upload nothing (p_target in device memory, d_target NULL, no peer exchange, no mask) go
 * alternately to two private streams and OVERLAP on the device: workgroup b of launch k + 1 takes its envs as soon as workgroup b of launch k
 * has published them (a per-workgroup sequence number in device memory; an env block is always processed behind the same XCD's L2, release =
 * s_waitcnt vmcnt(0), acquire = buffer_inv sc1 - or agent-scope fences when the host's probe does not find the round-robin XCD pattern), whatever the other workgroups of
 * launch k are still doing; a one-thread gate kernel in front of launch k + 1 keeps it off the chip until launch k has been dispatched
 * completely, so that a waiting workgroup never holds a slot its predecessor needs.  Results are bit-identical to the un-pipelined sequence
 * (envs are independent; each env's steps still run in order).  Every other entry point that touches the world's stream JOINS the pipeline
 * first (the world's stream waits for both private streams), so reads, uploads, plain rsb_integrate calls and rsb_synchronize see completed
 * steps as before.  What the caller must know: work it enqueues ITSELF on a borrowed stream (rsb_set_stream) between two control steps is not
 * ordered after them unless it calls rsb_get_stream / rsb_synchronize (both join) first; and a consumer that needs every env of step k before
 * step k + 1 may start joins at every step and gains nothing - the overlap pays in open-loop stepping (the benchmark's random PD targets,
 * action sequences of sampling-based MPC, replay) and, since round 5, in the CLOSED loop when the policy runs as an action stage per env block
 * (rsb_closed_loop_run in rsb_pipeline.h: 203 M against 147 M env-steps/s in lock-step on the headline workload).  Upstream counterpart: none (RaiSim steps its
 * worlds one after the other on CPU threads). */
int rsb_set_step_pipelining(rsb_world* w, int on);
/* 1 when control steps are pipelined.  rsb_set_step_pipelining(w, 1) leaves it at 0 (and says so once on stderr) with RSB_STEP_PIPELINING=0 in the
 * environment or under a profiler that SERIALISES dispatches (rocprofv3 --pmc sets ROCPROF_COUNTER_COLLECTION): such a tool runs one kernel at a
 * time in an order of its own, a pipelined launch would wait for a predecessor that is not allowed to start (its wait then times out after RSB_PIPE_TIMEOUT_MS,
 * default 10 s, and the fault path of rsb_pipeline.h replays the steps in lock-step).  Counter passes therefore see the plain kernel classes; kernel traces (no serialisation) see the pipeline. */
int rsb_step_pipelining_enabled(const rsb_world* w);
/* Consumers and producers on OTHER streams while the pipeline keeps running (the obs all-gather of a multi-GPU run on its own stream):
 *   rsb_step_pipeline_publish(w, stream)     `stream` waits for the most recent pipelined control step (and nothing else of the pipeline);
 *   rsb_step_pipeline_wait_event(w, event)   the NEXT control step additionally waits for `event` (a hipEvent_t recorded by the caller, e.g.
 *                                            behind the collective that still reads the buffer this step overwrites).
 * Neither joins.  No pipelined step in flight (pipelining off, or just joined): `stream` waits for the world's stream instead, the event is
 * honoured by the next launch all the same - a caller can use the pair unconditionally. */
int rsb_step_pipeline_publish(rsb_world* w, void* hip_stream);
int rsb_step_pipeline_wait_event(rsb_world* w, void* hip_event);
/* pipelined launches enqueued so far and the number of times other calls joined them (diagnostics).  Returns 1 instead of RSB_OK when the
 * library found no two streams whose kernels run concurrently (HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues, default 4; a
 * probe picks the pair when the pipeline is first used): results are the same, the launches run in order */
int rsb_step_pipelining_stats(rsb_world* w, long long* launches, long long* joins);
/* Round 5: (a) no device trap anywhere in the pipeline - a fault (an env-block ticket outside its XCD's range, a wait past RSB_PIPE_TIMEOUT_MS) sets a
 * device error word, the pipelined launches drain without touching their envs, and the next JOINING call restores the state of the last join,
 * switches pipelining off, replays the steps in lock-step and returns RSB_E_PIPELINE once (the inputs of pipelined steps - p_target buffers,
 * reset states - must therefore stay unchanged until the next joining call); a join now waits on the host.  (b) The closed loop: K control
 * steps with an ACTION STAGE between them, handed over env block by env block so that the steps still overlap although every step's
 * actions depend on the step before: rsb_closed_loop_run, rsb_closed_loop_run_linear, rsb_step_pipeline_join / _fault,
 * rsb_debug_pipeline_fault and the device-side half of a caller's stage kernel are declared in rsb_pipeline.h. */

/* Debug aid: where the HOST spends its time inside rsb_view_exchange, accumulated since the last reset - out[0] ns enqueueing the uploads (+ the masked
 * state-row kernels), [1] the launches, [2] the downloads, [3] waiting for the stream, [4] calls (tools/prof_template_path.py). */
int rsb_debug_view_profile(rsb_world* w, long long out[5], int reset);


/* on != 0: rsb_control_steps and the closed-loop runs with an in-repo stage (rsb_closed_loop_run_linear / _mlp, rsb_pipeline.h) use ONE resident launch
 * per call when the world's kernel class has a resident twin; a caller-supplied stage (rsb_closed_loop_run) cannot be compiled into the step kernel and
 * keeps the pipelined path. */
int rsb_set_step_residency(rsb_world* w, int on);
int rsb_step_residency_enabled(const rsb_world* w);
/* 1 when a resident launch exists for this world as it is configured now (stage: 0 open loop, 1 linear policy, 2 actor network; K = control steps per
 * launch), else 0 with the reason in rsb_last_error(): floating base, the plain contact / integration / slip rules, no peer exchange, N a multiple of the
 * envs per workgroup, and one of the two compiled model sizes (tree depth <= 5 with <= 8 contact slots at 16 lanes per env; tree depth <= 13 with 16 slots at 32). */
int rsb_step_residency_status(rsb_world* w, int stage);
/* resident launches so far */
long long rsb_step_residency_launches(const rsb_world* w);
/* debug aid: 1 = every control step of a resident launch writes everything a separate launch writes (state rows, warm records, contact records) */
int rsb_debug_resident_full_writes(rsb_world* w, int on);

/* host only (no GPU): LDS bytes of ONE workgroup of the step kernel for this model (kmax contact slots, self-collision on / off, lanes_per_env 16 / 32 / 64 or 0 =
 * the library's choice).  A CU holds min(4, 160 KiB / this) workgroups of one wave each: the layout of the benchmark's models sits close to such a boundary
 * (4 x 40 048 B for the ANYmal-like model, 3 workgroups for the Atlas-like one), and a table that grows by a few hundred bytes can cost a quarter or a third of the
 * resident waves (tests/test_kernel_budget.py pins the counts). */
int rsb_model_lds_bytes(const rsb_model* m, int kmax, int self_collision, int lanes_per_env);

/* ---- specialised step kernels.  The ahead-of-time kernel classes read the model's dimensions (bodies, coordinates, tree depth, collision primitives,
 * self-collision pairs) and the world's switches (terrain kind, sub-steps per call, warm start, solver lags) from their kernel arguments.  A SPECIALISED code
 * object is the same kernel compiled with those values as constants (-30 % instructions, -40 % branches; +13 % env-steps/s on the benchmark; results bit for
 * bit the same).  Code objects live in rsb_spec_dir() ($RSB_SPEC_DIR, default spec/ next to librsb.so), one per (kernel class, values, source hash of the
 * library).  mode: RSB_SPEC_OFF - ahead-of-time classes only; RSB_SPEC_CACHED (default; $RSB_SPECIALIZE=0 / compile change it) - a code object found there is
 * used, a miss runs the ahead-of-time class; RSB_SPEC_COMPILE - a miss compiles it first (hipcc over the kernel sources, $RSB_SRC_DIR / $RSB_INCLUDE_DIR when
 * they are not where the in-tree build left them; ~25 s once per key).  A code object that fails to load or was compiled against another argument layout is
 * refused (message on stderr), never launched. */
#define RSB_SPEC_OFF 0
#define RSB_SPEC_CACHED 1
#define RSB_SPEC_COMPILE 2
int rsb_set_specialization(rsb_world* w, int mode);
/* returns the mode; step launches so far that ran a specialised code object / an ahead-of-time class */
int rsb_specialization_status(const rsb_world* w, long long* specialized_launches, long long* generic_launches);
const char* rsb_spec_dir(void);
/* host only (no GPU): compile the code object of one manifest line "<lpe> <kmax> <cl> <ml> | -DRSB_SPECIALIZED -DRSB_SPEC_NB=13 ..." into rsb_spec_dir()
 * (what a miss appends to $RSB_SPEC_RECORD; raisimlib_amd/spec_manifest.txt holds the shipped workloads' lines and build() compiles them); its file name */
int rsb_spec_compile(const char* manifest_line);
int rsb_spec_file_name(const char* manifest_line, char* out, int capacity);


/* elapsed device time (ms) of the most recent rsb_integrate launch, measured with HIP events
 * on the handle's stream; also the kernel's static resource usage for reports. */
int rsb_last_kernel_ms(rsb_world* w, float* ms);
/* on = 0: no events; 1: one event pair (rsb_last_kernel_ms); n > 1: a ring of n event pairs, one per launch,
 * read back after the fact with rsb_read_kernel_ms (no per-launch synchronisation). */
int rsb_enable_timing(rsb_world* w, int on);
/* bracket only every stride-th launch (default 1): an event pair costs ~7 us of stream time per launch, 5 % of a
 * 0.16 ms control step, so a benchmark samples its timed region instead of bracketing all of it */
int rsb_set_timing_stride(rsb_world* w, int stride);
/* durations (ms) of the last min(n, launches recorded) step-kernel launches, oldest first; synchronises the
 * stream; returns how many were written (or a negative status) */
int rsb_read_kernel_ms(rsb_world* w, float* ms, int n);

/* Debug aid (tests): dump one env's contact problem of the last sub-step of the next launches:
 * nc, Delassus matrix G [3nc,3nc] row-major, free contact velocity c [3nc], impulses lam [3nc], all in
 * contact-frame coordinates [t1 t2 n] per contact.  env < 0 disables the dump. */
int rsb_debug_select_env(rsb_world* w, int env);
int rsb_debug_read_contact_problem(rsb_world* w, int* nc, float* G, float* c, float* lam);
/* Debug aid (profiling): shader-clock stamps at the phase boundaries of workgroup 0's last sub-step:
 * out16[0..9] = stamps, [10] = solver iterations, [11] = wave-max contact count. */
int rsb_debug_phase_cycles(rsb_world* w, int enable, long long* out16);
/* per-workgroup profile of the last launch (needs rsb_debug_phase_cycles(w, 1, ...) first): out [16*n_blocks] =
 * {total cycles, Gauss-Seidel cycles, sum over sub-steps of the wave's max sweep count, max contact count,
 *  global slip searches run, Newton refinements run, contact solves (sweeps x wave contact count), cycles in searches,
 *  cycles of the solver set-up (G rows -> registers), cycles in Newton refinements, cycles in the per-sweep epilogue, 0...} */
int rsb_debug_wave_profile(rsb_world* w, long long* out, int n_blocks);


#ifdef __cplusplus
}
#endif

#endif /* RSB_EXT_H_ */
