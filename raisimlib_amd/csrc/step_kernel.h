// step_kernel.h — the fused World::integrate() kernel for gfx950 (CDNA4, wave64).
//
// Replaces, for N independent envs at once, the hot path of SURVEY.md §8a (rows a1-a15):
// raisim::World::integrate1() (kinematics, collision detection, per-object dynamics) and
// World::integrate2() (Delassus blocks, per-contact solver, time integration).  None of those files exist
// in /root/reference (3-file stub) — the algorithm follows the published sources cited in
// oracle/rsb_oracle.h and is checked against that oracle.
//
// Mapping.  One workgroup = one wavefront (64 lanes).  A group of LPE lanes (16, 32 or 64) owns one env;
// LPE=64 is the north star's "one wavefront per env", smaller LPE packs 64/LPE envs into a wave (a wave
// instruction costs the same for 16 or 64 active lanes, so packing is what fills the chip at N=4096).
// At N=4096 there is one wave per SIMD, so the kernel is LATENCY bound: every design choice below
// minimises dependent LDS round trips and barriers rather than instruction count.
//   tree passes   : lane = BODY.  What does not depend on the parent (joint transform, rigid inertia, bias force, actuation)
//                   runs once for all bodies; the propagation is level-synchronous, one LDS hand-over per tree LEVEL.
//                   The floating base is computed redundantly by every lane (no exchange needed).
//   collisions    : lane = collision sphere.      contact columns : lane = (contact, axis).
//   Delassus      : lane = contact pair.          Gauss-Seidel    : lane = contact (its G rows, velocity and
//                   impulse live in registers); impulse changes are broadcast with DPP row_newbcast, the slip
//                   case's candidate directions are spread over the 16 lanes of the row (ballot + DPP min).
// All per-env intermediates live in LDS / registers; HBM is touched only for the state rows at launch
// start / end ([N, dim] row-major rows: consecutive lanes read consecutive floats).
//
// Algorithm (fp32).  Common-frame spatial algebra with origin at the base position (see oracle):
//   down pass : R, r, S, V, bias acceleration A per body; rigid inertia and bias force
//   up pass   : articulated-body inertia IA (RBDA Table 7.1), U = IA S, D = S.U; the same pass
//               propagates Z = dt*(bias force) so that yhat_k = dt*tau_k - S_k.Z_k is the k-th entry
//               of L^-T b (M = L^T D L).  The base's 6x6 articulated inertia is Cholesky-factored.
//   columns   : for each contact axis the unit impulse [x×t; t] is propagated up the support chain
//               (same recursion) giving a sparse column W_c = D^-1/2 L^-T J_c^T; G = W W^T,
//               c = J u + W_c.W_b.
//   solver    : per-contact Gauss-Seidel, open/stick/slip(minimum-energy point of the cone boundary)
//               (Hwangbo et al. 2018), same rules and constants as oracle solve_one_contact().
//   update    : du = L^-1 D^-1/2 (W_b + sum W_c lam) by one root->leaf pass; semi-implicit Euler.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "env_task.h"
#include "rsb_types.h"
#include "step_types.h"
#include "step_spec.h"     // RSB_DIM: model dimensions / world switches as compile-time constants in specialised code objects
#include "step_math.h"      // vectors, spatial algebra, LDS access
#include "step_terrain.h"   // sphere x height map narrow phase
#include "step_slip.h"      // slip case of the one-contact rule, DPP row reductions
#include "stage_bodies.h"   // the in-repo action stages' per-block bodies (resident closed-loop classes)

// The resident classes (kernel class bit 64) wrap the sub-step loop and the epilogue in a loop over control steps.  The wrapper is PREPROCESSOR-selected
// (every instance is its own translation unit, compiled with -DRSB_I_CL=...): written as `if constexpr` / a one-trip loop it moved the register
// allocation of every other class (+4 VGPRs, +2 spilled SGPRs in the benchmark's instance) - those stay instruction for instruction what they were.
#ifndef RSB_X_RES_RING   /* groups in the weight ring of the actor network inside the resident classes (stage_bodies.h: mlp_block); 0 = the stage kernels' ring.
                            8 (56 loads in flight instead of 32) measured no faster - 198.2 against 198.6 M with the 34-128-128-12 network in the loop: the block's network
                            is bound by its dependent matrix-instruction chains, not by the ring (profiles/r06_ab_log.txt #3) */
#define RSB_X_RES_RING 0
#endif
#if defined(RSB_I_CL) && ((RSB_I_CL) & 64)
#define RSB_RESIDENT 1
#else
#define RSB_RESIDENT 0
#endif

namespace rsbk {

// Kernel arguments are read through the kernarg segment pointer, one "view" per phase: RSB_ARGS(x) declares a reference
// whose loads cannot move above that point (the empty asm makes the pointer opaque), so an argument lives in SGPRs only
// inside the phase that uses it.  Taken by value and referenced directly, all ~150 dwords of StepArgs are loaded at kernel
// entry and the ones needed late (the epilogue's 14 pointers, the solver's parameters ...) are carried across every phase:
// 217 spilled SGPRs and ~1000 v_readlane reloads in round 1's ISA - each one an issue slot of the one wave a SIMD holds.
typedef const __attribute__((address_space(4))) StepArgs* KArgs;
__device__ __forceinline__ KArgs rsb_cold(KArgs p) { asm volatile("" : "+s"(p)); return p; }
#define RSB_ARGS(name) const __attribute__((address_space(4))) StepArgs& name = *rsb_cold(ka)

#define RSB_STAMP(i) \
  if (PROF && a.prof && blockIdx.x == 0 && lane == 0 && sub == a.nsub - 1) a.prof[i] = clock64();

// rigid inertia (10 parameters about O) + bias force of a body given (R, r, V, A) and its constants MF
// (DevModel::bodyf layout); Zout = dt * f
__device__ __forceinline__ void body_inertia(const float* Rb, const float* rb, const float* Vb, const float* Ab,
                                             const float* MF, float dt, float* I10, float* Zout) {
  const float mass = MF[7];
  float c[3], t[3], T[9], Iw[6];
  mat3_vec(Rb, MF + 17, t);
  c[0] = rb[0] + t[0]; c[1] = rb[1] + t[1]; c[2] = rb[2] + t[2];
  const float Il[9] = {MF[20], MF[21], MF[22], MF[21], MF[23], MF[24], MF[22], MF[24], MF[25]};
  mat3_mul(Rb, Il, T);
  Iw[0] = T[0] * Rb[0] + T[1] * Rb[1] + T[2] * Rb[2];
  Iw[1] = T[0] * Rb[3] + T[1] * Rb[4] + T[2] * Rb[5];
  Iw[2] = T[0] * Rb[6] + T[1] * Rb[7] + T[2] * Rb[8];
  Iw[3] = T[3] * Rb[3] + T[4] * Rb[4] + T[5] * Rb[5];
  Iw[4] = T[3] * Rb[6] + T[4] * Rb[7] + T[5] * Rb[8];
  Iw[5] = T[6] * Rb[6] + T[7] * Rb[7] + T[8] * Rb[8];
  const float cc = dot3(c, c);
  I10[0] = Iw[0] + mass * (cc - c[0] * c[0]); I10[1] = Iw[1] - mass * c[0] * c[1]; I10[2] = Iw[2] - mass * c[0] * c[2];
  I10[3] = Iw[3] + mass * (cc - c[1] * c[1]); I10[4] = Iw[4] - mass * c[1] * c[2];
  I10[5] = Iw[5] + mass * (cc - c[2] * c[2]);
  I10[6] = mass * c[0]; I10[7] = mass * c[1]; I10[8] = mass * c[2]; I10[9] = mass;
  float IV[6], IAc[6], n1[3], n2[3], n3[3];
  rigid_mul(I10, I10 + 6, mass, Vb, IV);
  rigid_mul(I10, I10 + 6, mass, Ab, IAc);
  // f = I A + V x* (I V) ;  [w;v] x* [n;f] = [w x n + v x f ; w x f]
  cross3(Vb, IV, n1); cross3(Vb + 3, IV + 3, n2); cross3(Vb, IV + 3, n3);
  RSB_UNROLL for (int i = 0; i < 3; ++i) { Zout[i] = dt * (IAc[i] + n1[i] + n2[i]); Zout[3 + i] = dt * (IAc[3 + i] + n3[i]); }
}

// ---- Delassus storage of the large-contact classes (KMAX > 8): only the blocks (i, j) with i >= j exist, packed at
// ((i (i + 1)) / 2 + j) * 12 floats (3 rows on a 4-float pitch): 1632 floats at KMAX 16 where the square layout takes 3264 -
// what lets a 31-body humanoid run two envs per wave (LPE 32).  tri_load returns M[ra][rb] = G[3 a + ra][3 b + rb] for any
// (a, b): the stored block or its transpose (G is symmetric); tri_store writes it back the same way.
__device__ __forceinline__ int tri_off(int hi, int lo) { return ((hi * (hi + 1)) / 2 + lo) * 12; }
__device__ __forceinline__ void tri_load(const float* G, int a, int b, float (&M)[3][3]) {
  const bool tr = b > a;
  const float* p = G + tri_off(tr ? b : a, tr ? a : b);
  float t[3][4];
  RSB_UNROLL for (int r = 0; r < 3; ++r) ld4(p + 4 * r, t[r]);
  RSB_UNROLL for (int r = 0; r < 3; ++r)
    RSB_UNROLL for (int c = 0; c < 3; ++c) M[r][c] = tr ? t[c][r] : t[r][c];
}
__device__ __forceinline__ void tri_store(float* G, int a, int b, const float (&M)[3][3]) {
  const bool tr = b > a;
  float* p = G + tri_off(tr ? b : a, tr ? a : b);
  RSB_UNROLL for (int r = 0; r < 3; ++r) {
    const float row[4] = {tr ? M[0][r] : M[r][0], tr ? M[1][r] : M[r][1], tr ? M[2][r] : M[r][2], 0.f};
    st4(p + 4 * r, row);
  }
}

// ---- resident closed loop: pass `pass` of the action stage for env block `blk`, evaluated by the block's own wave (stage_bodies.h).  The rows the
// body reads (observation, reward, done) were written by this wave's epilogue: the stores have to have arrived (vmcnt) and the wave's L1 must not
// serve an older copy of the lines (buffer_inv); likewise behind it for the action rows the next control step reads.
template <int STG, int EPW>
__device__ __forceinline__ void resident_stage(KArgs ka, int blk, int pass, int n_steps) {
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass of a translation unit that includes this header cannot copy a struct out of the kernarg address space)
  RSB_ARGS(as);
  asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_inv sc1" ::: "memory");
  rsb_stage_ctx c;
  c.ob = as.env_ob; c.act = const_cast<float*>(as.act); c.reward = as.env_reward; c.done = as.done_out;
  c.n_envs = as.N; c.ob_dim = 10 + 2 * (as.nv - 6); c.act_dim = as.nv - 6;
  c.n_steps = n_steps; c.pass_global0 = as.res_pass_global0;
  const int env0 = blk * EPW, n_env = min(EPW, as.N - env0);
  if constexpr (STG == 1) rsb_stage_body::linear_block(c, as.res_pol.lin, env0, n_env, pass, pass == n_steps);
  else rsb_stage_body::mlp_block<(STG == 2 ? 2 : 4), std::remove_reference_t<decltype(as.res_pol.mlp)>, RSB_X_RES_RING>(c, as.res_pol.mlp, env0, n_env, pass, pass == n_steps);
  asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_inv sc1" ::: "memory");
#endif
}

// ------------------------------------------------------------------------------- the kernel
// LPE : lanes per env.  KMAX : contact capacity.  CL : kernel class bits: 1 = fixed-base systems (their contact blocks get a compliance, see the Delassus
// phase), 2 = peer-mapped obs exchange in the epilogue (rsb_obs_peer_*); 0 = floating base, no exchange - the benchmark's class stays what it was,
// instruction for instruction (code added to the shared epilogue moved the register allocation of the sub-step loop: +26 spill moves).
// ML : body-level capacity (>= depth-1).  PROF : compile the cycle stamps / contact-problem dump / LDS poisoning of the
// rsb_debug_* entry points in (the production instances carry none of it: fewer SGPRs, no branches in the solver loop).
#ifdef RSB_X_WPE2   /* experiment: cap the instance at 256 registers (two waves per SIMD by registers; the allocator spills the rest to scratch) */
#define RSB_X_WPE_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#else
#define RSB_X_WPE_ATTR
#endif
// Register budget of the pipelined classes (| 16): they share their SIMD with a wave of the action stage in a closed-loop run (rsb_pipeline.hip: the stage
// kernels are held under 96 registers for this), so they may take 416 of the SIMD's 512 - tests/test_kernel_budget.py reads the compiler's report.  (An
// __attribute__((amdgpu_num_vgpr(416))) does nothing here: on gfx950 it bounds the 256 architectural registers only.  What keeps the classes inside the budget
// is the number of coupling blocks the solver holds in registers: step_phase_solver.inc, NPK.)
template <int LPE, int KMAX, int CL, int ML, bool PROF>
__global__ void __launch_bounds__(64) RSB_X_WPE_ATTR rsb_step_kernel(const StepArgs) {
  const KArgs ka = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();   // the by-value StepArgs sits at offset 0 of the kernarg segment
#ifdef RSB_X_NOPS   /* experiment: shift the whole kernel's code by RSB_X_NOPS x 4 bytes (alignment of the hot loops' fetch windows) */
  static_for<0, RSB_X_NOPS>([&](auto) { asm volatile("s_nop 0"); });
#endif
  RSB_ARGS(a);                                                     // the prologue's view (and the PROF-only fields)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  long long t_entry = 0; if (PROF) t_entry = clock64();
  constexpr int EPW = 64 / LPE;
  constexpr bool FIXED = (CL & 1) != 0, PEER = (CL & 2) != 0, HM2 = (CL & 4) != 0, TH = (CL & 8) != 0, PIPE = (CL & 16) != 0;
  constexpr bool COUL = (CL & 32) != 0;   // classical Coulomb slip rule (rsb_set_slip_rule) instead of the published least-energy point
  // RESIDENT launch (StepArgs::res_steps): several control steps per launch, the env block's state stays in LDS; STG: the action stage the block's own
  // wave evaluates between two control steps (0 = open loop: PD targets from a bank; 1 = linear policy; 2 / 3 = actor network, widths <= 128 / <= 256)
  constexpr bool RES = (CL & 64) != 0;
  [[maybe_unused]] constexpr int STG = RES ? ((CL >> 7) & 3) : 0;
  static_assert(!RES || (CL & ~(64 | 128 | 256)) == 0, "resident launches exist for the plain floating-base class");
  static_assert(RES == (RSB_RESIDENT != 0), "the control-step loop of the resident classes is selected by the preprocessor (RSB_I_CL)");
  constexpr bool TRI = KMAX > 8;    // packed lower-triangular Delassus blocks (see tri_off); the quadruped classes keep the square layout
  const int lane = threadIdx.x;
  const int el = lane / LPE;
  const int s = lane - el * LPE;
  // XCD-aware block -> env mapping: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs (each with its own
  // L2); consecutive env blocks share the cache lines at their row boundaries, so block b of XCD x takes env block
  // x * (blocks / 8) + b / 8 and every XCD reads (and writes) one contiguous slice of every state array
  // (profiles/r02_traffic_calibration.txt: 1.4x over-fetch of the 76-B rows without it)
  int blk = blockIdx.x;
  if ((gridDim.x & 7) == 0) blk = (blk & 7) * (gridDim.x >> 3) + (blk >> 3);
  if constexpr (PIPE) {
    // pipelined control steps (StepArgs::pipe_prog).  The gate of the next launch counts the workgroups of this one that are on the chip.
    // pipe_xcds > 0: the env block is chosen by the XCD this workgroup landed on (block = XCD x (blocks / XCDs) + a ticket of that XCD), so that
    // every block is always processed behind the same L2 and the hand-over needs no L2 write-back (see the wait below)
    if (lane == 0) __hip_atomic_fetch_add(a.pipe_started, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // a fault earlier in the pipeline (the error word is set: the gates no longer hold the launches apart, tickets of several launches mix): leave at once
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(a.pipe_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) return;
    if (a.pipe_xcds > 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      xcc &= 15u;
      unsigned t = 0;
      if (lane == 0) t = __hip_atomic_fetch_add(a.pipe_xcc_ctr + (xcc << 6), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.pipe_xcc_base;   // (one 256-byte line per XCD's counter)
      t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
      const unsigned per = gridDim.x / (unsigned)a.pipe_xcds;
      if (t >= per || xcc >= (unsigned)a.pipe_xcds) {
        // the dispatcher did not deal this launch's workgroups round-robin over the XCDs (the host's probe saw it do so on an idle device): no
        // env block can be assigned.  Error word instead of a trap: this workgroup leaves without touching anything, the others follow (below),
        // the host's next join replays the steps in lock-step
        if (lane == 0 && atomicCAS(a.pipe_err, 0, 1 /* RSB_PIPE_ERR_TICKET; the first code stays */) == 0) __hip_atomic_store(a.pipe_err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
      }
      blk = (int)(xcc * per + t);
    }
  }
  int env = blk * EPW + el;
  bool env_valid = env < a.N;
  if (!env_valid) env = a.N - 1;
  // masked launch (per-env raisim::World views, rsb_integrate_masked): a masked-off env runs along but writes nothing back
  if (a.env_mask && !a.env_mask[env]) env_valid = false;
  // model dimensions travel in the kernel arguments (read through a.model they cost one more dependent load before anything can start)
  // (RSB_DIM: a compile-time constant in a specialised code object, the kernel argument in the ahead-of-time classes - step_spec.h)
  const int nb = RSB_DIM(NB, a.nb), nq = RSB_DIM(NQ, a.nq), nv = RSB_DIM(NV, a.nv), depth = RSB_DIM(DEPTH, a.depth), ncol = RSB_DIM(NCOL, a.ncol);
#ifdef RSB_SPECIALIZED
  constexpr int cw = (6 + RSB_SPEC_DEPTH - 1 + 3) & ~3;   // (round4(6 + depth - 1): rsb_world.hip)
#else
  const int cw = a.cw;
#endif
  const bool fixed_base = RSB_DIM(FIXED_BASE, a.fixed_base != 0);
  const auto& L = a.L;

  float* MODELF = lds + L.t_model;
  float* GAIN = lds + L.t_gain;                                  // [nb][2] kp, kd of the body's joint
  int* PARLV = reinterpret_cast<int*>(lds + L.t_parlv);          // [nb] (parent+1) | level << 8
  int* ANC = reinterpret_cast<int*>(lds + L.t_anc);              // [nb*depth]
  float* DIR16 = lds + L.t_dir;                                  // float4 [16] slip-search brackets
  float* COLT = lds + L.t_col;                                   // [ncol][kColSlot] sphere centre (body frame), radius | body, mu, restitution, res_threshold | axis, rim
  int* KIDS = reinterpret_cast<int*>(lds + L.t_kids);            // [nb] child bodies, grouped by parent (DevModel::kid_start / kid_count)
  int* KIDX = reinterpret_cast<int*>(lds + L.t_kidx);            // [nb] kid_start | kid_count << 16
  float* E = lds + L.shared_total + el * L.per_env;
  float* Q = E + L.q;
  float* U = E + L.u;
  float* PT = E + L.pt;
  float* DTG = E + L.dtg;
  float* TF = E + L.tf;
  float* BODY = E + L.body;
  float* UPS = E + L.g;                                          // [nb][28] articulated inertia + bias handed to the parent body; ALIASES G (dead before the Delassus phase)
  float* FACT = E + L.fact;
  float* WB = E + L.wb;
  float* CON = E + L.con;
  float* WC = E + L.wc;
  float* CV = E + L.cv;
  float* G = E + L.g;
  float* GINV = E + L.ginv;
  float* LAM = E + L.lam;
  float* TACT = E + L.tact;                                      // [nv] actuator torque of every joint in the current sub-step
  float* WARM = E + L.warm;                                      // [ncol][6] warm state of the contact solver (see StepArgs::warm)
  const int* SPAIR = reinterpret_cast<const int*>(lds + L.t_spair);   // [n_self + 1] candidate pairs of self-collision: byte offset of centre i in CEN | of centre j << 16; the last entry pairs primitive 0 with itself (never a hit)
  float* CEN = E + L.cen;                                        // [ncol][4] primitive centres (relative to the base position) + radius; may alias WC
  float* SELFT = E + L.selft;                                    // [kmax][4] per contact slot of a self-collision: mu, restitution, threshold | J u of the slot's normal row
  const int n_self = RSB_DIM(N_SELF, a.n_self);                                   // candidate pairs; 0 = self-collision off
  const int nwarm = 6 * ncol;
  const int GS = L.gstride;

  if (PROF && a.poison_lds) {
    for (int i = lane; i < a.lds_floats; i += 64) lds[i] = __int_as_float(0x7fc00000);
    __syncthreads();
  }
#if RSB_RESIDENT
  const int ncs = a.res_steps;   // control steps of this launch
  // closed loop: pass 0 of the action stage - the actions of the first control step from the observation the host left in env_ob - before the
  // prologue reads the action rows
  if constexpr (STG != 0) resident_stage<STG, EPW>(ka, blk, 0, ncs);
#endif
#include "step_phase_prologue.inc"
#if RSB_RESIDENT
  for (int cs = 0; cs < ncs; ++cs) {
#endif
  for (int sub = 0; sub < nsub; ++sub) {
    RSB_STAMP(0)
    RSB_ARGS(ab);
    tsq = 0.f;
#include "step_phase_tree_down.inc"
#include "step_phase_collision.inc"
#include "step_phase_tree_up.inc"
    iters_used = 0;
    float wlam[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // base part of sum_c W_c lam_c (left by the solver on the lanes of the env's first row)
    if (ncw > 0) {
      RSB_ARGS(aw);
      const float erp = aw.erp;
#include "step_phase_columns.inc"
#include "step_phase_delassus.inc"
#include "step_phase_solver.inc"
    } else if (has_warm) {
      for (int i = s; i < nwarm; i += LPE) WARM[i] = 0.f;   // no contact anywhere in this wave: nothing survives
    }
    RSB_STAMP(6)

#include "step_phase_update.inc"
  }  // substeps

  if (PROF && a.prof && lane == 0) { long long* P = a.prof + 16 + 16 * (long long)blk; P[8] = t_setup; P[9] = t_newt; P[10] = t_epi; P[11] = t_rule; P[12] = t_exch; P[13] = t_end; P[14] = t_start - t_entry; P[15] = t_mag; P[0] = clock64() - t_start; P[1] = t_gs; P[2] = p_iters; P[3] = p_ncw; P[4] = p_search; P[5] = p_newton; P[6] = p_solves; P[7] = t_srch; }
#include "step_phase_epilogue.inc"
#if RSB_RESIDENT
  }  // control steps
#endif
}

}  // namespace rsbk