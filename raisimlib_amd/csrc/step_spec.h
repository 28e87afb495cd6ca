// step_spec.h — SPECIALISED compilation of the step kernel (rsb_specialize; rsb_spec.hip).
//
// The ahead-of-time kernel classes of step_launch.h read the model's dimensions (bodies, coordinates, tree depth, collision primitives, candidate
// pairs of self-collision ...) and the world's switches (terrain kind, sub-steps per call, warm start, solver lags ...) from their kernel arguments:
// every one of them is a scalar load, a live SGPR and - worst for the ONE wave a SIMD holds - a branch or a loop bound the compiler cannot resolve
// (a branch costs a lone wave 20-45 cycles taken or not: profiles/r02_ubench_lone_wave_latency.txt; the generic quadruped class executes ~245 of
// them per sub-step: profiles/r06_spec_log.txt).  A specialised code object is the SAME template instance compiled once more with those values as
// compile-time constants (-DRSB_SPECIALIZED -DRSB_SPEC_NB=13 ...): 30 % fewer instructions, 40 % fewer branches, less than half the spilled SGPRs,
// bit-identical results, +13 % env-steps/s on the benchmark.  The code object is loaded as a HIP module and launched with the same StepArgs.
//
// ONE list names what is specialised: RSB_SPEC_FIELDS(X) calls X(MACRO_SUFFIX, value-expression over a StepArgs `a`).  The host builds a launch's key
// and the compiler's -D flags from it (rsb_spec.hip); the kernel reads a field as RSB_DIM(SUFFIX, generic expression).
#pragma once

#define RSB_SPEC_FIELDS(X)                                                                                                              \
  X(NB, a.nb) X(NQ, a.nq) X(NV, a.nv) X(DEPTH, a.depth) X(NCOL, a.ncol) X(MAX_KID, a.max_kid) X(FIXED_BASE, (a.fixed_base != 0))      \
  X(N_SELF, a.n_self) X(NSUB, a.nsub) X(KMAX, a.kmax) X(HAS_WARM, (a.warm != nullptr)) X(TERRAIN, a.terrain_type)                      \
  X(EARLY_TERM, (a.early_term != 0)) X(SECTION_ROUNDS, a.section_rounds) X(STALL_WINDOW, a.stall_window) X(FREEZE_AFTER, a.freeze_after) \
  X(REFINE, a.refine) X(MULTI_FA, a.multi_freeze_after) X(MULTI_DEPTH, a.multi_depth) X(MULTI_LIGHT, a.multi_light)                    \
  X(MULTI_SW, a.multi_stall_window) X(CHAIN, (a.chain != 0)) X(MODEL_PITCH, a.L.model_pitch)

#ifdef RSB_SPECIALIZED
#define RSB_DIM(NAME, expr) (RSB_SPEC_##NAME)
#else
#define RSB_DIM(NAME, expr) (expr)
#endif

namespace rsbk {
#define RSB_SPEC_COUNT_ONE(NAME, expr) +1
constexpr int kSpecFields = 0 RSB_SPEC_FIELDS(RSB_SPEC_COUNT_ONE);
#undef RSB_SPEC_COUNT_ONE
}  // namespace rsbk
