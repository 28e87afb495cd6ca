// step_launch.h — host-callable launchers of the step kernel's instances.
//
// Each (LPE, KMAX, CL, ML, PROF) combination of rsbk::rsb_step_kernel is compiled in its own object file from
// step_instance.hip (raisimlib_amd/build.py passes the five values as -D macros and builds the objects in parallel);
// this header only declares the launcher template, so rsb_world.hip links against whichever instances were built.
#pragma once

#include <hip/hip_runtime.h>

#include "step_types.h"

namespace rsbk {

// the instance list, single source of truth for build.py (parsed there) and the dispatch in rsb_world.hip:
// (third value: kernel class bits - 0 = floating base, 1 = fixed-base systems, 2 = floating base + peer-mapped obs exchange in the epilogue,
//  4 = floating base + a second contact per primitive against a height map (both flanks of a valley),
//  8 = floating base + an integration scheme other than semi-implicit Euler;
//  +16 = the pipelined twin of a class (rsb_set_step_pipelining: per-workgroup hand-over between consecutive launches; every class but the peer exchange's);
//  32 = floating base + the CLASSICAL COULOMB slip rule (rsb_set_slip_rule; quadruped-sized models: tree depth <= 5, <= 8 contacts);
//  64 = RESIDENT launch of the plain floating-base class (rsb_set_step_residency: several control steps per launch, the env block stays in LDS), open loop;
//  64 + 128 / + 256 / + 384 = ... with the action stage inside: the linear policy / the actor network of widths <= 128 / <= 256 (stage_bodies.h).
//  Built for the benchmark's two model sizes (quadruped: 16,8,.,4; humanoid: 32,16,.,12); everything else runs its control steps as separate launches)
// RSB_STEP_INSTANCES: 16,8,0,4 32,8,0,4 64,8,0,4 16,16,0,4 32,16,0,4 64,16,0,4 16,16,0,12 32,16,0,12 64,16,0,12 16,16,0,16 32,16,0,16 64,16,0,16 16,8,1,4 32,8,1,4 64,8,1,4 16,16,1,4 32,16,1,4 64,16,1,4 16,16,1,12 32,16,1,12 64,16,1,12 16,16,1,16 32,16,1,16 64,16,1,16 16,8,2,4 32,8,2,4 64,8,2,4 16,16,2,4 32,16,2,4 64,16,2,4 16,16,2,12 32,16,2,12 64,16,2,12 16,8,4,4 32,8,4,4 64,8,4,4 16,16,4,4 32,16,4,4 64,16,4,4 16,16,4,12 32,16,4,12 64,16,4,12 16,8,8,4 32,8,8,4 64,8,8,4 16,16,8,4 32,16,8,4 64,16,8,4 16,16,8,12 32,16,8,12 64,16,8,12 16,8,16,4 32,8,16,4 64,8,16,4 16,16,16,4 32,16,16,4 64,16,16,4 16,16,16,12 32,16,16,12 64,16,16,12 16,16,16,16 32,16,16,16 64,16,16,16 16,8,17,4 32,8,17,4 64,8,17,4 16,16,17,4 32,16,17,4 64,16,17,4 16,16,17,12 32,16,17,12 64,16,17,12 16,16,17,16 32,16,17,16 64,16,17,16 16,8,20,4 32,8,20,4 64,8,20,4 16,16,20,4 32,16,20,4 64,16,20,4 16,16,20,12 32,16,20,12 64,16,20,12 16,8,24,4 32,8,24,4 64,8,24,4 16,16,24,4 32,16,24,4 64,16,24,4 16,16,24,12 32,16,24,12 64,16,24,12 16,8,32,4 32,8,32,4 64,8,32,4 16,8,48,4 32,8,48,4 64,8,48,4 16,8,64,4 16,8,192,4 16,8,320,4 16,8,448,4 32,16,64,12 32,16,192,12 32,16,320,12

// sets the dynamic-LDS attribute and launches `blocks` workgroups of one wavefront on `stream`
template <int LPE, int KMAX, int CL, int ML, bool PROF>
hipError_t launch_step_instance(const StepArgs& a, int blocks, size_t lds_bytes, hipStream_t stream);

}  // namespace rsbk
