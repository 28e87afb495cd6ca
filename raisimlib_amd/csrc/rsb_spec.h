// rsb_spec.h — specialised code objects of the step kernel (rsb_spec.hip; step_spec.h says what is specialised and why).  Not installed.
#pragma once

#include <hip/hip_runtime.h>

#include "step_types.h"

struct rsb_world;

namespace rsbw {
struct SpecClass { int lpe, kmax, cl, ml, prof = 0; };   // (prof: the instance with the cycle stamps of the rsb_debug_* entry points)   // the template arguments of rsbk::rsb_step_kernel a launch would run with
// the specialised kernel for this launch: from the world's memo, else from the cache directory, else (RSB_SPEC_COMPILE) compiled now; nullptr = run the ahead-of-time class
hipFunction_t spec_find(rsb_world* w, const SpecClass& c, const rsbk::StepArgs& a);
int spec_launch(hipFunction_t fn, const rsbk::StepArgs& a, int blocks, size_t lds_bytes, hipStream_t stream);
int spec_default_mode();                       // $RSB_SPECIALIZE: 0 / off, compile, anything else or unset: cached
}  // namespace rsbw
