// step_instance.hip — one instance of the fused step kernel (step_kernel.h) per object file.
// Built by raisimlib_amd/build.py as   hipcc -c step_instance.hip -DRSB_I_LPE=16 -DRSB_I_KMAX=8 -DRSB_I_CL=0 -DRSB_I_ML=4 -DRSB_I_PROF=0
// so that the kernel classes of step_launch.h (lanes per env x contact capacity x base kind x tree depth) x {production, profiling}
// compile in parallel instead of in one translation unit of several minutes.
#include "step_kernel.h"
#include "step_launch.h"

#include <cstddef>

#if !defined(RSB_I_LPE) || !defined(RSB_I_KMAX) || !defined(RSB_I_CL) || !defined(RSB_I_ML) || !defined(RSB_I_PROF)
#error "step_instance.hip needs -DRSB_I_LPE= -DRSB_I_KMAX= -DRSB_I_CL= -DRSB_I_ML= -DRSB_I_PROF="
#endif

#ifdef RSB_SPECIALIZED
// the layout this code object was compiled against: the loader (rsb_spec.hip) compares it with the library's own before it launches anything
extern "C" __device__ __attribute__((used)) const unsigned rsb_spec_abi[4] = {(unsigned)sizeof(rsbk::StepArgs), (unsigned)sizeof(rsbk::LdsLayout),
                                                                              (unsigned)offsetof(rsbk::StepArgs, L), (unsigned)rsbk::kSpecFields};
#endif

namespace rsbk {

template <int LPE, int KMAX, int CL, int ML, bool PROF>
hipError_t launch_step_instance(const StepArgs& a, int blocks, size_t lds_bytes, hipStream_t stream) {
  auto kern = rsb_step_kernel<LPE, KMAX, CL, ML, PROF>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds_bytes, stream, a);
  return hipGetLastError();
}

template hipError_t launch_step_instance<RSB_I_LPE, RSB_I_KMAX, RSB_I_CL, RSB_I_ML, (RSB_I_PROF != 0)>(const StepArgs&, int, size_t, hipStream_t);

}  // namespace rsbk
