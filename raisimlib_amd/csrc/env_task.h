// env_task.h — rsg_anymal task arithmetic shared by the step kernel's fused epilogue (step_kernel.h) and the stand-alone
// observation kernel (rsb_world.hip), so that the two produce bit-identical observations: every operation is rounded on its own
// (no FMA contraction), whatever the surrounding code looks like.
// Task semantics [RECALL raisimGymTorch/env/envs/rsg_anymal/Environment.hpp, absent from /root/reference]: observation =
// [base height, third row of the base rotation (world z in the body frame), joint angles, body-frame linear velocity,
//  body-frame angular velocity, joint velocities]; reward = forward_vel_coeff * min(clip, v_x body) + torque_coeff * |tau|^2.
#pragma once

#include <hip/hip_runtime.h>

namespace rsbk {

// entry i of the (10 + 2 nj)-dimensional observation; qs(k) / us(k) return entry k of gc / gv
template <class QF, class UF>
__device__ __forceinline__ float env_ob_entry(int i, int nj, QF qs, UF us) {
#pragma clang fp contract(off)
  const float w = qs(3), x = qs(4), y = qs(5), z = qs(6);
  // world -> body rotation R^T, row-major
  const float Rt[9] = {1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y),
                       2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x),
                       2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)};
  if (i == 0) return qs(2);
  if (i < 4) return Rt[3 * (i - 1) + 2];                      // third ROW of the body -> world rotation: rot.e().row(2)
  if (i < 4 + nj) return qs(7 + i - 4);
  if (i < 10 + nj) {
    const int k = i - 4 - nj, rr = k % 3, o3 = k < 3 ? 0 : 3;   // linear, then angular velocity in the body frame
    return (Rt[3 * rr] * us(o3) + Rt[3 * rr + 1] * us(o3 + 1)) + Rt[3 * rr + 2] * us(o3 + 2);
  }
  return us(6 + i - 10 - nj);
}

// forward velocity in the body frame (first row of R^T times the linear velocity)
template <class QF, class UF>
__device__ __forceinline__ float env_forward_velocity(QF qs, UF us) {
#pragma clang fp contract(off)
  const float w = qs(3), x = qs(4), y = qs(5), z = qs(6);
  return ((1 - 2 * (y * y + z * z)) * us(0) + (2 * (x * y + w * z)) * us(1)) + (2 * (x * z - w * y)) * us(2);
}

}  // namespace rsbk
