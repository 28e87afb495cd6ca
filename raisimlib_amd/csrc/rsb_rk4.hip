// rsb_rk4.hip — IntegrationScheme::RUNGE_KUTTA_4 [RECALL raisim::ArticulatedSystem::setIntegrationScheme; upstream file absent from
// /root/reference], host-driven over the query kernels: the slow, accurate path.  Correctness only - four dynamics
// evaluations with dense M^-1 per integrate(); the benchmark kernels know nothing of it.
//
// What one integrate() does in this scheme (this repo's definition: what RaiSim's does with contacts cannot be read from the reference):
//   (1) the classical four-stage Runge-Kutta step of the SMOOTH equations of motion  q' = u,  u' = M(q)^-1 (tau(q, u) - h(q, u))  from (q0, u0):
//       tau = explicit PD at the stage's own state (+ feed-forward, effort clip, passive joint damping), h from the RNE query, M^-1 from the CRBA query.
//       The base orientation is advanced on SO(3) (Runge-Kutta-Munthe-Kaas: stage rotation vectors, dexp^-1 truncated after the second-order term), so
//       the order holds for a spinning body.  Result: a velocity increment du and a configuration increment Theta (velocity space).
//   (2) contacts and joint limits as in every other scheme - detected at q0, one per-contact solve - on top of that free motion: the step kernel runs
//       ONCE in FORCE_AND_TORQUE mode with the generalized force  tau_eff = M(q0) du / dt + h(q0, u0) (+ the damping it subtracts again), which makes
//       its own free-motion update equal du exactly:  u+ = u0 + du + M^-1 J^T lambda.
//   (3) positions:  q+ = q0 (+) (Theta + dt (u+ - u0 - du)) - the Runge-Kutta increment plus the contact impulses' share, semi-implicitly.
// Without contacts (2) adds nothing and the step is the fourth-order one: tests/test_gpu_kat.py pins the energy drift of a free asymmetric top and
// of a pendulum at O(dt^4), tests/test_gpu_parity.py the step against an fp64 numpy restatement over the oracle's M and h.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "rsb_world.h"

namespace rsbw {
namespace {

__device__ inline void quat_left_exp(const float* th, const float* q, float* out) {      // out = exp(th) (x) q, th a world-frame rotation vector
  const float a = sqrtf(th[0] * th[0] + th[1] * th[1] + th[2] * th[2]);
  const float h = 0.5f * a;
  const float sc = a > 1e-8f ? sinf(h) / a : 0.5f;
  const float d0 = cosf(h), d1 = sc * th[0], d2 = sc * th[1], d3 = sc * th[2];
  const float r0 = d0 * q[0] - d1 * q[1] - d2 * q[2] - d3 * q[3];
  const float r1 = d0 * q[1] + d1 * q[0] + d2 * q[3] - d3 * q[2];
  const float r2 = d0 * q[2] - d1 * q[3] + d2 * q[0] + d3 * q[1];
  const float r3 = d0 * q[3] + d1 * q[2] - d2 * q[1] + d3 * q[0];
  const float in = 1.0f / sqrtf(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3);
  out[0] = r0 * in; out[1] = r1 * in; out[2] = r2 * in; out[3] = r3 * in;
}
// q = q0 (+) th  (th in velocity space: base linear, base rotation vector (world frame), joints)
__device__ inline void config_add(const float* q0, const float* th, int nq, int nv, int fixed_base, float* q) {
  if (fixed_base) { for (int i = 0; i < 7; ++i) q[i] = q0[i]; }
  else {
    for (int i = 0; i < 3; ++i) q[i] = q0[i] + th[i];
    quat_left_exp(th + 3, q0 + 3, q + 3);
  }
  for (int i = 6; i < nv; ++i) q[i + 1] = q0[i + 1] + th[i];
}

struct Rk4Args {
  const rsbk::DevModel* model;
  float *gc, *gv;                      // the world's state rows: the stage state on entry, the NEXT stage's state on exit
  const float *q0, *u0;                // the step's initial state
  const float *Minv, *h;               // of the stage state (query kernels)
  const float *kp, *kd, *pt, *dtg, *tff;
  float *ka, *kv;                      // [4][N, nv] accelerations / velocity-space increments of the stages
  const uint8_t* mask;
  int N, nq, nv, stage, pd;
  float dt;
};

// stage `stage` (0..3): a = M^-1 (tau - h) at the stage state, k_v = the stage's velocity-space increment, then the next stage's state
__global__ void rk4_stage_kernel(const Rk4Args a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.N || (a.mask && !a.mask[e])) return;
  const rsbk::DevModel& m = *a.model;
  const int nq = a.nq, nv = a.nv;
  float* q = a.gc + (size_t)e * nq;
  float* u = a.gv + (size_t)e * nv;
  const float* q0 = a.q0 + (size_t)e * nq;
  const float* u0 = a.u0 + (size_t)e * nv;
  const float* Minv = a.Minv + (size_t)e * nv * nv;
  const float* h = a.h + (size_t)e * nv;
  const size_t plane = (size_t)a.N * nv;
  float* ka = a.ka + a.stage * plane + (size_t)e * nv;
  float* kv = a.kv + a.stage * plane + (size_t)e * nv;
  float rhs[RSB_MAX_DOF];
  for (int i = 0; i < nv; ++i) {
    float tau = a.tff[(size_t)e * nv + i];
    if (i >= 6) {
      const int b = i - 5;
      if (a.pd) tau += a.kp[i] * (a.pt[(size_t)e * nq + i + 1] - q[i + 1]) + a.kd[i] * (a.dtg[(size_t)e * nv + i] - u[i]);
      const float eff = m.bodyf[b][28];
      if (eff > 0.f && fabsf(tau) > eff) tau = tau > 0.f ? eff : -eff;
      tau -= m.bodyf[b][27] * u[i];
    }
    rhs[i] = tau - h[i];
  }
  const int i0 = m.fixed_base ? 6 : 0;
  for (int i = 0; i < nv; ++i) {
    float s = 0.f;
    for (int j = i0; j < nv; ++j) s += Minv[i * nv + j] * rhs[j];
    ka[i] = i < i0 ? 0.f : s;
  }
  // k_v: the stage velocity; the angular part through dexp^-1 of the stage's rotation vector (zero in stage 0)
  const float c_prev = a.stage == 0 ? 0.f : (a.stage == 3 ? 1.f : 0.5f);
  for (int i = 0; i < nv; ++i) kv[i] = i < i0 ? 0.f : u[i];
  if (a.stage > 0 && !m.fixed_base) {
    const float* kvp = a.kv + (a.stage - 1) * plane + (size_t)e * nv;
    const float th[3] = {c_prev * a.dt * kvp[3], c_prev * a.dt * kvp[4], c_prev * a.dt * kvp[5]};
    const float w[3] = {u[3], u[4], u[5]};
    const float c1[3] = {th[1] * w[2] - th[2] * w[1], th[2] * w[0] - th[0] * w[2], th[0] * w[1] - th[1] * w[0]};
    const float c2[3] = {th[1] * c1[2] - th[2] * c1[1], th[2] * c1[0] - th[0] * c1[2], th[0] * c1[1] - th[1] * c1[0]};
    for (int i = 0; i < 3; ++i) kv[3 + i] = w[i] - 0.5f * c1[i] + (1.0f / 12.0f) * c2[i];
  }
  if (a.stage < 3) {
    const float c = a.stage == 2 ? 1.f : 0.5f;
    float th[RSB_MAX_DOF];
    for (int i = 0; i < nv; ++i) th[i] = c * a.dt * kv[i];
    float ka_l[RSB_MAX_DOF];
    for (int i = 0; i < nv; ++i) ka_l[i] = ka[i];
    config_add(q0, th, nq, nv, m.fixed_base, q);
    for (int i = 0; i < nv; ++i) u[i] = u0[i] + c * a.dt * ka_l[i];
  }
}

struct Rk4Final {
  const rsbk::DevModel* model;
  float *gc, *gv, *tff_eff;
  const float *q0, *u0, *M0, *h0, *ka, *kv;
  float *theta, *du;
  const uint8_t* mask;
  int N, nq, nv;
  float dt;
};
// the combination of the four stages; the state goes back to (q0, u0); the generalized force that makes the step kernel's free motion equal du
__global__ void rk4_combine_kernel(const Rk4Final a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.N || (a.mask && !a.mask[e])) return;
  const rsbk::DevModel& m = *a.model;
  const int nq = a.nq, nv = a.nv;
  const size_t plane = (size_t)a.N * nv, r = (size_t)e * nv;
  float du[RSB_MAX_DOF];
  for (int i = 0; i < nv; ++i) {
    du[i] = a.dt * (1.0f / 6.0f) * (a.ka[r + i] + 2.f * a.ka[plane + r + i] + 2.f * a.ka[2 * plane + r + i] + a.ka[3 * plane + r + i]);
    a.theta[r + i] = a.dt * (1.0f / 6.0f) * (a.kv[r + i] + 2.f * a.kv[plane + r + i] + 2.f * a.kv[2 * plane + r + i] + a.kv[3 * plane + r + i]);
    a.du[r + i] = du[i];
  }
  const float* M = a.M0 + (size_t)e * nv * nv;
  const float idt = 1.0f / a.dt;
  for (int i = 0; i < nv; ++i) {
    float s = a.h0[r + i];
    for (int j = 0; j < nv; ++j) s += M[i * nv + j] * du[j] * idt;
    if (i >= 6) s += m.bodyf[i - 5][27] * a.u0[r + i];       // the step kernel subtracts the passive damping of u0 again
    a.tff_eff[r + i] = s;
  }
  for (int i = 0; i < nq; ++i) a.gc[(size_t)e * nq + i] = a.q0[(size_t)e * nq + i];
  for (int i = 0; i < nv; ++i) a.gv[r + i] = a.u0[r + i];
}
// q+ = q0 (+) (Theta + dt (u+ - u0 - du)) after the step kernel has left u+ in gv
__global__ void rk4_position_kernel(float* gc, const float* gv, const float* q0, const float* u0, const float* theta, const float* du, const uint8_t* mask,
                                    const rsbk::DevModel* model, int N, int nq, int nv, float dt) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N || (mask && !mask[e])) return;
  const size_t r = (size_t)e * nv;
  float th[RSB_MAX_DOF];
  for (int i = 0; i < nv; ++i) th[i] = theta[r + i] + dt * (gv[r + i] - u0[r + i] - du[r + i]);
  config_add(q0 + (size_t)e * nq, th, nq, nv, model->fixed_base, gc + (size_t)e * nq);
}

}  // namespace

int rk4_integrate(rsb_world* w, int nsub) {
  const size_t N = w->N, nq = w->blob.nq, nv = w->blob.nv;
  const rsb_world::Fuse f = w->fuse;
  if (f.ptarget_src || f.act || f.obs_out || f.do_reset || f.env_task || f.peer) {
    w->fuse = rsb_world::Fuse();
    rsb::set_error("RUNGE_KUTTA_4: plain integrate() calls only (rsb_integrate, rsb_integrate_masked, the World views) - the fused control step / env step run the one-evaluation schemes");
    return RSB_E_UNSUPPORTED;
  }
  hipStream_t s = stream_of(w);
  const uint8_t* mask = w->launch_mask;
  if (!w->d_rk) {
    // q0 | u0 | ka [4] | kv [4] | theta | du | h0 | tff saved | M0
    const size_t floats = N * nq + N * nv * (1 + 4 + 4 + 1 + 1 + 1 + 1) + N * nv * nv;
    HIP_TRY(hipMalloc(&w->d_rk, floats * sizeof(float)));
  }
  float* q0 = w->d_rk; float* u0 = q0 + N * nq; float* ka = u0 + N * nv; float* kv = ka + 4 * N * nv; float* theta = kv + 4 * N * nv;
  float* du = theta + N * nv; float* h0 = du + N * nv; float* tff_save = h0 + N * nv; float* M0 = tff_save + N * nv;
  const int blocks = (int)((N + 63) / 64);
  // One sub-step; on ANY failure part-way through the world is put back to the state the sub-step started from (gc / gv used to be left at
  // a stage state, d_tff at tau_eff, the launch mask set - the next integrate() then ran from corrupted inputs without a word)
  bool have_q0 = false, tff_dirty = false;
  auto substep = [&]() -> int {
    have_q0 = false; tff_dirty = false;
    HIP_TRY(hipMemcpyAsync(q0, w->d_gc, N * nq * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(u0, w->d_gv, N * nv * sizeof(float), hipMemcpyDeviceToDevice, s));
    have_q0 = true;
    for (int stage = 0; stage < 4; ++stage) {
      int st = launch_dynamics_query(w, s);        // M, h, M^-1 of the state in gc / gv
      if (st != RSB_OK) return st;
      if (stage == 0) {
        HIP_TRY(hipMemcpyAsync(M0, w->d_M, N * nv * nv * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(h0, w->d_h, N * nv * sizeof(float), hipMemcpyDeviceToDevice, s));
      }
      Rk4Args a;
      a.model = w->d_model; a.gc = w->d_gc; a.gv = w->d_gv; a.q0 = q0; a.u0 = u0; a.Minv = w->d_Minv; a.h = w->d_h;
      a.kp = w->d_kp; a.kd = w->d_kd; a.pt = w->d_pt; a.dtg = w->d_dt; a.tff = w->d_tff; a.ka = ka; a.kv = kv; a.mask = mask;
      a.N = (int)N; a.nq = (int)nq; a.nv = (int)nv; a.stage = stage; a.pd = w->control_mode == RSB_PD_PLUS_FEEDFORWARD_TORQUE ? 1 : 0; a.dt = (float)w->dt;
      hipLaunchKernelGGL(rk4_stage_kernel, dim3(blocks), dim3(64), 0, s, a);
      HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(tff_save, w->d_tff, N * nv * sizeof(float), hipMemcpyDeviceToDevice, s));
    Rk4Final c;
    c.model = w->d_model; c.gc = w->d_gc; c.gv = w->d_gv; c.tff_eff = w->d_tff; c.q0 = q0; c.u0 = u0; c.M0 = M0; c.h0 = h0; c.ka = ka; c.kv = kv;
    c.theta = theta; c.du = du; c.mask = mask; c.N = (int)N; c.nq = (int)nq; c.nv = (int)nv; c.dt = (float)w->dt;
    tff_dirty = true;       // (the combine kernel overwrites d_tff with tau_eff)
    hipLaunchKernelGGL(rk4_combine_kernel, dim3(blocks), dim3(64), 0, s, c);
    HIP_TRY(hipGetLastError());
    // the contact step: the one-evaluation kernel in force mode, without the effort clip (tau_eff carries inertial terms), semi-implicit positions
    const int mode = w->control_mode;
    const double theta_keep = w->integ_theta;
    w->control_mode = RSB_FORCE_AND_TORQUE; w->rk4_inner = true; w->integ_theta = 1.0; w->image_dirty = true;
    w->launch_mask = mask;
    const bool tz = w->tff_zero;
    w->tff_zero = false;          // (tau_eff lives in d_tff for this launch)
    int st = do_integrate(w, 1);
    w->tff_zero = tz;
    w->control_mode = mode; w->rk4_inner = false; w->integ_theta = theta_keep; w->image_dirty = true;
    if (st != RSB_OK) return st;
    hipLaunchKernelGGL(rk4_position_kernel, dim3(blocks), dim3(64), 0, s, w->d_gc, w->d_gv, q0, u0, theta, du, mask, w->d_model, (int)N, (int)nq, (int)nv, (float)w->dt);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(w->d_tff, tff_save, N * nv * sizeof(float), hipMemcpyDeviceToDevice, s));
    tff_dirty = false;
    return RSB_OK;
  };
  int status = RSB_OK;
  for (int sub = 0; sub < nsub && status == RSB_OK; ++sub) status = substep();
  if (status != RSB_OK) {
    const std::string msg = rsb::last_error();
    if (have_q0) {
      (void)hipMemcpyAsync(w->d_gc, q0, N * nq * sizeof(float), hipMemcpyDeviceToDevice, s);
      (void)hipMemcpyAsync(w->d_gv, u0, N * nv * sizeof(float), hipMemcpyDeviceToDevice, s);
    }
    if (tff_dirty) (void)hipMemcpyAsync(w->d_tff, tff_save, N * nv * sizeof(float), hipMemcpyDeviceToDevice, s);
    (void)hipGetLastError();
    rsb::set_error(msg + " (RUNGE_KUTTA_4: the sub-step was rolled back)");
  }
  w->launch_mask = nullptr;
  w->integrate1_valid = false; w->env_ob_valid = false;
  return status;
}

}  // namespace rsbw
