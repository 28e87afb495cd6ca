// raisim/VectorizedEnvironment.hpp — the batched counterpart of raisimGymTorch's VectorizedEnvironment<ENV>.
//
// Upstream (raisimGymTorch/env/VectorizedEnvironment.hpp, absent from /root/reference — SURVEY.md §3.1, §8b) owns
// num_envs ENVIRONMENT objects, each with its own raisim::World, and fans `step` out with an OpenMP parallel-for.
// Here ONE BatchedWorld holds all replicas on the GPU and `step` is a single fused launch of
// control_dt/simulation_dt sub-steps; the method names, argument meaning and in-place caller-owned buffers
// (row-major float [num_envs, dim], bool [num_envs]) are upstream's, with (T*, rows, cols) spans instead of
// Eigen::Ref (Eigen is not available here).
//
// Task semantics are the rsg_anymal ones [RECALL]: action -> PD position targets (actionMean + action*actionStd on
// the actuated joints), observation = [height, body z-axis(3), joint angles, body lin vel(3), body ang vel(3),
// joint velocities] (obDim = 10 + 2*nJoints), reward = forward velocity - torque cost (coefficients in Config),
// termination on any non-foot contact followed by reset to the initial state.
#pragma once

#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "raisim/World.hpp"

namespace raisim {

struct VecEnvConfig {
  int num_envs = 4096;
  double simulation_dt = 0.0025, control_dt = 0.01;
  double action_std = 0.3, p_gain = 50.0, d_gain = 0.2;
  double forward_vel_reward_coeff = 0.3, torque_reward_coeff = -4e-5, terminal_reward = -10.0;
  std::vector<double> gc_init;            // size gcDim; default set by the constructor for ANYmal-like models
  std::vector<std::string> foot_collision_suffixes = {"_foot"};
  int device = 0;
};

class VectorizedEnvironment {
 public:
  VectorizedEnvironment(const std::string& urdfPath, const VecEnvConfig& cfg) : cfg_(cfg), world_(urdfPath, cfg.num_envs, cfg.device) {}

  void init() {
    n_ = world_.numEnvs(); nq_ = world_.gcDim(); nv_ = world_.dof(); nj_ = nv_ - 6;
    obDim_ = 10 + 2 * nj_; actionDim_ = nj_;
    world_.setTimeStep(cfg_.simulation_dt);
    world_.addGround(0.0);
    substeps_ = (int)(cfg_.control_dt / cfg_.simulation_dt + 1e-10);
    std::vector<float> kp(nv_, 0.f), kd(nv_, 0.f);
    for (int i = 6; i < nv_; ++i) { kp[i] = (float)cfg_.p_gain; kd[i] = (float)cfg_.d_gain; }
    world_.setPdGains(kp.data(), kd.data());
    gcInit_.assign(nq_, 0.f); gvInit_.assign(nv_, 0.f);
    if ((int)cfg_.gc_init.size() == nq_) for (int i = 0; i < nq_; ++i) gcInit_[i] = (float)cfg_.gc_init[i];
    else { gcInit_[2] = 0.6f; gcInit_[3] = 1.f; }
    const rsb_model_blob& b = world_.blob();
    for (int c = 0; c < b.ncol; ++c)
      for (const auto& suf : cfg_.foot_collision_suffixes) {
        std::string nm = b.col_name[c];
        if (nm.size() >= suf.size() && nm.compare(nm.size() - suf.size(), suf.size(), suf) == 0) feet_.push_back(c);
      }
    gc_.assign((size_t)n_ * nq_, 0.f); gv_.assign((size_t)n_ * nv_, 0.f);
    pTarget_.assign((size_t)n_ * nq_, 0.f); dTarget_.assign((size_t)n_ * nv_, 0.f);
    done_.assign(n_, 0);
    reset();
  }

  void reset() {
    for (int e = 0; e < n_; ++e) {
      std::copy(gcInit_.begin(), gcInit_.end(), gc_.begin() + (size_t)e * nq_);
      std::copy(gvInit_.begin(), gvInit_.end(), gv_.begin() + (size_t)e * nv_);
    }
    world_.setState(gc_.data(), gv_.data());
  }

  /// ob: float [num_envs, obDim] row-major, written in place (updateStatistics is accepted for source compatibility)
  void observe(float* ob, int rows, int cols, bool /*updateStatistics*/ = false) {
    RSFATAL_IF(rows != n_ || cols != obDim_, "observe: buffer must be [num_envs, obDim]");
    world_.getState(gc_.data(), gv_.data());
    for (int e = 0; e < n_; ++e) writeObs(e, ob + (size_t)e * obDim_);
  }

  /// action: float [num_envs, actionDim]; reward: float [num_envs]; done: bool [num_envs] — all written in place
  void step(const float* action, int rows, int cols, float* reward, bool* done) {
    RSFATAL_IF(rows != n_ || cols != actionDim_, "step: action must be [num_envs, actionDim]");
    for (int e = 0; e < n_; ++e) {
      float* pt = pTarget_.data() + (size_t)e * nq_;
      for (int j = 0; j < nj_; ++j) pt[7 + j] = gcInit_[7 + j] + (float)cfg_.action_std * action[(size_t)e * actionDim_ + j];
    }
    world_.setPdTarget(pTarget_.data(), dTarget_.data());
    world_.integrate(substeps_);                                           // ONE fused launch for all envs
    RSB_CHECK(rsb_reset_terminated(world_.handle(), feet_.data(), (int)feet_.size(), gcInit_.data(), gvInit_.data(), 1,
                                   done_.data(), RSB_HOST));
    world_.getState(gc_.data(), gv_.data());
    for (int e = 0; e < n_; ++e) {
      const float* u = gv_.data() + (size_t)e * nv_;
      const float* q = gc_.data() + (size_t)e * nq_;
      const float* pt = pTarget_.data() + (size_t)e * nq_;
      double torque2 = 0;
      for (int j = 0; j < nj_; ++j) { const double t = cfg_.p_gain * (pt[7 + j] - q[7 + j]) - cfg_.d_gain * u[6 + j]; torque2 += t * t; }
      double r = cfg_.forward_vel_reward_coeff * std::fmin(4.0, bodyVelX(q, u)) + cfg_.torque_reward_coeff * torque2;
      done[e] = done_[e] != 0;
      reward[e] = (float)(done[e] ? cfg_.terminal_reward : r);
    }
  }

  void isTerminalState(bool* terminalState) { for (int e = 0; e < n_; ++e) terminalState[e] = done_[e] != 0; }
  void setSeed(int) {}
  void close() {}
  void curriculumUpdate() {}
  void turnOnVisualization() {}
  void turnOffVisualization() {}
  void setSimulationTimeStep(double dt) { cfg_.simulation_dt = dt; world_.setTimeStep(dt); substeps_ = (int)(cfg_.control_dt / dt + 1e-10); }
  void setControlTimeStep(double dt) { cfg_.control_dt = dt; substeps_ = (int)(dt / cfg_.simulation_dt + 1e-10); }
  int getObDim() const { return obDim_; }
  int getActionDim() const { return actionDim_; }
  int getNumOfEnvs() const { return n_; }
  BatchedWorld& world() { return world_; }

 private:
  static void rotT(const float* q, double Rt[9]) {  // world -> body rotation from the base quaternion
    const double w = q[3], x = q[4], y = q[5], z = q[6];
    Rt[0] = 1 - 2 * (y * y + z * z); Rt[3] = 2 * (x * y - w * z);     Rt[6] = 2 * (x * z + w * y);
    Rt[1] = 2 * (x * y + w * z);     Rt[4] = 1 - 2 * (x * x + z * z); Rt[7] = 2 * (y * z - w * x);
    Rt[2] = 2 * (x * z - w * y);     Rt[5] = 2 * (y * z + w * x);     Rt[8] = 1 - 2 * (x * x + y * y);
  }
  static double bodyVelX(const float* q, const float* u) { double Rt[9]; rotT(q, Rt); return Rt[0] * u[0] + Rt[1] * u[1] + Rt[2] * u[2]; }
  void writeObs(int e, float* ob) const {
    const float* q = gc_.data() + (size_t)e * nq_;
    const float* u = gv_.data() + (size_t)e * nv_;
    double Rt[9];
    rotT(q, Rt);
    int k = 0;
    ob[k++] = q[2];
    ob[k++] = (float)Rt[6]; ob[k++] = (float)Rt[7]; ob[k++] = (float)Rt[8];      // body z-axis in the world = R row 2
    for (int j = 0; j < nj_; ++j) ob[k++] = q[7 + j];
    for (int i = 0; i < 3; ++i) ob[k++] = (float)(Rt[3 * i] * u[0] + Rt[3 * i + 1] * u[1] + Rt[3 * i + 2] * u[2]);
    for (int i = 0; i < 3; ++i) ob[k++] = (float)(Rt[3 * i] * u[3] + Rt[3 * i + 1] * u[4] + Rt[3 * i + 2] * u[5]);
    for (int j = 0; j < nj_; ++j) ob[k++] = u[6 + j];
  }

  VecEnvConfig cfg_;
  BatchedWorld world_;
  int n_ = 0, nq_ = 0, nv_ = 0, nj_ = 0, obDim_ = 0, actionDim_ = 0, substeps_ = 4;
  std::vector<float> gcInit_, gvInit_, gc_, gv_, pTarget_, dTarget_;
  std::vector<int32_t> feet_;
  std::vector<uint8_t> done_;
};

}  // namespace raisim
