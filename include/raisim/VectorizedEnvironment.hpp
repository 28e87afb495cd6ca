// raisim/VectorizedEnvironment.hpp — the batched counterpart of raisimGymTorch's VectorizedEnvironment<ENV>.
//
// Upstream (raisimGymTorch/env/VectorizedEnvironment.hpp, absent from /root/reference — SURVEY.md §3.1, §8b) owns
// num_envs ENVIRONMENT objects, each with its own raisim::World, and fans `step` out with an OpenMP parallel-for.
// Here ONE BatchedWorld holds all replicas on the GPU; `step` is a single fused launch of control_dt/simulation_dt
// sub-steps plus two tiny kernels (action -> PD targets, reward/termination/reset), all through the C-ABI's rsb_env_*
// entry points, and `stepDevice` / `observeDevice` take device buffers so a GPU-resident policy never crosses PCIe; the method names, argument meaning and in-place caller-owned buffers
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
  bool early_termination = false;         // rsb_set_early_termination: not upstream's rule, see include/rsb.h
};

class VectorizedEnvironment {
 public:
  VectorizedEnvironment(const std::string& urdfPath, const VecEnvConfig& cfg) : cfg_(cfg), world_(urdfPath, cfg.num_envs, cfg.device) {}

  void init() {
    n_ = world_.numEnvs(); nq_ = world_.gcDim(); nv_ = world_.dof(); nj_ = nv_ - 6;
    obDim_ = 10 + 2 * nj_; actionDim_ = nj_;
    world_.setTimeStep(cfg_.simulation_dt);
    world_.addGround(0.0);
    RSB_CHECK(rsb_set_early_termination(world_.handle(), cfg_.early_termination ? 1 : 0));
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
    done_.assign(n_, 0);
    // the task itself (action scaling, observation, reward, termination, reset) runs on the GPU: rsb_env_*
    std::vector<float> pt0((size_t)n_ * nq_, 0.f), dt0((size_t)n_ * nv_, 0.f);
    for (int e = 0; e < n_; ++e) pt0[(size_t)e * nq_ + 3] = 1.f;
    world_.setPdTarget(pt0.data(), dt0.data());
    configureEnv();
    reset();
  }

  void reset() { RSB_CHECK(rsb_env_reset(world_.handle())); }

  /// ob: float [num_envs, obDim] row-major, written in place (updateStatistics is accepted for source compatibility)
  void observe(float* ob, int rows, int cols, bool /*updateStatistics*/ = false) {
    RSFATAL_IF(rows != n_ || cols != obDim_, "observe: buffer must be [num_envs, obDim]");
    RSB_CHECK(rsb_env_observe(world_.handle(), ob, RSB_HOST));
  }
  /// the same with a device buffer (e.g. a torch CUDA tensor's data_ptr): nothing crosses PCIe, nothing synchronises
  void observeDevice(float* ob_device) { RSB_CHECK(rsb_env_observe(world_.handle(), ob_device, RSB_DEVICE)); }

  /// action: float [num_envs, actionDim]; reward: float [num_envs]; done: bool [num_envs] — all written in place
  void step(const float* action, int rows, int cols, float* reward, bool* done) {
    RSFATAL_IF(rows != n_ || cols != actionDim_, "step: action must be [num_envs, actionDim]");
    RSB_CHECK(rsb_env_step(world_.handle(), action, reward, done_.data(), nullptr, RSB_HOST));   // action kernel + ONE fused launch + reward/reset kernel
    for (int e = 0; e < n_; ++e) done[e] = done_[e] != 0;
  }
  /// device buffers: action float [num_envs, actionDim], reward float [num_envs], done uint8 [num_envs]
  /// ob_next_device (optional, float [num_envs, obDim]): the observation the next step starts from, from the same launch
  void stepDevice(const float* action_device, float* reward_device, uint8_t* done_device, float* ob_next_device = nullptr) {
    RSB_CHECK(rsb_env_step(world_.handle(), action_device, reward_device, done_device, ob_next_device, RSB_DEVICE));
  }

  void isTerminalState(bool* terminalState) { for (int e = 0; e < n_; ++e) terminalState[e] = done_[e] != 0; }
  void setSeed(int) {}
  void close() {}
  void curriculumUpdate() {}
  void turnOnVisualization() {}
  void turnOffVisualization() {}
  void setSimulationTimeStep(double dt) { cfg_.simulation_dt = dt; world_.setTimeStep(dt); substeps_ = (int)(cfg_.control_dt / dt + 1e-10); configureEnv(); }
  void setControlTimeStep(double dt) { cfg_.control_dt = dt; substeps_ = (int)(dt / cfg_.simulation_dt + 1e-10); configureEnv(); }
  int getObDim() const { return obDim_; }
  int getActionDim() const { return actionDim_; }
  int getNumOfEnvs() const { return n_; }
  BatchedWorld& world() { return world_; }

 private:
  void configureEnv() {
    rsb_env_config ec{};
    ec.n_substeps = substeps_;
    ec.action_std = (float)cfg_.action_std;
    ec.forward_vel_coeff = (float)cfg_.forward_vel_reward_coeff; ec.forward_vel_clip = 4.0f;
    ec.torque_coeff = (float)cfg_.torque_reward_coeff; ec.terminal_reward = (float)cfg_.terminal_reward;
    ec.n_foot = (int)feet_.size();
    for (size_t i = 0; i < feet_.size(); ++i) ec.foot_collisions[i] = feet_[i];
    RSB_CHECK(rsb_env_configure(world_.handle(), &ec, gcInit_.data() + 7, gcInit_.data(), gvInit_.data()));
  }
  VecEnvConfig cfg_;
  BatchedWorld world_;
  int n_ = 0, nq_ = 0, nv_ = 0, nj_ = 0, obDim_ = 0, actionDim_ = 0, substeps_ = 4;
  std::vector<float> gcInit_, gvInit_;
  std::vector<int32_t> feet_;
  std::vector<uint8_t> done_;
};

}  // namespace raisim
