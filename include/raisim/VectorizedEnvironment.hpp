// raisim/VectorizedEnvironment.hpp — the two batched counterparts of raisimGymTorch's VectorizedEnvironment<ENV>.
//
// Upstream (raisimGymTorch/env/VectorizedEnvironment.hpp, absent from /root/reference — SURVEY.md §3.1, §8b) owns
// num_envs ENVIRONMENT objects, each with its own raisim::World, and fans `step` out with an OpenMP parallel-for.
//
//   raisim::VectorizedEnvironment<ChildEnvironment>   upstream's template, same constructor (resourceDir, cfg yaml text),
//       same methods and in-place caller-owned buffers ((T*, rows, cols) spans where upstream takes Eigen::Ref).  The
//       ChildEnvironment is ARBITRARY user code written against raisim::World / ArticulatedSystem (an rsg_anymal-style
//       Environment.hpp): its N instances are constructed inside a raisim::BatchScope, so their N Worlds are the N
//       replicas of ONE BatchedWorld, and step() runs the N env->step() bodies as fibers (raisim/Fiber.hpp).  World::integrate()
//       only records a sub-step; the first read after it parks the fiber, and once every env is parked the batch is flushed:
//       the control_dt / simulation_dt integrate() calls of an rsg_anymal-style step() are ONE fused kernel launch for the
//       whole batch (the launch the benchmark times), bracketed by one upload of the staged PD targets and one download of
//       what the environments read (state, contacts, generalized force) - a single rsb_view_exchange per control step.
//
//   raisim::DeviceVectorizedEnvironment   rsg_anymal's task compiled into the library (rsb_env_*): action scaling,
//       observation, reward, termination and reset run on the GPU, a control step is one fused launch of
//       control_dt/simulation_dt sub-steps (reward, termination, reset and the next observation in its epilogue), and
//       `stepDevice` / `observeDevice` take device buffers
//       so that a GPU-resident policy never crosses PCIe.  Task semantics [RECALL rsg_anymal]: action -> PD position
//       targets (actionMean + action*actionStd on the actuated joints), observation = [height, third row of the base
//       rotation (3), joint angles, body lin vel(3), body ang vel(3), joint velocities] (obDim = 10 + 2*nJoints),
//       reward = forward velocity - torque cost, termination on any non-foot contact followed by reset.
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "raisim/RaisimGymEnv.hpp"
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

class DeviceVectorizedEnvironment {
 public:
  DeviceVectorizedEnvironment(const std::string& urdfPath, const VecEnvConfig& cfg) : cfg_(cfg), world_(urdfPath, cfg.num_envs, cfg.device) {}
  ~DeviceVectorizedEnvironment() { if (dW_) rsb_device_free(world_.handle(), dW_); if (dBias_) rsb_device_free(world_.handle(), dBias_); for (void* d : mlpDev_) rsb_device_free(world_.handle(), d); }
  DeviceVectorizedEnvironment(const DeviceVectorizedEnvironment&) = delete;
  DeviceVectorizedEnvironment& operator=(const DeviceVectorizedEnvironment&) = delete;

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

  // ---- the policy in the loop ON THE DEVICE (round 5; include/rsb_pipeline.h).  K control steps with an action stage between every two, handed
  // over env block by env block: with setStepPipelining(true) consecutive steps overlap although each step's actions depend on the one before.
  /// consecutive control steps overlap on the device (bit-identical results); returns what the library granted (false under a serialising profiler)
  bool setStepPipelining(bool on) { RSB_CHECK(rsb_set_step_pipelining(world_.handle(), on ? 1 : 0)); return rsb_step_pipelining_enabled(world_.handle()) != 0; }
  /// round 6: K control steps of rolloutLinear / rolloutMlp run as ONE resident launch of the step kernel - the env blocks stay in LDS, each block's own wave
  /// evaluates the policy between two control steps (rsb_set_step_residency; bit-identical results).  Returns whether this world has a resident kernel class.
  bool setStepResidency(bool on) { RSB_CHECK(rsb_set_step_residency(world_.handle(), on ? 1 : 0)); return rsb_step_residency_status(world_.handle(), 1) != 0; }
  /// specialised step kernels (rsb_ext.h: RSB_SPEC_OFF / RSB_SPEC_CACHED (default) / RSB_SPEC_COMPILE; bit-identical results)
  void setKernelSpecialization(int mode) { RSB_CHECK(rsb_set_specialization(world_.handle(), mode)); }
  /// K control steps with the CALLER's stage kernel (a HIP kernel built around rsb_stage::serve).  Nothing synchronises.
  void closedLoopRun(int steps, rsb_stage_launch_fn launch, void* user) { RSB_CHECK(rsb_closed_loop_run(world_.handle(), steps, launch, user)); }
  /// K control steps with the in-repo linear policy  action = clip(bias + W ob):  W [actionDim, obDim] row-major and bias [actionDim] are HOST
  /// arrays, uploaded at EVERY call - a learner updates its weights in place, so the pointers say nothing about the contents; the copy is
  /// 1.6 KB next to a K-step run.  clip <= 0: none.  The run itself does not synchronise.
  void rolloutLinear(int steps, const float* W, const float* bias = nullptr, float clip = 0.f) {
    RSFATAL_IF(!W, "rolloutLinear: W is null");
    const size_t wb = (size_t)actionDim_ * obDim_ * sizeof(float), bb = (size_t)actionDim_ * sizeof(float);
    if (!dW_) { RSB_CHECK(rsb_device_alloc(world_.handle(), wb, &dW_)); RSB_CHECK(rsb_device_alloc(world_.handle(), bb, &dBias_)); }
    RSB_CHECK(rsb_device_copy(world_.handle(), dW_, W, wb, 0));
    if (bias) RSB_CHECK(rsb_device_copy(world_.handle(), dBias_, bias, bb, 0));
    rsb_linear_policy p{};
    p.W = static_cast<const float*>(dW_); p.bias = bias ? static_cast<const float*>(dBias_) : nullptr; p.clip = clip;
    RSB_CHECK(rsb_closed_loop_run_linear(world_.handle(), steps, &p));
  }
  /// K control steps with an ACTOR NETWORK in the loop (the in-repo MLP stage, rsb_closed_loop_run_mlp): layer l = (weights[l] [dims[l + 1], dims[l]]
  /// row-major as torch.nn.Linear stores them, biases[l] [dims[l + 1]] or null), HOST arrays, uploaded (weights transposed) at EVERY call (in-place updates of
  /// the learner's buffers are the normal case: ADVICE r05; 89 KB for 34-128-128-12); the device buffers are re-allocated only when `dims` changes;
  /// dims.front() = obDim, dims.back() = actionDim, widths <= 256; activation RSB_ACT_TANH / _RELU / _LEAKY_RELU on the hidden layers.
  void rolloutMlp(int steps, const std::vector<int>& dims, const std::vector<const float*>& weights, const std::vector<const float*>& biases,
                  int activation = RSB_ACT_LEAKY_RELU, float clip = 0.f) {
    const int L = (int)dims.size() - 1;
    RSFATAL_IF(L < 1 || L > RSB_MLP_MAX_LAYERS || (int)weights.size() != L || (int)biases.size() != L, "rolloutMlp: 1 .. 4 layers, one weight and one bias pointer per layer");
    for (int l = 0; l < L; ++l) RSFATAL_IF(!weights[l], "rolloutMlp: a layer's weight pointer is null");
    if (dims != mlpDims_) {      // device buffers: one weight and one bias buffer per layer, sized by dims
      for (void* d : mlpDev_) rsb_device_free(world_.handle(), d);
      mlpDev_.assign(2 * (size_t)L, nullptr);
      for (int l = 0; l < L; ++l) {
        RSB_CHECK(rsb_device_alloc(world_.handle(), (size_t)dims[l] * dims[l + 1] * sizeof(float), &mlpDev_[2 * l]));
        RSB_CHECK(rsb_device_alloc(world_.handle(), (size_t)dims[l + 1] * sizeof(float), &mlpDev_[2 * l + 1]));
      }
      mlpDims_ = dims;
    }
    mlp_ = rsb_mlp_policy{};
    mlp_.n_layers = L;
    for (int l = 0; l <= L; ++l) mlp_.dims[l] = dims[l];
    for (int l = 0; l < L; ++l) {
      const int in = dims[l], out = dims[l + 1];
      mlpStage_.resize((size_t)in * out);
      for (int o = 0; o < out; ++o) for (int i = 0; i < in; ++i) mlpStage_[(size_t)i * out + o] = weights[l][(size_t)o * in + i];
      RSB_CHECK(rsb_device_copy(world_.handle(), mlpDev_[2 * l], mlpStage_.data(), mlpStage_.size() * sizeof(float), 0));      // (joins and waits: the staging vector is free again)
      mlp_.Wt[l] = static_cast<const float*>(mlpDev_[2 * l]);
      if (biases[l]) {
        RSB_CHECK(rsb_device_copy(world_.handle(), mlpDev_[2 * l + 1], biases[l], (size_t)out * sizeof(float), 0));
        mlp_.bias[l] = static_cast<const float*>(mlpDev_[2 * l + 1]);
      }
    }
    mlp_.activation = activation; mlp_.leaky_slope = 0.01f; mlp_.clip = clip;
    RSB_CHECK(rsb_closed_loop_run_mlp(world_.handle(), steps, &mlp_));
  }
  /// waits for everything in flight; RSB_OK, or RSB_E_PIPELINE once after a pipeline fault (the steps were then replayed in lock-step: results are valid)
  int join() { return rsb_step_pipeline_join(world_.handle()); }

  void isTerminalState(bool* terminalState) { for (int e = 0; e < n_; ++e) terminalState[e] = done_[e] != 0; }
  void setSeed(int) {}            // the simulation is deterministic; randomness lives in the caller's actions
  void close() {}
  void curriculumUpdate() {}      // rsg_anymal has no curriculum
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
  void* dW_ = nullptr; void* dBias_ = nullptr;       // rolloutLinear's weights on the device (freed with the world's context)
  rsb_mlp_policy mlp_{};                             // rolloutMlp's network on the device
  std::vector<void*> mlpDev_; std::vector<int> mlpDims_; std::vector<float> mlpStage_;
};

/// Upstream's template: N arbitrary ChildEnvironment objects on one GPU batch (see the header comment).
template <class ChildEnvironment>
class VectorizedEnvironment {
 public:
  explicit VectorizedEnvironment(std::string resourceDir, std::string cfg, bool normalizeObservation = true)
      : resourceDir_(std::move(resourceDir)), cfgString_(std::move(cfg)), normalizeObservation_(normalizeObservation) {
    Yaml::Parse(cfg_, cfgString_);
    if (!cfg_["render"].IsNone()) render_ = cfg_["render"].template As<bool>();
    init();
  }
  ~VectorizedEnvironment() { for (auto* env : environments_) delete env; }
  VectorizedEnvironment(const VectorizedEnvironment&) = delete;
  VectorizedEnvironment& operator=(const VectorizedEnvironment&) = delete;

  const std::string& getResourceDir() const { return resourceDir_; }
  const std::string& getCfgString() const { return cfgString_; }

  void init() {
    if (!environments_.empty()) return;
    num_envs_ = cfg_["num_envs"].template As<int>();
    const int device = cfg_["device"].template As<int>(0);      // (new key) GPU that holds the batch
    // cfg["num_threads"] (upstream: the OpenMP team of the per-env loop): host threads that run the N step() bodies between two
    // flushes of the batch, capped by the machine; RSB_FIBER_THREADS overrides it, 1 = everything on the caller's thread
    threads_ = cfg_["num_threads"].IsNone() ? 1 : cfg_["num_threads"].template As<int>();
    if (const char* ft = std::getenv("RSB_FIBER_THREADS")) threads_ = std::atoi(ft);
    threads_ = std::max(1, std::min(threads_, (int)std::max(1u, std::thread::hardware_concurrency())));
    environments_.reserve(num_envs_);
    rewardInformation_.reserve(num_envs_);
    {
      BatchScope scope(num_envs_, device);     // the N Worlds constructed below become the N replicas of one BatchedWorld
      for (int i = 0; i < num_envs_; i++) {
        environments_.push_back(new ChildEnvironment(resourceDir_, cfg_, render_ && i == 0));
        environments_.back()->setSimulationTimeStep(cfg_["simulation_dt"].template As<double>());
        environments_.back()->setControlTimeStep(cfg_["control_dt"].template As<double>());
        rewardInformation_.push_back(environments_.back()->getRewards().getStdMap());
      }
      batch_ = scope.shared();
    }
    setSeed(0);
    for (int i = 0; i < num_envs_; i++) {
      environments_[i]->init();
      environments_[i]->reset();
    }
    obDim_ = environments_[0]->getObDim();
    actionDim_ = environments_[0]->getActionDim();
    RSFATAL_IF(obDim_ == 0 || actionDim_ == 0, "Observation/Action dimension must be defined in the constructor of each environment!");
    if (normalizeObservation_) {
      obMean_.assign(obDim_, 0.f); obVar_.assign(obDim_, 1.f);
      recentMean_.assign(obDim_, 0.f); recentVar_.assign(obDim_, 0.f); delta_.assign(obDim_, 0.f);
    }
  }

  // resets all environments and returns observation
  void reset() { for (auto env : environments_) env->reset(); }

  /// ob: float [num_envs, obDim] row-major (upstream: Eigen::Ref<EigenRowMajorMat>&)
  void observe(float* ob, int rows, int cols, bool updateStatistics) {
    RSFATAL_IF(rows != num_envs_ || cols != obDim_, "observe: buffer must be [num_envs, obDim]");
    fibers_.forEach(num_envs_, [&](int i) { environments_[i]->observe(rowOf(ob + (size_t)i * obDim_, obDim_)); }, threads_);   // (upstream: an OpenMP parallel-for)
    if (normalizeObservation_) updateObservationStatisticsAndNormalize(ob, updateStatistics);
  }

  /// action: float [num_envs, actionDim]; reward: float [num_envs]; done: bool [num_envs] - written in place.
  /// The N step() bodies run as fibers; each World::integrate() inside them is one launch for the whole batch.
  void step(const float* action, int rows, int cols, float* reward, bool* done) {
    RSFATAL_IF(rows != num_envs_ || cols != actionDim_, "step: action must be [num_envs, actionDim]");
    auto body = [&](int i) { perAgentStep(i, action, reward, done); };
    if (!batch_) { for (int i = 0; i < num_envs_; i++) body(i); return; }     // envs that never created a World
    struct Guard { BatchedWorld* b; ~Guard() { b->setFiberBatch(false); b->abortViews(); } } guard{batch_.get()};   // (after a clean run nothing is pending)
    batch_->setFiberBatch(true);
    const auto t0 = std::chrono::steady_clock::now();
    long long inFlush = 0;
    fibers_.run(num_envs_, body, [this, &inFlush] {
      const auto f0 = std::chrono::steady_clock::now();
      batch_->flushViews();
      inFlush += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - f0).count();
      ++stepProfile_.flushes;
    }, threads_);
    batch_->flushViews();      // integrate() calls of bodies that ended without reading anything afterwards
    stepProfile_.total_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    stepProfile_.flush_ns += inFlush;
    ++stepProfile_.steps;
  }
  /// (new) where step() spends the host's time, accumulated: total, inside the flushes (BatchedWorld::flushPrepNs / flushExchangeNs split those further;
  /// the rest = the rounds of the N step() bodies on the fiber threads + the scheduler's hand-overs).  tools/prof_template_path.py prints the table.
  struct StepProfile { long long total_ns = 0, flush_ns = 0, steps = 0, flushes = 0; };
  const StepProfile& stepProfile() const { return stepProfile_; }
  void resetStepProfile() { stepProfile_ = StepProfile(); }

  void turnOnVisualization() { if (render_) environments_[0]->turnOnVisualization(); }
  void turnOffVisualization() { if (render_) environments_[0]->turnOffVisualization(); }
  void startRecordingVideo(const std::string& videoName) { if (render_) environments_[0]->startRecordingVideo(videoName); }
  void stopRecordingVideo() { if (render_) environments_[0]->stopRecordingVideo(); }
  void getObStatistics(float* mean, float* var, float& count) {
    for (int i = 0; i < obDim_; ++i) { mean[i] = obMean_[i]; var[i] = obVar_[i]; }
    count = obCount_;
  }
  void setObStatistics(const float* mean, const float* var, float count) {
    obMean_.assign(mean, mean + obDim_); obVar_.assign(var, var + obDim_); obCount_ = count;
  }
  void setSeed(int seed) { int seed_inc = seed; for (auto* env : environments_) env->setSeed(seed_inc++); }
  void close() { for (auto* env : environments_) env->close(); }
  void isTerminalState(bool* terminalState) {
    for (int i = 0; i < num_envs_; i++) { float terminalReward; terminalState[i] = environments_[i]->isTerminalState(terminalReward); }
  }
  void setSimulationTimeStep(double dt) { for (auto* env : environments_) env->setSimulationTimeStep(dt); }
  void setControlTimeStep(double dt) { for (auto* env : environments_) env->setControlTimeStep(dt); }
  int getObDim() { return obDim_; }
  int getActionDim() { return actionDim_; }
  int getNumOfEnvs() { return num_envs_; }
  int getNumOfThreads() const { return threads_; }     ///< (new) host threads the N step() bodies are dealt to
  void curriculumUpdate() { for (auto* env : environments_) env->curriculumUpdate(); }
  const std::vector<std::map<std::string, float>>& getRewardInfo() { return rewardInformation_; }

  /// (new) the batch behind the environments and how many launches its views have issued so far
  BatchedWorld* batch() { return batch_.get(); }
  ChildEnvironment* environment(int i) { return environments_[i]; }

 private:
  void updateObservationStatisticsAndNormalize(float* ob, bool updateStatistics) {
    const int n = num_envs_, d = obDim_;
    if (updateStatistics) {
      for (int j = 0; j < d; ++j) { double m = 0; for (int i = 0; i < n; ++i) m += ob[(size_t)i * d + j]; recentMean_[j] = (float)(m / n); }
      for (int j = 0; j < d; ++j) { double v = 0; for (int i = 0; i < n; ++i) { const double x = ob[(size_t)i * d + j] - recentMean_[j]; v += x * x; } recentVar_[j] = (float)(v / n); }
      const float totCount = obCount_ + n;
      for (int j = 0; j < d; ++j) {
        delta_[j] = obMean_[j] - recentMean_[j];
        delta_[j] = delta_[j] * delta_[j];
        obMean_[j] = obMean_[j] * (obCount_ / totCount) + recentMean_[j] * (n / totCount);
        obVar_[j] = (obVar_[j] * obCount_ + recentVar_[j] * n + delta_[j] * (obCount_ * n / totCount)) / totCount;
      }
      obCount_ = totCount;
    }
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < d; ++j) ob[(size_t)i * d + j] = (ob[(size_t)i * d + j] - obMean_[j]) / std::sqrt(obVar_[j] + 1e-8f);
  }

  inline void perAgentStep(int agentId, const float* action, float* reward, bool* done) {
    reward[agentId] = environments_[agentId]->step(rowOf(action + (size_t)agentId * actionDim_, actionDim_));
    rewardInformation_[agentId] = environments_[agentId]->getRewards().getStdMap();
    float terminalReward = 0;
    done[agentId] = environments_[agentId]->isTerminalState(terminalReward);
    if (done[agentId]) {
      environments_[agentId]->reset();
      reward[agentId] += terminalReward;
    }
  }

  std::vector<ChildEnvironment*> environments_;
  std::vector<std::map<std::string, float>> rewardInformation_;
  std::shared_ptr<BatchedWorld> batch_;
  detail::FiberScheduler fibers_;
  StepProfile stepProfile_;
  int num_envs_ = 1, obDim_ = 0, actionDim_ = 0, threads_ = 1;
  bool recordVideo_ = false, render_ = false;
  std::string resourceDir_;
  Yaml::Node cfg_;
  std::string cfgString_;
  /// observation running mean
  bool normalizeObservation_ = true;
  std::vector<float> obMean_, obVar_, recentMean_, recentVar_, delta_;
  float obCount_ = 1e-4f;
};

}  // namespace raisim
