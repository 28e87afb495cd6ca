// raisim/RaisimGymEnv.hpp — the base class raisimGymTorch environments derive from, re-authored from recollection
// [RECALL raisimGymTorch/env/RaisimGymEnv.hpp, Reward.hpp; absent from /root/reference, SURVEY.md §8b].
// Same members and virtuals as upstream; Eigen::Ref<EigenVec> is replaced by raisim::EigenVecRef, a (float*, size)
// span with operator[] / size() / setZero() (Eigen is not installed here; where <Eigen/Core> exists the span converts
// from and to Eigen maps).  `server_` is upstream's std::unique_ptr<raisim::RaisimServer>; the class behind it is a no-op (raisim/RaisimServer.hpp:
// visualisation is out of scope) so that an environment's `if (visualizable_) { server_ = std::make_unique<...> ... }` block and its lock / unlock
// pairs around integrate() compile and run as written.
#pragma once

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "raisim/RaisimServer.hpp"
#include "raisim/World.hpp"
#include "raisim/Yaml.hpp"

/// raisimGymTorch's Common.hpp [RECALL]: READ_YAML(double, action_std, cfg_["action_std"])
#define READ_YAML(type, dst, node) { RSFATAL_IF((node).IsNone(), "Node " #node " doesn't exist"); dst = (node).template As<type>(); }

namespace raisim {

/// stand-in for Eigen::Ref<Eigen::Matrix<float, -1, 1>> (one row of the [num_envs, dim] observation / action matrix)
template <typename T>
struct RowRef {
  T* p = nullptr;
  int n = 0;
  RowRef() = default;
  RowRef(T* data, int size) : p(data), n(size) {}
  T& operator[](int i) const { return p[i]; }
  T& operator()(int i) const { return p[i]; }
  int size() const { return n; }
  T* data() const { return p; }
  void setZero() const { for (int i = 0; i < n; ++i) p[i] = T(0); }
#ifdef RAISIM_HAS_EIGEN
  Eigen::Map<Eigen::Matrix<typename std::remove_const<T>::type, Eigen::Dynamic, 1>> e() const { return {const_cast<typename std::remove_const<T>::type*>(p), n}; }
#endif
};
#ifdef RAISIM_HAS_EIGEN
// with Eigen in the include path the boundary carries upstream's own types [RECALL raisimGymTorch/env/RaisimGymEnv.hpp, Common.hpp]:
//   virtual void observe(Eigen::Ref<EigenVec> ob);  virtual float step(const Eigen::Ref<EigenVec>& action);
// EigenVecRef / ConstEigenVecRef name the same types, so an environment written against the spans keeps overriding them
using EigenRowMajorMat = Eigen::Matrix<float, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>;
using EigenVec = Eigen::Matrix<float, Eigen::Dynamic, 1>;
using EigenBoolVec = Eigen::Matrix<bool, Eigen::Dynamic, 1>;
using EigenVecRef = Eigen::Ref<EigenVec>;
using ConstEigenVecRef = Eigen::Ref<EigenVec>;
/// one row of the caller's [num_envs, dim] float matrix as what observe() / step() take
inline Eigen::Map<EigenVec> rowOf(float* p, int n) { return Eigen::Map<EigenVec>(p, n); }
inline Eigen::Map<EigenVec> rowOf(const float* p, int n) { return Eigen::Map<EigenVec>(const_cast<float*>(p), n); }   // (upstream passes rows of a non-const Ref as well)
#else
using EigenVecRef = RowRef<float>;
using ConstEigenVecRef = RowRef<const float>;
inline EigenVecRef rowOf(float* p, int n) { return EigenVecRef(p, n); }
inline ConstEigenVecRef rowOf(const float* p, int n) { return ConstEigenVecRef(p, n); }
#endif

/// upstream raisim::Reward: named reward terms with coefficients read from cfg["reward"]
class Reward {
 public:
  void initializeFromConfigurationFile(const Yaml::Node& cfg) {
    terms_.clear();
    for (const std::string& k : cfg.Keys()) terms_[k] = Term{cfg[k]["coeff"].As<double>(), 0.0};
  }
  void record(const std::string& name, double reward, bool accumulate = false) {
    auto it = terms_.find(name);
    RSFATAL_IF(it == terms_.end(), "Reward::record: no such reward term: " + name);
    RSFATAL_IF(!std::isfinite(reward), "Reward::record: " + name + " is not finite");
    it->second.value = (accumulate ? it->second.value : 0.0) + reward * it->second.coeff;
  }
  float sum() const { double s = 0; for (const auto& t : terms_) s += t.second.value; return (float)s; }
  float operator[](const std::string& name) const { return (float)terms_.at(name).value; }
  void reset() { for (auto& t : terms_) t.second.value = 0.0; }
  /// the terms and their sum as upstream logs them; the map is a member updated in place (upstream returns a reference as well
  /// [RECALL]): VectorizedEnvironment copies it for every env in every step, and a freshly built map cost three allocations each time
  const std::map<std::string, float>& getStdMap() {
    if (log_.size() != terms_.size() + 1) { log_.clear(); for (const auto& t : terms_) log_[t.first] = 0.f; sumIt_ = log_.emplace("reward_sum", 0.f).first; }
    auto it = log_.begin();
    for (const auto& t : terms_) { if (it == sumIt_) ++it; it->second = (float)t.second.value; ++it; }   // (both maps iterate in key order)
    sumIt_->second = sum();
    return log_;
  }
 private:
  struct Term { double coeff, value; };
  std::map<std::string, Term> terms_;
  std::map<std::string, float> log_;
  std::map<std::string, float>::iterator sumIt_;
};

class RaisimGymEnv {
 public:
  explicit RaisimGymEnv(std::string resourceDir, const Yaml::Node& cfg) : resourceDir_(std::move(resourceDir)), cfg_(cfg) {}
  virtual ~RaisimGymEnv() { if (server_) server_->killServer(); }      // (as upstream's destructor does [RECALL])

  /////// implement these methods /////////
  virtual void init() = 0;
  virtual void reset() = 0;
  virtual void observe(EigenVecRef ob) = 0;
  virtual float step(const ConstEigenVecRef& action) = 0;
  virtual bool isTerminalState(float& terminalReward) = 0;
  ////////////////////////////////////////

  /////// optional methods ///////
  virtual void curriculumUpdate() {}
  virtual void close() {}
  virtual void setSeed(int) {}
  ////////////////////////////////

  void setSimulationTimeStep(double dt) { simulation_dt_ = dt; world_->setTimeStep(dt); }
  void setControlTimeStep(double dt) { control_dt_ = dt; }
  int getObDim() { return obDim_; }
  int getActionDim() { return actionDim_; }
  double getControlTimeStep() { return control_dt_; }
  double getSimulationTimeStep() { return simulation_dt_; }
  raisim::World* getWorld() { return world_.get(); }
  void turnOffVisualization() { if (server_) server_->hibernate(); }      // [RECALL upstream: server_->hibernate() / wakeup()]
  void turnOnVisualization() { if (server_) server_->wakeup(); }
  void startRecordingVideo(const std::string& videoName) { if (server_) server_->startRecordingVideo(videoName); }
  void stopRecordingVideo() { if (server_) server_->stopRecordingVideo(); }
  raisim::Reward& getRewards() { return rewards_; }

 protected:
  std::unique_ptr<raisim::World> world_;
  double simulation_dt_ = 0.001;
  double control_dt_ = 0.01;
  std::string resourceDir_;
  Yaml::Node cfg_;
  int obDim_ = 0, actionDim_ = 0;
  std::unique_ptr<raisim::RaisimServer> server_;
  raisim::Reward rewards_;
};

}  // namespace raisim
