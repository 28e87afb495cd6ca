// An rsg_anymal-style ENVIRONMENT written only against the raisim::World / ArticulatedSystem / RaisimGymEnv surface
// (what raisimGymTorch/env/envs/rsg_anymal/Environment.hpp does [RECALL; absent from /root/reference]), with VecDyn
// loops where upstream uses Eigen expressions.  It knows nothing about batching: it owns "its" raisim::World, calls
// world_->integrate() control_dt/simulation_dt times per step and reads its own state and contacts afterwards.
// tests/cpp/facade_test.cpp runs N of these under raisim::VectorizedEnvironment<ENVIRONMENT> and checks that every
// integrate() of the control step is ONE launch for the whole batch and that the results equal the device-resident env's.
#pragma once

#include <cmath>
#include <set>
#include <string>

#include "raisim/RaisimGymEnv.hpp"

namespace raisim {

class ENVIRONMENT : public RaisimGymEnv {
 public:
  explicit ENVIRONMENT(const std::string& resourceDir, const Yaml::Node& cfg, bool visualizable)
      : RaisimGymEnv(resourceDir, cfg), visualizable_(visualizable) {
    /// create world
    world_ = std::make_unique<raisim::World>();

    /// add objects
    anymal_ = world_->addArticulatedSystem(resourceDir_ + "/anymal_c_like.urdf");
    anymal_->setName("anymal");
    anymal_->setControlMode(raisim::ControlMode::PD_PLUS_FEEDFORWARD_TORQUE);
    world_->addGround();

    /// get robot data
    gcDim_ = (int)anymal_->getGeneralizedCoordinateDim();
    gvDim_ = (int)anymal_->getDOF();
    nJoints_ = gvDim_ - 6;

    /// initialize containers
    gc_.resize(gcDim_); gc_init_.resize(gcDim_);
    gv_.resize(gvDim_); gv_init_.resize(gvDim_);
    pTarget_.resize(gcDim_); vTarget_.resize(gvDim_); pTarget12_.resize(nJoints_);

    /// this is nominal configuration of anymal
    const double init[19] = {0, 0, 0.57, 1.0, 0.0, 0.0, 0.0, 0.03, 0.4, -0.8, -0.03, 0.4, -0.8, 0.03, -0.4, 0.8, -0.03, -0.4, 0.8};
    for (int i = 0; i < 19; ++i) gc_init_[i] = init[i];

    /// set pd gains
    raisim::VecDyn jointPgain(gvDim_), jointDgain(gvDim_);
    for (int i = 6; i < gvDim_; ++i) { jointPgain[i] = 50.0; jointDgain[i] = 0.2; }
    anymal_->setPdGains(jointPgain, jointDgain);
    anymal_->setGeneralizedForce(raisim::VecDyn(gvDim_));

    /// MUST BE DONE FOR ALL ENVIRONMENTS
    obDim_ = 34;
    actionDim_ = nJoints_;
    actionMean_.resize(actionDim_); actionStd_.resize(actionDim_);
    obDouble_.resize(obDim_);

    /// action scaling
    for (int i = 0; i < nJoints_; ++i) { actionMean_[i] = gc_init_[7 + i]; actionStd_[i] = cfg["action_std"].As<double>(0.3); }

    /// Reward coefficients
    rewards_.initializeFromConfigurationFile(cfg["reward"]);

    /// indices of links that should not make contact with ground
    footIndices_.insert(anymal_->getBodyIdx("LF_SHANK"));
    footIndices_.insert(anymal_->getBodyIdx("RF_SHANK"));
    footIndices_.insert(anymal_->getBodyIdx("LH_SHANK"));
    footIndices_.insert(anymal_->getBodyIdx("RH_SHANK"));
    footCollisions_ = {7, 11, 15, 19};   // the stand-in URDF also has knee spheres on the shanks: feet by collision primitive

    /// visualize if it is the first environment
    if (visualizable_) {
      server_ = std::make_unique<raisim::RaisimServer>(world_.get());
      server_->launchServer();
      server_->focusOn(anymal_);
    }
  }

  void init() final {}

  void reset() final {
    anymal_->setState(gc_init_, gv_init_);
    updateObservation();
  }

  float step(const ConstEigenVecRef& action) final {
    /// action scaling (float arithmetic, two roundings: the same expression the device-resident env evaluates)
    pTarget_.setZero(); pTarget_[3] = 1.0;
    for (int i = 0; i < nJoints_; ++i) {
      const float scaled = (float)actionStd_[i] * action[i];
      pTarget12_[i] = (double)((float)actionMean_[i] + scaled);
      pTarget_[7 + i] = pTarget12_[i];
    }
    anymal_->setPdTarget(pTarget_, vTarget_);

    for (int i = 0; i < int(control_dt_ / simulation_dt_ + 1e-10); i++) {
      if (server_) server_->lockVisualizationServerMutex();
      world_->integrate();
      if (server_) server_->unlockVisualizationServerMutex();
    }

    updateObservation();

    rewards_.record("torque", anymal_->getGeneralizedForce().squaredNorm());
    rewards_.record("forwardVel", std::min(4.0, bodyLinearVel_[0]));
    return rewards_.sum();
  }

  void updateObservation() {
    anymal_->getState(gc_, gv_);
    raisim::Mat<3, 3> rot;
    anymal_->getBaseOrientation(rot);
    for (int r = 0; r < 3; ++r) {      // bodyLinearVel_ = rot^T * gv_.segment(0, 3), bodyAngularVel_ = rot^T * gv_.segment(3, 3)
      bodyLinearVel_[r] = rot(0, r) * gv_[0] + rot(1, r) * gv_[1] + rot(2, r) * gv_[2];
      bodyAngularVel_[r] = rot(0, r) * gv_[3] + rot(1, r) * gv_[4] + rot(2, r) * gv_[5];
    }
    int k = 0;
    obDouble_[k++] = gc_[2];                                       /// body height
    for (int c = 0; c < 3; ++c) obDouble_[k++] = rot(2, c);        /// body orientation: rot.e().row(2)
    for (int j = 0; j < nJoints_; ++j) obDouble_[k++] = gc_[7 + j];  /// joint angles
    for (int c = 0; c < 3; ++c) obDouble_[k++] = bodyLinearVel_[c];
    for (int c = 0; c < 3; ++c) obDouble_[k++] = bodyAngularVel_[c];  /// body linear&angular velocity
    for (int j = 0; j < nJoints_; ++j) obDouble_[k++] = gv_[6 + j];  /// joint velocity
  }

  void observe(EigenVecRef ob) final {
    /// convert it to float
    for (int i = 0; i < obDim_; ++i) ob[i] = (float)obDouble_[i];
  }

  bool isTerminalState(float& terminalReward) final {
    terminalReward = float(terminalRewardCoeff_);

    /// if the contact body is not feet
    for (auto& contact : anymal_->getContacts())
      if (footCollisions_.find(contact.getCollisionIndex()) == footCollisions_.end())
        return true;

    terminalReward = 0.f;
    return false;
  }

  void curriculumUpdate() {}

 private:
  int gcDim_, gvDim_, nJoints_;
  bool visualizable_ = false;
  raisim::ArticulatedSystem* anymal_;
  raisim::VecDyn gc_init_, gv_init_, gc_, gv_, pTarget_, pTarget12_, vTarget_;
  double terminalRewardCoeff_ = -10.;
  raisim::VecDyn actionMean_, actionStd_, obDouble_;
  raisim::Vec<3> bodyLinearVel_, bodyAngularVel_;
  std::set<size_t> footIndices_;
  std::set<int> footCollisions_;
};

}  // namespace raisim
