// raisim/RaisimServer.hpp — a NO-OP raisim::RaisimServer, so that an upstream Environment.hpp compiles UNMODIFIED.
//
// Every raisimGymTorch environment holds `std::unique_ptr<raisim::RaisimServer> server_` (RaisimGymEnv), creates it under
// `if (visualizable_)` - `server_ = std::make_unique<raisim::RaisimServer>(world_.get()); server_->launchServer(); server_->focusOn(robot);` -
// and brackets every `world_->integrate()` with `if (server_) server_->lockVisualizationServerMutex(); ... unlock...` [RECALL
// raisimGymTorch/env/envs/rsg_anymal/Environment.hpp, raisim/RaisimServer.hpp; both absent from /root/reference - its .travis.yml:11 builds the
// examples that use them].  Visualisation itself (the TCP protocol to raisimUnity / raisimUnreal, video recording) is OUT OF SCOPE (SURVEY.md §2 rows
// 14-21): this class has upstream's method names and does nothing - no socket is opened, isConnected() is always false, the mutex is a real
// std::mutex so that the lock / unlock pairs of user code keep their meaning (integrateWorldThreadSafe serialises against them as upstream's does).
// (VERDICT r05 missing #3 / next #3.)
#pragma once

#include <mutex>
#include <string>

#include "raisim/World.hpp"

namespace raisim {

class RaisimServer {
 public:
  static constexpr int SEND_BUFFER_SIZE = 33554432;      // upstream's constants, for code that names them [RECALL]
  static constexpr int RECEIVE_BUFFER_SIZE = 33554432;

  explicit RaisimServer(World* world) : world_(world) {}
  RaisimServer(const RaisimServer&) = delete;
  RaisimServer& operator=(const RaisimServer&) = delete;
  ~RaisimServer() { killServer(); }

  /// upstream: starts the server thread on `port`.  Here: remembers that it was asked to (nothing listens)
  void launchServer(int port = 8080) { port_ = port; launched_ = true; }
  void killServer() { launched_ = false; }
  bool isConnected() const { return false; }
  bool isTerminateRequested() const { return false; }
  int getPort() const { return port_; }
  bool isLaunched() const { return launched_; }

  /// upstream: the camera follows this object.  Any object pointer of the facade is accepted (ArticulatedSystem*, Ground*, HeightMap*)
  template <class OBJECT>
  void focusOn(OBJECT* obj) { focused_ = static_cast<const void*>(obj); }
  const void* focusedObject() const { return focused_; }

  void lockVisualizationServerMutex() { mtx_.lock(); }
  void unlockVisualizationServerMutex() { mtx_.unlock(); }
  /// upstream: world.integrate() under the server's mutex
  void integrateWorldThreadSafe() { std::lock_guard<std::mutex> g(mtx_); world_->integrate(); }

  void hibernate() { hibernating_ = true; }
  void wakeup() { hibernating_ = false; }
  bool isHibernating() const { return hibernating_; }

  void startRecordingVideo(const std::string& videoName) { recording_ = true; videoName_ = videoName; }
  void stopRecordingVideo() { recording_ = false; }
  bool isRecording() const { return recording_; }

  void setCameraPositionAndLookAt(const Vec<3>&, const Vec<3>&) {}
  void setMap(const std::string&) {}

 private:
  World* world_ = nullptr;
  std::mutex mtx_;
  const void* focused_ = nullptr;
  std::string videoName_;
  int port_ = 8080;
  bool launched_ = false, hibernating_ = false, recording_ = false;
};

}  // namespace raisim
