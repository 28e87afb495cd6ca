// raisim/World.hpp — raisim::World / ArticulatedSystem / Ground / HeightMap / Contact facade over the C-ABI (rsb.h).
//
// Re-authored from recollection of upstream raisim/World.hpp, object/ArticulatedSystem/ArticulatedSystem.hpp,
// object/terrain/HeightMap.hpp, contact/Contact.hpp [RECALL — all absent from /root/reference, SURVEY.md §8b].
// The engine behind these classes is the batched, GPU-resident world of librsb.so:
//
//   raisim::BatchedWorld   N replicas of {World + one ArticulatedSystem + terrain} on one GPU (new type; what a
//                          batched VectorizedEnvironment drives directly — the fast path).
//   raisim::World          the upstream per-env class, always a VIEW of one replica of a BatchedWorld:
//                            - default-constructed outside a batch scope it owns a 1-replica BatchedWorld (an unmodified
//                              Environment.hpp runs, one launch per env: correctness path);
//                            - default-constructed inside a raisim::BatchScope (what VectorizedEnvironment<ENV> opens
//                              around the construction of its N environments) the k-th World becomes replica k of ONE
//                              shared BatchedWorld with N replicas;
//                            - World(BatchedWorld&, env) names the replica explicitly.
//                          integrate() on a view advances THAT replica only: the call is recorded, and the batch is
//                          flushed with a single launch once every replica has one pending.  Under VectorizedEnvironment<ENV>
//                          (Fiber.hpp) the call is recorded and the env's step() body simply CONTINUES: integrate() returns
//                          nothing an env could look at, so k integrate() calls in a row are k recorded sub-steps, and only the
//                          first read (or staged write) that follows parks the fiber.  Once every live env is parked the whole
//                          batch is flushed by ONE rsb_view_exchange: staged uploads, ONE launch of k sub-steps (the fused
//                          kernel the benchmark runs; envs with different counts go in masked launches), the downloads the
//                          environments have been reading, one stream synchronisation.  RSB_VIEW_FUSE=0 restores round 3's
//                          behaviour (a flush per integrate()).
//                          Global setters (setTimeStep, setGravity, addGround, setERP, materials, solver parameters,
//                          PD gains) act on the shared world, i.e. on ALL replicas - every env of a vectorised
//                          environment calls them with the same values, as upstream's ENVs do.
//   raisim::ArticulatedSystem  per-env view.  Row writes (state, PD targets, feed-forward force) are staged on the host
//                          and uploaded as whole arrays at the next flush; reads come from a host copy of the batch
//                          refreshed once per flush - N envs cost a constant number of PCIe transfers per integrate().
//
// Errors follow upstream's RSFATAL: a failed call throws std::runtime_error with rsb_last_error().
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <map>
#include <vector>

#include "raisim/Fiber.hpp"
#include "raisim/math.hpp"
#include "rsb.h"

// raisim_message.hpp's macros [RECALL; absent from /root/reference - SURVEY.md section 5: "same names for source compat"]: stream-style messages,
//   RSINFO("dt " << dt);  RSWARN_IF(n > 8, "contacts: " << n);  RSFATAL_IF(!ok, "cannot open " << path);
// INFO / WARN print "[file:line] [INFO] message" to stderr.  FATAL upstream prints and exits; here it THROWS std::runtime_error with the same text
// (an abort inside a Python extension module would take the interpreter down; pybind11 turns the exception into a Python one).
#define RSB_MSG_(level, msg) do { std::ostringstream rs_os_; rs_os_ << "[" << __FILE__ << ":" << __LINE__ << "] [" level "] " << msg; std::cerr << rs_os_.str() << std::endl; } while (0)
#define RSINFO(msg) RSB_MSG_("INFO", msg)
#define RSWARN(msg) RSB_MSG_("WARN", msg)
#define RSINFO_IF(cond, msg) do { if (cond) RSINFO(msg); } while (0)
#define RSWARN_IF(cond, msg) do { if (cond) RSWARN(msg); } while (0)
#define RSFATAL(msg) do { std::ostringstream rs_os_; rs_os_ << msg; throw std::runtime_error(rs_os_.str()); } while (0)
#define RSFATAL_IF(cond, msg) do { if (cond) RSFATAL(msg); } while (0)
#define RSASSERT(cond, msg) RSFATAL_IF(!(cond), msg)
#define RSRETURN_IF(cond, msg) do { if (cond) { RSWARN(msg); return; } } while (0)
#define RSISNAN(val) RSFATAL_IF(std::isnan(val), #val " is nan")
#define RSB_CHECK(expr) do { int st_ = (expr); if (st_ != RSB_OK) throw std::runtime_error(std::string(#expr) + ": " + rsb_last_error()); } while (0)

namespace raisim {

namespace IntegrationScheme {
enum Type : int { TRAPEZOID = 0, SEMI_IMPLICIT = 1, EULER = 2, RUNGE_KUTTA_4 = 3 };   // RUNGE_KUTTA_4: host-driven, four dynamics evaluations per integrate() (rsb_set_integration_scheme)
}

namespace ControlMode {
enum Type : int { FORCE_AND_TORQUE = RSB_FORCE_AND_TORQUE, PD_PLUS_FEEDFORWARD_TORQUE = RSB_PD_PLUS_FEEDFORWARD_TORQUE };
}

/// Reader-writer lock whose readers do not share a cache line (a "big-reader" lock): N env bodies on T threads stage their rows as
/// readers thousands of times per control step - through one pthread rwlock word that was 16 000 contended atomic operations per step
/// and the reason the bodies did not scale with the threads -, a writer (whole-array upload / refresh of the mirrors) is rare.
class BigReaderLock {
 public:
  void lock_shared() {
    Slot& s = slots_[slot()];
    for (;;) {
      s.readers.fetch_add(1, std::memory_order_seq_cst);
      if (!writer_.load(std::memory_order_seq_cst)) return;
      s.readers.fetch_sub(1, std::memory_order_seq_cst);        // a writer is in or waiting: step back and let it through
      while (writer_.load(std::memory_order_acquire)) std::this_thread::yield();
    }
  }
  void unlock_shared() { slots_[slot()].readers.fetch_sub(1, std::memory_order_release); }
  void lock() {
    wmu_.lock();
    writer_.store(true, std::memory_order_seq_cst);
    for (Slot& s : slots_) while (s.readers.load(std::memory_order_seq_cst) != 0) std::this_thread::yield();
  }
  void unlock() { writer_.store(false, std::memory_order_release); wmu_.unlock(); }
 private:
  struct alignas(64) Slot { std::atomic<int> readers{0}; };
  static constexpr int kSlots = 64;
  static int slot() { static std::atomic<int> next{0}; static thread_local int mine = next.fetch_add(1) % kSlots; return mine; }
  Slot slots_[kSlots];
  std::atomic<bool> writer_{false};
  std::mutex wmu_;
};

/// Page-locked host array (rsb_host_alloc): the mirrors behind the per-env views, so that a flush's copies run asynchronously on
/// the world's stream up to its single synchronisation.
template <class T>
class PinnedArray {
 public:
  PinnedArray() = default;
  ~PinnedArray() { if (p_) rsb_host_free(p_); }
  PinnedArray(const PinnedArray&) = delete;
  PinnedArray& operator=(const PinnedArray&) = delete;
  /// grow-only storage: shrinking or re-growing within the capacity touches no allocator (a flush whose launch masks change size from step to
  /// step must not pay a hipHostFree / hipHostMalloc pair each time); new storage starts zeroed, kept storage keeps its contents
  void resize(size_t n) {
    if (n <= cap_) { n_ = n; return; }
    void* m = nullptr;
    RSB_CHECK(rsb_host_alloc(n * sizeof(T), &m));
    std::memset(m, 0, n * sizeof(T));
    if (p_) { std::memcpy(m, static_cast<void*>(p_), n_ * sizeof(T)); rsb_host_free(p_); }
    p_ = static_cast<T*>(m); n_ = cap_ = n;
  }
  T* data() { return p_; }
  const T* data() const { return p_; }
  size_t size() const { return n_; }
  T& operator[](size_t i) { return p_[i]; }
  const T& operator[](size_t i) const { return p_[i]; }
 private:
  T* p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
};

/// One solved contact of an articulated system (upstream raisim::Contact).
class Contact {
 public:
  explicit Contact(const rsb_contact& c) : c_(c) {}
  Vec<3> getPosition() const { return v3(c_.position); }
  Vec<3> getNormal() const { return v3(c_.normal); }
  Vec<3> getImpulse() const { return v3(c_.impulse); }   // world frame (upstream: contact frame + getContactFrame())
  double getDepth() const { return c_.depth; }
  size_t getlocalBodyIndex() const { return (size_t)c_.body; }
  int getCollisionIndex() const { return RSB_CONTACT_PRIMITIVE(c_.collision); }
  /// a self-collision is listed once per body: the two entries are neighbours in getContacts(), object A first
  bool isSelfCollision() const { return (c_.collision & (RSB_CONTACT_SELF_A | RSB_CONTACT_SELF_B)) != 0; }
  bool isSecondTerrainContact() const { return (c_.collision & RSB_CONTACT_SECOND) != 0; }   // extension: rsb_set_heightmap_contacts
  bool isCapsuleCylinderContact() const { return (c_.collision & RSB_CONTACT_CAPSULE) != 0; }  // extension: rsb_set_capsule_contacts (a capsule's / cylinder's barrel, a box's face or edge; the id is the first end's / corner's)
  bool isObjectA() const { return (c_.collision & RSB_CONTACT_SELF_B) == 0; }
  bool skip() const { return false; }
 private:
  static Vec<3> v3(const float* p) { Vec<3> v; v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; return v; }
  rsb_contact c_;
};

/// N lock-stepped replicas on one GPU.
class BatchedWorld {
 public:
  BatchedWorld(const std::string& urdfPath, int numEnvs, int device = 0) {
    RSB_CHECK(rsb_model_from_urdf_file(urdfPath.c_str(), &model_));
    init(numEnvs, device);
  }
  ~BatchedWorld() { if (world_) rsb_destroy(world_); if (model_) rsb_model_destroy(model_); }
  BatchedWorld(const BatchedWorld&) = delete;
  BatchedWorld& operator=(const BatchedWorld&) = delete;

  rsb_world* handle() { return world_; }
  const rsb_model* model() const { return model_; }
  const rsb_model_blob& blob() const { return blob_; }
  int numEnvs() const { return n_; }
  int gcDim() const { return blob_.nq; }
  int dof() const { return blob_.nv; }

  void setTimeStep(double dt) { std::lock_guard<std::recursive_mutex> lk(mu_); RSB_CHECK(rsb_set_timestep(world_, dt)); }
  double getTimeStep() const { return rsb_get_timestep(world_); }
  double getWorldTime() const { return rsb_get_world_time(world_); }
  void setGravity(const Vec<3>& g) { std::lock_guard<std::recursive_mutex> lk(mu_); RSB_CHECK(rsb_set_gravity(world_, g.data())); }
  void setERP(double erp, double = 0) { std::lock_guard<std::recursive_mutex> lk(mu_); RSB_CHECK(rsb_set_erp(world_, erp)); }
  void setDefaultMaterial(double friction, double restitution = 0, double resThreshold = 0) { std::lock_guard<std::recursive_mutex> lk(mu_); RSB_CHECK(rsb_set_material(world_, friction, restitution, resThreshold)); }
  /// World::setMaterialPairProp [RECALL; upstream Materials.hpp absent]: (mu, restitution, resThreshold) of the material pair,
  /// order-free.  The terrain is the only other object of an env, so the pair table is resolved into one triple per collision
  /// primitive of the robot: (primitive's URDF material, terrain's material) -> rsb_set_collision_materials.
  void setMaterialPairProp(const std::string& m1, const std::string& m2, double friction, double restitution, double resThreshold) {
    RSFATAL_IF(friction < 0 || restitution < 0 || restitution > 1 || resThreshold < 0, "setMaterialPairProp: friction >= 0, 0 <= restitution <= 1, resThreshold >= 0");
    pairProps_[pairKey(m1, m2)] = {friction, restitution, resThreshold};
    resolveMaterials();
  }
  /// ArticulatedSystem::ignoreCollisionBetween(bodyIdx1, bodyIdx2) / self-collision on-off for every replica
  void ignoreCollisionBetween(size_t bodyIdx1, size_t bodyIdx2) { std::lock_guard<std::recursive_mutex> lk(mu_); RSB_CHECK(rsb_ignore_collision_between(world_, (int)bodyIdx1, (int)bodyIdx2)); resolveMaterials(); }
  void setSelfCollision(bool on) { std::lock_guard<std::recursive_mutex> lk(mu_); RSB_CHECK(rsb_set_self_collision(world_, on ? 1 : 0)); }
  /// material of the terrain (addGround / addHeightMap's material argument)
  void setTerrainMaterial(const std::string& material) { terrainMaterial_ = material; resolveMaterials(); }
  void setContactSolverParam(double alpha_init, double alpha_min, double alpha_decay, int maxIter, double threshold) {
    RSB_CHECK(rsb_set_contact_solver_param(world_, alpha_init, alpha_min, alpha_decay, maxIter, threshold));
  }
  /// extensions (no upstream counterpart; rsb.h): solver settings of redundant contact sets, the Anderson step, two contacts per
  /// primitive against a height map
  void setIntegrationScheme(int scheme) { RSB_CHECK(rsb_set_integration_scheme(world_, scheme)); }
  void setMultiContactSolverParam(int depth, bool lightPasses, int freezeAfter, int stallWindow) { RSB_CHECK(rsb_set_solver_multi_contact(world_, depth, lightPasses ? 1 : 0, freezeAfter, stallWindow)); }
  void setSolverAcceleration(int firstSweep, double clip = 20.0) { RSB_CHECK(rsb_set_solver_anderson(world_, firstSweep, clip)); }
  void setHeightMapContactsPerPrimitive(int n, double minAngleDeg = 45.0) { RSB_CHECK(rsb_set_heightmap_contacts(world_, n, minAngleDeg)); }
  void setExactCapsuleContacts(bool on) { RSB_CHECK(rsb_set_capsule_contacts(world_, on ? 1 : 0)); }
  /// consecutive control steps of the BATCH (rsb_control_step: the device-resident loop, not the per-env views, whose flush reads every step's output)
  /// overlap on the device; any other call joins first (rsb.h)
  void setStepPipelining(bool on) { std::lock_guard<std::recursive_mutex> lk(mu_); RSB_CHECK(rsb_set_step_pipelining(world_, on ? 1 : 0)); }
  /// specialised step kernels (rsb_ext.h: RSB_SPEC_OFF / RSB_SPEC_CACHED (default) / RSB_SPEC_COMPILE): the kernel compiled for THIS model and world
  /// configuration - same results, ~10 % faster; RSB_SPEC_COMPILE compiles a missing code object at the first integrate() (~3 s, cached on disk)
  void setKernelSpecialization(int mode) { std::lock_guard<std::recursive_mutex> lk(mu_); RSB_CHECK(rsb_set_specialization(world_, mode)); }
  void addGround(double zHeight = 0.0) { std::lock_guard<std::recursive_mutex> lk(mu_); RSB_CHECK(rsb_set_ground(world_, zHeight)); }
  void addHeightMap(int xSamples, int ySamples, double xSize, double ySize, double centerX, double centerY,
                    const std::vector<double>& height) {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    std::vector<float> h(height.begin(), height.end());
    RSFATAL_IF((int)h.size() != xSamples * ySamples, "addHeightMap: height.size() != xSamples*ySamples");
    RSB_CHECK(rsb_set_heightmap(world_, xSamples, ySamples, xSize, ySize, centerX, centerY, h.data()));
  }
  /// the whole batch at once (fast path; staged view writes are uploaded first)
  void integrate(int nSubsteps = 1) { std::lock_guard<std::recursive_mutex> lk(mu_); uploadStaged(); RSB_CHECK(rsb_integrate(world_, nSubsteps)); stateCacheValid_ = false; contactsValid_ = false; genfValid_ = false; queryValid_ = false; }
  /// the whole-batch M / h query; N views calling it between two flushes cost ONE launch (valid until a launch or a staged write)
  void integrate1() {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    uploadStaged();
    if (queryValid_) return;
    RSB_CHECK(rsb_integrate1(world_));
    queryValid_ = true; ++queryLaunches_;
  }
  void integrate2() { std::lock_guard<std::recursive_mutex> lk(mu_); uploadStaged(); RSB_CHECK(rsb_integrate2(world_)); stateCacheValid_ = false; contactsValid_ = false; genfValid_ = false; queryValid_ = false; }
  long queryLaunches() const { return queryLaunches_; }   ///< launches issued by integrate1() (tests: N views -> 1 launch)
  /// the whole-batch query results, serialised like every other call into the handle (env bodies on several threads ask for them)
  void getMassMatrices(float* M) { std::lock_guard<std::recursive_mutex> lk(mu_); RSB_CHECK(rsb_get_mass_matrix(world_, M, RSB_HOST)); }
  void getInverseMassMatrices(float* Mi) { std::lock_guard<std::recursive_mutex> lk(mu_); RSB_CHECK(rsb_get_inverse_mass_matrix(world_, Mi, RSB_HOST)); }
  void getNonlinearitiesAll(float* h) { std::lock_guard<std::recursive_mutex> lk(mu_); RSB_CHECK(rsb_get_nonlinearities(world_, h, RSB_HOST)); }

  // batched, caller-owned host buffers (row-major [N, dim] float32, the raisimGymTorch matrix layout)
  void setState(const float* gc, const float* gv) {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    RSB_CHECK(rsb_set_state(world_, gc, gv, nullptr, RSB_HOST)); dropStage(RSB_F_GC); dropStage(RSB_F_GV); stateCacheValid_ = false; queryValid_ = false;
    std::fill(gcMask_.begin(), gcMask_.end(), 0); std::fill(gvMask_.begin(), gvMask_.end(), 0);
  }
  void getState(float* gc, float* gv) { std::lock_guard<std::recursive_mutex> lk(mu_); uploadStaged(); RSB_CHECK(rsb_get_state(world_, gc, gv, RSB_HOST)); }
  void setPdGains(const float* kp, const float* kd) { std::lock_guard<std::recursive_mutex> lk(mu_); RSB_CHECK(rsb_set_pd_gains(world_, kp, kd)); }
  void setPdTarget(const float* pTarget, const float* dTarget) { std::lock_guard<std::recursive_mutex> lk(mu_); RSB_CHECK(rsb_set_pd_target(world_, pTarget, dTarget, RSB_HOST)); if (pTarget) dropStage(RSB_F_PTARGET); if (dTarget) dropStage(RSB_F_DTARGET); }
  void setGeneralizedForce(const float* tau) { std::lock_guard<std::recursive_mutex> lk(mu_); RSB_CHECK(rsb_set_generalized_force(world_, tau, RSB_HOST)); dropStage(RSB_F_TAU_FF); }
  void setControlMode(ControlMode::Type m) { std::lock_guard<std::recursive_mutex> lk(mu_); RSB_CHECK(rsb_set_control_mode(world_, (int)m)); }

 public:
  // ---- staging and caches behind the per-env views (raisim::World / raisim::ArticulatedSystem) ---------------------
  /// stage one env's row of GC / GV / PTARGET / DTARGET / TAU_FF; uploaded as a whole array at the next flush.
  /// The host arrays double as the read mirror, so an env reading back what it just wrote costs no transfer.
  /// Thread-safe against the whole-array operations: N step() bodies stage their rows concurrently (shared lock, distinct rows),
  /// an upload / refresh of the mirrors takes the lock exclusively and so never sees a half-written row or clears the flags of
  /// a row that is being written (a staged write used to be able to vanish that way).
  void stageRow(int field, int env, const double* v, int dim) {
    syncView(env, "a staged write");          // integrate() calls recorded before this write must run before it
    Stage& st = stage(field);                  // (creates the mirror on first use; takes mu_, so before the stage lock)
    std::shared_lock<BigReaderLock> sl(stageMu_);
    for (int i = 0; i < dim; ++i) st.host[(size_t)env * dim + i] = (float)v[i];
    if (field == RSB_F_GC) gcMask_[env] = 1;
    if (field == RSB_F_GV) gvMask_[env] = 1;
    raise(st.dirty);
  }
  /// one env's row of any of the five fields, as the device holds (or is about to hold) it
  void readRow(int field, int env, double* out, int dim) {
    if (field == RSB_F_GC || field == RSB_F_GV) raise(wantState_);
    syncView(env, "a state read");
    if (field == RSB_F_GC || field == RSB_F_GV) refreshState();
    const float* src = stage(field).host.data();
    for (int i = 0; i < dim; ++i) out[i] = src[(size_t)env * dim + i];
  }
  /// the generalized force the actuators applied to replica `env` in its last integrate() (one download per launch)
  void readGeneralizedForce(int env, double* out, int dim) {
    raise(wantGenf_);
    syncView(env, "getGeneralizedForce()");
    if (!genfValid_.load(std::memory_order_acquire)) {
      std::lock_guard<std::recursive_mutex> lk(mu_);
      if (!genfValid_) {
        genf_.resize((size_t)n_ * dim);
        RSB_CHECK(rsb_get_field(world_, RSB_F_GENERALIZED_FORCE, genf_.data(), RSB_HOST));
        genfValid_ = true;
      }
    }
    for (int i = 0; i < dim; ++i) out[i] = genf_[(size_t)env * dim + i];
  }
  /// World::integrate() of replica `env`: recorded.  Outside a fiber batch it is flushed when every replica has one pending;
  /// inside one (VectorizedEnvironment<ENV>) the body continues and the record is flushed at the env's next read / staged write
  /// (syncView), together with everybody else's - k integrate() calls in a row become ONE launch of k sub-steps.
  void integrateView(int env) {
    detail::FiberScheduler* fs = detail::FiberScheduler::current();
    if (fs && fiberBatch_) {
      ++pending_[env];                                   // (the env's own counter: no shared word is touched; flushViews scans them)
      if (!fuse_) fs->park();                            // RSB_VIEW_FUSE=0: a flush per integrate(), as in round 3
      return;
    }
    if (pending_[env]) throw std::runtime_error("raisim::World::integrate(): this replica already has an un-flushed integrate(); "
                                                "every World of the batch must call integrate() before the next one (or drive the envs through VectorizedEnvironment<ENV>)");
    pending_[env] = 1; ++nPending_;
    if (nPending_ == n_) flushViews();
  }
  /// integrate1() of a view: the whole-batch query, after this replica's recorded integrate() calls have run
  void integrate1View(int env) { syncView(env, "integrate1()"); integrate1(); }
  /// time of replica `env`: the batch's clock plus the integrate() calls this replica has recorded but not yet run
  double worldTimeOf(int env) const { return rsb_get_world_time(world_) + pending_[env] * rsb_get_timestep(world_); }
  /// one rsb_view_exchange for everything that is pending: staged rows up, the recorded integrate() calls (ONE launch when every
  /// replica recorded the same number, else masked launches by count), the fields the environments read down; one synchronisation
  void flushViews() {
    const auto tf0 = std::chrono::steady_clock::now();
    std::lock_guard<std::recursive_mutex> lk(mu_);
    int cmin = 1 << 30, cmax = 0;
    for (int e = 0; e < n_; ++e) { cmin = std::min(cmin, pending_[e]); cmax = std::max(cmax, pending_[e]); }
    if (cmax == 0) return;
    std::unique_lock<BigReaderLock> sl(stageMu_);
    rsb_view_io io{};
    collectUploads(io);
    // launches: distinct counts c_1 < c_2 < ...; launch i runs c_i - c_(i-1) sub-steps for the envs with count >= c_i
    launchSub_.clear();
    if (cmin == cmax) launchSub_.push_back(cmax);         // (the usual case: every env of the batch ran the same step() body)
    else {
      std::vector<int> levels(pending_.begin(), pending_.end());
      std::sort(levels.begin(), levels.end());
      levels.erase(std::unique(levels.begin(), levels.end()), levels.end());
      if (levels.front() == 0) levels.erase(levels.begin());
      launchMasks_.resize((size_t)levels.size() * n_);
      int prev = 0;
      for (size_t i = 0; i < levels.size(); ++i) {
        launchSub_.push_back(levels[i] - prev); prev = levels[i];
        for (int e = 0; e < n_; ++e) launchMasks_[i * n_ + e] = pending_[e] >= levels[i] ? 1 : 0;
      }
      io.launch_masks = launchMasks_.data();
    }
    io.n_launches = (int32_t)launchSub_.size(); io.launch_substeps = launchSub_.data();
    // downloads: what the environments have been reading since the batch exists (sticky: an env that reads its contacts in one
    // control step reads them in the next).  The mirrors' staged rows have just been uploaded, so whole arrays can be overwritten.
    const bool st = wantState_, ct = wantContacts_, gf = wantGenf_;
    if (st) { Stage& gc = stageLocked(RSB_F_GC); Stage& gv = stageLocked(RSB_F_GV); io.gc_out = gc.host.data(); io.gv_out = gv.host.data(); }
    if (ct) {
      if (kmax_ == 0) RSB_CHECK(rsb_dims(world_, nullptr, nullptr, nullptr, nullptr, &kmax_));
      cnt_.resize(n_); con_.resize((size_t)n_ * kmax_);
      io.contact_counts = cnt_.data(); io.contacts = con_.data();
    }
    if (gf) { genf_.resize((size_t)n_ * blob_.nv); io.generalized_force = genf_.data(); }
    const auto tf1 = std::chrono::steady_clock::now();
    RSB_CHECK(rsb_view_exchange(world_, &io));
    const auto tf2 = std::chrono::steady_clock::now();
    flushPrepNs_ += std::chrono::duration_cast<std::chrono::nanoseconds>(tf1 - tf0).count();
    flushExchangeNs_ += std::chrono::duration_cast<std::chrono::nanoseconds>(tf2 - tf1).count();
    std::fill(pending_.begin(), pending_.end(), 0);
    nPending_ = 0;
    viewLaunches_ += (long)launchSub_.size();
    ++viewFlushes_;
    stateCacheValid_ = st; contactsValid_ = ct; genfValid_ = gf; queryValid_ = false;
  }
  /// drop every recorded-but-unflushed integrate() (error path of the fiber scheduler: the parked fibers are gone, their
  /// pending flags must not outlive them or every later step() would throw "already has an un-flushed integrate()")
  void abortViews() { std::fill(pending_.begin(), pending_.end(), 0); nPending_ = 0; }
  void setFiberBatch(bool on) { fiberBatch_ = on; }
  void setFuseIntegrateCalls(bool on) { fuse_ = on; }     ///< (tests, A/B) false = a flush per integrate()
  long viewLaunches() const { return viewLaunches_; }     ///< launches issued by flushViews() (tests: N views x k integrate() -> 1 launch)
  long viewFlushes() const { return viewFlushes_; }       ///< rsb_view_exchange calls issued by flushViews()
  /// host time (ns) flushViews() has spent preparing its exchanges (scan of the pending counters, upload lists) and inside rsb_view_exchange
  long long flushPrepNs() const { return flushPrepNs_; }
  long long flushExchangeNs() const { return flushExchangeNs_; }
  int pendingViews() const { int k = 0; for (int c : pending_) k += c != 0; return k; }
  /// contacts of the last integrate() of every env, downloaded once per flush
  const PinnedArray<rsb_contact>& contactsOf(int env, int& count, int& kmax) {
    raise(wantContacts_);
    syncView(env, "getContacts()");
    if (!contactsValid_.load(std::memory_order_acquire)) {
      std::lock_guard<std::recursive_mutex> lk(mu_);
      if (!contactsValid_) {
        RSB_CHECK(rsb_dims(world_, nullptr, nullptr, nullptr, nullptr, &kmax_));
        cnt_.resize(n_); con_.resize((size_t)n_ * kmax_);
        RSB_CHECK(rsb_get_contacts(world_, cnt_.data(), con_.data(), RSB_HOST));
        contactsValid_ = true;
      }
    }
    count = cnt_[env]; kmax = kmax_;
    return con_;
  }
  /// staged rows -> device (before a launch, or before a query that must see them)
  void uploadStaged() {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    std::unique_lock<BigReaderLock> sl(stageMu_);
    rsb_view_io io{};
    if (collectUploads(io)) RSB_CHECK(rsb_view_exchange(world_, &io));
  }

 private:
  struct Stage { PinnedArray<float> host; std::atomic<bool> init{false}, dirty{false}; };
  /// set a flag that N env bodies on several threads raise over and over: a plain load when it is already up (a store - let alone a
  /// sequentially consistent one - would bounce the flag's cache line between the threads on every read of every env)
  static void raise(std::atomic<bool>& f) { if (!f.load(std::memory_order_relaxed)) f.store(true, std::memory_order_release); }
  Stage& stage(int field) {
    Stage& st = stages_[field];
    if (st.init.load(std::memory_order_acquire)) return st;  // (fast path: no lock once the mirror exists)
    std::lock_guard<std::recursive_mutex> lk(mu_);
    return stageLocked(field);
  }
  Stage& stageLocked(int field) {     // mu_ held
    Stage& st = stages_[field];
    if (!st.init) {          // the host copy starts as what the device holds
      const int dim = (field == RSB_F_GC || field == RSB_F_PTARGET) ? blob_.nq : blob_.nv;
      st.host.resize((size_t)n_ * dim);
      RSB_CHECK(rsb_get_field(world_, field, st.host.data(), RSB_HOST));
      st.init = true;
    }
    return st;
  }
  /// the staged rows as uploads of one rsb_view_exchange (mu_ and stageMu_ held; the exchange must follow): true if there are any
  bool collectUploads(rsb_view_io& io) {
    bool any = false;
    Stage& gc = stages_[RSB_F_GC]; Stage& gv = stages_[RSB_F_GV];
    if (gc.dirty || gv.dirty) {
      // masked upload: only the rows a view wrote are overwritten (and their solver warm state cleared).  A row written in
      // only one of the two fields takes its other half from the mirror, which must then be current.
      bool half = false;
      for (int e = 0; e < n_ && !half; ++e) half = gcMask_[e] != gvMask_[e];
      if (half) refreshStateLocked();
      stateMask_.resize(n_);
      for (int e = 0; e < n_; ++e) stateMask_[e] = gcMask_[e] | gvMask_[e];
      io.gc = stageLocked(RSB_F_GC).host.data(); io.gv = stageLocked(RSB_F_GV).host.data(); io.state_mask = stateMask_.data();
      gc.dirty = gv.dirty = false;
      std::fill(gcMask_.begin(), gcMask_.end(), 0); std::fill(gvMask_.begin(), gvMask_.end(), 0);
      any = true;
    }
    Stage& pt = stages_[RSB_F_PTARGET]; Stage& dt = stages_[RSB_F_DTARGET]; Stage& tf = stages_[RSB_F_TAU_FF];
    if (pt.dirty) { io.p_target = pt.host.data(); pt.dirty = false; any = true; }
    if (dt.dirty) { io.d_target = dt.host.data(); dt.dirty = false; any = true; }
    if (tf.dirty) { io.tau_ff = tf.host.data(); tf.dirty = false; any = true; }
    if (any) queryValid_ = false;
    return any;
  }
  void dropStage(int field) { stages_[field].init = false; stages_[field].dirty = false; }
  /// make the GC / GV mirrors current: one download after a launch; rows staged since then keep their staged values
  void refreshState() {
    if (stateCacheValid_.load(std::memory_order_acquire) && stages_[RSB_F_GC].init.load(std::memory_order_acquire) && stages_[RSB_F_GV].init.load(std::memory_order_acquire)) return;     // (fast path without the lock)
    std::lock_guard<std::recursive_mutex> lk(mu_);
    std::unique_lock<BigReaderLock> sl(stageMu_);       // no row is being staged while the mirrors are overwritten
    refreshStateLocked();
  }
  void refreshStateLocked() {       // mu_ and stageMu_ (exclusively) held
    if (stateCacheValid_ && stages_[RSB_F_GC].init && stages_[RSB_F_GV].init) return;
    Stage& gc = stageLocked(RSB_F_GC); Stage& gv = stageLocked(RSB_F_GV);
    tmpGc_.resize(gc.host.size()); tmpGv_.resize(gv.host.size());
    RSB_CHECK(rsb_get_state(world_, tmpGc_.data(), tmpGv_.data(), RSB_HOST));
    const int nq = blob_.nq, nv = blob_.nv;
    for (int e = 0; e < n_; ++e) {
      if (!gcMask_[e]) std::copy_n(&tmpGc_[(size_t)e * nq], nq, &gc.host[(size_t)e * nq]);
      if (!gvMask_[e]) std::copy_n(&tmpGv_[(size_t)e * nv], nv, &gv.host[(size_t)e * nv]);
    }
    stateCacheValid_ = true;
  }
  /// a read or staged write of replica `env`: its recorded integrate() calls run first.  In a fiber batch the fiber parks until the
  /// scheduler has flushed the batch; outside one the caller has broken the "every view integrates, then reads" protocol
  void syncView(int env, const char* what) {
    if (!pending_[env]) return;
    detail::FiberScheduler* fs = detail::FiberScheduler::current();
    if (fs && fiberBatch_) { while (pending_[env]) fs->park(); return; }
    throw std::runtime_error(std::string("raisim::World view: ") + what + " while this replica's integrate() is still waiting for the "
                                                "other replicas of the batch (all views must call integrate() first, or use VectorizedEnvironment<ENV>)");
  }
  void init(int numEnvs, int device) {
    RSB_CHECK(rsb_model_get_blob(model_, &blob_));
    RSB_CHECK(rsb_create(model_, numEnvs, device, &world_));
    n_ = numEnvs;
    pending_.assign(n_, 0); gcMask_.assign(n_, 0); gvMask_.assign(n_, 0);
    RSB_CHECK(rsb_enable_generalized_force_output(world_, 1));   // ArticulatedSystem::getGeneralizedForce() (rsg_anymal's torque reward reads it)
  }
  struct PairProp { double mu, restitution, resThreshold; };
  static std::string pairKey(const std::string& a, const std::string& b) { return a < b ? a + "\n" + b : b + "\n" + a; }
  void resolveMaterials() {
    std::lock_guard<std::recursive_mutex> lk(mu_);
    const int nc = blob_.ncol;
    std::vector<double> mu(nc, -1.0), e(nc, -1.0), thr(nc, -1.0);   // negative = the world's default material
    for (int i = 0; i < nc; ++i) {
      const char* cm = rsb_model_collision_material(model_, i);
      auto it = pairProps_.find(pairKey(cm ? cm : "default", terrainMaterial_));
      if (it != pairProps_.end()) { mu[i] = it->second.mu; e[i] = it->second.restitution; thr[i] = it->second.resThreshold; }
    }
    RSB_CHECK(rsb_set_collision_materials(world_, mu.data(), e.data(), thr.data()));
    // self-collisions: the pair (material of primitive i, material of primitive j)
    const int np = rsb_self_collision_pairs(world_, nullptr, 0);
    std::vector<int32_t> pairs((size_t)2 * np);
    if (np > 0) rsb_self_collision_pairs(world_, pairs.data(), np);
    std::vector<double> smu(np, -1.0), se(np, -1.0), sthr(np, -1.0);
    for (int k = 0; k < np; ++k) {
      const char* mi = rsb_model_collision_material(model_, pairs[2 * k]);
      const char* mj = rsb_model_collision_material(model_, pairs[2 * k + 1]);
      auto it = pairProps_.find(pairKey(mi ? mi : "default", mj ? mj : "default"));
      if (it != pairProps_.end()) { smu[k] = it->second.mu; se[k] = it->second.restitution; sthr[k] = it->second.resThreshold; }
    }
    RSB_CHECK(rsb_set_self_collision_materials(world_, smu.data(), se.data(), sthr.data()));
  }
  std::map<std::string, PairProp> pairProps_;
  std::string terrainMaterial_ = "default";
  rsb_model* model_ = nullptr;
  rsb_world* world_ = nullptr;
  rsb_model_blob blob_;
  int n_ = 0;
  Stage stages_[5];
  std::vector<int> pending_;                  // integrate() calls each replica has recorded since the last flush
  std::vector<uint8_t> gcMask_, gvMask_;
  std::atomic<int> nPending_{0};              // replicas with a non-zero count
  std::vector<int32_t> launchSub_;
  PinnedArray<uint8_t> launchMasks_, stateMask_;
  int kmax_ = 0;
  long viewLaunches_ = 0, viewFlushes_ = 0;
  long long flushPrepNs_ = 0, flushExchangeNs_ = 0;
  bool fuse_ = !(std::getenv("RSB_VIEW_FUSE") && std::atoi(std::getenv("RSB_VIEW_FUSE")) == 0);
  std::atomic<bool> wantState_{false}, wantContacts_{false}, wantGenf_{false};   // what the environments read: downloaded by every flush
  BigReaderLock stageMu_;                    // shared: an env stages one of its rows; exclusive: whole-array upload / refresh of the mirrors (after mu_)
  long queryLaunches_ = 0;
  bool fiberBatch_ = false;
  // set once per flush by the first env that asks, read by all: the N env bodies between two flushes may run on several threads
  std::atomic<bool> stateCacheValid_{false}, contactsValid_{false}, genfValid_{false}, queryValid_{false};
  std::recursive_mutex mu_;   // serialises every call into the C-ABI handle (not re-entrant per handle) and the lazy downloads
  PinnedArray<float> genf_;
  std::vector<float> tmpGc_, tmpGv_;
  PinnedArray<int32_t> cnt_;
  PinnedArray<rsb_contact> con_;
};

class Ground {};
/// raisim::TerrainProperties [RECALL raisim/object/terrain/HeightMap.hpp]: parameters of a Perlin-noise terrain
struct TerrainProperties {
  double frequency = 0.1, zScale = 1.0, xSize = 10.0, ySize = 10.0;
  size_t xSamples = 100, ySamples = 100, fractalOctaves = 5;
  double fractalLacunarity = 2.0, fractalGain = 0.5, stepSize = 0.0;
  std::uint32_t seed = 6479;
  double heightOffset = 0.0;
};

class HeightMap {
 public:
  HeightMap(BatchedWorld* w, int xs, int ys, double xSize, double ySize, double cx, double cy, std::vector<double> h)
      : w_(w), xs_(xs), ys_(ys), xSize_(xSize), ySize_(ySize), cx_(cx), cy_(cy), h_(std::move(h)) {}
  double getHeight(double x, double y) const {  // same triangulation as the device collider
    const double dx = xSize_ / (xs_ - 1), dy = ySize_ / (ys_ - 1);
    double gx = (x - (cx_ - 0.5 * xSize_)) / dx, gy = (y - (cy_ - 0.5 * ySize_)) / dy;
    gx = gx < 0 ? 0 : (gx > xs_ - 1 ? xs_ - 1 : gx);
    gy = gy < 0 ? 0 : (gy > ys_ - 1 ? ys_ - 1 : gy);
    int ix = (int)std::floor(gx), iy = (int)std::floor(gy);
    if (ix > xs_ - 2) ix = xs_ - 2;
    if (iy > ys_ - 2) iy = ys_ - 2;
    const double fx = gx - ix, fy = gy - iy;
    const double h00 = h_[iy * xs_ + ix], h10 = h_[iy * xs_ + ix + 1], h01 = h_[(iy + 1) * xs_ + ix], h11 = h_[(iy + 1) * xs_ + ix + 1];
    return fx >= fy ? h00 + (h10 - h00) * fx + (h11 - h10) * fy : h00 + (h11 - h01) * fx + (h01 - h00) * fy;
  }
  const std::vector<double>& getHeightVector() const { return h_; }
 private:
  BatchedWorld* w_;
  int xs_, ys_;
  double xSize_, ySize_, cx_, cy_;
  std::vector<double> h_;
};

/// Per-env view of the articulated system (upstream raisim::ArticulatedSystem).
class ArticulatedSystem {
 public:
  ArticulatedSystem(BatchedWorld* w, int env) : w_(w), env_(env), gc_(w->gcDim()), gv_(w->dof()) {}
  /// fixed-base systems (URDF root link "world") expose the joints only, as upstream does: gcDim = dof = number of joints.
  /// (The batch keeps 7 + 6 inert base entries in front of them; putRow / getRow below pad and strip.)
  bool isFixedBase() const { return w_->blob().fixed_base != 0; }
  size_t getGeneralizedCoordinateDim() const { return (size_t)(w_->gcDim() - gcOff()); }
  size_t getDOF() const { return (size_t)(w_->dof() - gvOff()); }
  void setName(const std::string& n) { name_ = n; }
  const std::string& getName() const { return name_; }
  double getTotalMass() const { return rsb_model_total_mass(w_->model()); }
  /// upstream ArticulatedSystem::ignoreCollisionBetween; the replicas share one model, so it applies to all of them
  void ignoreCollisionBetween(size_t bodyIdx1, size_t bodyIdx2) { w_->ignoreCollisionBetween(bodyIdx1, bodyIdx2); }
  size_t getBodyIdx(const std::string& link) const {
    int i = rsb_model_body_index(w_->model(), link.c_str());
    RSFATAL_IF(i < 0, "getBodyIdx: no such body: " + link);
    return (size_t)i;
  }
  std::vector<std::string> getBodyNames() const {
    std::vector<std::string> n;
    for (int i = 0; i < w_->blob().nb; ++i) n.emplace_back(w_->blob().body_name[i]);
    return n;
  }

  void setGeneralizedCoordinate(const VecDyn& gc) { putRow(RSB_F_GC, gc); }
  void setGeneralizedVelocity(const VecDyn& gv) { putRow(RSB_F_GV, gv); }
  void setState(const VecDyn& gc, const VecDyn& gv) { putRow(RSB_F_GC, gc); putRow(RSB_F_GV, gv); }
  void getState(VecDyn& gc, VecDyn& gv) { getRow(RSB_F_GC, gc, w_->gcDim()); getRow(RSB_F_GV, gv, w_->dof()); }
#ifdef RAISIM_HAS_EIGEN
  /// upstream's getState(Eigen::VectorXd&, Eigen::VectorXd&) [RECALL]; the setters take Eigen vectors through VecDyn's converting constructor
  template <class A, class B> void getState(Eigen::MatrixBase<A>& gc, Eigen::MatrixBase<B>& gv) {
    getRow(RSB_F_GC, gc_, w_->gcDim()); getRow(RSB_F_GV, gv_, w_->dof());
    gc.derived() = gc_.e(); gv.derived() = gv_.e();
  }
#endif
  const VecDyn& getGeneralizedCoordinate() { getRow(RSB_F_GC, gc_, w_->gcDim()); return gc_; }
  const VecDyn& getGeneralizedVelocity() { getRow(RSB_F_GV, gv_, w_->dof()); return gv_; }

  void setControlMode(ControlMode::Type m) { w_->setControlMode(m); }
  void setIntegrationScheme(IntegrationScheme::Type scheme) { w_->setIntegrationScheme((int)scheme); }
  /// gains are shared by all replicas of the batched world (one robot model, one controller tuning)
  void setPdGains(const VecDyn& p, const VecDyn& d) {
    RSFATAL_IF(p.size() != getDOF() || d.size() != getDOF(), "setPdGains: gain vectors must have DOF entries");
    std::vector<float> kp((size_t)w_->dof(), 0.f), kd((size_t)w_->dof(), 0.f);     // (a fixed base's six inert entries lead the batch's rows)
    for (size_t i = 0; i < p.size(); ++i) { kp[gvOff() + i] = (float)p[i]; kd[gvOff() + i] = (float)d[i]; }
    w_->setPdGains(kp.data(), kd.data());
  }
  void setPdTarget(const VecDyn& pTarget, const VecDyn& dTarget) { putRow(RSB_F_PTARGET, pTarget); putRow(RSB_F_DTARGET, dTarget); }
  void setGeneralizedForce(const VecDyn& tau) { putRow(RSB_F_TAU_FF, tau); }
  /// upstream's one-sided setters and the common dimension aliases
  void setPTarget(const VecDyn& pTarget) { putRow(RSB_F_PTARGET, pTarget); }
  void setDTarget(const VecDyn& dTarget) { putRow(RSB_F_DTARGET, dTarget); }
  size_t getGeneralizedVelocityDim() const { return getDOF(); }
  /// upstream ArticulatedSystem::getPosition(bodyIdx, point_B, point_W): a point given in the body frame, in the world frame
  void getPosition(size_t body, const Vec<3>& pointB, Vec<3>& pointW) {
    Vec<3> p; Mat<3, 3> R;
    getBodyPosition(body, p); getBodyOrientation(body, R);
    for (int r = 0; r < 3; ++r) pointW[r] = p[r] + R(r, 0) * pointB[0] + R(r, 1) * pointB[1] + R(r, 2) * pointB[2];
  }
  /// upstream getVelocity(bodyIdx, vel_w) / getAngularVelocity(bodyIdx, angVel_w): the body frame's origin
  void getVelocity(size_t body, Vec<3>& velW) { getFrameVelocity(body, velW); }
  void getAngularVelocity(size_t body, Vec<3>& angVelW) { getFrameAngularVelocity(body, angVelW); }
  /// ... and of a point given in the body frame: v_origin + w x (R point_B)
  void getVelocity(size_t body, const Vec<3>& pointB, Vec<3>& velW) {
    Vec<3> v, w; Mat<3, 3> R;
    getFrameVelocity(body, v); getFrameAngularVelocity(body, w); getBodyOrientation(body, R);
    double r[3];
    for (int k = 0; k < 3; ++k) r[k] = R(k, 0) * pointB[0] + R(k, 1) * pointB[1] + R(k, 2) * pointB[2];
    velW[0] = v[0] + w[1] * r[2] - w[2] * r[1]; velW[1] = v[1] + w[2] * r[0] - w[0] * r[2]; velW[2] = v[2] + w[0] * r[1] - w[1] * r[0];
  }
  void getBasePosition(Vec<3>& pos) { getBodyPosition(0, pos); }

  /// valid after World::integrate1() (upstream semantics): M(q) and h(q,u) of this env
  const MatDyn& getMassMatrix() {
    const int nv = w_->dof(), o = gvOff();
    std::vector<float> M((size_t)w_->numEnvs() * nv * nv);
    w_->getMassMatrices(M.data());
    M_.resize(nv - o, nv - o);
    for (int i = o; i < nv; ++i) for (int j = o; j < nv; ++j) M_(i - o, j - o) = M[((size_t)env_ * nv + i) * nv + j];
    return M_;
  }
  const MatDyn& getInverseMassMatrix() {
    const int nv = w_->dof(), o = gvOff();
    RSFATAL_IF(o != 0, "getInverseMassMatrix: not available for fixed-base systems (the batch inverts the floating-base matrix)");
    std::vector<float> Mi((size_t)w_->numEnvs() * nv * nv);
    w_->getInverseMassMatrices(Mi.data());
    Minv_.resize(nv, nv);
    for (int i = 0; i < nv; ++i) for (int j = 0; j < nv; ++j) Minv_(i, j) = Mi[((size_t)env_ * nv + i) * nv + j];
    return Minv_;
  }
  const VecDyn& getNonlinearities(const Vec<3>& /*gravity*/ = Vec<3>()) {
    const int nv = w_->dof(), o = gvOff();
    std::vector<float> h((size_t)w_->numEnvs() * nv);
    w_->getNonlinearitiesAll(h.data());
    h_.resize(nv - o);
    for (int i = o; i < nv; ++i) h_[i - o] = h[(size_t)env_ * nv + i];
    return h_;
  }
  /// contacts of the last integrate() of this env
  std::vector<Contact>& getContacts() {
    int kmax = 0, count = 0;
    const PinnedArray<rsb_contact>& con = w_->contactsOf(env_, count, kmax);
    contacts_.clear();
    for (int k = 0; k < count; ++k) contacts_.emplace_back(con[(size_t)env_ * kmax + k]);
    return contacts_;
  }
  // ---- frame queries (slow path, correctness only: forward kinematics of this env on the host from the model blob).
  // A "frame" is a body's joint frame, indexed like the bodies (getFrameIdxByName: joint name -> its child body).
  size_t getFrameIdxByName(const std::string& jointName) const {
    int i = rsb_model_joint_index(w_->model(), jointName.c_str());
    RSFATAL_IF(i < 0, "getFrameIdxByName: no such joint: " + jointName);
    return (size_t)i;
  }
  void getFramePosition(size_t frame, Vec<3>& p) { fk(); for (int c = 0; c < 3; ++c) p[c] = fkP_[3 * frame + c]; }
  void getFrameOrientation(size_t frame, Mat<3, 3>& R) { fk(); for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R(r, c) = fkR_[9 * frame + 3 * r + c]; }
  void getBodyPosition(size_t body, Vec<3>& p) { getFramePosition(body, p); }
  void getBodyOrientation(size_t body, Mat<3, 3>& R) { getFrameOrientation(body, R); }
  /// positional Jacobian (3 x DOF, world frame) of the frame origin: v_point = J * gv
  void getDenseFrameJacobian(size_t frame, MatDyn& J) { jac(frame, J, false); }
  /// rotational Jacobian (3 x DOF): w_body = J * gv
  void getDenseFrameRotationalJacobian(size_t frame, MatDyn& J) { jac(frame, J, true); }
  void getFrameVelocity(size_t frame, Vec<3>& v) { MatDyn J; jac(frame, J, false); mulJ(J, v); }
  void getFrameAngularVelocity(size_t frame, Vec<3>& w) { MatDyn J; jac(frame, J, true); mulJ(J, w); }
  /// External force (world frame) at the origin of `body`'s frame, applied as the generalized force J^T f through
  /// the feed-forward channel.  Unlike upstream it is NOT cleared after the next integrate(): call
  /// clearExternalForces() (or setGeneralizedForce) to remove it.
  void setExternalForce(size_t body, const Vec<3>& force) {
    MatDyn J; jac(body, J, false);
    VecDyn tau(getDOF());
    getRow(RSB_F_TAU_FF, tau, w_->dof());
    for (int d = 0; d < (int)getDOF(); ++d) tau[d] += J(0, d) * force[0] + J(1, d) * force[1] + J(2, d) * force[2];
    putRow(RSB_F_TAU_FF, tau);
  }
  /// External torque (world frame) on `body`: generalized force J_rot^T t, same lifetime rule as setExternalForce
  void setExternalTorque(size_t body, const Vec<3>& torque) {
    MatDyn J; jac(body, J, true);
    VecDyn tau(getDOF());
    getRow(RSB_F_TAU_FF, tau, w_->dof());
    for (int d = 0; d < (int)getDOF(); ++d) tau[d] += J(0, d) * torque[0] + J(1, d) * torque[1] + J(2, d) * torque[2];
    putRow(RSB_F_TAU_FF, tau);
  }
  /// upstream ArticulatedSystem::getGeneralizedForce(): what the actuators applied in the last integrate() - clipped PD +
  /// feed-forward on the joints, the feed-forward wrench on a floating base's six rows (zero before the first integrate())
  const VecDyn& getGeneralizedForce() {
    const int dim = w_->dof(), off = gvOff();
    if ((int)gf_.size() != dim - off) gf_.resize(dim - off);
    if (off == 0) { w_->readGeneralizedForce(env_, gf_.data(), dim); return gf_; }
    std::vector<double> full((size_t)dim);
    w_->readGeneralizedForce(env_, full.data(), dim);
    for (int i = off; i < dim; ++i) gf_[i - off] = full[i];
    return gf_;
  }
  void clearExternalForces() { VecDyn tau(getDOF()); putRow(RSB_F_TAU_FF, tau); }

  void getBaseOrientation(Mat<3, 3>& rot) {
    const VecDyn& q = fullGc();
    const double w = q[3], x = q[4], y = q[5], z = q[6];
    rot(0, 0) = 1 - 2 * (y * y + z * z); rot(0, 1) = 2 * (x * y - w * z);     rot(0, 2) = 2 * (x * z + w * y);
    rot(1, 0) = 2 * (x * y + w * z);     rot(1, 1) = 1 - 2 * (x * x + z * z); rot(1, 2) = 2 * (y * z - w * x);
    rot(2, 0) = 2 * (x * z - w * y);     rot(2, 1) = 2 * (y * z + w * x);     rot(2, 2) = 1 - 2 * (x * x + y * y);
  }

 private:
  const VecDyn& fullGc() { if ((int)fullq_.size() != w_->gcDim()) fullq_.resize(w_->gcDim()); w_->readRow(RSB_F_GC, env_, fullq_.data(), w_->gcDim()); return fullq_; }   // incl. the base entries of a fixed-base system
  int gcOff() const { return isFixedBase() ? 7 : 0; }
  int gvOff() const { return isFixedBase() ? 6 : 0; }
  void putRow(int field, const VecDyn& v) {
    const bool isq = field == RSB_F_GC || field == RSB_F_PTARGET;
    const int dim = isq ? w_->gcDim() : w_->dof(), off = isq ? gcOff() : gvOff();
    RSFATAL_IF((int)v.size() != dim - off, "ArticulatedSystem: vector has the wrong dimension");
    if (off == 0) { w_->stageRow(field, env_, v.data(), dim); return; }
    std::vector<double> full((size_t)dim, 0.0);          // fixed base: identity pose / zero velocity in front of the joints
    if (isq) full[3] = 1.0;
    for (int i = off; i < dim; ++i) full[i] = v[i - off];
    w_->stageRow(field, env_, full.data(), dim);
  }
  void getRow(int field, VecDyn& v, int dim) {
    const bool isq = field == RSB_F_GC || field == RSB_F_PTARGET;
    const int off = isq ? gcOff() : gvOff();
    if (off == 0) { if ((int)v.size() != dim) v.resize(dim); w_->readRow(field, env_, v.data(), dim); return; }
    std::vector<double> full((size_t)dim);
    w_->readRow(field, env_, full.data(), dim);
    v.resize(dim - off);
    for (int i = off; i < dim; ++i) v[i - off] = full[i];
  }
  // host forward kinematics of this env (world frame): fkR_ [nb][9] row-major, fkP_ [nb][3], fkA_ [nb][3] joint axes
  void fk() {
    const rsb_model_blob& b = w_->blob();
    Mat<3, 3> R0; getBaseOrientation(R0);
    const VecDyn& q = fullGc();
    fkR_.assign(9 * b.nb, 0.0); fkP_.assign(3 * b.nb, 0.0); fkA_.assign(3 * b.nb, 0.0);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) fkR_[3 * r + c] = R0(r, c);
    for (int c = 0; c < 3; ++c) fkP_[c] = q[c];
    for (int i = 1; i < b.nb; ++i) {
      const int p = b.parent[i];
      const double* Rp = &fkR_[9 * p];
      double Rt[9], Rj[9];                                   // Rt = Rp * rtree, then the joint rotation about `axis`
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
        double sacc = 0; for (int k = 0; k < 3; ++k) sacc += Rp[3 * r + k] * b.rtree[i][3 * k + c];
        Rt[3 * r + c] = sacc;
      }
      const double* ax = b.axis[i];
      const double qi = q[6 + i];
      for (int c = 0; c < 3; ++c) {
        double t = 0; for (int k = 0; k < 3; ++k) t += Rp[3 * c + k] * b.ptree[i][k];
        fkP_[3 * i + c] = fkP_[3 * p + c] + t;
      }
      for (int r = 0; r < 3; ++r) fkA_[3 * i + r] = Rt[3 * r] * ax[0] + Rt[3 * r + 1] * ax[1] + Rt[3 * r + 2] * ax[2];
      if (b.jtype[i] == RSB_JOINT_REVOLUTE) {
        const double cs = std::cos(qi), sn = std::sin(qi), v = 1 - cs, x = ax[0], y = ax[1], z = ax[2];
        const double Rq[9] = {cs + x * x * v, x * y * v - z * sn, x * z * v + y * sn, y * x * v + z * sn, cs + y * y * v,
                              y * z * v - x * sn, z * x * v - y * sn, z * y * v + x * sn, cs + z * z * v};
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
          double sacc = 0; for (int k = 0; k < 3; ++k) sacc += Rt[3 * r + k] * Rq[3 * k + c];
          Rj[3 * r + c] = sacc;
        }
      } else {
        for (int k = 0; k < 9; ++k) Rj[k] = Rt[k];
        for (int c = 0; c < 3; ++c) fkP_[3 * i + c] += fkA_[3 * i + c] * qi;
      }
      for (int k = 0; k < 9; ++k) fkR_[9 * i + k] = Rj[k];
    }
  }
  void jac(size_t frame, MatDyn& J, bool rotational) {
    fk();
    const rsb_model_blob& b = w_->blob();
    const int nv = w_->dof();
    J.resize(3, nv);
    const double* pf = &fkP_[3 * frame];
    if (rotational) { for (int c = 0; c < 3; ++c) J(c, 3 + c) = 1.0; }
    else {
      const double r[3] = {pf[0] - fkP_[0], pf[1] - fkP_[1], pf[2] - fkP_[2]};
      for (int c = 0; c < 3; ++c) J(c, c) = 1.0;
      J(0, 4) = r[2]; J(0, 5) = -r[1]; J(1, 3) = -r[2]; J(1, 5) = r[0]; J(2, 3) = r[1]; J(2, 4) = -r[0];   // -[r]x
    }
    for (int j = (int)frame; j >= 1; j = b.parent[j]) {
      const double* a = &fkA_[3 * j];
      const int d = 5 + j;
      if (b.jtype[j] == RSB_JOINT_REVOLUTE) {
        if (rotational) { for (int c = 0; c < 3; ++c) J(c, d) = a[c]; }
        else {
          const double r[3] = {pf[0] - fkP_[3 * j], pf[1] - fkP_[3 * j + 1], pf[2] - fkP_[3 * j + 2]};
          J(0, d) = a[1] * r[2] - a[2] * r[1]; J(1, d) = a[2] * r[0] - a[0] * r[2]; J(2, d) = a[0] * r[1] - a[1] * r[0];
        }
      } else if (!rotational) {
        for (int c = 0; c < 3; ++c) J(c, d) = a[c];
      }
    }
    if (isFixedBase()) {     // upstream's Jacobians of a fixed-base system have one column per joint
      MatDyn Jj; Jj.resize(3, nv - 6);
      for (int c = 0; c < 3; ++c) for (int d = 6; d < nv; ++d) Jj(c, d - 6) = J(c, d);
      J = Jj;
    }
  }
  void mulJ(const MatDyn& J, Vec<3>& out) {
    const VecDyn& u = getGeneralizedVelocity();
    for (int c = 0; c < 3; ++c) { double sacc = 0; for (size_t d = 0; d < J.cols(); ++d) sacc += J(c, d) * u[d]; out[c] = sacc; }
  }
  BatchedWorld* w_;
  int env_;
  std::string name_;
  VecDyn gc_, gv_, h_, fullq_, gf_;
  std::vector<double> fkR_, fkP_, fkA_;
  MatDyn M_, Minv_;
  std::vector<Contact> contacts_;
};

/// Opens a scope in which default-constructed raisim::World objects become replicas 0, 1, 2, ... of ONE shared
/// BatchedWorld with `numEnvs` replicas (created by the first addArticulatedSystem call inside the scope).  This is how
/// VectorizedEnvironment<ENV> puts N unmodified Environment objects - each of which does
/// `world_ = std::make_unique<raisim::World>()` - onto one GPU batch.
class BatchScope {
 public:
  BatchScope(int numEnvs, int device = 0) : n_(numEnvs), device_(device), prev_(active()) { active() = this; }
  ~BatchScope() { active() = prev_; }
  BatchScope(const BatchScope&) = delete;
  BatchScope& operator=(const BatchScope&) = delete;
  static BatchScope*& active() { static thread_local BatchScope* a = nullptr; return a; }
  /// the shared world (null until the first World of the scope has loaded its URDF); outlives the scope
  std::shared_ptr<BatchedWorld> shared() const { return shared_; }
  int numEnvs() const { return n_; }
 private:
  friend class World;
  int n_, device_, next_ = 0;
  std::string urdf_;
  std::shared_ptr<BatchedWorld> shared_;
  BatchScope* prev_;
};

/// Upstream raisim::World: one env = one replica of a BatchedWorld (see the header comment for the three ways it binds).
class World {
 public:
  World() {
    if (BatchScope* sc = BatchScope::active()) {
      RSFATAL_IF(sc->next_ >= sc->n_, "raisim::World: more Worlds constructed than the BatchScope has replicas");
      scope_ = sc; env_ = sc->next_++;
      if (sc->shared_) { keep_ = sc->shared_; shared_ = keep_.get(); }
    }
  }
  World(BatchedWorld& shared, int env) : shared_(&shared), env_(env) {
    RSFATAL_IF(env < 0 || env >= shared.numEnvs(), "raisim::World: replica index out of range");
  }
  static void setActivationKey(const std::string&) {}  // nothing to activate in a from-scratch build
  int replica() const { return env_; }
  BatchedWorld* batch() { return shared_; }

  ArticulatedSystem* addArticulatedSystem(const std::string& urdfPath, const std::string& /*resDir*/ = "") {
    RSFATAL_IF(robot_ != nullptr, "this build supports one ArticulatedSystem per World");
    if (!shared_) {
      if (scope_) {      // the first World of a BatchScope creates the shared batch, the others attach to it
        if (!scope_->shared_) { scope_->shared_ = std::make_shared<BatchedWorld>(urdfPath, scope_->n_, scope_->device_); scope_->urdf_ = urdfPath; }
        RSFATAL_IF(scope_->urdf_ != urdfPath, "raisim::World: every env of one batch must load the same URDF");
        keep_ = scope_->shared_;
      } else {
        keep_ = std::make_shared<BatchedWorld>(urdfPath, 1); env_ = 0;
      }
      shared_ = keep_.get();
      applyPending();
    }
    robot_ = std::make_unique<ArticulatedSystem>(shared_, env_);
    return robot_.get();
  }
  Ground* addGround(double zHeight = 0.0, const std::string& material = "default") {
    groundZ_ = zHeight; hasGround_ = true; terrainMaterial_ = material;
    if (shared_) { shared_->addGround(zHeight); shared_->setTerrainMaterial(material); }
    return &ground_;
  }
  HeightMap* addHeightMap(int xSamples, int ySamples, double xSize, double ySize, double centerX, double centerY,
                          const std::vector<double>& height, const std::string& material = "default") {
    RSFATAL_IF(!shared_, "addHeightMap: add the ArticulatedSystem first");
    shared_->addHeightMap(xSamples, ySamples, xSize, ySize, centerX, centerY, height);
    terrainMaterial_ = material; shared_->setTerrainMaterial(material);
    hm_ = std::make_unique<HeightMap>(shared_, xSamples, ySamples, xSize, ySize, centerX, centerY, height);
    return hm_.get();
  }
  /// Perlin-noise terrain (rsb_heightmap_perlin)
  HeightMap* addHeightMap(double centerX, double centerY, TerrainProperties& tp, const std::string& material = "default") {
    rsb_terrain_properties c{tp.frequency, tp.zScale, tp.xSize, tp.ySize, (int32_t)tp.xSamples, (int32_t)tp.ySamples,
                             (int32_t)tp.fractalOctaves, tp.seed, tp.fractalLacunarity, tp.fractalGain, tp.stepSize, tp.heightOffset};
    std::vector<float> h(tp.xSamples * tp.ySamples);
    RSB_CHECK(rsb_heightmap_perlin(&c, h.data()));
    return addHeightMap((int)tp.xSamples, (int)tp.ySamples, tp.xSize, tp.ySize, centerX, centerY, std::vector<double>(h.begin(), h.end()), material);
  }
  /// PNG terrain: height = pixel / max_pixel * heightScale + heightOffset (rsb_heightmap_png_*)
  HeightMap* addHeightMap(const std::string& pngFileName, double centerX, double centerY, double xSize, double ySize,
                          double heightScale, double heightOffset, const std::string& material = "default") {
    int xs = 0, ys = 0;
    RSB_CHECK(rsb_heightmap_png_size(pngFileName.c_str(), &xs, &ys));
    std::vector<float> h((size_t)xs * ys);
    RSB_CHECK(rsb_heightmap_png_read(pngFileName.c_str(), heightScale, heightOffset, h.data(), xs * ys));
    return addHeightMap(xs, ys, xSize, ySize, centerX, centerY, std::vector<double>(h.begin(), h.end()), material);
  }
  /// text terrain file: "xSamples ySamples xSize ySize" + heights (rsb_heightmap_text_*)
  HeightMap* addHeightMap(const std::string& raisimHeightMapFileName, double centerX, double centerY,
                          const std::string& material = "default") {
    int xs = 0, ys = 0; double sx = 0, sy = 0;
    RSB_CHECK(rsb_heightmap_text_size(raisimHeightMapFileName.c_str(), &xs, &ys, &sx, &sy));
    std::vector<float> h((size_t)xs * ys);
    RSB_CHECK(rsb_heightmap_text_read(raisimHeightMapFileName.c_str(), h.data(), xs * ys));
    return addHeightMap(xs, ys, sx, sy, centerX, centerY, std::vector<double>(h.begin(), h.end()), material);
  }
  void setTimeStep(double dt) { dt_ = dt; if (shared_) shared_->setTimeStep(dt); }
  double getTimeStep() const { return shared_ ? shared_->getTimeStep() : dt_; }
  double getWorldTime() const { return shared_ ? shared_->worldTimeOf(env_) : 0.0; }
  void setGravity(const Vec<3>& g) { need().setGravity(g); }
  void setERP(double erp, double erp2 = 0) { need().setERP(erp, erp2); }
  void setDefaultMaterial(double mu, double r = 0, double t = 0) { need().setDefaultMaterial(mu, r, t); }
  /// friction / restitution of a material pair; materials are named by <collision><material name=../> in the URDF and by the
  /// material argument of addGround / addHeightMap (in a batch every replica shares the table, like every other world parameter)
  void setMaterialPairProp(const std::string& m1, const std::string& m2, double mu, double r, double t) { need().setMaterialPairProp(m1, m2, mu, r, t); }
  void setContactSolverParam(double a0, double amin, double adec, int maxIter, double thr) { need().setContactSolverParam(a0, amin, adec, maxIter, thr); }
  /// advances THIS replica by one time step; launched together with the other replicas' pending integrate() calls
  void integrate() { need().integrateView(env_); }
  /// M, h queries of the current state (computed for the whole batch, idempotent); integrate2() then advances this replica
  void integrate1() { need().integrate1View(env_); }
  void integrate2() { need().integrateView(env_); }

 private:
  BatchedWorld& need() { RSFATAL_IF(!shared_, "World: add the ArticulatedSystem before configuring the solver"); return *shared_; }
  void applyPending() { if (dt_ > 0) shared_->setTimeStep(dt_); if (hasGround_) { shared_->addGround(groundZ_); shared_->setTerrainMaterial(terrainMaterial_); } }
  std::string terrainMaterial_ = "default";
  std::shared_ptr<BatchedWorld> keep_;     // owned 1-replica world, or a share of the BatchScope's world
  BatchedWorld* shared_ = nullptr;
  BatchScope* scope_ = nullptr;
  int env_ = 0;
  std::unique_ptr<ArticulatedSystem> robot_;
  std::unique_ptr<HeightMap> hm_;
  Ground ground_;
  double dt_ = 0, groundZ_ = 0;
  bool hasGround_ = false;
};

}  // namespace raisim
