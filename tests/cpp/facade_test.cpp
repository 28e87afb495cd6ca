// Compiles against include/raisim/*.hpp only (the way an upstream Environment.hpp would) and links librsb.so.
// Exercises: per-env raisim::World (N = 1 replica), ArticulatedSystem views, integrate1/2 queries, contacts,
// and the batched VectorizedEnvironment.  Exit code 0 = all checks passed.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <memory>
#include <vector>

#include "anymal_env/Environment.hpp"
#include "raisim/VectorizedEnvironment.hpp"
#include "raisim/World.hpp"

#define CHECK(c) do { if (!(c)) { std::printf("CHECK failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

// a minimal env whose step() throws on demand between its two integrate() calls (error path of the fiber scheduler)
class FlakyEnv : public raisim::RaisimGymEnv {
 public:
  FlakyEnv(const std::string& resourceDir, const Yaml::Node& cfg, bool) : RaisimGymEnv(resourceDir, cfg) {
    world_ = std::make_unique<raisim::World>();
    robot_ = world_->addArticulatedSystem(resourceDir_ + "/anymal_c_like.urdf");
    world_->addGround();
    obDim_ = 1; actionDim_ = 1;
    gc_.resize(robot_->getGeneralizedCoordinateDim()); gv_.resize(robot_->getDOF());
    gc_[2] = 0.6; gc_[3] = 1.0;
  }
  void init() final {}
  void reset() final { robot_->setState(gc_, gv_); }
  void observe(raisim::EigenVecRef ob) final { ob[0] = (float)robot_->getGeneralizedCoordinate()[2]; }
  float step(const raisim::ConstEigenVecRef& action) final {
    world_->integrate();
    if (action[0] > 100.f) throw std::runtime_error("flaky");
    world_->integrate();
    return 0.f;
  }
  bool isTerminalState(float& terminalReward) final { terminalReward = 0.f; return false; }

 private:
  raisim::ArticulatedSystem* robot_;
  raisim::VecDyn gc_, gv_;
};

int main(int argc, char** argv) {
  if (argc < 2) { std::printf("usage: facade_test <urdf>\n"); return 2; }
  const std::string urdf = argv[1];
  if (argc >= 3 && std::string(argv[2]) == "config1") {
    // BASELINE.json configs[0] as SURVEY.md 8d writes it: ONE env through raisim::World (N = 1 on the device), flat ground, dt 0.0025,
    // 4000 integrate() calls from gc_init, gv = 0, kp 50 / kd 0.2 on the twelve joints, targets = gc_init.  Prints what tests/test_cpp_facade.py compares
    // with the oracle's run of the same configuration: base height, velocity norm, the collision primitives in contact at the end.
    try {
      raisim::World world;
      world.setTimeStep(0.0025);
      auto* anymal = world.addArticulatedSystem(urdf);
      world.addGround();
      raisim::VecDyn gc(19), gv(18), kp(18), kd(18), dT(18);
      const double init[19] = {0, 0, 0.50, 1, 0, 0, 0, 0.03, 0.4, -0.8, -0.03, 0.4, -0.8, 0.03, -0.4, 0.8, -0.03, -0.4, 0.8};
      for (int i = 0; i < 19; ++i) gc[i] = init[i];
      for (int j = 0; j < 12; ++j) { kp[6 + j] = 50.0; kd[6 + j] = 0.2; }
      anymal->setState(gc, gv);
      anymal->setControlMode(raisim::ControlMode::PD_PLUS_FEEDFORWARD_TORQUE);
      anymal->setPdGains(kp, kd);
      anymal->setPdTarget(gc, dT);
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 4000; ++i) world.integrate();
      anymal->getState(gc, gv);
      const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      double vmax = 0;
      for (int i = 0; i < 18; ++i) vmax = std::max(vmax, std::fabs(gv[i]));
      std::printf("config1 z=%.6f qw=%.6f vmax=%.6f t=%.4f steps_per_s=%.0f contacts=", gc[2], gc[3], vmax, world.getWorldTime(), 4000.0 / sec);
      for (auto& c : anymal->getContacts()) std::printf("%d,", (int)c.getCollisionIndex());
      std::printf("\n");
      return 0;
    } catch (const std::exception& e) { std::printf("config1 failed: %s\n", e.what()); return 1; }
  }
  try {
    // ---- the upstream single-env pattern (what an rsg_anymal Environment.hpp does)
    raisim::World world;
    world.setTimeStep(0.0025);
    auto* anymal = world.addArticulatedSystem(urdf);
    world.addGround();
    anymal->setName("anymal");
    const int gcDim = (int)anymal->getGeneralizedCoordinateDim(), gvDim = (int)anymal->getDOF();
    CHECK(gcDim == 19 && gvDim == 18);
    raisim::VecDyn gc(gcDim), gv(gvDim), pT(gcDim), dT(gvDim), kp(gvDim), kd(gvDim);
    const double nominal[12] = {0.03, 0.4, -0.8, -0.03, 0.4, -0.8, 0.03, -0.4, 0.8, -0.03, -0.4, 0.8};
    gc[2] = 0.57; gc[3] = 1.0;
    for (int j = 0; j < 12; ++j) { gc[7 + j] = nominal[j]; kp[6 + j] = 400.0; kd[6 + j] = 10.0; }
    pT = gc.v;
    anymal->setState(gc, gv);
    anymal->setControlMode(raisim::ControlMode::PD_PLUS_FEEDFORWARD_TORQUE);
    anymal->setPdGains(kp, kd);
    anymal->setPdTarget(pT, dT);
    anymal->setPTarget(pT); anymal->setDTarget(dT);   // (upstream's one-sided setters: the same targets again)
    world.integrate1();
    const auto& M = anymal->getMassMatrix();
    CHECK(std::fabs(M(0, 0) - anymal->getTotalMass()) < 1e-3 && std::fabs(M(3, 7) - M(7, 3)) < 1e-6);
    const auto& h = anymal->getNonlinearities();
    CHECK(std::fabs(h[2] - anymal->getTotalMass() * 9.81) < 1e-2);
    world.integrate2();
    for (int i = 0; i < 799; ++i) world.integrate();
    CHECK(std::fabs(world.getWorldTime() - 2.0) < 1e-9);
    auto& contacts = anymal->getContacts();
    CHECK(contacts.size() == 4);
    double fz = 0;
    for (auto& c : contacts) { fz += c.getImpulse()[2] / world.getTimeStep(); CHECK(c.getlocalBodyIndex() % 3 == 0); }
    CHECK(std::fabs(fz / (anymal->getTotalMass() * 9.81) - 1.0) < 0.05);
    const auto& q = anymal->getGeneralizedCoordinate();
    CHECK(q[2] > 0.45 && q[2] < 0.62);
    raisim::Mat<3, 3> rot;
    anymal->getBaseOrientation(rot);
    CHECK(rot(2, 2) > 0.99);
    // ---- frame queries (host FK): J * gv = frame velocity; finite-difference of the frame position over one step
    {
      const size_t foot = anymal->getFrameIdxByName("LF_KFE");
      CHECK(foot == anymal->getBodyIdx("LF_SHANK") || foot > 0);
      raisim::VecDyn g2(gcDim), v2(gvDim);
      g2 = gc.v; g2[2] = 2.0; g2[3] = 0.9; g2[4] = 0.1; g2[5] = -0.2; g2[6] = 0.3;
      { double nrm = std::sqrt(g2[3] * g2[3] + g2[4] * g2[4] + g2[5] * g2[5] + g2[6] * g2[6]); for (int k = 3; k < 7; ++k) g2[k] /= nrm; }
      for (int d = 0; d < gvDim; ++d) v2[d] = 0.3 * std::sin(1.0 + d);
      anymal->setState(g2, v2);
      raisim::Vec<3> p0, p1, vel, w;
      anymal->getFramePosition(foot, p0);
      world.integrate();
      anymal->getFramePosition(foot, p1);
      anymal->getFrameVelocity(foot, vel);          // J(q+) * u+
      anymal->getFrameAngularVelocity(foot, w);
      for (int c = 0; c < 3; ++c) CHECK(std::fabs((p1[c] - p0[c]) / world.getTimeStep() - vel[c]) < 2e-2);
      {   // upstream's point queries: a point of the body frame in the world, its velocity = v_origin + w x r; finite difference over the step
        raisim::Vec<3> pb, pw, vw, vo, bp;
        pb[0] = 0.05; pb[1] = -0.02; pb[2] = -0.1;
        anymal->getPosition(foot, pb, pw);
        anymal->getVelocity(foot, pb, vw);
        anymal->getVelocity(foot, vo);
        CHECK(std::fabs(vo[0] - vel[0]) < 1e-12 && std::fabs(vo[2] - vel[2]) < 1e-12);
        raisim::Mat<3, 3> Rb;
        anymal->getBodyOrientation(foot, Rb);
        double r[3];
        for (int k = 0; k < 3; ++k) r[k] = Rb(k, 0) * pb[0] + Rb(k, 1) * pb[1] + Rb(k, 2) * pb[2];
        CHECK(std::fabs(pw[0] - (p1[0] + r[0])) < 1e-12 && std::fabs(pw[2] - (p1[2] + r[2])) < 1e-12);
        CHECK(std::fabs(vw[0] - (vel[0] + w[1] * r[2] - w[2] * r[1])) < 1e-12);
        raisim::Vec<3> wb;
        anymal->getAngularVelocity(foot, wb);
        CHECK(std::fabs(wb[1] - w[1]) < 1e-12);
        anymal->getBasePosition(bp);
        CHECK(std::fabs(bp[2] - anymal->getGeneralizedCoordinate()[2]) < 1e-6);
        CHECK(anymal->getGeneralizedVelocityDim() == (size_t)gvDim);
      }
      raisim::MatDyn J;
      anymal->getDenseFrameJacobian(foot, J);
      CHECK(J.rows() == 3 && (int)J.cols() == gvDim && std::fabs(J(0, 0) - 1.0) < 1e-12 && std::fabs(J(2, 17)) < 1e-12);   // RH joints do not move the LF foot
      raisim::Mat<3, 3> Rf;
      anymal->getFrameOrientation(foot, Rf);
      double det = Rf(0, 0) * (Rf(1, 1) * Rf(2, 2) - Rf(1, 2) * Rf(2, 1)) - Rf(0, 1) * (Rf(1, 0) * Rf(2, 2) - Rf(1, 2) * Rf(2, 0)) +
                   Rf(0, 2) * (Rf(1, 0) * Rf(2, 1) - Rf(1, 1) * Rf(2, 0));
      CHECK(std::fabs(det - 1.0) < 1e-6);
      // an upward external force m*g at the base origin cancels gravity for the base (PD holds the legs): hover
      raisim::VecDyn gz(gcDim), vz(gvDim);
      gz = gc.v; gz[2] = 3.0;
      anymal->setState(gz, vz);
      raisim::Vec<3> f; f[2] = anymal->getTotalMass() * 9.81;
      anymal->setExternalForce(0, f);
      for (int i = 0; i < 40; ++i) world.integrate();
      CHECK(std::fabs(anymal->getGeneralizedVelocity()[2]) < 0.05);     // free fall would be at -0.98 m/s by now
      anymal->clearExternalForces();
      // a pure yaw torque on the floating base spins it about z and nothing else (no gravity coupling in free fall)
      anymal->setState(gz, vz);
      raisim::Vec<3> tq; tq[2] = 20.0;
      anymal->setExternalTorque(0, tq);
      anymal->setIntegrationScheme(raisim::IntegrationScheme::SEMI_IMPLICIT);
      for (int i = 0; i < 20; ++i) world.integrate();
      { const auto& uu = anymal->getGeneralizedVelocity(); CHECK(uu[5] > 0.05 && std::fabs(uu[3]) < 0.05 * uu[5] && std::fabs(uu[4]) < 0.05 * uu[5]); }
      anymal->clearExternalForces();
      anymal->setState(gc, gv);
      for (int i = 0; i < 400; ++i) world.integrate();
      anymal->getGeneralizedCoordinate();   // refresh the cached row that `q` refers to
    }
    std::printf("single-env World: z=%.3f fz/mg=%.3f contacts=%zu\n", q[2], fz / (anymal->getTotalMass() * 9.81), contacts.size());

    // ---- the batched vectorised environment (one fused launch per step for all envs)
    raisim::VecEnvConfig cfg;
    cfg.num_envs = 512;
    cfg.gc_init.assign(19, 0.0);
    cfg.gc_init[2] = 0.57; cfg.gc_init[3] = 1.0;
    for (int j = 0; j < 12; ++j) cfg.gc_init[7 + j] = nominal[j];
    raisim::DeviceVectorizedEnvironment env(urdf, cfg);
    env.init();
    CHECK(env.getObDim() == 34 && env.getActionDim() == 12 && env.getNumOfEnvs() == 512);
    std::vector<float> ob((size_t)512 * 34), act((size_t)512 * 12, 0.f), rew(512);
    std::unique_ptr<bool[]> done(new bool[512]);
    unsigned s = 12345u;
    int resets = 0;
    for (int it = 0; it < 50; ++it) {
      for (auto& a : act) { s = s * 1664525u + 1013904223u; a = ((s >> 8) / 16777216.0f - 0.5f) * 2.0f; }
      env.step(act.data(), 512, 12, rew.data(), done.get());
      for (int e = 0; e < 512; ++e) resets += done[e] ? 1 : 0;
    }
    env.observe(ob.data(), 512, 34);
    for (int e = 0; e < 512; ++e) {
      CHECK(std::isfinite(ob[(size_t)e * 34]) && ob[(size_t)e * 34] > 0.2f && ob[(size_t)e * 34] < 0.7f);
      CHECK(ob[(size_t)e * 34 + 3] > 0.5f);  // body z-axis still points up
      CHECK(std::isfinite(rew[e]));
    }
    std::printf("DeviceVectorizedEnvironment: 50 control steps x 512 envs, %d resets, mean height %.3f\n", resets, ob[0]);

    // ---- the policy in the loop on the device: K control steps with the in-repo linear stage, pipelined == lock-step bit for bit
    {
      std::vector<float> W0((size_t)12 * 34), bias(12, 0.05f), obA((size_t)512 * 34), obB((size_t)512 * 34), obC((size_t)512 * 34), obD((size_t)512 * 34);
      unsigned ws = 99u;
      for (auto& x : W0) { ws = ws * 1664525u + 1013904223u; x = ((ws >> 8) / 16777216.0f - 0.5f) * 0.6f; }     // strong enough to make robots fall and reset
      // mode 0 lock-step, 1 pipelined, 2 RESIDENT (round 6: each run is ONE launch of the step kernel), 3 lock-step WITHOUT the in-place weight update
      for (int mode = 0; mode < 4; ++mode) {
        std::vector<float> W = W0;
        raisim::DeviceVectorizedEnvironment cl(urdf, cfg);
        cl.init();
        const bool granted = cl.setStepPipelining(mode == 1);
        if (mode == 1) CHECK(granted);
        if (mode == 2) CHECK(cl.setStepResidency(true));
        cl.rolloutLinear(40, W.data(), bias.data(), 2.0f);
        // the learner updates its weights IN PLACE - same pointers, new contents (the upload used to be skipped on pointer identity)
        if (mode != 3) for (auto& x : W) x = -x;
        cl.rolloutLinear(25, W.data(), bias.data(), 2.0f);
        CHECK(cl.join() == RSB_OK);
        if (mode == 2) CHECK(rsb_step_residency_launches(cl.world().handle()) == 2);
        cl.observe(mode == 0 ? obA.data() : mode == 1 ? obB.data() : mode == 2 ? obC.data() : obD.data(), 512, 34);
      }
      int moved = 0, differs = 0;
      for (size_t i = 0; i < obA.size(); ++i) { CHECK(obA[i] == obB[i]); CHECK(obA[i] == obC[i]); CHECK(std::isfinite(obA[i])); differs += obA[i] != obD[i] ? 1 : 0; }
      CHECK(differs > 1000);       // the second run saw the updated weights
      for (int e = 0; e < 512; ++e) moved += std::fabs(obA[(size_t)e * 34] - ob[(size_t)e * 34]) > 1e-3f ? 1 : 0;
      CHECK(moved > 256);
      std::printf("DeviceVectorizedEnvironment::rolloutLinear: 65 control steps x 512 envs, resident == pipelined == lock-step bit for bit; in-place weight update seen\n");
    }
    // ---- ... and with an actor network (34 -> 64 -> 32 -> 12, tanh) as the stage
    {
      const std::vector<int> dims = {34, 64, 32, 12};
      std::vector<std::vector<float>> Wm(3), Bm(3);
      unsigned ws = 4242u;
      for (int l = 0; l < 3; ++l) {
        Wm[l].resize((size_t)dims[l] * dims[l + 1]); Bm[l].assign(dims[l + 1], 0.01f * (l + 1));
        const float bound = (l == 2 ? 0.5f : 1.0f) / std::sqrt((float)dims[l]);
        for (auto& x : Wm[l]) { ws = ws * 1664525u + 1013904223u; x = ((ws >> 8) / 16777216.0f - 0.5f) * 2.0f * bound; }
      }
      const std::vector<const float*> wp = {Wm[0].data(), Wm[1].data(), Wm[2].data()}, bp = {Bm[0].data(), Bm[1].data(), Bm[2].data()};
      std::vector<float> obA((size_t)512 * 34), obB((size_t)512 * 34), obC((size_t)512 * 34);
      for (int mode = 0; mode < 3; ++mode) {      // lock-step, pipelined, resident
        for (int l = 0; l < 3; ++l) for (auto& x : Bm[l]) x = 0.01f * (l + 1);
        raisim::DeviceVectorizedEnvironment cl(urdf, cfg);
        cl.init();
        cl.setStepPipelining(mode == 1);
        if (mode == 2) CHECK(cl.setStepResidency(true));
        cl.rolloutMlp(30, dims, wp, bp, RSB_ACT_TANH, 2.0f);
        for (int l = 0; l < 3; ++l) for (auto& x : Bm[l]) x = -x;      // in-place update of the BIASES alone (the cache key used to ignore them)
        cl.rolloutMlp(20, dims, wp, bp, RSB_ACT_TANH, 2.0f);
        CHECK(cl.join() == RSB_OK);
        cl.observe(mode == 0 ? obA.data() : mode == 1 ? obB.data() : obC.data(), 512, 34);
      }
      for (size_t i = 0; i < obA.size(); ++i) { CHECK(obA[i] == obB[i]); CHECK(obA[i] == obC[i]); CHECK(std::isfinite(obA[i])); }
      std::printf("DeviceVectorizedEnvironment::rolloutMlp: 50 control steps x 512 envs, resident == pipelined == lock-step bit for bit\n");
    }

    // ---- N per-env World VIEWS of one batch: N integrate() calls = ONE launch in which every replica advances once
    {
      const int NV = 8;
      raisim::BatchedWorld batch(urdf, NV), twin(urdf, NV);
      batch.setTimeStep(0.0025); twin.setTimeStep(0.0025);
      std::vector<std::unique_ptr<raisim::World>> views;
      std::vector<raisim::ArticulatedSystem*> robots;
      std::vector<float> kpf(gvDim, 0.f), kdf(gvDim, 0.f);
      for (int j = 6; j < gvDim; ++j) { kpf[j] = 50.f; kdf[j] = 0.2f; }
      twin.setPdGains(kpf.data(), kdf.data());
      std::vector<float> tgc((size_t)NV * gcDim), tgv((size_t)NV * gvDim, 0.f), tpt((size_t)NV * gcDim, 0.f), tdt((size_t)NV * gvDim, 0.f);
      for (int e = 0; e < NV; ++e) {
        views.push_back(std::make_unique<raisim::World>(batch, e));
        robots.push_back(views.back()->addArticulatedSystem(urdf));
        raisim::VecDyn g(gcDim), v(gvDim), p(gcDim), d(gvDim);
        g = gc.v; g[2] = 0.54 + 0.01 * e; g[0] = 0.3 * e;
        p = g.v; p[7] += 0.05 * e;
        for (int i = 0; i < gcDim; ++i) { tgc[(size_t)e * gcDim + i] = (float)g[i]; tpt[(size_t)e * gcDim + i] = (float)p[i]; }
        robots.back()->setState(g, v);
        robots.back()->setPdTarget(p, d);
      }
      robots[0]->setPdGains(kp, kd);   // gains are shared by the replicas (kp 400 / kd 10 from above) ...
      for (int j = 6; j < gvDim; ++j) { kpf[j] = (float)kp[j]; kdf[j] = (float)kd[j]; }
      twin.setPdGains(kpf.data(), kdf.data());
      twin.setState(tgc.data(), tgv.data());
      twin.setPdTarget(tpt.data(), tdt.data());
      for (int rep = 0; rep < 3; ++rep) {
        const long before = batch.viewLaunches();
        for (int e = 0; e < NV; ++e) {
          views[e]->integrate();
          if (e == 2) {               // a read while the batch is incomplete must fail loudly, not return a stale or half-stepped state
            bool threw = false;
            try { robots[e]->getGeneralizedCoordinate(); } catch (const std::exception&) { threw = true; }
            CHECK(threw);
          }
        }
        CHECK(batch.viewLaunches() == before + 1 && batch.pendingViews() == 0);
        twin.integrate(1);
      }
      CHECK(std::fabs(batch.getWorldTime() - 3 * 0.0025) < 1e-12);      // 3 steps of world time, not 3 * NV
      std::vector<float> bgc((size_t)NV * gcDim), bgv((size_t)NV * gvDim);
      twin.getState(bgc.data(), bgv.data());
      for (int e = 0; e < NV; ++e) {
        const auto& qe = robots[e]->getGeneralizedCoordinate();
        for (int i = 0; i < gcDim; ++i) CHECK((float)qe[i] == bgc[(size_t)e * gcDim + i]);   // bit-identical to the batched path
      }
      // a partial batch: only replicas 1 and 5 step (masked launch); the others must not move
      std::vector<float> before_gc((size_t)NV * gcDim), after_gc((size_t)NV * gcDim), tmp_gv((size_t)NV * gvDim);
      batch.getState(before_gc.data(), tmp_gv.data());
      views[1]->integrate(); views[5]->integrate();
      CHECK(batch.pendingViews() == 2);
      batch.flushViews();
      batch.getState(after_gc.data(), tmp_gv.data());
      for (int e = 0; e < NV; ++e) {
        bool moved = false;
        for (int i = 0; i < gcDim; ++i) moved |= before_gc[(size_t)e * gcDim + i] != after_gc[(size_t)e * gcDim + i];
        CHECK(moved == (e == 1 || e == 5));
      }
      // integrate1() of N views between two flushes is ONE whole-batch query launch, renewed after the next launch / staged write
      {
        const long q0 = batch.queryLaunches();
        for (int e = 0; e < NV; ++e) views[e]->integrate1();
        CHECK(batch.queryLaunches() == q0 + 1);
        const double m00 = robots[3]->getMassMatrix()(0, 0);
        CHECK(std::fabs(m00 - robots[3]->getTotalMass()) < 1e-3 * m00);
        for (int e = 0; e < NV; ++e) views[e]->integrate2();
        CHECK(batch.pendingViews() == 0);
        for (int e = 0; e < NV; ++e) views[e]->integrate1();
        CHECK(batch.queryLaunches() == q0 + 2);
        raisim::VecDyn g2(gcDim), v2(gvDim);
        g2 = robots[0]->getGeneralizedCoordinate().v; v2 = robots[0]->getGeneralizedVelocity().v;
        robots[0]->setState(g2, v2);                       // a staged write invalidates the query
        views[0]->integrate1();
        CHECK(batch.queryLaunches() == q0 + 3);
      }
      std::printf("World views: %d replicas, one launch per round of integrate() / integrate1() calls, masked partial flush OK\n", NV);
    }

    // ---- upstream's template: N arbitrary ENVIRONMENT objects (tests/cpp/anymal_env/Environment.hpp) on one batch
    {
      const int NE = 64;
      const std::string resourceDir = urdf.substr(0, urdf.find_last_of('/'));
      const std::string yaml =
          "num_envs: 64\nnum_threads: 8   # ignored\nsimulation_dt: 0.0025\ncontrol_dt: 0.01\nrender: false\naction_std: 0.3\n"
          "reward:\n  forwardVel:\n    coeff: 0.3\n  torque:\n    coeff: -4e-5\n";
      raisim::VectorizedEnvironment<raisim::ENVIRONMENT> venv(resourceDir, yaml, /*normalizeObservation=*/false);
      CHECK(venv.getNumOfEnvs() == NE && venv.getObDim() == 34 && venv.getActionDim() == 12);
      raisim::VecEnvConfig dc;
      dc.num_envs = NE; dc.gc_init = cfg.gc_init; dc.torque_reward_coeff = -4e-5; dc.forward_vel_reward_coeff = 0.3;
      raisim::DeviceVectorizedEnvironment denv(urdf, dc);
      denv.init();
      std::vector<float> a((size_t)NE * 12), r1(NE), r2(NE), o1((size_t)NE * 34), o2((size_t)NE * 34);
      std::unique_ptr<bool[]> d1(new bool[NE]), d2(new bool[NE]);
      unsigned sd = 777u;
      int ndone = 0;
      const long l0 = venv.batch()->viewLaunches();
      const int STEPS = 30;
      for (int it = 0; it < STEPS; ++it) {
        for (auto& x : a) { sd = sd * 1664525u + 1013904223u; x = ((sd >> 8) / 16777216.0f - 0.5f) * (it % 7 == 6 ? 8.0f : 2.0f); }
        venv.step(a.data(), NE, 12, r1.data(), d1.get());
        denv.step(a.data(), NE, 12, r2.data(), d2.get());
        venv.observe(o1.data(), NE, 34, false);
        denv.observe(o2.data(), NE, 34);
        for (int e = 0; e < NE; ++e) {
          CHECK(d1[e] == d2[e]);
          CHECK(std::fabs(r1[e] - r2[e]) < 1e-4f);
          ndone += d1[e] ? 1 : 0;
          for (int k = 0; k < 34; ++k) CHECK(std::fabs(o1[(size_t)e * 34 + k] - o2[(size_t)e * 34 + k]) < 1e-4f);
        }
      }
      CHECK(venv.batch()->viewLaunches() - l0 == STEPS && venv.batch()->viewFlushes() >= STEPS);       // the 4 integrate() calls of a control step: ONE fused launch of 4 sub-steps for all 64 envs
      // an exception inside one env's step() surfaces from venv.step() and must leave the batch usable: env 5 throws between
      // its two integrate() calls, when envs 0-4 have finished their bodies with two recorded, un-flushed integrate() calls each
      // (dropped with the failed step: nothing is launched)
      {
        const std::string yaml8 = "num_envs: 8\nsimulation_dt: 0.0025\ncontrol_dt: 0.005\nrender: false\n";
        raisim::VectorizedEnvironment<FlakyEnv> fenv(resourceDir, yaml8, /*normalizeObservation=*/false);
        std::vector<float> fa(8, 0.f), fr(8);
        std::unique_ptr<bool[]> fd(new bool[8]);
        fenv.reset();
        fenv.step(fa.data(), 8, 1, fr.data(), fd.get());
        const long fl0 = fenv.batch()->viewLaunches();
        fa[5] = 1000.f;
        bool threw = false;
        try { fenv.step(fa.data(), 8, 1, fr.data(), fd.get()); } catch (const std::exception& e) { threw = std::string(e.what()) == "flaky"; }
        CHECK(threw && fenv.batch()->viewLaunches() == fl0 && fenv.batch()->pendingViews() == 0);
        fa[5] = 0.f;
        fenv.reset();
        fenv.step(fa.data(), 8, 1, fr.data(), fd.get());            // works again: one fused launch of two sub-steps, nothing left pending
        CHECK(fenv.batch()->viewLaunches() == fl0 + 1 && fenv.batch()->pendingViews() == 0);
        std::printf("VectorizedEnvironment: an exception in one env's step() leaves the batch usable\n");
      }
      CHECK(ndone > 0);                                            // the big kicks made some robots fall and reset
      std::printf("VectorizedEnvironment<ENVIRONMENT>: %d envs x %d control steps = %ld launches, %d resets, equal to the device-resident env\n",
                  NE, STEPS, venv.batch()->viewLaunches() - l0, ndone);
    }

    // ---- a Perlin-noise terrain through the TerrainProperties overload: the robot lands ON it
    raisim::World hills;
    hills.setTimeStep(0.0025);
    auto* robot = hills.addArticulatedSystem(urdf);
    raisim::TerrainProperties tp;
    tp.frequency = 0.3; tp.zScale = 0.6; tp.xSize = 8.0; tp.ySize = 8.0; tp.xSamples = 81; tp.ySamples = 81; tp.seed = 11;
    auto* hm = hills.addHeightMap(0.0, 0.0, tp);
    CHECK(hm->getHeightVector().size() == 81u * 81u);
    const double ground = hm->getHeight(0.0, 0.0);
    raisim::VecDyn gch(gcDim), gvh(gvDim);
    gch[2] = ground + 0.62; gch[3] = 1.0;
    for (int j = 0; j < 12; ++j) gch[7 + j] = nominal[j];
    robot->setState(gch, gvh);
    robot->setPdGains(kp, kd);
    pT = gch.v;
    robot->setPdTarget(pT, dT);
    for (int i = 0; i < 600; ++i) hills.integrate();
    const auto& qh = robot->getGeneralizedCoordinate();
    const double under = hm->getHeight(qh[0], qh[1]);
    CHECK(robot->getContacts().size() >= 2);
    CHECK(qh[2] - under > 0.25 && qh[2] - under < 0.75);
    std::printf("Perlin terrain: ground %.3f under the robot, base %.3f above it, %zu contacts\n", under, qh[2] - under, robot->getContacts().size());

    // ---- material pairs: World::setMaterialPairProp(material of the primitive, material of the ground) resolves to one
    //      friction coefficient per collision primitive; a sliding ball decelerates at the PAIR's mu * g
    {
      const char* ball = "<robot name=\"ball\"><link name=\"ball\"><inertial><origin xyz=\"0 0 0\"/><mass value=\"2\"/>"
                         "<inertia ixx=\"0.008\" ixy=\"0\" ixz=\"0\" iyy=\"0.008\" iyz=\"0\" izz=\"0.008\"/></inertial>"
                         "<collision><origin xyz=\"0 0 0\"/><geometry><sphere radius=\"0.1\"/></geometry><material name=\"rubber\"/></collision>"
                         "</link></robot>";
      const std::string path = "/tmp/rsb_facade_ball.urdf";
      { FILE* f = std::fopen(path.c_str(), "w"); CHECK(f != nullptr); std::fputs(ball, f); std::fclose(f); }
      for (int variant = 0; variant < 3; ++variant) {
        raisim::World w;
        w.setTimeStep(0.0025);
        auto* b = w.addArticulatedSystem(path);
        w.addGround(0.0, variant == 2 ? "ice" : "concrete");
        w.setMaterialPairProp("concrete", "rubber", 1.1, 0.0, 0.0);     // order-free
        w.setMaterialPairProp("rubber", "ice", 0.05, 0.0, 0.0);
        if (variant == 1) w.setMaterialPairProp("rubber", "concrete", 0.4, 0.0, 0.0);   // overwrites the same pair
        const double mu = variant == 0 ? 1.1 : (variant == 1 ? 0.4 : 0.05);
        raisim::VecDyn q(7), u(6);
        q[2] = 0.1 - 1e-5; q[3] = 1.0; u[0] = 3.0;
        b->setState(q, u);
        double prev = 3.0;
        for (int i = 0; i < 5; ++i) {
          w.integrate();
          const double v = b->getGeneralizedVelocity()[0];
          CHECK(std::fabs((v - prev) + mu * 9.81 * 0.0025) < 1e-5);
          prev = v;
        }
      }
      std::printf("material pairs: rubber on concrete / ice slide at their own mu\n");
    }

    // ---- self-collision: a three-link chain folds its hand onto its own torso; the contact is listed once per body
    //      (isSelfCollision, object A first), ignoreCollisionBetween removes it
    {
      const char* folder = "<robot name=\"folder\">"
        "<link name=\"torso\"><inertial><origin xyz=\"0 0 0\"/><mass value=\"5\"/><inertia ixx=\"0.05\" ixy=\"0\" ixz=\"0\" iyy=\"0.05\" iyz=\"0\" izz=\"0.05\"/></inertial>"
        "<collision><origin xyz=\"0 0 0\"/><geometry><sphere radius=\"0.1\"/></geometry></collision></link>"
        "<link name=\"upper\"><inertial><origin xyz=\"0 0 -0.15\"/><mass value=\"1\"/><inertia ixx=\"0.01\" ixy=\"0\" ixz=\"0\" iyy=\"0.01\" iyz=\"0\" izz=\"0.002\"/></inertial></link>"
        "<link name=\"lower\"><inertial><origin xyz=\"0 0 -0.15\"/><mass value=\"1\"/><inertia ixx=\"0.01\" ixy=\"0\" ixz=\"0\" iyy=\"0.01\" iyz=\"0\" izz=\"0.002\"/></inertial>"
        "<collision><origin xyz=\"0 0 -0.3\"/><geometry><sphere radius=\"0.05\"/></geometry></collision></link>"
        "<joint name=\"shoulder\" type=\"revolute\"><origin xyz=\"0.15 0 0\"/><parent link=\"torso\"/><child link=\"upper\"/><axis xyz=\"0 1 0\"/><limit effort=\"100\" velocity=\"100\" lower=\"-10\" upper=\"10\"/></joint>"
        "<joint name=\"elbow\" type=\"revolute\"><origin xyz=\"0 0 -0.3\"/><parent link=\"upper\"/><child link=\"lower\"/><axis xyz=\"0 1 0\"/><limit effort=\"100\" velocity=\"100\" lower=\"-10\" upper=\"10\"/></joint>"
        "</robot>";
      const std::string path = "/tmp/rsb_facade_folder.urdf";
      { FILE* f = std::fopen(path.c_str(), "w"); CHECK(f != nullptr); std::fputs(folder, f); std::fclose(f); }
      for (int variant = 0; variant < 2; ++variant) {
        raisim::World w;
        w.setTimeStep(0.0025);
        auto* r = w.addArticulatedSystem(path);
        w.setGravity({0, 0, 0});
        w.addGround(-10.0);
        if (variant == 1) r->ignoreCollisionBetween(r->getBodyIdx("torso"), r->getBodyIdx("lower"));
        raisim::VecDyn q(9), u(8), kp(8), kd(8), pt(9), dtg(8);
        q[2] = 1.0; q[3] = 1.0; q[8] = 2.0;
        kp[6] = kp[7] = 40.0; kd[6] = kd[7] = 2.0;
        pt[7] = 0.3; pt[8] = 3.1;
        r->setState(q, u);
        r->setPdGains(kp, kd);
        r->setPdTarget(pt, dtg);
        int touched = 0;
        for (int i = 0; i < 300; ++i) {
          w.integrate();
          auto& cs = r->getContacts();
          if (cs.empty()) continue;
          ++touched;
          CHECK(cs.size() == 2 && cs[0].isSelfCollision() && cs[1].isSelfCollision() && cs[0].isObjectA() && !cs[1].isObjectA());
          CHECK(cs[0].getlocalBodyIndex() == 0 && cs[1].getlocalBodyIndex() == 2 && cs[0].getCollisionIndex() == 0 && cs[1].getCollisionIndex() == 1);
          for (int c = 0; c < 3; ++c) CHECK(std::fabs(cs[0].getImpulse()[c] + cs[1].getImpulse()[c]) < 1e-9 && std::fabs(cs[0].getNormal()[c] + cs[1].getNormal()[c]) < 1e-9);
        }
        CHECK(variant == 0 ? touched > 100 : touched == 0);
        const double elbow = r->getGeneralizedCoordinate()[8];
        CHECK(variant == 0 ? elbow < 3.0 : elbow > 3.0);     // the torso is in the way / the hand passes through it
      }
      std::printf("self-collision: listed per body, ignoreCollisionBetween removes it\n");
    }

    // ---- fixed-base system (URDF root link "world"): the facade exposes the joints only, as upstream does
    {
      const char* arm = "<robot name=\"arm\"><link name=\"world\"/>"
                        "<link name=\"bob\"><inertial><origin xyz=\"0 0 -0.5\"/><mass value=\"1\"/>"
                        "<inertia ixx=\"1e-9\" ixy=\"0\" ixz=\"0\" iyy=\"1e-9\" iyz=\"0\" izz=\"1e-9\"/></inertial></link>"
                        "<joint name=\"hinge\" type=\"revolute\"><origin xyz=\"0 0 2\"/><parent link=\"world\"/><child link=\"bob\"/><axis xyz=\"0 1 0\"/>"
                        "<limit effort=\"0\" velocity=\"100\" lower=\"-10\" upper=\"10\"/></joint></robot>";
      const std::string path = "/tmp/rsb_facade_arm.urdf";
      { FILE* f = std::fopen(path.c_str(), "w"); CHECK(f != nullptr); std::fputs(arm, f); std::fclose(f); }
      raisim::World w;
      w.setTimeStep(0.0025);
      auto* a = w.addArticulatedSystem(path);
      w.addGround(-5.0);
      CHECK(a->isFixedBase() && a->getGeneralizedCoordinateDim() == 1 && a->getDOF() == 1);
      a->setControlMode(raisim::ControlMode::FORCE_AND_TORQUE);
      raisim::VecDyn q(1), u(1);
      q[0] = 0.05;
      a->setState(q, u);
      w.integrate1();
      const auto& M = a->getMassMatrix();
      CHECK(M.rows() == 1 && std::fabs(M(0, 0) - 0.25) < 1e-4);                    // m l^2
      raisim::Vec<3> p; a->getFramePosition(1, p);
      CHECK(std::fabs(p[2] - 2.0) < 1e-9);                                          // the hinge sits where the fixed base put it
      double prev = 0.05, t = 0, t0 = -1, t1 = -1;
      for (int i = 0; i < 1300; ++i) {
        w.integrate();
        t += 0.0025;
        const double qi = a->getGeneralizedCoordinate()[0];
        if (prev > 0 && qi <= 0) { const double tc = t - 0.0025 * qi / (qi - prev); if (t0 < 0) t0 = tc; else if (t1 < 0) t1 = tc; }
        prev = qi;
      }
      CHECK(t1 > t0 && std::fabs((t1 - t0) / (2 * M_PI * std::sqrt(0.5 / 9.81)) - 1.0) < 1e-2);
      std::printf("fixed base: 1-DoF pendulum period %.4f s\n", t1 - t0);
    }
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 1;
  }
  std::printf("facade_test OK\n");
  return 0;
}
