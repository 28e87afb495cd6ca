// CPU-tier test of the drop-in boundary's host side against the C-ABI TEST DOUBLE (tests/cpp/rsb_host_double.cpp: toy update rule,
// NOT the simulator): raisim::VectorizedEnvironment<ENVIRONMENT> over the unmodified rsg_anymal-style Environment.hpp, the per-env
// World views, their staging / fused flush / lazy downloads and the threaded fiber scheduler.  What it pins:
//   1. a control step of N envs (setPdTarget, 4 x integrate(), state / contact / generalized-force reads) is ONE fused launch of 4
//      sub-steps and ONE rsb_view_exchange - and bit-identical to RSB_VIEW_FUSE=0 (a flush per integrate(), round 3's behaviour);
//   2. the same on 1 and on T host threads; the C-ABI handle is never entered concurrently (the double counts re-entries);
//   3. envs that integrate a different number of times, write between integrate() calls or call integrate1() inside step() are
//      flushed correctly (masked launches by count, order of writes and sub-steps preserved);
//   4. ADVICE r03 (medium): envs staging rows while other envs of the same round force uploads (integrate1(), batched getState)
//      lose no staged write.
// usage: facade_host_test <resource dir> [bench N steps threads]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "anymal_env/Environment.hpp"
#include "raisim/VectorizedEnvironment.hpp"

extern "C" {
struct rsbd_counter_block { long launches, substeps, masked_launches, uploads, downloads, syncs, exchanges, reentries; };
rsbd_counter_block* rsbd_counters(void);
}

#define CHECK(c) do { if (!(c)) { std::printf("CHECK failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

using VecEnv = raisim::VectorizedEnvironment<raisim::ENVIRONMENT>;

static bool g_render = false;      // render: true -> env 0 is built `visualizable`: the upstream-style block creates its raisim::RaisimServer (a no-op here)
static std::string cfg(int n, int threads) {
  return "num_envs: " + std::to_string(n) + "\nnum_threads: " + std::to_string(threads) +
         "\nsimulation_dt: 0.0025\ncontrol_dt: 0.01\nrender: " + (g_render ? "true" : "false") + "\naction_std: 0.3\nreward:\n  forwardVel:\n    coeff: 0.3\n  torque:\n    coeff: -4e-5\n";
}

struct Run { std::vector<float> ob, rew; std::vector<char> done; long launches = 0, exchanges = 0, substeps = 0, reentries = 0; };

static Run run_env(const std::string& rsc, int n, int threads, bool fuse, int steps) {
  VecEnv env(rsc, cfg(n, threads), false);
  env.batch()->setFuseIntegrateCalls(fuse);
  std::vector<float> act((size_t)n * 12), ob((size_t)n * 34), rew(n);
  std::vector<char> done(n);
  env.reset();
  const rsbd_counter_block c0 = *rsbd_counters();
  Run r;
  for (int k = 0; k < steps; ++k) {
    for (int i = 0; i < n * 12; ++i) act[i] = 0.01f * (float)((i * 7 + k * 13) % 41 - 20);
    env.step(act.data(), n, 12, rew.data(), reinterpret_cast<bool*>(done.data()));
    env.observe(ob.data(), n, 34, false);
    r.ob.insert(r.ob.end(), ob.begin(), ob.end()); r.rew.insert(r.rew.end(), rew.begin(), rew.end()); r.done.insert(r.done.end(), done.begin(), done.end());
  }
  const rsbd_counter_block c1 = *rsbd_counters();
  r.launches = c1.launches - c0.launches; r.exchanges = c1.exchanges - c0.exchanges; r.substeps = c1.substeps - c0.substeps; r.reentries = c1.reentries - c0.reentries;
  return r;
}

int main(int argc, char** argv) {
  if (argc < 2) { std::printf("usage: facade_host_test <resource dir> [bench N steps threads]\n"); return 2; }
  const std::string rsc = argv[1];
  if (argc >= 6 && std::string(argv[2]) == "bench") {
    // host cost of a control step of the template path (the double's "launch" is a few memory passes: what is timed is the facade)
    const int n = std::atoi(argv[3]), steps = std::atoi(argv[4]), threads = std::atoi(argv[5]);
    for (int fuse = 1; fuse >= 0; --fuse) {
      VecEnv env(rsc, cfg(n, threads), false);
      env.batch()->setFuseIntegrateCalls(fuse != 0);
      std::vector<float> act((size_t)n * 12, 0.1f), ob((size_t)n * 34), rew(n);
      std::vector<char> done(n);
      env.reset();
      for (int k = 0; k < 5; ++k) { env.step(act.data(), n, 12, rew.data(), reinterpret_cast<bool*>(done.data())); env.observe(ob.data(), n, 34, false); }
      env.resetStepProfile();
      const auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < steps; ++k) { env.step(act.data(), n, 12, rew.data(), reinterpret_cast<bool*>(done.data())); env.observe(ob.data(), n, 34, false); }
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / steps;
      const auto& sp = env.stepProfile();
      std::printf("host cost of the template path (C-ABI double), N = %d, %d threads, fuse = %d: %.3f ms per control step (step(): rounds %.3f ms + flushes %.3f ms)\n", n, threads, fuse, ms,
                  (sp.total_ns - sp.flush_ns) / 1e6 / sp.steps, sp.flush_ns / 1e6 / sp.steps);
    }
    return 0;
  }
  const int N = 257, STEPS = 6;
  // 1 + 2: fused vs per-integrate flush, 1 thread vs 5 threads: identical results, 1 launch of 4 sub-steps per control step
  const Run a = run_env(rsc, N, 1, true, STEPS), b = run_env(rsc, N, 1, false, STEPS), c = run_env(rsc, N, 5, true, STEPS), d = run_env(rsc, N, 5, false, STEPS);
  CHECK(a.launches == STEPS && a.exchanges == STEPS && a.substeps == 4 * STEPS);
  CHECK(b.launches == 4 * STEPS && b.substeps == 4 * STEPS);
  CHECK(c.launches == STEPS && c.exchanges == STEPS);
  CHECK(a.reentries == 0 && b.reentries == 0 && c.reentries == 0 && d.reentries == 0);
  CHECK(a.ob == b.ob && a.rew == b.rew && a.done == b.done);
  CHECK(a.ob == c.ob && a.rew == c.rew && a.ob == d.ob && a.rew == d.rew);
  bool moved = false;
  for (size_t i = 0; i < (size_t)N * 34; ++i) moved |= a.ob[i] != a.ob[(size_t)(STEPS - 1) * N * 34 + i];
  CHECK(moved);

  // 3: bodies of different shapes on one batch, through the scheduler directly
  {
    const int n = 64;
    raisim::BatchedWorld bw(rsc + "/anymal_c_like.urdf", n);
    bw.setTimeStep(0.0025);
    std::vector<float> kp(18, 0.f), kd(18, 0.f);
    for (int j = 6; j < 18; ++j) { kp[j] = 50.f; kd[j] = 0.2f; }
    bw.setPdGains(kp.data(), kd.data());
    std::vector<std::unique_ptr<raisim::World>> worlds;
    std::vector<std::unique_ptr<raisim::ArticulatedSystem>> robots;
    for (int e = 0; e < n; ++e) { worlds.emplace_back(new raisim::World(bw, e)); robots.emplace_back(new raisim::ArticulatedSystem(&bw, e)); }
    // reference: the same programme env by env on a batch of its own, flushed per integrate()
    auto programme = [&](raisim::BatchedWorld& w, std::vector<std::unique_ptr<raisim::World>>& ws, std::vector<std::unique_ptr<raisim::ArticulatedSystem>>& rs, bool fuse, int threads,
                         std::vector<std::vector<double>>& out) {
      w.setFuseIntegrateCalls(fuse);
      raisim::detail::FiberScheduler fs;
      out.assign(n, {});
      auto body = [&](int e) {
        raisim::VecDyn pt(19), dt(18), gc(19), gv(18);
        pt[3] = 1.0;
        for (int j = 0; j < 12; ++j) pt[7 + j] = 0.1 * (e % 5) + 0.01 * j;
        rs[e]->setPdTarget(pt, dt);
        const int k = 2 + e % 3;                                  // 2, 3 or 4 sub-steps
        for (int s = 0; s < k; ++s) {
          ws[e]->integrate();
          if (e % 7 == 3 && s == 0) { pt[7] += 0.5; rs[e]->setPTarget(pt); }     // a write BETWEEN two integrate() calls
          if (e % 11 == 5 && s == 1) { ws[e]->integrate1(); out[e].push_back(rs[e]->getMassMatrix()(0, 0)); }   // a query in the middle
        }
        rs[e]->getState(gc, gv);
        for (int j = 0; j < 19; ++j) out[e].push_back(gc[j]);
        for (int j = 0; j < 18; ++j) out[e].push_back(gv[j]);
        out[e].push_back((double)rs[e]->getContacts().size());
        out[e].push_back(ws[e]->getWorldTime());
      };
      w.setFiberBatch(true);
      fs.run(n, body, [&] { w.flushViews(); }, threads);
      w.flushViews();
      w.setFiberBatch(false);
    };
    std::vector<std::vector<double>> fused, plain, fusedT;
    const long l0 = bw.viewLaunches();
    programme(bw, worlds, robots, true, 1, fused);
    const long lf = bw.viewLaunches() - l0;
    raisim::BatchedWorld bw2(rsc + "/anymal_c_like.urdf", n);
    bw2.setTimeStep(0.0025); bw2.setPdGains(kp.data(), kd.data());
    std::vector<std::unique_ptr<raisim::World>> w2; std::vector<std::unique_ptr<raisim::ArticulatedSystem>> r2;
    for (int e = 0; e < n; ++e) { w2.emplace_back(new raisim::World(bw2, e)); r2.emplace_back(new raisim::ArticulatedSystem(&bw2, e)); }
    programme(bw2, w2, r2, false, 1, plain);
    raisim::BatchedWorld bw3(rsc + "/anymal_c_like.urdf", n);
    bw3.setTimeStep(0.0025); bw3.setPdGains(kp.data(), kd.data());
    std::vector<std::unique_ptr<raisim::World>> w3; std::vector<std::unique_ptr<raisim::ArticulatedSystem>> r3;
    for (int e = 0; e < n; ++e) { w3.emplace_back(new raisim::World(bw3, e)); r3.emplace_back(new raisim::ArticulatedSystem(&bw3, e)); }
    programme(bw3, w3, r3, true, 4, fusedT);
    for (int e = 0; e < n; ++e) {
      CHECK(fused[e].size() == plain[e].size() && fused[e].size() == fusedT[e].size());
      // the clock is the batch's: compare everything but the last entry (world time) exactly
      for (size_t i = 0; i + 1 < fused[e].size(); ++i) { CHECK(fused[e][i] == plain[e][i]); CHECK(fused[e][i] == fusedT[e][i]); }
    }
    CHECK(lf < 4 * 3);      // fused: a handful of masked launches, not one per integrate() and count class
    CHECK(bw.pendingViews() == 0 && rsbd_counters()->reentries == 0);
  }

  // 4: staged writes racing with uploads forced by other envs of the same round (World.hpp:160)
  {
    const int n = 512, rounds = 40, T = 8;
    raisim::BatchedWorld bw(rsc + "/anymal_c_like.urdf", n);
    bw.setTimeStep(0.0025);
    std::vector<std::unique_ptr<raisim::World>> ws; std::vector<std::unique_ptr<raisim::ArticulatedSystem>> rs;
    for (int e = 0; e < n; ++e) { ws.emplace_back(new raisim::World(bw, e)); rs.emplace_back(new raisim::ArticulatedSystem(&bw, e)); }
    raisim::detail::FiberScheduler fs;
    std::atomic<int> bad{0};
    for (int r = 0; r < rounds; ++r) {
      auto body = [&](int e) {
        raisim::VecDyn pt(19), dt(18), gc(19), gv(18);
        pt[3] = 1.0; gc[3] = 1.0;
        for (int j = 0; j < 12; ++j) { pt[7 + j] = e + 0.001 * r + j; gc[7 + j] = -e - 0.001 * r - j; }
        gc[2] = 0.5 + e; gv[0] = r;
        if (e % 3 == 0) ws[e]->integrate1();                       // forces an upload of whatever is staged right now
        rs[e]->setPdTarget(pt, dt);
        rs[e]->setState(gc, gv);
        if (e % 5 == 0) { std::vector<float> a((size_t)n * 19), b((size_t)n * 18); bw.getState(a.data(), b.data()); }   // a batched read: upload + download
        ws[e]->integrate();
        raisim::VecDyn q(19), u(18), p2(19);
        rs[e]->getState(q, u);
        // the toy rule moves a joint by dt * dt * 0 (gains are zero here): the state read back is the state staged
        for (int j = 0; j < 12; ++j) if (q[7 + j] != (double)(float)gc[7 + j]) ++bad;
        if (q[2] != (double)(float)(gc[2] + 0.0025 * 0)) ++bad;
      };
      bw.setFiberBatch(true);
      fs.run(n, body, [&] { bw.flushViews(); }, T);
      bw.flushViews();
      bw.setFiberBatch(false);
      // every env's staged PD target reached the "device"
      std::vector<float> pt((size_t)n * 19);
      CHECK(rsb_get_field(bw.handle(), RSB_F_PTARGET, pt.data(), RSB_HOST) == RSB_OK);
      for (int e = 0; e < n; ++e) for (int j = 0; j < 12; ++j) CHECK(pt[(size_t)e * 19 + 7 + j] == (float)(e + 0.001 * r + j));
    }
    CHECK(bad.load() == 0);
    CHECK(rsbd_counters()->reentries == 0);
  }
  {
    // 5. raisim::RaisimServer (include/raisim/RaisimServer.hpp, a no-op: visualisation is out of scope).  With `render: true` env 0 runs upstream's
    // `if (visualizable_) { server_ = std::make_unique<RaisimServer>(world_.get()); launchServer(); focusOn(robot); }` block and brackets every
    // integrate() with the server's lock / unlock - the results are those of the run without it, nothing listens, the mutex is a real one
    g_render = true;
    const Run v = run_env(rsc, N, 5, true, STEPS);
    g_render = false;
    CHECK(v.ob == a.ob && v.rew == a.rew && v.done == a.done && v.launches == STEPS);
    raisim::World w0;
    raisim::RaisimServer srv(&w0);
    CHECK(!srv.isLaunched() && !srv.isConnected());
    srv.launchServer();
    CHECK(srv.isLaunched() && srv.getPort() == 8080 && !srv.isConnected());
    srv.focusOn(&w0);
    CHECK(srv.focusedObject() == &w0);
    srv.lockVisualizationServerMutex(); srv.unlockVisualizationServerMutex();
    srv.hibernate(); CHECK(srv.isHibernating()); srv.wakeup(); CHECK(!srv.isHibernating());
    srv.startRecordingVideo("x.mp4"); CHECK(srv.isRecording()); srv.stopRecordingVideo(); CHECK(!srv.isRecording());
    srv.killServer(); CHECK(!srv.isLaunched());
  }
  std::printf("facade_host_test OK\n");
  return 0;
}
