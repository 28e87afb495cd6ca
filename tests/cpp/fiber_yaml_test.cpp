// CPU-only checks of the host-side machinery behind VectorizedEnvironment<ENV>: the fiber scheduler (raisim/Fiber.hpp)
// and the cfg.yaml subset parser (raisim/Yaml.hpp).  No GPU, no librsb.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "raisim/Fiber.hpp"
#include "raisim/Yaml.hpp"

#define CHECK(c) do { if (!(c)) { std::printf("CHECK failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main() {
  using raisim::detail::FiberScheduler;
  {  // N bodies that "integrate" a different number of times: a flush happens exactly when every live fiber is parked
    const int N = 1000;
    FiberScheduler fs(64 * 1024);
    std::vector<int> parks(N, 0), done(N, 0);
    std::vector<int> order;
    int flushes = 0, parked_now = 0;
    std::vector<int> parked_at_flush;
    auto body = [&](int i) {
      char pad[2048]; pad[0] = (char)i;                 // some stack use per fiber
      const int k = 4 + (i % 3 == 0 ? 1 : 0);          // every third env takes one more sub-step
      for (int s = 0; s < k; ++s) { ++parks[i]; ++parked_now; FiberScheduler::current()->park(); }
      done[i] = 1 + (pad[0] == (char)i ? 0 : 100);
    };
    fs.run(N, body, [&] { ++flushes; parked_at_flush.push_back(parked_now); parked_now = 0; });
    CHECK(flushes == 5);
    CHECK(parked_at_flush[0] == N && parked_at_flush[3] == N && parked_at_flush[4] == (N + 2) / 3);
    for (int i = 0; i < N; ++i) CHECK(done[i] == 1 && parks[i] == 4 + (i % 3 == 0 ? 1 : 0));
    CHECK(FiberScheduler::current() == nullptr);
    // the scheduler is reusable, and bodies that never park need no flush
    int f2 = 0, ran = 0;
    fs.run(10, [&](int) { ++ran; }, [&] { ++f2; });
    CHECK(ran == 10 && f2 == 0);
    // an exception inside a fiber surfaces in run()
    bool threw = false;
    try { fs.run(4, [&](int i) { if (i == 2) throw std::runtime_error("boom"); FiberScheduler::current()->park(); }, [] {}); }
    catch (const std::runtime_error& e) { threw = std::string(e.what()) == "boom"; }
    CHECK(threw && FiberScheduler::current() == nullptr);
  }
  {  // the same on a pool of threads (cfg["num_threads"]): the bodies of a round run concurrently, the rounds stay lock-step,
     // every fiber stays on the thread it started on
    const int N = 1003, T = 7;
    FiberScheduler fs(64 * 1024);
    std::vector<int> parks(N, 0), done(N, 0);
    std::vector<std::thread::id> first(N), last(N);
    std::atomic<int> parked_now{0};
    int flushes = 0;
    std::vector<int> parked_at_flush;
    const auto caller = std::this_thread::get_id();
    bool flush_on_caller = true;
    auto body = [&](int i) {
      first[i] = std::this_thread::get_id();
      const int k = 4 + (i % 3 == 0 ? 1 : 0);
      for (int s = 0; s < k; ++s) { ++parks[i]; ++parked_now; FiberScheduler::current()->park(); }
      last[i] = std::this_thread::get_id();
      done[i] = 1;
    };
    fs.run(N, body, [&] { ++flushes; parked_at_flush.push_back(parked_now.exchange(0)); flush_on_caller = flush_on_caller && std::this_thread::get_id() == caller; }, T);
    CHECK(flushes == 5 && flush_on_caller);
    CHECK(parked_at_flush[0] == N && parked_at_flush[3] == N && parked_at_flush[4] == (N + 2) / 3);
    std::vector<std::thread::id> ids;
    for (int i = 0; i < N; ++i) {
      CHECK(done[i] == 1 && parks[i] == 4 + (i % 3 == 0 ? 1 : 0) && first[i] == last[i]);
      if (std::find(ids.begin(), ids.end(), first[i]) == ids.end()) ids.push_back(first[i]);
    }
    CHECK((int)ids.size() == T && FiberScheduler::current() == nullptr);
    bool threw = false;     // an exception on a worker thread surfaces on the caller; the pool is joined
    try { fs.run(64, [&](int i) { if (i == 50) throw std::runtime_error("boom"); FiberScheduler::current()->park(); }, [] {}, 4); }
    catch (const std::runtime_error& e) { threw = std::string(e.what()) == "boom"; }
    CHECK(threw && FiberScheduler::current() == nullptr);
    // forEach: the same pool without fibers (observe()); every index once, exceptions surface, run() still works afterwards
    std::vector<std::atomic<int>> hits(1000);
    fs.forEach(1000, [&](int i) { ++hits[i]; }, 6);
    for (int i = 0; i < 1000; ++i) CHECK(hits[i] == 1);
    threw = false;
    try { fs.forEach(100, [&](int i) { if (i == 77) throw std::runtime_error("each"); }, 3); } catch (const std::runtime_error& e) { threw = std::string(e.what()) == "each"; }
    CHECK(threw);
    int ran = 0, fl = 0;
    fs.run(8, [&](int) { FiberScheduler::current()->park(); }, [&] { ++fl; }, 2);
    fs.forEach(5, [&](int) { ++ran; }, 1);
    CHECK(fl == 1 && ran == 5);
  }
  {  // the cfg.yaml subset
    const std::string text =
        "seed: 1\nrecord_video: yes\n\nenvironment:\n  render: True   # comment\n  num_envs: 100\n  eval_every_n: 200\n"
        "  simulation_dt: 0.0025\n  control_dt: 0.01\n  max_time: 4.0\n  action_std: 0.3\n  name: \"rsg # anymal\"\n  reward:\n    forwardVel:\n      coeff: 0.3\n"
        "    torque:\n      coeff: -4e-5\n\narchitecture:\n  policy_net: [128, 128]\n";
    Yaml::Node root;
    Yaml::Parse(root, text);
    const Yaml::Node& env = root["environment"];
    CHECK(root["seed"].As<int>() == 1 && root["record_video"].As<bool>());
    CHECK(env["num_envs"].As<int>() == 100 && env["render"].As<bool>());
    CHECK(env["simulation_dt"].As<double>() == 0.0025 && env["name"].As<std::string>() == "rsg # anymal");
    CHECK(env["reward"]["torque"]["coeff"].As<double>() == -4e-5);
    CHECK(env["reward"].Keys().size() == 2 && env["reward"].Keys()[0] == "forwardVel");
    CHECK(env["missing"].IsNone() && env["missing"].As<int>(7) == 7);
    CHECK(root["architecture"]["policy_net"].As<std::string>() == "[128, 128]");
    bool threw = false;
    try { env["reward"].As<int>(); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);
  }
  std::printf("fiber_yaml_test OK\n");
  return 0;
}
