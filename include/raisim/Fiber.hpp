// raisim/Fiber.hpp — cooperative fibers that let N unmodified per-env Environment::step() bodies run in lock-step.
//
// Upstream's VectorizedEnvironment<ENV>::step fans `environments_[i]->step(action.row(i))` out with an OpenMP
// parallel-for: every env sets its PD target, calls world_->integrate() control_dt/simulation_dt times and then reads
// its state [RECALL raisimGymTorch/env/VectorizedEnvironment.hpp, absent from /root/reference].  With ONE batched world
// behind all envs, the i-th integrate() of env 0 must not run before every other env has issued its own i-th
// integrate().  So each env's step() body runs on its own fiber (ucontext); raisim::World::integrate() on a view parks
// the fiber, and when every live fiber is parked the scheduler flushes the batch with ONE launch and resumes them all.
// No threads, no locks: fibers run one at a time on the caller's thread, exactly like the serial loop they replace.
#pragma once

#include <ucontext.h>

#include <cstddef>
#include <cstdlib>
#include <exception>
#include <functional>
#include <stdexcept>
#include <vector>

namespace raisim {
namespace detail {

class FiberScheduler {
 public:
  /// the scheduler whose fiber is running on this thread right now (nullptr outside of run())
  static FiberScheduler*& current() { static thread_local FiberScheduler* c = nullptr; return c; }

  explicit FiberScheduler(size_t stackBytes = 128 * 1024) : stackBytes_(stackBytes) {}
  ~FiberScheduler() { std::free(stacks_); }
  FiberScheduler(const FiberScheduler&) = delete;
  FiberScheduler& operator=(const FiberScheduler&) = delete;

  /// Runs body(i) for i in [0, n) as fibers.  Whenever all unfinished fibers are parked, onAllParked() is called
  /// (the batch flush) and they are resumed.  An exception thrown inside a fiber is re-thrown here.
  void run(int n, const std::function<void(int)>& body, const std::function<void()>& onAllParked) {
    if (current()) throw std::runtime_error("FiberScheduler::run: nested fiber schedulers are not supported");
    if (n > cap_) {
      std::free(stacks_);
      stacks_ = static_cast<char*>(std::malloc((size_t)n * stackBytes_));   // virtual; pages are touched on demand
      if (!stacks_) throw std::bad_alloc();
      cap_ = n;
      ctx_.resize(n);
    }
    body_ = &body;
    state_.assign(n, kReady);
    error_ = nullptr;
    for (int i = 0; i < n; ++i) {
      getcontext(&ctx_[i]);
      ctx_[i].uc_stack.ss_sp = stacks_ + (size_t)i * stackBytes_;
      ctx_[i].uc_stack.ss_size = stackBytes_;
      ctx_[i].uc_link = &main_;
      makecontext(&ctx_[i], reinterpret_cast<void (*)()>(&FiberScheduler::trampoline), 0);
    }
    current() = this;
    int live = n;
    while (live > 0 && !error_) {
      int parked = 0;
      for (int i = 0; i < n && !error_; ++i) {
        if (state_[i] == kDone) continue;
        running_ = i;
        state_[i] = kRunning;
        swapcontext(&main_, &ctx_[i]);
        if (state_[i] == kRunning) { state_[i] = kDone; --live; }   // returned through uc_link: the body finished
        else ++parked;
      }
      if (parked > 0 && !error_) {
        try { onAllParked(); } catch (...) { error_ = std::current_exception(); }
      }
    }
    current() = nullptr;
    running_ = -1;
    if (error_) { auto e = error_; error_ = nullptr; std::rethrow_exception(e); }   // fibers left parked are simply dropped
  }

  /// called from inside a fiber: give control back to the scheduler until the next flush
  void park() {
    const int i = running_;
    state_[i] = kParked;
    swapcontext(&ctx_[i], &main_);
  }
  int running() const { return running_; }

 private:
  enum State : char { kReady, kRunning, kParked, kDone };
  static void trampoline() {
    FiberScheduler* s = current();
    try { (*s->body_)(s->running_); } catch (...) { s->error_ = std::current_exception(); }
    // falling off the end switches to uc_link (= main_) with state_ still kRunning, which run() reads as "finished"
  }
  size_t stackBytes_;
  char* stacks_ = nullptr;
  int cap_ = 0;
  std::vector<ucontext_t> ctx_;
  ucontext_t main_;
  std::vector<char> state_;
  const std::function<void(int)>* body_ = nullptr;
  std::exception_ptr error_;
  int running_ = -1;
};

}  // namespace detail
}  // namespace raisim
