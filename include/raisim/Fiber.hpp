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

#include <sys/mman.h>
#include <ucontext.h>
#include <unistd.h>

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

  /// stackBytes: usable stack of one fiber (RSB_FIBER_STACK_KB overrides the default at run time: a ChildEnvironment::step()
  /// with large locals or deep Eigen expression temporaries needs more).  Every stack is followed by a PROT_NONE guard page:
  /// an overflow faults instead of silently overwriting the neighbouring env's stack.
  explicit FiberScheduler(size_t stackBytes = 128 * 1024) {
    if (const char* kb = std::getenv("RSB_FIBER_STACK_KB")) { const long v = std::atol(kb); if (v >= 16) stackBytes = (size_t)v * 1024; }
    page_ = (size_t)sysconf(_SC_PAGESIZE);
    stackBytes_ = (stackBytes + page_ - 1) / page_ * page_;
  }
  ~FiberScheduler() { release(); }
  FiberScheduler(const FiberScheduler&) = delete;
  FiberScheduler& operator=(const FiberScheduler&) = delete;

  /// Runs body(i) for i in [0, n) as fibers.  Whenever all unfinished fibers are parked, onAllParked() is called
  /// (the batch flush) and they are resumed.  An exception thrown inside a fiber is re-thrown here.
  void run(int n, const std::function<void(int)>& body, const std::function<void()>& onAllParked) {
    if (current()) throw std::runtime_error("FiberScheduler::run: nested fiber schedulers are not supported");
    if (n > cap_) {
      release();
      // [guard | stack 0 | guard | stack 1 | ... ]: stacks grow downwards INTO the guard page below them.  Virtual memory;
      // pages are touched on demand
      const size_t slot = stackBytes_ + page_, total = (size_t)n * slot;
      void* m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (m == MAP_FAILED) throw std::bad_alloc();
      stacks_ = static_cast<char*>(m); mapped_ = total;
      for (int i = 0; i < n; ++i) mprotect(stacks_ + (size_t)i * slot, page_, PROT_NONE);
      cap_ = n;
      ctx_.resize(n);
    }
    body_ = &body;
    state_.assign(n, kReady);
    error_ = nullptr;
    for (int i = 0; i < n; ++i) {
      getcontext(&ctx_[i]);
      ctx_[i].uc_stack.ss_sp = stacks_ + (size_t)i * (stackBytes_ + page_) + page_;
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
  void release() { if (stacks_) munmap(stacks_, mapped_); stacks_ = nullptr; mapped_ = 0; cap_ = 0; }
  size_t stackBytes_ = 0, page_ = 4096, mapped_ = 0;
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
