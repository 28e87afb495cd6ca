// raisim/Fiber.hpp — cooperative fibers that let N unmodified per-env Environment::step() bodies run in lock-step.
//
// Upstream's VectorizedEnvironment<ENV>::step fans `environments_[i]->step(action.row(i))` out with an OpenMP
// parallel-for: every env sets its PD target, calls world_->integrate() control_dt/simulation_dt times and then reads
// its state [RECALL raisimGymTorch/env/VectorizedEnvironment.hpp, absent from /root/reference].  With ONE batched world
// behind all envs, the i-th integrate() of env 0 must not run before every other env has issued its own i-th
// integrate().  So each env's step() body runs on its own fiber (ucontext); raisim::World::integrate() on a view parks
// the fiber, and when every live fiber is parked the scheduler flushes the batch with ONE launch and resumes them all.
// By default the fibers run one at a time on the caller's thread, exactly like the serial loop they replace; with threads > 1
// they are dealt to a pool (cfg["num_threads"], what upstream's OpenMP loop uses) and the host part of the N step() bodies -
// observation, reward, contact queries - runs in parallel between two flushes.
//
// Context switch.  glibc's swapcontext saves and restores the signal mask: one rt_sigprocmask system call per switch, i.e.
// ~10 N system calls per control step of N envs (40 000 at N = 4096: round 3's drop-in path spent more time there than on the
// GPU).  On x86-64 the switch is therefore hand-written (detail::fiber_switch below: the callee-saved registers of the SysV ABI
// + the stack pointer, no system call, ~20 instructions); other architectures keep ucontext.
#pragma once

#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>
#if !defined(__x86_64__) || defined(__SHSTK__)
#include <ucontext.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <stdexcept>
#include <vector>

namespace raisim {
namespace detail {

// The hand-written switch changes stacks with a jmp and later returns on the stack it was parked on: a CET shadow stack (-fcf-protection=return /
// full with user-mode shadow stacks enabled) would fault on that return.  Where the compiler says shadow stacks are on (__SHSTK__) the portable
// ucontext path below is used instead (it saves the shadow-stack pointer with the context).
#if defined(__x86_64__) && !defined(__SHSTK__)
/// A parked context is its stack pointer; the stack holds (from the pointer upwards) rbx, rbp and the address to continue at.
struct FiberContext { void* sp = nullptr; };
/// Saves the running context into `from`, continues `to`.  Every register the SysV ABI lets a callee keep is either saved on the
/// stack (rbx, rbp: not allowed in a clobber list when they are the PIC / frame register) or declared clobbered, so the compiler
/// spills what is live around the call; 128 bytes are skipped first because the enclosing function may keep data in the red zone.
/// Not switched: the x87 control word and MXCSR (env bodies that change rounding or exception masks change them for their thread, as a
/// plain function call would).
__attribute__((noinline)) inline void fiber_switch(FiberContext* from, const FiberContext* to) {
  void** save = &from->sp;
  void* next = to->sp;
  asm volatile(
      "leaq -128(%%rsp), %%rsp\n\t"
      "leaq 1f(%%rip), %%rax\n\t"
      "pushq %%rax\n\t"
      "pushq %%rbp\n\t"
      "pushq %%rbx\n\t"
      "movq %%rsp, (%0)\n\t"
      "movq %1, %%rsp\n\t"
      "popq %%rbx\n\t"
      "popq %%rbp\n\t"
      "popq %%rax\n\t"
      "jmpq *%%rax\n\t"
      "1:\n\t"
      "endbr64\n\t"
      "leaq 128(%%rsp), %%rsp\n\t"
      : "+D"(save), "+S"(next)
      :
      : "rax", "rcx", "rdx", "r8", "r9", "r10", "r11", "r12", "r13", "r14", "r15", "memory", "cc",
        "xmm0", "xmm1", "xmm2", "xmm3", "xmm4", "xmm5", "xmm6", "xmm7", "xmm8", "xmm9", "xmm10", "xmm11", "xmm12", "xmm13", "xmm14", "xmm15");
}
/// A fresh context that starts `entry` (which must never return) on the stack [stack, stack + bytes)
inline void fiber_make(FiberContext* c, void* stack, size_t bytes, void (*entry)()) {
  // at `entry`'s first instruction rsp + 8 must be 16-byte aligned, as after a call: [top - 8] is the (null) return address
  uintptr_t top = (reinterpret_cast<uintptr_t>(stack) + bytes) & ~static_cast<uintptr_t>(15);
  void** sp = reinterpret_cast<void**>(top);
  *--sp = nullptr;                                 // return address of `entry`: never used
  *--sp = reinterpret_cast<void*>(entry);          // continue here
  *--sp = nullptr;                                 // rbp
  *--sp = nullptr;                                 // rbx
  c->sp = sp;
}
#else
struct FiberContext { ucontext_t uc; };
inline void fiber_switch(FiberContext* from, const FiberContext* to) { swapcontext(&from->uc, &to->uc); }
inline void fiber_make(FiberContext* c, void* stack, size_t bytes, void (*entry)()) {
  getcontext(&c->uc);
  c->uc.uc_stack.ss_sp = stack; c->uc.uc_stack.ss_size = bytes; c->uc.uc_link = nullptr;
  makecontext(&c->uc, entry, 0);
}
#endif

class FiberScheduler {
 public:
  /// the scheduler whose fiber is running on this thread right now (nullptr outside of run())
  static FiberScheduler*& current() { return tls().sched; }

  /// stackBytes: usable stack of one fiber (RSB_FIBER_STACK_KB overrides the default at run time: a ChildEnvironment::step()
  /// with large locals or deep Eigen expression temporaries needs more).  Every stack is followed by a PROT_NONE guard page:
  /// an overflow faults instead of silently overwriting the neighbouring env's stack.
  explicit FiberScheduler(size_t stackBytes = 128 * 1024) {
    if (const char* kb = std::getenv("RSB_FIBER_STACK_KB")) { const long v = std::atol(kb); if (v >= 16) stackBytes = (size_t)v * 1024; }
    page_ = (size_t)sysconf(_SC_PAGESIZE);
    stackBytes_ = (stackBytes + page_ - 1) / page_ * page_;
  }
  ~FiberScheduler() { stopPool(); release(); }
  FiberScheduler(const FiberScheduler&) = delete;
  FiberScheduler& operator=(const FiberScheduler&) = delete;

  /// Runs body(i) for i in [0, n) as fibers.  Whenever all unfinished fibers are parked, onAllParked() is called
  /// (the batch flush, on the calling thread) and they are resumed.  An exception thrown inside a fiber is re-thrown here.
  /// threads > 1: the fibers are dealt to that many threads in contiguous blocks (upstream fans the N step() bodies out with
  /// an OpenMP parallel-for over cfg["num_threads"]); a fiber always runs on the thread it was dealt to, the bodies of one round
  /// run concurrently - whatever they share must be thread-safe (raisim::BatchedWorld is) - and the rounds stay lock-step.
  void run(int n, const std::function<void(int)>& body, const std::function<void()>& onAllParked, int threads = 1) {
    if (current()) throw std::runtime_error("FiberScheduler::run: nested fiber schedulers are not supported");
    threads = std::max(1, std::min(threads, n));
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
    failedFlag_.store(false);
    mains_.assign(threads, FiberContext());
    auto owner = [&](int i) { return (int)((long long)i * threads / n); };
    // one round of thread t: resume each of its unfinished fibers once; returns (still live, parked now)
    auto round = [&](int t, int& live, int& parked) {
      Tls& me = tls();
      me.sched = this; me.thread = t;
      const int lo = (int)(((long long)t * n + threads - 1) / threads), hi = (int)(((long long)(t + 1) * n + threads - 1) / threads);
      live = parked = 0;
      for (int i = lo; i < hi && !failed(); ++i) {
        if (owner(i) != t || state_[i] == kDone) continue;
        me.running = i;
        if (state_[i] == kReady)      // a fresh context, made by the thread that runs it right before its (cold) stack is touched anyway
          fiber_make(&ctx_[i], stacks_ + (size_t)i * (stackBytes_ + page_) + page_, stackBytes_, &FiberScheduler::trampoline);
        state_[i] = kRunning;
        fiber_switch(&mains_[t], &ctx_[i]);
        if (state_[i] != kDone) { ++parked; ++live; }
      }
      me.sched = nullptr; me.running = -1;
    };
    if (threads == 1) {
      int live = n, parked = 0;
      while (live > 0 && !failed()) {
        round(0, live, parked);
        if (parked > 0 && !failed()) { try { onAllParked(); } catch (...) { fail(std::current_exception()); } }
      }
    } else {
      // workers 1 .. threads-1 (a pool kept across run() calls: a control step is one run()) wait for a round number, run their
      // block, report; the caller is worker 0 and runs the flush
      ensurePool(threads);
      CallerPin pin(cpus_);
      roundFn_ = [&](int t, int& live, int& parked) { round(t, live, parked); };
      for (;;) {
        postRound();
        int live, parked;
        round(0, live, parked);
        liveSum_ += live; parkedSum_ += parked;
        awaitWorkers(threads);
        if (failed() || liveSum_ == 0) break;
        if (parkedSum_ > 0) { try { onAllParked(); } catch (...) { fail(std::current_exception()); break; } }
      }
      roundFn_ = nullptr;
    }
    if (error_) { auto e = error_; error_ = nullptr; std::rethrow_exception(e); }   // fibers left parked are simply dropped
  }

  /// fn(i) for i in [0, n) on the same pool of threads, without fibers: for the per-env loops that never touch the batch's launch
  /// (observe(): upstream runs it as an OpenMP parallel-for as well).  fn must not call park().
  void forEach(int n, const std::function<void(int)>& fn, int threads = 1) {
    if (current()) throw std::runtime_error("FiberScheduler::forEach: called from inside a fiber");
    threads = std::max(1, std::min(threads, n));
    if (threads == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
    error_ = nullptr; failedFlag_.store(false);
    auto block = [&](int t, int& live, int& parked) {
      live = parked = 0;
      const int lo = (int)((long long)t * n / threads), hi = (int)((long long)(t + 1) * n / threads);
      try { for (int i = lo; i < hi && !failed(); ++i) fn(i); } catch (...) { fail(std::current_exception()); }
    };
    ensurePool(threads);
    CallerPin pin(cpus_);
    roundFn_ = block;
    postRound();
    int live, parked;
    block(0, live, parked);
    awaitWorkers(threads);
    roundFn_ = nullptr;
    if (error_) { auto e = error_; error_ = nullptr; std::rethrow_exception(e); }
  }

  /// called from inside a fiber: give control back to the scheduler until the next flush
  void park() {
    Tls& me = tls();
    const int i = me.running;
    state_[i] = kParked;
    fiber_switch(&ctx_[i], &mains_[me.thread]);
  }
  int running() const { return tls().running; }

 private:
  enum State : char { kReady, kRunning, kParked, kDone };
  struct Tls { FiberScheduler* sched = nullptr; int running = -1, thread = 0; };
  static Tls& tls() { static thread_local Tls t; return t; }
  static void trampoline() {
    FiberScheduler* s = current();
    try { (*s->body_)(tls().running); } catch (...) { s->fail(std::current_exception()); }
    // the body is finished: back to the owner thread's main context for good (this context is never resumed)
    Tls& me = tls();
    s->state_[me.running] = kDone;
    fiber_switch(&s->ctx_[me.running], &s->mains_[me.thread]);
    std::abort();
  }
  bool failed() const { return failedFlag_.load(std::memory_order_acquire); }
  void fail(std::exception_ptr e) {
    std::lock_guard<std::mutex> lk(errMutex_);
    if (!error_) error_ = e;
    failedFlag_.store(true, std::memory_order_release);
  }
  void release() { if (stacks_) munmap(stacks_, mapped_); stacks_ = nullptr; mapped_ = 0; cap_ = 0; }
  void ensurePool(int threads) {
    if ((int)pool_.size() == threads - 1) return;
    stopPool();
    quit_.store(false);
    const int start = roundNo_.load();
    // Workers are pinned to distinct CPUs of the process's affinity mask (RSB_FIBER_PIN=0 switches it off) and spin for a while before
    // they sleep: a round lasts a few hundred microseconds, and threads woken through a futex all start on the waker's CPU and run one
    // after the other before the kernel's load balancer has moved them (measured: eight workers of 0.7 ms each started 0.8 ms apart,
    // a "parallel" round took as long as the serial one).  OpenMP runtimes wait actively between regions for the same reason.
    std::vector<int>& cpus = cpus_;
    cpus.clear();
    const char* pin = std::getenv("RSB_FIBER_PIN");
    if (!(pin && std::atoi(pin) == 0)) {
      cpu_set_t cs;
      if (pthread_getaffinity_np(pthread_self(), sizeof cs, &cs) == 0) for (int c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &cs)) cpus.push_back(c);
      if ((int)cpus.size() < threads) cpus.clear();      // fewer CPUs than threads: pinning would stack spinning threads on one core
    }
    for (int t = 1; t < threads; ++t) {
      pool_.emplace_back([this, t, start] {
        int seen = start;     // (the round number at creation, NOT at first run: a worker that starts late must not miss round one)
        for (;;) {
          if (!spinUntil([&] { return quit_.load(std::memory_order_acquire) || roundNo_.load(std::memory_order_acquire) != seen; })) {
            std::unique_lock<std::mutex> lk(poolMutex_);
            poolCv_.wait(lk, [&] { return quit_.load() || roundNo_.load() != seen; });
          }
          if (quit_.load()) return;
          seen = roundNo_.load();
          int live = 0, parked = 0;
          roundFn_(t, live, parked);
          liveSum_ += live; parkedSum_ += parked;
          // (store-then-load across two variables: the worker's increment must be ordered before its look at mainSleeps_, and the main thread's
          //  mainSleeps_ = true before its re-check of reported_ - sequential consistency on both sides, not acquire / release: ADVICE r04)
          reported_.fetch_add(1, std::memory_order_seq_cst);
          if (mainSleeps_.load(std::memory_order_seq_cst)) { std::lock_guard<std::mutex> lk(poolMutex_); poolCv_.notify_all(); }
        }
      });
      if (!cpus.empty()) {
        cpu_set_t one; CPU_ZERO(&one); CPU_SET(cpus[(size_t)t], &one);
        pthread_setaffinity_np(pool_.back().native_handle(), sizeof one, &one);
      }
    }
  }
  /// The caller is worker 0 of every round: while a step's rounds run it sits on the one CPU of the mask no pool thread is pinned to (it would
  /// otherwise share a core with a pinned worker whenever the scheduler had left it there - measured inside bench.py: half the rate of the
  /// same loop in a fresh process); its own mask is restored afterwards.
  struct CallerPin {
    cpu_set_t old; bool on = false;
    explicit CallerPin(const std::vector<int>& cpus) {
      if (cpus.empty() || pthread_getaffinity_np(pthread_self(), sizeof old, &old) != 0) return;
      cpu_set_t one; CPU_ZERO(&one); CPU_SET(cpus[0], &one);
      on = pthread_setaffinity_np(pthread_self(), sizeof one, &one) == 0;
    }
    ~CallerPin() { if (on) pthread_setaffinity_np(pthread_self(), sizeof old, &old); }
  };
  /// a new round for the workers (the caller is worker 0)
  void postRound() {
    reported_.store(1); liveSum_.store(0); parkedSum_.store(0);
    roundNo_.fetch_add(1, std::memory_order_acq_rel);
    { std::lock_guard<std::mutex> lk(poolMutex_); }      // a worker between its predicate check and its wait holds the mutex: it sees the new round
    poolCv_.notify_all();
  }
  void awaitWorkers(int threads) {
    if (spinUntil([&] { return reported_.load(std::memory_order_acquire) == threads; })) return;
    std::unique_lock<std::mutex> lk(poolMutex_);
    mainSleeps_.store(true, std::memory_order_seq_cst);
    poolCv_.wait(lk, [&] { return reported_.load(std::memory_order_seq_cst) == threads; });
    mainSleeps_.store(false);
  }
  /// polls `ready` for up to ~200 us (RSB_FIBER_SPIN_US); false = not yet, the caller goes to sleep on the condition variable
  template <class F>
  bool spinUntil(F ready) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0;; ++i) {
      if (ready()) return true;
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
      if ((i & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spinUs_)) return false;
    }
  }
  void stopPool() {
    quit_.store(true);
    { std::lock_guard<std::mutex> lk(poolMutex_); }
    poolCv_.notify_all();
    for (auto& th : pool_) th.join();
    pool_.clear();
  }
  size_t stackBytes_ = 0, page_ = 4096, mapped_ = 0;
  char* stacks_ = nullptr;
  int cap_ = 0;
  std::vector<FiberContext> ctx_, mains_;
  std::vector<char> state_;
  const std::function<void(int)>* body_ = nullptr;
  std::exception_ptr error_;
  std::mutex errMutex_;
  std::atomic<bool> failedFlag_{false};
  // worker pool of the threaded rounds
  std::vector<std::thread> pool_;
  std::vector<int> cpus_;       // CPUs of the creating thread's affinity mask: worker t is pinned to cpus_[t], the caller to cpus_[0] during a step
  std::mutex poolMutex_;
  std::condition_variable poolCv_;
  std::function<void(int, int&, int&)> roundFn_;
  std::atomic<int> roundNo_{0}, reported_{0}, liveSum_{0}, parkedSum_{0};
  std::atomic<bool> quit_{false}, mainSleeps_{false};
  long spinUs_ = std::getenv("RSB_FIBER_SPIN_US") ? std::atol(std::getenv("RSB_FIBER_SPIN_US")) : 200;
};

}  // namespace detail
}  // namespace raisim
