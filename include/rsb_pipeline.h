/*
 * rsb_pipeline.h — the CLOSED-LOOP pipeline: control steps that overlap on the device although a policy sits in the loop.
 *
 * rsb_set_step_pipelining (rsb.h) lets workgroup b of control step k + 1 start as soon as workgroup b of step k has published its env
 * block; the chip then runs at its mean wave instead of waiting for the slowest one of every launch.  Round 4 could use that only for OPEN-loop
 * callers (targets that do not depend on the last step).  What raisimGymTorch's VectorizedEnvironment::step / observe cycle
 * [RECALL raisimGymTorch/env/VectorizedEnvironment.hpp; absent from /root/reference] needs is a policy between two steps.  The hand-over
 * therefore becomes a THREE-stage one at the same granularity, one ENV BLOCK (the envs of one workgroup of the step kernel):
 *
 *     step k,   workgroup b : integrates block b, writes its observation / reward / done rows,      publishes step_prog[b] = seq(k)
 *     ACTION STAGE, block b : waits for step_prog[b] = seq(k), reads block b's observation rows,
 *                             writes block b's action rows,                                          publishes act_prog[b]  = seq(k)
 *     step k+1, workgroup b : waits for act_prog[b] = seq(k), reads block b's action rows, integrates ...
 *
 * Block b never waits for another block: a robot that has just fallen onto a knee (three times the solver sweeps) delays its own block
 * only.  The action stage is a kernel of the CALLER - this header is its device-side half - or one of the in-repo stages: a fixed
 * linear policy (rsb_closed_loop_run_linear) or an actor network (rsb_closed_loop_run_mlp).
 *
 * The action stage is ONE launch per run of K steps: `grid` workgroups of 64 threads that stay resident next to the step kernel's
 * waves (which leave 96 VGPRs per SIMD lane and no LDS: keep a stage under 96 VGPRs, no LDS, or it takes SIMDs from the steps),
 * poll the hand-over words of THEIR share of the blocks of their XCD, run the caller's body on a block whose step has finished and publish.
 * rsb_stage::serve() below is that loop; the caller writes
 *
 *     __global__ void my_stage(rsb_stage_ctx c, MyPolicy p) {
 *       rsb_stage::serve(c, [&](int block, int env0, int n_env, int pass, bool final) {
 *         // 64 lanes; observation rows c.ob[env0 .. env0 + n_env) are ready (after step `pass`, before step `pass + 1`);
 *         // unless `final`, write the action rows c.act[env0 .. env0 + n_env); c.reward / c.done of the step just finished (pass > 0)
 *       });
 *     }
 *     static int launch_my_stage(void* user, const rsb_stage_ctx* c) {
 *       hipLaunchKernelGGL(my_stage, dim3(c->grid), dim3(64), 0, (hipStream_t)c->stream, *c, *(MyPolicy*)user);
 *       return hipGetLastError() == hipSuccess ? 0 : -1;
 *     }
 *     rsb_closed_loop_run(world, K, launch_my_stage, &policy);
 *
 * Passes: pass t (t = 0 .. K) sees the observation after t steps of the run; passes 0 .. K-1 produce the action of step t + 1, pass K
 * is FINAL (nothing consumes an action: record the last reward / done / observation, or do nothing).
 * Memory model: within serve() the body's loads of c.ob / c.reward / c.done rows and its stores to c.act rows are ordered with the
 * steps by the claim's acquire and the publication's release.  Read rows written by the steps with VECTOR loads (lane-dependent
 * addresses): the scalar cache is not invalidated.
 *
 * With pipelining off (or refused: a dispatch-serialising profiler, RSB_STEP_PIPELINING=0, a pipeline fault) the library runs the same
 * run in LOCK-STEP on the world's stream - pass 0, step 1, pass 1, ... - and calls the launch function once per pass with
 * c->lockstep = 1: serve() then walks all blocks without waiting.  Results are bit-identical (tests/test_gpu_closed_loop.py).
 *
 * No device traps: a wait that runs past the time-out, a stage on an XCD the step kernel does not use, a bad launch geometry store a
 * code into the error word and leave; everything else drains, the next joining call (rsb_step_pipeline_join, any read, rsb_synchronize)
 * returns RSB_E_PIPELINE once after it has restored the state of the last join and replayed the steps in lock-step.
 */
#ifndef RSB_PIPELINE_H_
#define RSB_PIPELINE_H_

#include <stddef.h>
#include <stdint.h>

#include "rsb_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* codes of the pipeline's device error word (rsb_step_pipeline_fault reports the last one) */
#define RSB_PIPE_ERR_TICKET 1    /* a step workgroup drew an env-block ticket outside its XCD's range (the dispatcher did not deal round-robin) */
#define RSB_PIPE_ERR_TIMEOUT 2   /* a wait ran past the time-out (RSB_PIPE_TIMEOUT_MS, default 10 000 ms)                                     */
#define RSB_PIPE_ERR_STAGE 3     /* the action stage: workgroups of other than 64 threads, or a workgroup on an XCD outside the step's range  */
#define RSB_PIPE_ERR_INJECTED 4  /* rsb_debug_pipeline_fault                                                                                  */
#define RSB_PIPE_ERR_TIMEOUT_GATE 5   /* ... the time-out of a gate (the launch before it was not dispatched completely in time)              */
#define RSB_PIPE_ERR_TIMEOUT_STAGE 6  /* ... of an action-stage wave (no block of its XCD moved in time)                                      */

/* What an action-stage kernel receives (by value).  Filled by the library; the caller only reads it. */
typedef struct rsb_stage_ctx {
  /* hand-over words, device memory, one per env block */
  const int32_t* step_prog;   /* sequence number of the last step whose workgroup b has finished                    */
  int32_t* act_prog;          /* ... of the last pass served for block b                                             */
  uint32_t* ticket;           /* arrival counters of the stage's waves per XCD (monotonic; XCD x's at [64 x]); ticket_base: their value at the start of this run */
  int32_t* err;               /* the pipeline's error word (0 = fine)                                                */
  int32_t* err_host;          /* ... its copy for the host (page-locked host memory)                                 */
  uint32_t* started;          /* stage workgroups that have started (the first step of a run is gated on all `grid`) */
  int32_t blocks;             /* env blocks = workgroups of the step kernel                                          */
  int32_t envs_per_block;     /* block b holds envs [b * envs_per_block, min(n_envs, (b + 1) * envs_per_block))      */
  int32_t n_envs;
  int32_t xcds;               /* > 0: block b is always processed on XCD b / (blocks / xcds) - hand-over inside one L2 */
  uint32_t ticket_base;
  int32_t poll_sleep;         /* units of s_sleep 8 (512 cycles) an idle stage wave sleeps between two polls */
  int32_t word_stride;        /* block b's hand-over words: step_prog[b * word_stride], act_prog[b * word_stride] (spread over the memory channels) */
  int32_t seq0;               /* sequence number that pass 0 waits for                                               */
  int32_t pass_first, pass_last;   /* the passes THIS launch serves (pipelined: 0 .. K; lock-step: one pass)          */
  int32_t n_steps;            /* K of the run: pass K is the final one                                               */
  int32_t lockstep;           /* 1: a plain launch of one pass on the world's stream, nothing to wait for            */
  int64_t pass_global0;       /* passes served by earlier runs of this world (index of this run's pass 0)            */
  int64_t timeout_ticks;      /* time-out of the waits in ticks of the 100 MHz wall clock                            */
  /* rows of the env task (rsb.h: rsb_env_*), [n_envs, dim] row-major, device memory */
  const float* ob;            /* [n_envs, ob_dim] observation the next step starts from (a terminated env: its reset state) */
  float* act;                 /* [n_envs, act_dim] actions of the next step (PD target = action_mean + action_std * action) */
  const float* reward;        /* [n_envs] reward of the step just finished (pass > 0)                                */
  const uint8_t* done;        /* [n_envs] 1: the step just finished ended the env's episode                          */
  int32_t ob_dim, act_dim;
  /* launch geometry (host side): launch exactly `grid` workgroups of 64 threads on `stream` */
  int32_t grid;
  void* stream;               /* hipStream_t */
} rsb_stage_ctx;

/* Enqueues the caller's stage kernel for the passes *ctx describes (on ctx->stream, ctx->grid x 64 threads); returns 0 on success.
 * Called once per run when the steps are pipelined, once per pass in lock-step (and again, in lock-step, when a faulted run is replayed:
 * what `user` points to must stay valid and unchanged until the next joining call has returned). */
typedef int (*rsb_stage_launch_fn)(void* user, const rsb_stage_ctx* ctx);

struct rsb_world;
/* K control steps of the device-resident vectorised env (rsb_env_configure first; rsb.h) with the caller's action stage in the loop:
 * pass 0 -> step 1 -> pass 1 -> ... -> step K -> pass K.  Nothing synchronises; with rsb_set_step_pipelining on the steps overlap at
 * env-block granularity (see the top of this file), any other call on the world joins.  The step's outputs live in the world's own
 * buffers (ctx->ob / reward / done; rsb_closed_loop_buffers hands them out), overwritten by every step: a stage that needs a rollout
 * copies its block's rows in its pass. */
int rsb_closed_loop_run(struct rsb_world* w, int n_steps, rsb_stage_launch_fn launch, void* user);

/* The in-repo reference stage: a fixed linear policy  action = clip(bias + W ob + noise[pass]).  All pointers device memory; any but W may
 * be NULL.  W [act_dim, ob_dim] row-major, bias [act_dim], noise [noise_period, n_envs, act_dim] (pass p of the world's life uses slice
 * p % noise_period: exploration noise, or the benchmark's random PD targets with W = 0), clip <= 0: none.
 * Rollout (optional): what an on-policy learner stores - ob [K + 1, n_envs, ob_dim], act [K, n_envs, act_dim] (act[t] = the action
 * computed from ob[t], consumed by step t + 1), reward / done [K, n_envs] (of step t + 1). */
typedef struct rsb_linear_policy {
  const float* W; const float* bias; const float* noise;
  int32_t noise_period;
  float clip;
  float* rollout_ob; float* rollout_act; float* rollout_reward; uint8_t* rollout_done;
} rsb_linear_policy;
int rsb_closed_loop_run_linear(struct rsb_world* w, int n_steps, const rsb_linear_policy* policy);

/* The in-repo MLP stage: the actor network of a raisimGymTorch-style PPO run (upstream's default: MLP ob -> 128 -> 128 -> act, LeakyReLU
 * [RECALL raisimGymTorch/algo/ppo/module.py; absent from /root/reference]) evaluated per env block between every two control steps - the policy
 * a rollout actually has in its loop.  All pointers device memory; any but Wt may be NULL.
 *   layer l:  y = f(W_l x + b_l),  dims[0] = ob_dim, dims[n_layers] = act_dim, 1 <= n_layers <= RSB_MLP_MAX_LAYERS, every width <= 256;
 *             hidden layers use `activation`, the last layer is linear;
 *   Wt[l]     the layer's weight TRANSPOSED: [dims[l], dims[l + 1]] row-major (torch: linear.weight.t().contiguous()) - unit u of a layer is lane
 *             u & 63 of the stage's wave, so a row of Wt is one coalesced load and the B operand of a matrix instruction as it stands;  bias[l] [dims[l + 1]];
 *   input     x = clamp((ob - ob_mean) * ob_inv_std, -ob_clip, ob_clip)  (RaisimGymVecEnv's normalize_ob with frozen statistics [RECALL];
 *             ob_mean / ob_inv_std [ob_dim], NULL: the raw observation; ob_clip <= 0: none);
 *   output    action = clip(y + noise[pass % noise_period]), noise [noise_period, n_envs, act_dim] (pre-sampled exploration noise), clip <= 0: none;
 *   rollout   as rsb_linear_policy's.
 * Arithmetic: fp32 on the matrix cores (v_mfma_f32_4x4x1_16B_f32: a block's four envs x 64 units per instruction, one rank-1 update per input in
 * index order), the same instruction sequence pipelined and in lock-step (bit-identical runs); against a torch fp32 forward pass the actions agree
 * to rounding (tests/test_gpu_closed_loop.py). */
#define RSB_MLP_MAX_LAYERS 4
#define RSB_ACT_TANH 0
#define RSB_ACT_RELU 1
#define RSB_ACT_LEAKY_RELU 2
typedef struct rsb_mlp_policy {
  int32_t n_layers;
  int32_t dims[RSB_MLP_MAX_LAYERS + 1];
  const float* Wt[RSB_MLP_MAX_LAYERS];
  const float* bias[RSB_MLP_MAX_LAYERS];
  int32_t activation;
  float leaky_slope;            /* RSB_ACT_LEAKY_RELU: f(x) = x > 0 ? x : leaky_slope x  (torch's default 0.01) */
  const float* ob_mean; const float* ob_inv_std; float ob_clip;
  const float* noise; int32_t noise_period; float clip;
  float* rollout_ob; float* rollout_act; float* rollout_reward; uint8_t* rollout_done;
} rsb_mlp_policy;
int rsb_closed_loop_run_mlp(struct rsb_world* w, int n_steps, const rsb_mlp_policy* policy);

/* the world's own env-task buffers (device memory): ob [N, ob_dim], act [N, act_dim], reward [N], done [N] */
int rsb_closed_loop_buffers(struct rsb_world* w, float** ob, float** act, float** reward, uint8_t** done);
/* workgroups of the action stage per run (default 256 = one per CU; 0 restores the default) */
int rsb_closed_loop_set_stage_grid(struct rsb_world* w, int workgroups);

/* Joins the pipeline (waits for every pipelined step and stage pass in flight) and reports a fault: RSB_OK, or RSB_E_PIPELINE ONCE after a
 * fault - by then the library has restored the state of the last join, switched pipelining off and replayed the steps in lock-step,
 * so the handle is usable and holds the results the steps were meant to produce.  Every other joining call reports the same way. */
int rsb_step_pipeline_join(struct rsb_world* w);
/* faults so far and the device code of the last one (RSB_PIPE_ERR_*) */
int rsb_step_pipeline_fault(const struct rsb_world* w, int* faults, int* last_code);
/* Debug aid (tests): makes the NEXT pipelined launch fail on the device: kind 1 = its per-XCD ticket base is off by one (the last ticket of
 * every XCD falls outside the range: RSB_PIPE_ERR_TICKET), kind 2 = it waits for a sequence number nobody will publish
 * (RSB_PIPE_ERR_TIMEOUT after RSB_PIPE_TIMEOUT_MS), kind 4 = the error word is set outright (RSB_PIPE_ERR_INJECTED). */
int rsb_debug_pipeline_fault(struct rsb_world* w, int kind);
/* Debug aid (RSB_PIPE_STATS=1 in the environment when the world is created): mean time (us) a pipelined step workgroup waited for its env block and the
 * share of workgroups that waited at all, over the pipelined launches since the last call.  Joins. */
int rsb_debug_pipeline_wait_stats(struct rsb_world* w, double* mean_wait_us, double* waited_frac);

#ifdef __cplusplus
}
#endif

/* ------------------------------------------------------------------------------------------------------------------------------------
 * Device side (HIP, gfx950): the serve loop of an action stage.  Header-only; compiled into the caller's kernel.
 * ---------------------------------------------------------------------------------------------------------------------------------- */
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>

namespace rsb_stage {

__device__ __forceinline__ int ld_word(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fail(const rsb_stage_ctx& c, int code) {
  if ((threadIdx.x & 63) == 0 && atomicCAS(c.err, 0, code) == 0)      /* the first code stays; the host reads its own copy */
    __hip_atomic_store(c.err_host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

/* body(block, env0, n_env, pass, final): called by all 64 lanes of the wave that serves `block`, once per pass, in pass order */
template <class Body>
__device__ __forceinline__ void serve(const rsb_stage_ctx& c, Body&& body) {
  const int lane = (int)(threadIdx.x & 63u);
  if (c.lockstep) {
    /* one pass, launched behind the step it follows on the same stream: nothing to wait for, blocks dealt by workgroup index */
    for (int b = (int)blockIdx.x; b < c.blocks; b += (int)gridDim.x) {
      const int env0 = b * c.envs_per_block;
      body(b, env0, min(c.envs_per_block, c.n_envs - env0), c.pass_first, c.pass_first == c.n_steps);
    }
    return;
  }
  if (threadIdx.x == 0) __hip_atomic_fetch_add(c.started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (blockDim.x != 64u) { fail(c, RSB_PIPE_ERR_STAGE); return; }
  /* This wave's blocks: a FIXED share of the blocks of its XCD (every block is always processed behind the same L2: the hand-over needs no L2
   * write-back).  rank = the order in which the stage's waves arrived on this XCD (one atomic per wave and run), share = blocks rank, rank + T, ...
   * with T = the stage's waves per XCD (the dispatcher deals workgroups round-robin over the XCDs: a rank past T is a geometry fault).
   * No claims, no scanning of other waves' blocks: round 5's first version let every wave of an XCD scan and claim any of its blocks - 32 waves
   * raced for each block with memory-side atomics and a pass over 1024 blocks took 90 us. */
  int base = 0, per = c.blocks, T = (int)gridDim.x, rank = (int)blockIdx.x;
  if (c.xcds > 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    per = c.blocks / c.xcds;
    base = (int)xcc * per;
    T = (int)gridDim.x / c.xcds;
    unsigned t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(c.ticket + ((xcc & 15u) << 6), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - c.ticket_base;
    rank = __builtin_amdgcn_readfirstlane((int)t);
    if (xcc >= (unsigned)c.xcds || rank >= T || T < 1) { fail(c, RSB_PIPE_ERR_STAGE); return; }
  }
  /* lane i of the wave tracks block base + rank + i T (at most 64 blocks per wave: the host sizes the grid) */
  const int mine = rank + lane * T;
  const bool in = mine < per;
  const int b = base + (in ? mine : rank);
  if (rank + 64 * T < per) { fail(c, RSB_PIPE_ERR_STAGE); return; }
  const int last_seq = c.seq0 + c.pass_last;
  int served = c.seq0 + c.pass_first - 1;       /* sequence number of the last pass this wave has served for lane's block */
  long long t_progress = wall_clock64();
  int idle = 0;
  for (;;) {
    const unsigned long long open_ = __ballot(in && served - last_seq < 0);
    if (open_ == 0ull) return;                   /* every block of the share has had its final pass */
    const int pp = ld_word(c.step_prog + (size_t)b * c.word_stride);
    /* pass 0 sees the state the run starts from: nothing to wait for (the host has made the observation rows current before the launch) */
    unsigned long long ready = __ballot(in && served - last_seq < 0 && (pp - served >= 1 || (c.pass_first == 0 && served - (c.seq0 - 1) == 0)));
    if (ready == 0ull) {
      if ((++idle & 15) == 0 && __builtin_amdgcn_readfirstlane(ld_word(c.err)) != 0) return;      /* somebody failed: leave, the host replays in lock-step */
      for (int k = 0; k < c.poll_sleep; ++k) __builtin_amdgcn_s_sleep(8);      /* an idle wave's poll goes to the memory side: not too often */
      if (wall_clock64() - t_progress > c.timeout_ticks) { fail(c, RSB_PIPE_ERR_TIMEOUT_STAGE); return; }
      continue;
    }
    while (ready) {
      const int l = __builtin_ctzll(ready);
      ready &= ready - 1ull;
      const int bb = __builtin_amdgcn_readlane(b, l);
      const int seq = __builtin_amdgcn_readlane(served, l) + 1;
      /* acquire: the rows the step wrote (same XCD: the vector L1 alone; else agent scope) */
      if (c.xcds > 0) asm volatile("buffer_inv sc1" ::: "memory");
      else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      const int env0 = bb * c.envs_per_block;
      const int pass = seq - c.seq0;
      body(bb, env0, min(c.envs_per_block, c.n_envs - env0), pass, pass == c.n_steps);
      /* release: the action rows have arrived in the L2 the next step's workgroup reads from, then the word */
      if (c.xcds > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (lane == 0) __hip_atomic_store(c.act_prog + (size_t)bb * c.word_stride, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lane == l) served = seq;
    }
    t_progress = wall_clock64();
  }
}

}  /* namespace rsb_stage */
#endif /* device side */

#endif /* RSB_PIPELINE_H_ */
