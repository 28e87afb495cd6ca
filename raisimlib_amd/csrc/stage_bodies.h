// stage_bodies.h — the in-repo action stages' per-env-block bodies (rsb_linear_policy, rsb_mlp_policy of include/rsb_pipeline.h), shared by
//   rsb_pipeline.hip   the stage KERNELS: rsb_stage::serve(ctx, body) - one persistent launch next to the pipelined steps, or one launch per pass in lock-step;
//   step_kernel.h      the RESIDENT step classes (CL bit 64): the env block's own wave evaluates the same body between two control steps of ONE launch
//                      (at that boundary the step's registers are dead), so a resident run is bit-identical to the pipelined and the lock-step run.
// A body is called by all 64 lanes of a wave: body(ctx, policy, env0, n_env, pass, final) - observation rows ctx.ob[env0 .. env0 + n_env) are ready,
// unless `final` it writes the action rows ctx.act[env0 ..) (and the rollout rows of `pass`).
// Upstream counterpart: none (raisimGymTorch evaluates its actor in PyTorch on the learner's side [RECALL]; absent from /root/reference).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "rsb_pipeline.h"

#ifndef RSB_PRAGMA_UNROLL
#define RSB_PRAGMA_UNROLL _Pragma("unroll")
#endif

namespace rsb_stage_body {

// ---- the in-repo reference stage: a fixed linear policy (rsb_linear_policy) --------------------------------------------------------
// lane = (env of the block, action entry); the sum runs over the observation in index order with one FMA per term - the same instruction
// sequence whether the pass is served from the pipeline or launched in lock-step, so the two produce the same bits.
// what an on-policy learner stores of a pass: the block's observation rows, and the reward / done flags of the step just finished
__device__ __forceinline__ void record_rollout(const rsb_stage_ctx& c, float* rollout_ob, float* rollout_reward, uint8_t* rollout_done, int env0, int n_env, int pass) {
  const int lane = (int)threadIdx.x, od = c.ob_dim;
  const size_t N = (size_t)c.n_envs;
  if (rollout_ob) {      // (four rows of a block in flight at once: a stage wave pays the full L2 latency for every load -> wait round)
    const float* src = c.ob + (size_t)env0 * od;
    float* dst = rollout_ob + ((size_t)pass * N + env0) * od;
    const int n = n_env * od;
    for (int i0 = lane; i0 < n; i0 += 256) {
      float v[4];
      RSB_PRAGMA_UNROLL for (int k = 0; k < 4; ++k) v[k] = src[min(i0 + 64 * k, n - 1)];
      RSB_PRAGMA_UNROLL for (int k = 0; k < 4; ++k) if (i0 + 64 * k < n) dst[i0 + 64 * k] = v[k];
    }
  }
  if (pass > 0 && lane < n_env) {
    if (rollout_reward) rollout_reward[(size_t)(pass - 1) * N + env0 + lane] = c.reward[env0 + lane];
    if (rollout_done) rollout_done[(size_t)(pass - 1) * N + env0 + lane] = c.done[env0 + lane];
  }
}
// (P: rsb_linear_policy, or the same struct in the kernarg address space - the resident step classes read the policy straight from their arguments)
template <class P>
__device__ __forceinline__ void linear_block(const rsb_stage_ctx& c, const P& p, int env0, int n_env, int pass, bool final) {
  const int lane = (int)threadIdx.x;
  const int od = c.ob_dim, ad = c.act_dim;
  const size_t N = (size_t)c.n_envs;
  record_rollout(c, p.rollout_ob, p.rollout_reward, p.rollout_done, env0, n_env, pass);
  if (final) return;
  const long long gp = c.pass_global0 + pass;
  const float* nz = p.noise ? p.noise + (size_t)(gp % (p.noise_period > 0 ? p.noise_period : 1)) * N * ad : nullptr;
  constexpr int CH = 16;     // terms of the sum loaded together (weights and observation entries: 32 loads in flight, then 16 FMAs in index order)
  for (int idx = lane; idx < n_env * ad; idx += 64) {
    const int e = idx / ad, j = idx - e * ad;
    const float* ob = c.ob + (size_t)(env0 + e) * od;
    const float* wr = p.W + (size_t)j * od;
    float acc = p.bias ? p.bias[j] : 0.f;
    const float noise = nz ? nz[(size_t)(env0 + e) * ad + j] : 0.f;
    for (int i0 = 0; i0 < od; i0 += CH) {
      float wv[CH], ov[CH];
      RSB_PRAGMA_UNROLL for (int k = 0; k < CH; ++k) { const int i = min(i0 + k, od - 1); wv[k] = wr[i]; ov[k] = ob[i]; }
      RSB_PRAGMA_UNROLL for (int k = 0; k < CH; ++k) if (i0 + k < od) acc = fmaf(wv[k], ov[k], acc);
    }
    acc += noise;
    if (p.clip > 0.f) acc = fminf(fmaxf(acc, -p.clip), p.clip);
    c.act[(size_t)(env0 + e) * ad + j] = acc;
    if (p.rollout_act) p.rollout_act[((size_t)pass * N + env0 + e) * ad + j] = acc;
  }
}

// ---- the in-repo MLP stage (rsb_mlp_policy): the actor network of a PPO rollout, per env block, on the matrix cores ------------------------
// A block is FOUR envs: y[unit][env] += W[unit][k] x[k][env] is a rank-1 update of a (units x 4) matrix per input k - v_mfma_f32_4x4x1_16B_f32
// does sixteen 4 x 4 blocks of it at once = 64 units x 4 envs per instruction, every multiplier busy:
//   B (1 x 4 per block, lane 4 b + j)  = the weights of input k for units 4 b + j: lane = unit, ONE coalesced row of the transposed matrix;
//   A (4 x 1 per block, lane 4 b' + i) = the four envs' input k - the same for all sixteen blocks: the instruction's A-BROADCAST (cbsz 4, abid q)
//                                        feeds every block from quad q of the register, so ONE register holds sixteen inputs x four envs
//                                        (lane l: env l & 3, input 16 r + (l >> 2)) and nothing is broadcast by hand;
//   D (4 x 4 per block)                = register i: env i, lane = unit - the layout the activation, the bias and the action rows want.
// (tools/ubench/mfma_4x4_bcast.hip reads both layouts and the broadcast off the hardware.)  Between two layers the (env, unit) registers are
// turned into (input, env) registers by ds_bpermute (the LDS crossbar, no LDS memory: the step kernel's workgroups own all of it).
// The first versions ran on the vector ALU: the input had to become a scalar per env (v_readlane) - 65 cycles per input row, half of them
// the readlanes (tools/ubench/mlp_inner.hip), 20 us per block; here a row is two matrix instructions.
// The weight rows still come from L2 (the stage has nowhere to keep 89 KB) through a RING of NB groups of KU rows with explicit loads and waits
// (inline asm: the compiler's own schedule waited for every group): NB - 1 groups are in flight while one is consumed.  No LDS memory, <= 96
// registers: what a step wave leaves of its SIMD (tests/test_gpu_closed_loop.py reads both off the ISA).
typedef float f4v __attribute__((ext_vector_type(4)));
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {      // f(integral_constant<int, I>) for I = 0 .. N - 1: register indices and abid stay compile-time
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
__device__ __forceinline__ void ring_load(float& dst, unsigned voff, const char* sbase) {
  asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
// the wait passes the group's INPUT REGISTER through itself (a tied operand, no instruction): every matrix instruction of the group reads it, so
// none of them moves above the wait (tying the ring registers cost 2 v_mov per row, tying the accumulators 8 v_accvgpr_mov per group)
template <int CNT>
__device__ __forceinline__ void ring_wait(float& xr) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(xr) : "n"(CNT));
}
template <int NS, class P, int RING = 0>   // NS sets of 64 units: layer widths up to 64 NS; P: rsb_mlp_policy, or the same struct in the kernarg address space (a COPY of it
                                 // in registers would be indexed by the layer loop - private memory, and the weight rows' base no longer a scalar);
                                 // RING: groups in the weight ring (0 = the stage kernels' 5 / 2: they must stay under 96 registers next to a step wave; the resident step
                                 // classes evaluate the stage while the step's own registers are dead and take 8 = 56 loads in flight, the most s_waitcnt vmcnt can count:
                                 // the network is bound by L2 latency per round trip of the ring.  The order of the accumulations - the result - does not depend on it)
__device__ __forceinline__ void mlp_block(const rsb_stage_ctx& c, const P& p, int env0, int n_env, int pass, bool final) {
  constexpr int XR = 4 * NS;                  // input registers: 16 inputs x 4 envs each
  constexpr int KU = NS == 2 ? 4 : 2;         // weight rows per group (8 loads)
  constexpr int NB = RING > 0 ? RING : (NS == 2 ? 5 : 2);         // groups in the ring: 16 / 2 rows in flight = 40 / 16 registers
  static_assert((NB - 1) * KU * NS <= 63, "s_waitcnt vmcnt counts 63 outstanding loads at most");
  constexpr int LOADS = KU * NS;
  const int lane = (int)threadIdx.x;
  const int od = c.ob_dim, ad = c.act_dim;
  const size_t N = (size_t)c.n_envs;
  record_rollout(c, p.rollout_ob, p.rollout_reward, p.rollout_done, env0, n_env, pass);
  if (final) return;
  const int le = lane & 3, lq = lane >> 2;    // this lane's env and input slot in an input register
  float x[XR];
  RSB_PRAGMA_UNROLL for (int r = 0; r < XR; ++r) {
    const int k = 16 * r + lq;
    float v = 0.f;
    if (k < od && le < n_env) {
      v = c.ob[(size_t)(env0 + le) * od + k];
      if (p.ob_mean) v -= p.ob_mean[k];
      if (p.ob_inv_std) v *= p.ob_inv_std[k];
      if (p.ob_clip > 0.f) v = fminf(fmaxf(v, -p.ob_clip), p.ob_clip);
    }
    x[r] = v;
  }
  f4v acc[NS];                                // [set][env]: unit 64 s + lane
  for (int l = 0; l < p.n_layers; ++l) {
    const int in = p.dims[l], out = p.dims[l + 1];
    const char* Wb = reinterpret_cast<const char*>(p.Wt[l]);
    unsigned voff[NS];
    RSB_PRAGMA_UNROLL for (int s = 0; s < NS; ++s) {
      const int u = 64 * s + lane;
      voff[s] = 4u * (unsigned)min(u, out - 1);      // byte offset in a row (clamped: no load is predicated; units past the layer's width are zeroed below)
      const float b = (p.bias[l] && u < out) ? p.bias[l][u] : 0.f;
      acc[s] = f4v{b, b, b, b};
    }
    float ring[NB][KU][NS];
    // Rows run in chunks of 16 (one input register): a chunk that starts at or past `in` ends the layer, inside a chunk nothing branches - a lone
    // wave pays 20-45 cycles per branch, a group's eight matrix instructions 64.  Rows past the end are clamped to the last one and meet zero inputs
    // (so do the few groups the ring fetches ahead of the last chunk).  Uniform 64-bit row base in SGPRs + 32-bit lane offset: no address registers.
    const unsigned rowbytes = 4u * (unsigned)out;
    const int last = in - 1;
    auto issue = [&](auto gc, auto slot) {
      constexpr int g = decltype(gc)::value, S = decltype(slot)::value;
      static_for<0, KU>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const char* rowp = Wb + (size_t)((unsigned)min(g * KU + j, last) * rowbytes);
        static_for<0, NS>([&](auto sc) { ring_load(ring[S][j][decltype(sc)::value], voff[decltype(sc)::value], rowp); });
      });
    };
    static_for<0, NB - 1>([&](auto gc) { issue(gc, gc); });
    bool more = true;
    static_for<0, XR>([&](auto rc) {          // chunk r: inputs 16 r .. 16 r + 15 = 16 / KU groups, statically: ring slot g % NB, quad k & 15
      constexpr int r = decltype(rc)::value;
      if (more && 16 * r < in) {
        static_for<(16 / KU) * r, (16 / KU) * (r + 1)>([&](auto gc) {
          constexpr int g = decltype(gc)::value, S = g % NB;
          // one more group goes into the slot consumed last, then wait until THIS slot's loads - the oldest in flight - have landed
          issue(std::integral_constant<int, g + NB - 1>{}, std::integral_constant<int, (g + NB - 1) % NB>{});
          ring_wait<(NB - 1) * LOADS>(x[r]);
          static_for<0, KU>([&](auto jc) {
            constexpr int j = decltype(jc)::value, k = g * KU + j;
            RSB_PRAGMA_UNROLL for (int s = 0; s < NS; ++s) acc[s] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[r], ring[S][j][s], acc[s], 4, k & 15, 0);
          });
        });
      } else {
        more = false;
      }
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the groups fetched past the last chunk: the ring is reused by the next layer)
    const bool hidden = l + 1 < p.n_layers;
    RSB_PRAGMA_UNROLL for (int s = 0; s < NS; ++s) {
      const bool live = 64 * s + lane < out;
      RSB_PRAGMA_UNROLL for (int e = 0; e < 4; ++e) {
        float v = acc[s][e];
        if (hidden) v = p.activation == RSB_ACT_TANH ? tanhf(v) : p.activation == RSB_ACT_RELU ? fmaxf(v, 0.f) : (v > 0.f ? v : p.leaky_slope * v);
        acc[s][e] = live ? v : 0.f;         // (units past the layer's width loaded clamped weights)
      }
    }
    if (hidden) {
      // (env, unit) -> (input, env): input register r, lane l takes unit 16 r + (l >> 2) of env l & 3 = lane 16 (r & 3) + (l >> 2) of set r >> 2
      RSB_PRAGMA_UNROLL for (int r = 0; r < XR; ++r) {
        const int src = 4 * (16 * (r & 3) + lq);
        // (the permutes are inline asm ON PURPOSE.  Written with __builtin_amdgcn_ds_bpermute and a select over the destination lane's env - or a masked
        //  OR of the four results - this kernel came out with ONE permute per register instead of four: envs 1 .. 3 of a block received values computed
        //  for other envs.  The bit-identity tests passed (both runs wrong alike) and so did the comparison with torch while a block's envs still moved
        //  alike; it failed at 0.28 once they had parted (tests/test_gpu_closed_loop.py).  The pattern alone does not reproduce it
        //  (tools/ubench/bpermute_select_fold.hip: four permutes, right results), so it is this kernel's context; the asm form leaves nothing to fold.)
        float t0, t1, t2, t3;
        asm volatile("ds_bpermute_b32 %0, %4, %5\n\tds_bpermute_b32 %1, %4, %6\n\tds_bpermute_b32 %2, %4, %7\n\tds_bpermute_b32 %3, %4, %8\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                     : "v"(src), "v"(acc[r >> 2][0]), "v"(acc[r >> 2][1]), "v"(acc[r >> 2][2]), "v"(acc[r >> 2][3]));
        x[r] = le == 0 ? t0 : le == 1 ? t1 : le == 2 ? t2 : t3;
      }
    }
  }
  const long long gp = c.pass_global0 + pass;
  const float* nz = p.noise ? p.noise + (size_t)(gp % (p.noise_period > 0 ? p.noise_period : 1)) * N * ad : nullptr;
  RSB_PRAGMA_UNROLL for (int s = 0; s < NS; ++s) {
    const int j = 64 * s + lane;
    if (j < ad) {
      RSB_PRAGMA_UNROLL for (int e = 0; e < 4; ++e) {
        if (e < n_env) {
          float a = acc[s][e] + (nz ? nz[(size_t)(env0 + e) * ad + j] : 0.f);
          if (p.clip > 0.f) a = fminf(fmaxf(a, -p.clip), p.clip);
          c.act[(size_t)(env0 + e) * ad + j] = a;
          if (p.rollout_act) p.rollout_act[((size_t)pass * N + env0 + e) * ad + j] = a;
        }
      }
    }
  }
}
}  // namespace rsb_stage_body
