// Micro-benchmark (GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/ub lone_wave_latency.hip && /tmp/ub):
// what instructions cost a LONE wavefront on a SIMD (the step kernel's regime: 4096 envs = one wave per SIMD), in shader
// cycles (s_memtime).  Each pattern is repeated REP times between two time stamps; printed = cycles per pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
#define STR2(x) #x
#define STR(x) STR2(x)

#define TIMED(idx, body)                                                                 \
  {                                                                                      \
    unsigned long long t0, t1;                                                           \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); \
    asm volatile("s_mov_b32 s42, 3" ::: "s42");                                         \
    asm volatile(".rept " STR(REP) "\n" body "\n.endr" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(addr), "s"(sptr) : "vcc", "s40", "s41", "s42", "s43", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "memory"); \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); \
    if (threadIdx.x == 0) out[idx] = (long long)(t1 - t0);                               \
  }

__global__ void __launch_bounds__(64) ub(long long* out, float* sink, const float* sptr) {
  __shared__ float lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (float)i;
  __syncthreads();
  float x = threadIdx.x * 0.001f + 1.0f, y = 1.0001f, z = 0.5f, w = 2.0f;
  unsigned addr = (threadIdx.x & 15) * 16;
  TIMED(0, "")                                                                  // empty: stamp overhead
  TIMED(1, "v_fma_f32 %0, %0, %1, %2")                                          // dependent FMA chain
  TIMED(2, "v_fma_f32 %0, %1, %2, %3\n v_fma_f32 %1, %2, %3, %0")               // 2 FMAs, loosely dependent
  TIMED(3, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2")               // 2 independent chains
  TIMED(4, "v_rcp_f32 %0, %0")                                                  // dependent rcp
  TIMED(5, "v_rcp_f32 %0, %0\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %1, %1, %2, %3")   // rcp + 3 independent FMAs
  TIMED(6, "v_rsq_f32 %0, %0\n v_mul_f32 %0, %0, %1")                           // rsq then dependent mul
  TIMED(7, "v_fma_f32 %0, %0, %1, %2\n s_nop 1\n v_mov_b32_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf")   // VALU -> DPP on the result
  TIMED(8, "v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")          // compare + select
  TIMED(9, "v_cmp_gt_f32 vcc, %0, %1\n s_cbranch_vccz 1f\n v_fma_f32 %0, %0, %1, %2\n1:")      // branch not taken (x > y true -> vcc nonzero)
  TIMED(10, "v_cmp_lt_f32 vcc, %0, %1\n s_cbranch_vccz 1f\n v_fma_f32 %0, %0, %1, %2\n1:")     // branch taken (skips the FMA)
  TIMED(11, "v_cmp_gt_f32 s[40:41], %0, %1\n s_and_b64 s[42:43], s[40:41], exec\n s_cbranch_scc0 1f\n v_fma_f32 %0, %0, %1, %2\n1:")  // cmp -> SALU -> branch (not taken)
  TIMED(12, "ds_read_b128 v[200:203], %4\n s_waitcnt lgkmcnt(0)")               // LDS read latency, lone wave
  TIMED(13, "ds_read_b128 v[200:203], %4\n ds_read_b128 v[204:207], %4 offset:256\n ds_read_b128 v[208:211], %4 offset:512\n s_waitcnt lgkmcnt(0)")   // 3 reads in flight
  TIMED(14, "ds_read_b96 v[200:202], %4\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n s_waitcnt lgkmcnt(0)")   // read hidden behind 8 FMAs?
  TIMED(15, "v_readlane_b32 s40, %0, 5\n v_fma_f32 %1, %1, %2, %3")             // readlane
  TIMED(16, "v_readlane_b32 s40, %0, 5\n s_nop 3\n v_mul_f32 %1, s40, %1")      // readlane used by VALU
  TIMED(17, "s_load_dword s40, %5, 0x0\n s_waitcnt lgkmcnt(0)")                 // scalar load (cached) latency
  TIMED(18, "s_mov_b32 s40, 1\n s_mov_b32 s41, 2\n s_mov_b32 s42, 3\n s_mov_b32 s43, 4")   // 4 SALU
  TIMED(19, "ds_write_b128 %4, v[200:203]\n s_waitcnt lgkmcnt(0)")              // LDS write + wait
  TIMED(20, "ds_add_f32 %4, %1\n s_waitcnt lgkmcnt(0)")                         // LDS float atomic + wait
  TIMED(21, "v_fma_f32 %0, %0, %1, %2\n s_barrier")                             // barrier of a single-wave workgroup
  TIMED(22, "v_sqrt_f32 %0, %0")
  TIMED(23, "v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n s_branch 1f\n1:")   // unconditional taken branch every 4 VALU
  TIMED(24, "s_cmp_lt_i32 s42, 5\n s_cbranch_scc0 1f\n v_fma_f32 %0, %0, %1, %2\n1:")               // scalar compare + branch NOT taken (s42 = 3)
  TIMED(25, "s_cmp_gt_i32 s42, 5\n s_cbranch_scc0 1f\n v_fma_f32 %0, %0, %1, %2\n1:")               // scalar compare + branch TAKEN
  TIMED(26, "v_cmp_gt_f32 vcc, %0, %1\n s_and_saveexec_b64 s[40:41], vcc\n s_cbranch_execz 1f\n v_fma_f32 %0, %0, %1, %2\n1:\n s_or_b64 exec, exec, s[40:41]")   // divergent if with execz skip, not taken
  TIMED(27, "v_cmp_gt_f32 vcc, %0, %1\n s_and_saveexec_b64 s[40:41], vcc\n v_fma_f32 %0, %0, %1, %2\n s_or_b64 exec, exec, s[40:41]")   // divergent if, predication only (no branch)
  TIMED(28, "v_cmp_gt_f32 vcc, %0, %1\n s_nop 4\n s_cbranch_vccz 1f\n v_fma_f32 %0, %0, %1, %2\n1:")   // is the cost a VALU->branch hazard? pad with s_nop
  TIMED(29, "v_cmp_gt_f32 vcc, %0, %1\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %3, %3, %1, %2\n s_cbranch_vccz 1f\n v_fma_f32 %0, %0, %1, %2\n1:")   // 6 independent FMAs between compare and branch
  TIMED(30, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n s_cmp_lt_i32 s42, 5\n s_cbranch_scc0 1f\n v_fma_f32 %0, %0, %1, %2\n1:")   // 16 FMAs + scalar branch not taken: amortised cost
  TIMED(31, "v_pk_fma_f32 v[200:201], v[202:203], v[204:205], v[200:201]")                              // packed fp32 FMA, dependent
  TIMED(32, "v_pk_fma_f32 v[200:201], v[202:203], v[204:205], v[206:207]\n v_pk_fma_f32 v[208:209], v[202:203], v[204:205], v[206:207]")   // 2 independent packed FMAs
  TIMED(33, "ds_bpermute_b32 v200, %4, %0\n s_waitcnt lgkmcnt(0)")                                     // bpermute latency
  TIMED(34, "v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf")   // 3 independent DPP movs
  sink[threadIdx.x] = x + y + z + w;
}

int main() {
  long long* d; float* s; float* c;
  hipMalloc(&d, 64 * sizeof(long long)); hipMalloc(&s, 64 * sizeof(float)); hipMalloc(&c, 256);
  hipMemset(d, 0, 64 * sizeof(long long)); hipMemset(c, 0, 256);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(ub, dim3(1), dim3(64), 0, 0, d, s, c);
  hipDeviceSynchronize();
  std::vector<long long> h(64);
  hipMemcpy(h.data(), d, 64 * sizeof(long long), hipMemcpyDeviceToHost);
  const char* names[] = {"empty (stamp overhead, total cycles)", "dependent v_fma chain", "2 v_fma loosely dependent", "2 independent v_fma chains",
                         "dependent v_rcp", "v_rcp + 3 independent v_fma", "v_rsq + dependent v_mul", "v_fma -> s_nop 1 -> DPP row_newbcast of it",
                         "v_cmp + v_cndmask (vcc)", "v_cmp + s_cbranch_vccz NOT taken + v_fma", "v_cmp + s_cbranch_vccz TAKEN (skips v_fma)",
                         "v_cmp sgpr + s_and + s_cbranch_scc0 not taken + v_fma", "ds_read_b128 + wait (LDS latency)", "3 ds_read_b128 + wait",
                         "ds_read_b96 + 8 dependent v_fma + wait", "v_readlane + independent v_fma", "v_readlane + s_nop 3 + v_mul using it",
                         "s_load_dword (cached) + wait", "4 s_mov", "ds_write_b128 + wait", "ds_add_f32 + wait", "v_fma + s_barrier (1-wave workgroup)",
                         "dependent v_sqrt", "4 v_mul + s_branch taken",
                         "s_cmp + s_cbranch_scc NOT taken + v_fma", "s_cmp + s_cbranch_scc TAKEN (skips v_fma)",
                         "v_cmp + s_and_saveexec + s_cbranch_execz (not taken) + v_fma + s_or exec", "v_cmp + s_and_saveexec + v_fma + s_or exec (no branch)",
                         "v_cmp + s_nop 4 + s_cbranch_vccz not taken + v_fma", "v_cmp + 6 independent v_fma + s_cbranch_vccz + v_fma",
                         "16 dependent v_fma + s_cmp + s_cbranch_scc not taken + v_fma", "dependent v_pk_fma_f32", "2 independent v_pk_fma_f32",
                         "ds_bpermute_b32 + wait", "3 independent DPP row_newbcast movs"};
  const double base = (double)h[0];
  for (int i = 0; i < 35; ++i) std::printf("%2d %-58s %8.1f cycles per pattern\n", i, names[i], i == 0 ? base : (h[i] - base) / REP);
  return 0;
}
