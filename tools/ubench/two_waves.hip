// Micro-benchmark (GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/tw two_waves.hip && /tmp/tw):
// does a SECOND wave on a SIMD overlap with the first one?  (the step kernel runs one wave per SIMD at
// 415 registers; whether a <= 256-register variant with two waves per SIMD can win depends on what two co-resident waves share.)
//
// ONE workgroup of W x 64 threads on one CU: W = 4 puts one wave on each of the CU's four SIMDs, W = 8 two, W = 16 four (the SIMD of
// every wave is read from HW_ID and printed, so the placement is a measurement, not an assumption).  All waves run the same pattern REP
// times between two s_memtime stamps, after an s_barrier so that they start together.  Printed per pattern: cycles per pattern as seen by
// a wave (mean over the waves) at W = 4 / 8 / 16 and the ratio to the lone wave.  Ratio 1.0 = the co-resident waves overlap completely
// (the resource has room for them), 2.0 at W = 8 = they take turns (the resource was already full with one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
#define STR2(x) #x
#define STR(x) STR2(x)
#define NPAT 16

#define TIMED(idx, body)                                                                 \
  {                                                                                      \
    unsigned long long t0, t1;                                                           \
    __builtin_amdgcn_s_barrier();                                                        \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); \
    asm volatile("s_mov_b32 s42, 3" ::: "s42");                                         \
    asm volatile(".rept " STR(REP) "\n" body "\n.endr" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(addr), "s"(sptr) : "vcc", "s40", "s41", "s42", "s43", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "memory"); \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); \
    if ((threadIdx.x & 63) == 0) out[(threadIdx.x >> 6) * NPAT + idx] = (long long)(t1 - t0);  \
  }

// the sweep loop's instruction mix (step_kernel.h, ISA statistics of the quadruped instance): ~78 % VALU (a third of it dependent), 12 % SALU,
// 6 % LDS with a wait, 4 % DPP, one scalar branch per ~60 instructions
#define MIX                                                                                               \
  "ds_read_b128 v[100:103], %4\n"                                                                       \
  "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_mul_f32 %3, %3, %1\n" \
  "s_mov_b32 s40, 1\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2\n"                            \
  "v_mov_b32_dpp %2, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"                                    \
  "v_fma_f32 %0, %0, %1, %2\n v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %3, %3, %2, vcc\n"              \
  "s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, v100, %1, %0\n v_fma_f32 %3, v101, %1, %3\n s_add_u32 s41, s41, s40\n" \
  "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2\n"

__global__ void __launch_bounds__(1024) tw(long long* out, int* hwid, float* sink, const float* sptr) {
  extern __shared__ float lds[];      // 16 KB of dynamic LDS (the patterns address it by hand: a static array nobody reads is optimised away)
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  float x = threadIdx.x * 0.001f + 1.0f, y = 1.0001f, z = 0.5f, w = 2.0f;
  unsigned addr = (threadIdx.x & 15) * 16 + (threadIdx.x >> 6) * 256;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) hwid[threadIdx.x >> 6] = (int)id;
  TIMED(0, "")                                                                  // empty: stamp overhead
  TIMED(1, "v_fma_f32 %0, %0, %1, %2")                                          // dependent FMA chain
  TIMED(2, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2")               // 2 independent chains
  TIMED(3, "v_pk_fma_f32 v[100:101], v[102:103], v[104:105], v[100:101]")       // dependent packed FMA
  TIMED(4, "s_mov_b32 s40, 1\n s_mov_b32 s41, 2\n s_mov_b32 s42, 3\n s_mov_b32 s43, 4")   // 4 SALU
  TIMED(5, "ds_read_b128 v[100:103], %4\n s_waitcnt lgkmcnt(0)")               // LDS read + wait
  TIMED(6, "ds_read_b128 v[100:103], %4\n s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, v100, %1, %0\n v_fma_f32 %0, v101, %1, %0\n v_fma_f32 %0, v102, %1, %0\n v_fma_f32 %0, v103, %1, %0")   // read -> wait -> 4 dependent FMAs
  TIMED(7, "s_cmp_lt_i32 s42, 5\n s_cbranch_scc0 1f\n v_fma_f32 %0, %0, %1, %2\n1:")      // scalar branch not taken + FMA
  TIMED(8, "v_cmp_gt_f32 vcc, %0, %1\n s_cbranch_vccz 1f\n v_fma_f32 %0, %0, %1, %2\n1:") // vector compare -> branch not taken
  TIMED(9, MIX)                                                                 // the sweep loop's mix (19 instructions)
  TIMED(10, "v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf")
  TIMED(11, "v_rcp_f32 %0, %0\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %3, %3, %1, %2")
  TIMED(12, "ds_write_b128 %4, v[100:103]\n s_waitcnt lgkmcnt(0)")
  TIMED(13, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n s_mov_b32 s40, 1\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n s_mov_b32 s41, 1")   // 8 VALU : 2 SALU
  sink[threadIdx.x] = x + y + z + w;
}

int main() {
  long long* d; int* hw; float* s; float* c;
  hipMalloc(&d, 16 * NPAT * sizeof(long long)); hipMalloc(&hw, 16 * sizeof(int)); hipMalloc(&s, 1024 * sizeof(float)); hipMalloc(&c, 256);
  hipMemset(c, 0, 256);
  const char* names[NPAT] = {"empty (stamp overhead, total)", "dependent v_fma chain", "2 independent v_fma chains", "dependent v_pk_fma_f32", "4 s_mov",
                             "ds_read_b128 + wait", "ds_read_b128 + wait + 4 dependent v_fma", "s_cmp + s_cbranch_scc not taken + v_fma",
                             "v_cmp + s_cbranch_vccz not taken + v_fma", "sweep-loop mix (19 instr: 13 VALU 2 SALU 1 LDS 1 DPP 1 wait ..)",
                             "3 independent DPP row_newbcast", "v_rcp + 3 dependent-chain v_fma", "ds_write_b128 + wait", "8 dependent v_fma : 2 s_mov"};
  const int Ws[3] = {4, 8, 16};
  double res[3][NPAT];
  for (int wi = 0; wi < 3; ++wi) {
    const int W = Ws[wi];
    std::vector<long long> h(16 * NPAT);
    std::vector<int> hid(16);
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(d, 0, 16 * NPAT * sizeof(long long));
      hipLaunchKernelGGL(tw, dim3(1), dim3(64 * W), 16384, 0, d, hw, s, c);
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, 16 * NPAT * sizeof(long long), hipMemcpyDeviceToHost);
    hipMemcpy(hid.data(), hw, 16 * sizeof(int), hipMemcpyDeviceToHost);
    std::printf("W = %2d waves in one workgroup; (wave: simd, cu) =", W);
    int per_simd[4] = {0, 0, 0, 0};
    for (int k = 0; k < W; ++k) {
      const int simd = (hid[k] >> 4) & 3, cu = (hid[k] >> 8) & 15;   // HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8]
      std::printf(" (%d: %d, %d)", k, simd, cu);
      per_simd[simd]++;
    }
    std::printf("  -> waves per SIMD %d %d %d %d\n", per_simd[0], per_simd[1], per_simd[2], per_simd[3]);
    for (int i = 0; i < 14; ++i) {
      double m = 0;
      for (int k = 0; k < W; ++k) m += (double)(h[k * NPAT + i] - (i == 0 ? 0 : h[k * NPAT + 0]));
      res[wi][i] = m / W / (i == 0 ? 1 : REP);
    }
  }
  std::printf("%-66s %10s %10s %10s   %s\n", "pattern (cycles per pattern and wave)", "1 / SIMD", "2 / SIMD", "4 / SIMD", "ratio 2:1, 4:1");
  for (int i = 0; i < 14; ++i)
    std::printf("%2d %-63s %10.1f %10.1f %10.1f   %.2f %.2f\n", i, names[i], res[0][i], res[1][i], res[2][i], i ? res[1][i] / res[0][i] : 0.0, i ? res[2][i] / res[0][i] : 0.0);
  return 0;
}
