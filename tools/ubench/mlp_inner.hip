// mlp_inner.hip - what the inner loop of the MLP stage costs a LONE wave: per input row 4 v_readlane (one per env) + 4 v_pk_fma_f32 with the
// scalar as a broadcast operand.  Variants isolate the pieces.   hipcc --offload-arch=gfx950 -O3 -o /tmp/mlp_inner tools/ubench/mlp_inner.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int V>
__global__ void __launch_bounds__(64) k(float* out, long long* cyc, int rows, int reps) {
  const int lane = threadIdx.x;
  f2 x[4], acc[4], w = f2{1.0f + lane * 1e-3f, 0.5f};
  for (int e = 0; e < 4; ++e) { x[e] = f2{lane * 0.01f + e, lane * 0.02f - e}; acc[e] = f2{0.f, 0.f}; }
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    for (int k0 = 0; k0 < rows; k0 += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + j;
        if constexpr (V == 0) {          // the stage's loop: readlane -> SGPR -> v_pk_fma_f32
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (j & 1) ? x[e].y : x[e].x), (k & 127) >> 1)); acc[e] = __builtin_elementwise_fma(w, f2{s, s}, acc[e]); }
        } else if constexpr (V == 1) {   // readlane -> two plain v_fma_f32
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (j & 1) ? x[e].y : x[e].x), (k & 127) >> 1)); acc[e].x = fmaf(w.x, s, acc[e].x); acc[e].y = fmaf(w.y, s, acc[e].y); }
        } else if constexpr (V == 2) {   // no readlane: v_pk_fma_f32 on registers
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = __builtin_elementwise_fma(w, x[e], acc[e]);
        } else if constexpr (V == 3) {   // no readlane: two plain v_fma_f32 on registers
#pragma unroll
          for (int e = 0; e < 4; ++e) { acc[e].x = fmaf(w.x, x[e].x, acc[e].x); acc[e].y = fmaf(w.y, x[e].y, acc[e].y); }
        } else if constexpr (V == 4) {   // DPP row broadcast -> v_pk_fma_f32 (one env per row of 16 lanes: one broadcast, four pairs of units)
          const float s = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (j & 1) ? x[0].y : x[0].x), 0x150 + 5, 0xf, 0xf, true));
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = __builtin_elementwise_fma(w, f2{s, s}, acc[e]);
        }
        w.x += 1e-7f;                    // (one more VALU per row in every variant: keeps the rows from collapsing)
      }
    }
  }
  const long long t1 = clock64();
  float s = 0; for (int e = 0; e < 4; ++e) s += acc[e].x + acc[e].y;
  out[lane] = s;
  if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
  float* d; long long* c; CK(hipMalloc(&d, 256)); CK(hipMalloc(&c, 8));
  const int rows = 288, reps = 200;
  const char* name[5] = {"4 readlane + 4 v_pk_fma_f32(SGPR)", "4 readlane + 8 v_fma_f32(SGPR)", "4 v_pk_fma_f32 (registers)", "8 v_fma_f32 (registers)", "1 DPP row bcast + 4 v_pk_fma_f32"};
  for (int v = 0; v < 5; ++v) {
    for (int rep = 0; rep < 2; ++rep) {
      if (v == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, c, rows, reps);
      if (v == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, c, rows, reps);
      if (v == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, c, rows, reps);
      if (v == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d, c, rows, reps);
      if (v == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, d, c, rows, reps);
      CK(hipDeviceSynchronize());
    }
    long long h; CK(hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost));
    std::printf("%-40s %7.1f cycles per row (shader clock)   -> %6.2f us per 290 rows at 2.4 GHz\n", name[v], (double)h / (rows * reps), (double)h / (rows * reps) * 290 / 2400.0);
  }
  return 0;
}
