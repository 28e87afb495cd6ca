// mfma_4x4_bcast.hip - operand layout of v_mfma_f32_4x4x1_16B_f32 and its A-matrix broadcast (cbsz / abid), read off the hardware.
//   D_b[i][j] += A_b[i][0] * B_b[0][j]  for 16 blocks b;  with cbsz = 4, abid = q every block takes block q's A.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_4x4 tools/ubench/mfma_4x4_bcast.hip && /tmp/mfma_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  const int l = threadIdx.x;
  const float a = 100.f * (l / 4) + (l % 4) + 1.f;      // A: lane l = (block l / 4, row i = l % 4): 100 b + i + 1
  const float b = 0.01f * l + 1000.f;                   // B: lane l = (block l / 4, column j = l % 4): 1000 + 0.01 l
  f4 c = {0.f, 0.f, 0.f, 0.f};
  f4 d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  f4 d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 5, 0);      // every block: A of block 5
  for (int v = 0; v < 4; ++v) { out[v * 64 + l] = d0[v]; out[256 + v * 64 + l] = d1[v]; }
}
int main() {
  float* d; hipMalloc(&d, 512 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[512]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  // expected without broadcast: D_b[i][j] = (100 b + i + 1) * (1000 + 0.01 (4 b + j)); find which (register, lane) holds D_b[i][j]
  int ok0 = 0, ok1 = 0;
  for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
    const int b = l / 4, j = l % 4, i = v;
    const float e0 = (100.f * b + i + 1.f) * (1000.f + 0.01f * (4 * b + j)), e1 = (100.f * 5 + i + 1.f) * (1000.f + 0.01f * (4 * b + j));
    ok0 += h[v * 64 + l] == e0; ok1 += h[256 + v * 64 + l] == e1;
  }
  std::printf("layout 'register v = row i, lane 4 b + j = (block b, column j)': %d / 256 elements match without broadcast, %d / 256 with cbsz 4 abid 5 (= every block uses block 5's A)\n", ok0, ok1);
  std::printf("lane 9 (block 2, column 1): d0 = %.2f %.2f %.2f %.2f   d1 = %.2f %.2f %.2f %.2f\n", h[9], h[64 + 9], h[128 + 9], h[192 + 9], h[256 + 9], h[320 + 9], h[384 + 9], h[448 + 9]);
  return 0;
}
