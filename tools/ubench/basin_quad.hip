// basin_quad.hip — are the sixteen energies of the basin check the same numbers when four lanes scan four directions each (runtime x, y) as when one lane
// scans sixteen literal directions?  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-hip-fp32-correctly-rounded-divide-sqrt -I include -I raisimlib_amd/csrc tools/ubench/basin_quad.hip -o /tmp/basin_quad && /tmp/basin_quad
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "step_slip.h"
using namespace rsbk;
__global__ void k(const float* coef, float* out_serial, float* out_rt, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  SlipCoef c; const float* p = coef + 13 * i;
  c.a0 = p[0]; c.a1 = p[1]; c.a2 = p[2]; c.n00 = p[3]; c.n01 = p[4]; c.n02 = p[5]; c.n10 = p[6]; c.n11 = p[7]; c.n12 = p[8]; c.vn = p[9]; c.ls0 = p[10]; c.ls1 = p[11];
  const float mu = p[12];
  float es[16];
  es[0] = slip_E(c, mu, 1.0f, 0.0f);
  RSB_UNROLL for (int d = 1; d < 16; ++d) es[d] = slip_E(c, mu, kCos16[d], kSin16[d]);
  RSB_UNROLL for (int d = 0; d < 16; ++d) out_serial[16 * i + d] = es[d];
  // runtime directions: read from a table the compiler cannot see through
  for (int d = 0; d < 16; ++d) {
    float x = coef[13 * n + d], y = coef[13 * n + 16 + d];
    out_rt[16 * i + d] = slip_E(c, mu, x, y);
  }
}
int main() {
  const int n = 1 << 16;
  std::vector<float> h(13 * n + 32);
  srand(1);
  auto r = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (int i = 0; i < n; ++i) { float* p = &h[13 * i]; for (int j = 0; j < 12; ++j) p[j] = r(); p[0] = 1.f + r() * 0.5f; p[1] *= 0.3f; p[2] *= 0.3f; p[9] = -fabsf(p[9]); p[12] = 0.8f; }
  const float C[16] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.0f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f, 0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f};
  const float S[16] = {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f, 1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.0f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f};
  for (int d = 0; d < 16; ++d) { h[13 * n + d] = C[d]; h[13 * n + 16 + d] = S[d]; }
  float *dc, *d1, *d2;
  hipMalloc(&dc, h.size() * 4); hipMalloc(&d1, 16 * n * 4); hipMalloc(&d2, 16 * n * 4);
  hipMemcpy(dc, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dc, d1, d2, n);
  std::vector<float> a(16 * n), b(16 * n);
  hipMemcpy(a.data(), d1, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d2, b.size() * 4, hipMemcpyDeviceToHost);
  int bad[16] = {0}; int shown = 0;
  for (int i = 0; i < n; ++i) for (int d = 0; d < 16; ++d) if (memcmp(&a[16 * i + d], &b[16 * i + d], 4)) { ++bad[d]; if (shown++ < 6) printf("contact %d dir %d: literal %.9g runtime %.9g\n", i, d, a[16 * i + d], b[16 * i + d]); }
  printf("mismatches per direction:"); for (int d = 0; d < 16; ++d) printf(" %d", bad[d]); printf(" of %d\n", n);
}
