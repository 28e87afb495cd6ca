// Does hipExtAnyOrderLaunch let the next kernel of a stream start before the previous one has finished on gfx950?
// Kernel A: one wave spins for ~2 ms; kernel B: one wave stamps the time.  Launched A, B on one stream.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
__global__ void spin(uint64_t* out, uint64_t cycles) {
  const uint64_t t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { out[0] = t0; out[1] = wall_clock64(); }
}
__global__ void stamp(uint64_t* out) { if (threadIdx.x == 0) out[2] = wall_clock64(); }
int main() {
  uint64_t* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int flags = 0; flags < 2; ++flags) {
    for (int rep = 0; rep < 3; ++rep) {
      hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, d, (uint64_t)200000);   // 100 MHz wall clock: 2 ms
      hipExtLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s, nullptr, nullptr, flags, d);
      hipStreamSynchronize(s);
      uint64_t h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      printf("flags %d: spin %.1f us, stamp kernel ran %.1f us after the spin kernel started (%s)\n", flags, (h[1] - h[0]) / 100.0, ((double)h[2] - (double)h[0]) / 100.0,
             h[2] < h[1] ? "OVERLAPPED" : "after it ended");
    }
  }
  return 0;
}
