// ext_launch_events.hip - what a kernel-duration bracket costs: hipEventRecord pairs around a launch against the start / stop events of
// hipExtLaunchKernelGGL (bound to the dispatch packet itself: no barrier packets of their own).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/ext_launch_events tools/ubench/ext_launch_events.hip && /tmp/ext_launch_events
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(long long ticks, int* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}
int main() {
  int* d; CK(hipMalloc(&d, 64));
  hipStream_t s[2]; for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
  const int K = 20; const long long ticks = 5000;   // 50 us at 100 MHz
  std::vector<hipEvent_t> e0(K), e1(K);
  for (int i = 0; i < K; ++i) { CK(hipEventCreate(&e0[i])); CK(hipEventCreate(&e1[i])); }
  auto run = [&](int mode, double* total_us, double* mean_ms) -> int {
    std::vector<double> tot;
    for (int rep = 0; rep < 9; ++rep) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < K; ++i) {
        hipStream_t st = s[i & 1];
        if (mode == 1) CK(hipEventRecord(e0[i], st));
        if (mode == 2) hipExtLaunchKernelGGL(spin, dim3(1024), dim3(64), 0, st, e0[i], e1[i], 0, ticks, d);
        else hipLaunchKernelGGL(spin, dim3(1024), dim3(64), 0, st, ticks, d);
        if (mode == 1) CK(hipEventRecord(e1[i], st));
      }
      CK(hipDeviceSynchronize());
      tot.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(tot.begin(), tot.end());
    *total_us = tot[4];
    *mean_ms = 0;
    if (mode) { for (int i = 0; i < K; ++i) { float ms = 0; CK(hipEventElapsedTime(&ms, e0[i], e1[i])); *mean_ms += ms / K; } }
    return 0;
  };
  const char* name[3] = {"no events", "hipEventRecord pairs", "hipExtLaunchKernelGGL start/stop"};
  for (int mode = 0; mode < 3; ++mode) {
    double us, ms;
    if (run(mode, &us, &ms)) return 1;
    std::printf("%-34s 20 launches of a 50-us kernel on two alternating streams: %8.1f us (median of 9)   mean event-reported duration %.4f ms\n", name[mode], us, ms);
  }
  return 0;
}
