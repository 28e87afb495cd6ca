// Which XCD runs workgroup b?  Two dispatches of 1024 one-wave workgroups on two streams, overlapping in time (every workgroup spins ~40 us;
// 128 KB of LDS per workgroup keeps it to one workgroup per CU so that the second dispatch has to wait for slots of the first).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void rec(int* xcc, uint64_t cycles) {
  extern __shared__ float lds[];
  const uint64_t t0 = wall_clock64();
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (threadIdx.x == 0) { xcc[2 * blockIdx.x] = (int)(id & 0xf); xcc[2 * blockIdx.x + 1] = (int)hw; lds[0] = 1.f; }
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}
int main() {
  const int G = 1024;
  int *d0, *d1; hipMalloc(&d0, G * 8); hipMalloc(&d1, G * 8);
  hipStream_t s0, s1; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipFuncSetAttribute(reinterpret_cast<const void*>(rec), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(rec, dim3(G), dim3(64), 128 * 1024, s0, d0, (uint64_t)4000);
    hipLaunchKernelGGL(rec, dim3(G), dim3(64), 128 * 1024, s1, d1, (uint64_t)4000);
    hipDeviceSynchronize();
    std::vector<int> h0(2 * G), h1(2 * G);
    hipMemcpy(h0.data(), d0, G * 8, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), d1, G * 8, hipMemcpyDeviceToHost);
    int rr0 = 0, rr1 = 0, same = 0, hist[16] = {0};
    for (int b = 0; b < G; ++b) { rr0 += h0[2 * b] == (b & 7); rr1 += h1[2 * b] == (b & 7); same += h0[2 * b] == h1[2 * b]; hist[h0[2 * b] & 15]++; }
    printf("rep %d: dispatch A round-robin %d / %d, dispatch B round-robin %d / %d, same XCD for block b in A and B %d / %d; XCD histogram A:", rep, rr0, G, rr1, G, same, G);
    for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
    printf("; first blocks A:");
    for (int b = 0; b < 12; ++b) printf(" %d", h0[2 * b]);
    printf("\n");
  }
  return 0;
}
