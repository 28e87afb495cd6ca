// bpermute_select_fold.hip - a select of ds_bpermute results, as the compiler of this image builds it, against the same with opaque (inline-asm) permutes.
// Lane l wants   y[l & 3][src(l)]   (one of FOUR registers, chosen by the DESTINATION lane, read from another lane) - the (env, unit) -> (input, env)
// transposition of the MLP stage (raisimlib_amd/csrc/rsb_pipeline.hip).  Written with the builtin the four permutes come out as ONE: the select is
// folded into the permute's data operand, i.e. evaluated in the SOURCE lane.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bsf tools/ubench/bpermute_select_fold.hip && /tmp/bsf
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int l = threadIdx.x, le = l & 3, src = 4 * ((l * 7 + 3) & 63);
  int y[4];
  for (int e = 0; e < 4; ++e) y[e] = 1000 * e + l;                       // register e, lane l
  int t[4];
  for (int e = 0; e < 4; ++e) t[e] = __builtin_amdgcn_ds_bpermute(src, y[e]);
  out[l] = le == 0 ? t[0] : le == 1 ? t[1] : le == 2 ? t[2] : t[3];      // the builtin
  int a0, a1, a2, a3;
  asm volatile("ds_bpermute_b32 %0, %4, %5\n\tds_bpermute_b32 %1, %4, %6\n\tds_bpermute_b32 %2, %4, %7\n\tds_bpermute_b32 %3, %4, %8\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(src), "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]));
  out[64 + l] = le == 0 ? a0 : le == 1 ? a1 : le == 2 ? a2 : a3;          // opaque
}
int main() {
  int* d; if (hipMalloc(&d, 128 * 4) != hipSuccess) return 1;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[128]; if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  int bad_builtin = 0, bad_asm = 0;
  for (int l = 0; l < 64; ++l) {
    const int want = 1000 * (l & 3) + ((l * 7 + 3) & 63);
    bad_builtin += h[l] != want; bad_asm += h[64 + l] != want;
  }
  std::printf("select of four ds_bpermute results: builtin %d of 64 lanes wrong, inline asm %d of 64 lanes wrong   (lane 1: want %d, builtin %d, asm %d)\n",
              bad_builtin, bad_asm, 1000 + 10, h[1], h[65]);
  return 0;
}
