// Known-byte-count kernels in the step kernel's own access patterns, to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on
// gfx950 (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern before trusting an absolute").
//   rows_read  : [N, dim] row-major float32, a group of 16 lanes reads row e (4 B per lane, consecutive lanes = consecutive
//                floats) - how the step kernel reads gc / gv / targets
//   rows_write : the same pattern for stores (gc / gv / obs out)
//   rec_rw     : 32-B records, lane = record (warm state in / out, 2 x float4 per lane)
//   wide_read  : 16 B per lane fully coalesced stream (the guide's reference pattern: FETCH_SIZE reads 1/2 of it)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench/traffic_calib tools/ubench/traffic_calib.hip
// run  : rocprofv3 --pmc FETCH_SIZE -- ./traffic_calib ; rocprofv3 --pmc WRITE_SIZE -- ./traffic_calib   (separate passes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(64) rows_read(const float* in, float* sink, int N, int dim) {
  const int lane = threadIdx.x, el = lane >> 4, s = lane & 15;
  const int env = blockIdx.x * 4 + el;
  float acc = 0.f;
  if (env < N) for (int i = s; i < dim; i += 16) acc += in[(size_t)env * dim + i];
  if (acc == 123.456f) sink[0] = acc;
}
__global__ void __launch_bounds__(64) rows_write(float* out, int N, int dim) {
  const int lane = threadIdx.x, el = lane >> 4, s = lane & 15;
  const int env = blockIdx.x * 4 + el;
  if (env < N) for (int i = s; i < dim; i += 16) out[(size_t)env * dim + i] = (float)i;
}
__global__ void __launch_bounds__(64) rec_rw(const float4* in, float4* out, int N, int nrec) {
  const int lane = threadIdx.x, el = lane >> 4, s = lane & 15;
  const int env = blockIdx.x * 4 + el;
  if (env < N && s < nrec) {
    const size_t o = ((size_t)env * 16 + s) * 2;
    float4 a = in[o], b = in[o + 1];
    a.x += 1.f; b.y += 1.f;
    out[o] = a; out[o + 1] = b;
  }
}
__global__ void __launch_bounds__(256) wide_read(const float4* in, float* sink, size_t n4) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 123.456f) sink[0] = acc;
}

int main() {
  const int N = 1 << 20;            // 1 Mi rows (256 x the benchmark's 4096 envs: well past the 4 MB L2 per XCD)
  float *a, *b, *sink;
  const size_t bytes = (size_t)N * 128 * sizeof(float);
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 64);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  hipDeviceSynchronize();
  const int blocks = N / 4;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(rows_read, dim3(blocks), dim3(64), 0, 0, a, sink, N, 19);
    hipLaunchKernelGGL(rows_read, dim3(blocks), dim3(64), 0, 0, a, sink, N, 55);
    hipLaunchKernelGGL(rows_write, dim3(blocks), dim3(64), 0, 0, b, N, 19);
    hipLaunchKernelGGL(rows_write, dim3(blocks), dim3(64), 0, 0, b, N, 49);
    hipLaunchKernelGGL(rec_rw, dim3(blocks), dim3(64), 0, 0, (const float4*)a, (float4*)b, N, 8);
    hipLaunchKernelGGL(wide_read, dim3(4096), dim3(256), 0, 0, (const float4*)a, sink, bytes / 16);
  }
  hipDeviceSynchronize();
  std::printf("known bytes per dispatch: rows_read dim19 %.3f MB, dim55 %.3f MB | rows_write dim19 %.3f MB, dim49 %.3f MB | rec_rw read %.3f MB + written %.3f MB | wide_read %.3f MB\n",
              N * 19 * 4 / 1e6, N * 55 * 4 / 1e6, N * 19 * 4 / 1e6, N * 49 * 4 / 1e6, N * 8 * 32 / 1e6, N * 8 * 32 / 1e6, bytes / 1e6);
  return 0;
}
