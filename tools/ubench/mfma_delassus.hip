// Micro-benchmark (GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfd mfma_delassus.hip && /tmp/mfd):
// the ONE place of the step kernel with GEMM shape - the Delassus blocks G = W W^T of the Atlas-like class (kmax 16: W is
// 48 x 36, one row per contact axis, one column per velocity coordinate) - computed two ways by a LONE wave per SIMD (the
// kernel's regime), timed in shader cycles:
//   sparse  what the step kernel does: lane = contact pair, compact columns (6 base entries + one entry per support-chain
//           level, masked past the two chains' common prefix), 3 x 3 block per pair with VALU FMAs;
//   mfma    dense: the compact columns scattered into a zero-filled 48 x 36 (pitch 37) matrix in LDS, then
//           v_mfma_f32_16x16x4_f32 over the 6 tiles of the lower triangle (9 k-steps each), blocks written back to LDS.
// Both produce the same G (checked here against a host reference); nc = contacts of the env (8 = a standing humanoid on its
// eight foot spheres, 16 = the capacity).  north_star: "MFMA only for the small dense Delassus blocks".
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int KMAX = 16, NV = 36, ML = 10, CW = 16, PITCH = 37, GS = 4 * KMAX + 4;
typedef float v4f __attribute__((ext_vector_type(4)));

struct Problem {           // one env's contact set, as the column phase leaves it in LDS
  float wc[3 * KMAX][CW];  // compact columns: 6 base entries, then one per level of the contact body's support chain
  int chain[KMAX][ML];     // joint (velocity index 6..35) at each level of the contact's support chain, -1 past its end
  int nc;
};

__device__ __forceinline__ long long now() { long long t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

__global__ void __launch_bounds__(64) kern(const Problem* P, float* Gout_sparse, float* Gout_mfma, long long* cyc, int reps) {
  __shared__ __attribute__((aligned(16))) float WC[3 * KMAX * CW];
  __shared__ int CH[KMAX * ML];
  __shared__ __attribute__((aligned(16))) float G[3 * KMAX * GS];
  __shared__ __attribute__((aligned(16))) float WD[3 * KMAX * PITCH];
  const Problem& p = P[blockIdx.x];
  const int lane = threadIdx.x, nc = p.nc;
  for (int i = lane; i < 3 * KMAX * CW; i += 64) WC[i] = (&p.wc[0][0])[i];
  for (int i = lane; i < KMAX * ML; i += 64) CH[i] = (&p.chain[0][0])[i];
  __syncthreads();
  long long t_sparse = 0, t_mfma = 0;
  for (int rep = 0; rep < reps; ++rep) {
    // ---------------- sparse (the step kernel's formulation)
    long long t0 = now();
    const int npw = nc * (nc + 1) / 2;
    for (int p0 = 0; p0 < npw; p0 += 64) {
      const int pr = p0 + lane;
      int j = 0, rem = pr;
      while (rem > j) { rem -= j + 1; ++j; }
      const int i = rem;
      if (j < nc) {
        int lca = 0; bool same = true;
#pragma unroll
        for (int l = 0; l < ML; ++l) { const int a = CH[i * ML + l], b = CH[j * ML + l]; same = same && (a >= 0) && (a == b); lca += same ? 1 : 0; }
        float acc[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) acc[q] = 0.f;
        float wi[3][CW], wj[3][CW];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
          for (int q4 = 0; q4 < CW / 4; ++q4) {
            const v4f a = *reinterpret_cast<const v4f*>(&WC[(3 * i + rr) * CW + 4 * q4]), b = *reinterpret_cast<const v4f*>(&WC[(3 * j + rr) * CW + 4 * q4]);
            wi[rr][4 * q4] = a.x; wi[rr][4 * q4 + 1] = a.y; wi[rr][4 * q4 + 2] = a.z; wi[rr][4 * q4 + 3] = a.w;
            wj[rr][4 * q4] = b.x; wj[rr][4 * q4 + 1] = b.y; wj[rr][4 * q4 + 2] = b.z; wj[rr][4 * q4 + 3] = b.w;
          }
#pragma unroll
        for (int e = 0; e < CW; ++e) {
          const bool on = e < 6 + lca;
#pragma unroll
          for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) acc[3 * rr + cc] += (on ? wi[rr][e] : 0.f) * (on ? wj[cc][e] : 0.f);
        }
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) { G[(3 * i + rr) * GS + 4 * j + cc] = acc[3 * rr + cc]; G[(3 * j + cc) * GS + 4 * i + rr] = acc[3 * rr + cc]; }
      }
    }
    __syncthreads();
    t_sparse += now() - t0;
    if (rep == 0) for (int r = lane; r < 3 * nc; r += 64) for (int c = 0; c < 3 * nc; ++c) Gout_sparse[(size_t)blockIdx.x * 48 * 48 + r * 48 + c] = G[r * GS + 4 * (c / 3) + c % 3];
    __syncthreads();
    // ---------------- dense + MFMA
    t0 = now();
    for (int i = lane; i < 3 * KMAX * PITCH; i += 64) WD[i] = 0.f;
    __syncthreads();
    for (int e0 = 0; e0 < 3 * nc * CW; e0 += 64) {       // scatter: (row, compact entry) -> velocity coordinate
      const int e = e0 + lane;
      if (e < 3 * nc * CW) {
        const int row = e / CW, k = e - row * CW, con = row / 3;
        const int col = k < 6 ? k : CH[con * ML + (k - 6)];
        if (col >= 0) WD[row * PITCH + col] = WC[row * CW + k];
      }
    }
    __syncthreads();
    const int nt = (3 * nc + 15) / 16;                   // 16-row tiles in use (1..3)
    v4f acc[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) acc[q] = (v4f){0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, k4 = lane >> 4;
#pragma unroll
    for (int kb = 0; kb < NV / 4; ++kb) {
      float a[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) a[t] = WD[(16 * t + r16) * PITCH + 4 * kb + k4];   // A operand of row block t == B operand of column block t
      // lower-triangle tiles (I, J), J <= I:  0:(0,0) 1:(1,0) 2:(1,1) 3:(2,0) 4:(2,1) 5:(2,2)
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], a[0], acc[0], 0, 0, 0);
      if (nt > 1) {
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], a[0], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], a[1], acc[2], 0, 0, 0);
      }
      if (nt > 2) {
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], a[0], acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], a[1], acc[4], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], a[2], acc[5], 0, 0, 0);
      }
    }
    // D layout: lane holds D[4 * (lane / 16) + v][lane % 16], v = 0..3; written as G[row][4 * (col / 3) + col % 3] and mirrored
    const int TI[6] = {0, 1, 1, 2, 2, 2}, TJ[6] = {0, 0, 1, 0, 1, 2};
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      if (TI[q] < nt) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = 16 * TI[q] + 4 * k4 + v, col = 16 * TJ[q] + r16;
          const float val = acc[q][v];
          G[row * GS + 4 * (col / 3) + col % 3] = val;
          if (TI[q] != TJ[q]) G[col * GS + 4 * (row / 3) + row % 3] = val;
        }
      }
    }
    __syncthreads();
    t_mfma += now() - t0;
    if (rep == 0) for (int r = lane; r < 3 * nc; r += 64) for (int c = 0; c < 3 * nc; ++c) Gout_mfma[(size_t)blockIdx.x * 48 * 48 + r * 48 + c] = G[r * GS + 4 * (c / 3) + c % 3];
    __syncthreads();
  }
  if (lane == 0) { cyc[2 * blockIdx.x] = t_sparse / reps; cyc[2 * blockIdx.x + 1] = t_mfma / reps; }
}

int main() {
  const int B = 1024, reps = 20;     // one wave per SIMD of the chip
  for (int nc : {4, 8, 12, 16}) {
    std::vector<Problem> P(B);
    srand(7 + nc);
    for (auto& p : P) {
      p.nc = nc;
      for (int c = 0; c < KMAX; ++c) {
        // humanoid-like support chains: leg chains of 6 joints (two legs: 6..11, 12..17), arm / torso chains up to 10 (18..27)
        const int limb = c < nc ? (c / 4) % 3 : 0, len = limb < 2 ? 6 : 10, base = 6 + (limb < 2 ? 6 * limb : 12);
        for (int l = 0; l < ML; ++l) p.chain[c][l] = l < len ? base + l : -1;
        for (int rr = 0; rr < 3; ++rr)
          for (int k = 0; k < CW; ++k) p.wc[3 * c + rr][k] = (c < nc && k < 6 + len) ? (float)(rand() % 2001 - 1000) * 1e-3f : 0.f;
      }
    }
    Problem* dP; float *dGs, *dGm; long long* dC;
    hipMalloc(&dP, B * sizeof(Problem)); hipMalloc(&dGs, (size_t)B * 48 * 48 * 4); hipMalloc(&dGm, (size_t)B * 48 * 48 * 4); hipMalloc(&dC, 2 * B * sizeof(long long));
    hipMemset(dGs, 0, (size_t)B * 48 * 48 * 4); hipMemset(dGm, 0, (size_t)B * 48 * 48 * 4);
    hipMemcpy(dP, P.data(), B * sizeof(Problem), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kern, dim3(B), dim3(64), 0, 0, dP, dGs, dGm, dC, reps);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<float> Gs((size_t)B * 48 * 48), Gm((size_t)B * 48 * 48);
    std::vector<long long> C(2 * B);
    hipMemcpy(Gs.data(), dGs, Gs.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(Gm.data(), dGm, Gm.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost);
    double es = 0, em = 0;
    for (int b = 0; b < 8; ++b) {        // host reference in full velocity coordinates
      const Problem& p = P[b];
      std::vector<double> W(48 * NV, 0.0);
      for (int r = 0; r < 3 * nc; ++r) for (int k = 0; k < CW; ++k) { const int col = k < 6 ? k : p.chain[r / 3][k - 6]; if (col >= 0) W[r * NV + col] = p.wc[r][k]; }
      for (int r = 0; r < 3 * nc; ++r) for (int c = 0; c < 3 * nc; ++c) {
        double g = 0; for (int k = 0; k < NV; ++k) g += W[r * NV + k] * W[c * NV + k];
        es = fmax(es, fabs(g - Gs[(size_t)b * 48 * 48 + r * 48 + c])); em = fmax(em, fabs(g - Gm[(size_t)b * 48 * 48 + r * 48 + c]));
      }
    }
    double cs = 0, cm = 0; for (int b = 0; b < B; ++b) { cs += C[2 * b]; cm += C[2 * b + 1]; }
    printf("nc %2d (W %2d x %d): sparse VALU %6.0f cycles, dense MFMA (zero-fill + scatter + %2d v_mfma_f32_16x16x4_f32 + write-back) %6.0f cycles per env; max |G - ref| sparse %.1e mfma %.1e\n",
           nc, 3 * nc, NV, cs / B, cm / B, 9 * (((3 * nc + 15) / 16) * (((3 * nc + 15) / 16) + 1) / 2), es, em);
    hipFree(dP); hipFree(dGs); hipFree(dGm); hipFree(dC);
  }
  return 0;
}
