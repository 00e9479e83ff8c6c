// Probe: sustained rate of v_mfma_f64_4x4x4_4b_f64 (and 16x16x4) on gfx950 per SIMD, alone and with the LDS reads of
// the 20-state kernel's matrix phase (30 reads per 50 MFMAs), at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
typedef double v4d __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(1024) void k(double *out, long long *cyc, int iters)
{
  __shared__ __attribute__((aligned(16))) double A[2 * 1600];
  for (int i = threadIdx.x; i < 3200; i += blockDim.x) A[i] = 1.0 / (1 + i);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  double x1[5], x2[5];
  for (int t = 0; t < 5; ++t) { x1[t] = 1.0 + lane * 1e-3 + t; x2[t] = 2.0 - lane * 1e-3 + t; }
  double acc = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it)
  {
    if (MODE == 2)
    { // 16x16x4: 10 MFMAs of 64 cycles + 10 of 4x4x4 (the first generation's mix)
      v4d lo1 = {0, 0, 0, 0}, lo2 = {0, 0, 0, 0};
      double h1 = 0, h2 = 0;
#pragma unroll
      for (int t = 0; t < 5; ++t)
      {
        lo1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x2[t], x1[t], lo1, 0, 0, 0);
        lo2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[t], x2[t], lo2, 0, 0, 0);
        h1  = __builtin_amdgcn_mfma_f64_4x4x4f64(x2[t], x1[t], h1, 0, 0, 0);
        h2  = __builtin_amdgcn_mfma_f64_4x4x4f64(x1[t], x2[t], h2, 0, 0, 0);
      }
      acc += lo1[0] + lo2[1] + h1 + h2;
    }
    else
    {
      double u1[5] = {0, 0, 0, 0, 0}, u2[5] = {0, 0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < 5; ++t)
      {
        double a1[5], a2[5];
        if (MODE == 1)
        {
          const v2d *p1 = reinterpret_cast<const v2d *>(A + t * 320) + lane, *p2 = reinterpret_cast<const v2d *>(A + 1600 + t * 320) + lane;
          v2d q;
          q = p1[0]; a1[0] = q.x; a1[1] = q.y; q = p1[64]; a1[2] = q.x; a1[3] = q.y; a1[4] = A[t * 320 + 256 + lane];
          q = p2[0]; a2[0] = q.x; a2[1] = q.y; q = p2[64]; a2[2] = q.x; a2[3] = q.y; a2[4] = A[1600 + t * 320 + 256 + lane];
        }
        else
          for (int r = 0; r < 5; ++r) { a1[r] = x2[r]; a2[r] = x1[r]; }
#pragma unroll
        for (int r = 0; r < 5; ++r)
        {
          u1[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[r], x1[t], u1[r], 0, 0, 0);
          u2[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2[r], x2[t], u2[r], 0, 0, 0);
        }
      }
      for (int r = 0; r < 5; ++r) acc += u1[r] * u2[r];
      asm volatile("" ::: "memory");
    }
    x1[it % 5] += 1e-9; // keep the loop from being hoisted
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
int main()
{
  double *out; long long *cyc;
  hipMalloc(&out, 256 * 1024 * 8); hipMallocManaged(&cyc, 256 * 16 * 8);
  const int iters = 2000;
  for (int mode = 0; mode < 3; ++mode)
    for (int wpc : {4, 8, 12, 16})
    {
      for (int rep = 0; rep < 2; ++rep)
      {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * wpc), 0, 0, out, cyc, iters);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * wpc), 0, 0, out, cyc, iters);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * wpc), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
      }
      double mean = 0;
      for (int i = 0; i < 256 * wpc; ++i) mean += (double)cyc[i];
      mean /= 256.0 * wpc;
      const double per_group = mean / iters; // cycles a wave needs per group of 50 (mode 2: 20) MFMAs
      printf("mode %d (%s) waves/CU %2d: %.0f cycles per wave-group, %.0f per SIMD-group (ideal %d)\n", mode,
             mode == 0 ? "4x4x4 regs" : mode == 1 ? "4x4x4 + LDS A" : "16x16x4+4x4x4", wpc, per_group, per_group / (wpc / 4.0), 800);
    }
  return 0;
}
