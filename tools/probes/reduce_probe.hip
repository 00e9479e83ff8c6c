// tools/probes/reduce_probe.hip -- the wave sum of the evaluation kernels (lane 0 <- the 64 lanes, tree of offsets 32, 16, 8, 4, 2, 1) through
// ds_bpermute (__shfl_down) against the same tree on v_permlane32_swap / v_permlane16_swap / DPP row_shl: bit-equality of lane 0's sum on
// random doubles, and the latency of a dependent chain of each.  build: hipcc --offload-arch=gfx950 -O3 -o phyml_amd/lib/reduce_probe tools/probes/reduce_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
template <int CTRL> __device__ __forceinline__ double mov_dpp(double v)
{
  unsigned long long b; __builtin_memcpy(&b, &v, 8);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xF, 0xF, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, false);
  b = ((unsigned long long)hi << 32) | lo; double r; __builtin_memcpy(&r, &b, 8); return r;
}
__device__ __forceinline__ double from_plus32(double v) // lane i < 32: lane i + 32's value
{
  unsigned long long b; __builtin_memcpy(&b, &v, 8);
  const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
  const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  b = ((unsigned long long)(unsigned)r1[1] << 32) | (unsigned)r0[1]; double r; __builtin_memcpy(&r, &b, 8); return r;
}
__device__ __forceinline__ double from_plus16(double v) // lane i of rows 0 / 2: lane i + 16's value
{
  unsigned long long b; __builtin_memcpy(&b, &v, 8);
  const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
  const auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  b = ((unsigned long long)(unsigned)r1[1] << 32) | (unsigned)r0[1]; double r; __builtin_memcpy(&r, &b, 8); return r;
}
__device__ __forceinline__ double sum_new(double t)
{
  t += from_plus32(t);
  t += from_plus16(t);
  t += mov_dpp<0x108>(t); // row_shl:8
  t += mov_dpp<0x104>(t);
  t += mov_dpp<0x102>(t);
  t += mov_dpp<0x101>(t);
  return t;
}
__device__ __forceinline__ double sum_old(double t)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
  return t;
}
__global__ void check(const double *in, unsigned long long *bad, int n)
{
  for (int i = blockIdx.x; i < n; i += gridDim.x)
  {
    const double v = in[(size_t)i * 64 + threadIdx.x];
    const double a = sum_old(v), b = sum_new(v);
    unsigned long long x, y; __builtin_memcpy(&x, &a, 8); __builtin_memcpy(&y, &b, 8);
    if (threadIdx.x == 0 && x != y) atomicAdd(bad, 1ull);
  }
}
template <bool NEW> __global__ void chain(double *io, int reps, unsigned long long *ticks)
{
  double v = io[threadIdx.x];
  const unsigned long long t0 = wall_clock64();
  for (int r = 0; r < reps; ++r)
  {
    v = NEW ? sum_new(v) : sum_old(v);
    v = __shfl(v, 0, 64) * 1e-3 + (double)threadIdx.x; // (lane 0's sum feeds everybody: a dependent chain)
  }
  const unsigned long long t1 = wall_clock64();
  io[threadIdx.x] = v;
  if (threadIdx.x == 0) *ticks = t1 - t0;
}
int main()
{
  const int n = 200000;
  std::vector<double> h((size_t)n * 64);
  std::mt19937_64 g(7);
  for (auto &x : h) { const double m = (double)(g() >> 11) / 9007199254740992.0; x = (m - 0.5) * std::ldexp(1.0, (int)(g() % 80) - 40); }
  double *d; unsigned long long *bad, *ticks; hipMalloc(&d, h.size() * 8); hipMalloc(&bad, 8); hipMalloc(&ticks, 8);
  hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemset(bad, 0, 8);
  hipLaunchKernelGGL(check, dim3(1024), dim3(64), 0, 0, d, bad, n);
  unsigned long long nb = 0; hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost);
  printf("lane 0's sum: %d random vectors, %llu differ between the two trees\n", n, nb);
  int khz = 100000; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  const int reps = 20000;
  for (int k = 0; k < 2; ++k)
  {
    if (k) hipLaunchKernelGGL(chain<true>, dim3(1), dim3(64), 0, 0, d, reps, ticks); else hipLaunchKernelGGL(chain<false>, dim3(1), dim3(64), 0, 0, d, reps, ticks);
    unsigned long long t = 0; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    printf("%s: %.0f ns per reduction (+ one broadcast) in a dependent chain\n", k ? "permlane swaps + DPP" : "__shfl_down (ds_bpermute)", (double)t / reps * 1e9 / (khz * 1e3));
  }
  return nb != 0;
}
