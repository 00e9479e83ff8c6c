// tools/probes/l2_reuse_probe.hip -- does a persistent kernel that re-reads ITS OWN slice of a buffer find it in its XCD's L2?
// 256 workgroups x 12 waves stay resident; pass after pass every wave reads the same 4 KB (64 lanes x 4 x 16 B) of a 12.8 MB
// buffer (a dLk command's dot_prod traffic at 100 000 patterns), grid-synchronised by a counter so that every pass is the burst a
// command is.  Prints the time per pass for plain loads, for a footprint that cannot fit (x 16) and for sc1 loads.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_reuse_probe tools/probes/l2_reuse_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double2_ __attribute__((ext_vector_type(2)));
__global__ void read_clock(unsigned long long *o) { *o = wall_clock64(); }
__global__ __launch_bounds__(768) void probe(const double2_ *buf, size_t stride_tiles, int passes, int mode, unsigned long long start, unsigned long long *out, double *sink)
{
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const size_t tile = ((size_t)wid * gridDim.x + blockIdx.x) * stride_tiles;
  const double2_ *p = buf + tile * 256 + lane;
  double acc = 0.0;
  unsigned long long t0 = 0, tl = 0;
  __shared__ unsigned long long sh_tl[12];
  for (int it = 0; it < passes + 4; ++it)
  {
    // everybody starts the pass at the same tick of the device-wide real-time counter (a command arrives everywhere at once)
    __syncthreads();
    while (wall_clock64() < start + (unsigned long long)(it + 1) * 1000ull) {}
    __syncthreads();
    if (it == 4) t0 = wall_clock64();
    const unsigned long long t1 = wall_clock64();
    double2_ v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
      if (mode == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[j]) : "v"(p + j * 64) : "memory");
      else v[j] = p[j * 64];
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : : "memory");
    const unsigned long long t2 = wall_clock64();
    if (it >= 4) tl += t2 - t1;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc += v[j].x + v[j].y;
    asm volatile("" : "+v"(acc));
  }
  __syncthreads();
  if (lane == 0) sh_tl[wid] = tl;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    unsigned long long m = 0, sm = 0;
    for (int w = 0; w < 12; ++w) { m = sh_tl[w] > m ? sh_tl[w] : m; sm += sh_tl[w]; }
    out[blockIdx.x] = wall_clock64() - t0; out[gridDim.x + blockIdx.x] = m; out[2 * gridDim.x + blockIdx.x] = sm / 12;
  }
  if (acc == 12345.678) sink[0] = acc;
}
int main()
{
  const int wgs = 256, passes = 200;
  const size_t tiles = (size_t)wgs * 12;
  double2_ *buf; unsigned *ctr; unsigned long long *out; double *sink;
  const size_t big = tiles * 16 * 4096; // x16 footprint: 201 MB
  hipMalloc(&buf, big); hipMemset(buf, 0, big);
  hipMalloc(&ctr, 4); hipMalloc(&out, 8 * wgs * 3); hipMalloc(&sink, 8);
  int khz = 100000; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  const char *names[3] = {"plain loads, 12.6 MB (4 KB per wave, own slice)", "plain loads, same addresses, stride x16 (201 MB span)", "sc1 loads, 12.6 MB footprint"};
  for (int mode = 0; mode < 3; ++mode)
  {
    // (start: a little in the future of the device clock -- read by a one-thread kernel)
    unsigned long long *now_d, now_h = 0; hipMalloc(&now_d, 8);
    hipLaunchKernelGGL(read_clock, dim3(1), dim3(1), 0, 0, now_d); hipMemcpy(&now_h, now_d, 8, hipMemcpyDeviceToHost); hipFree(now_d);
    const unsigned long long start = now_h + 200000ull; // 2 ms
    hipLaunchKernelGGL(probe, dim3(wgs), dim3(768), 0, 0, buf, mode == 1 ? (size_t)16 : (size_t)1, passes, mode, start, out, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(wgs * 3);
    hipMemcpy(h.data(), out, 8 * wgs * 3, hipMemcpyDeviceToHost);
    double mx = 0, ml = 0, al = 0;
    for (int i = 0; i < wgs; ++i) { mx = h[i] > mx ? h[i] : mx; ml = h[wgs + i] > ml ? h[wgs + i] : ml; al += h[2 * wgs + i]; }
    const double us = 1e6 / (khz * 1e3) / passes;
    printf("%-70s pass every 10 us (%.2f); loads issued -> returned: slowest wave %.2f us, mean %.2f us\n", names[mode], mx * us, ml * us, al / wgs * us);
  }
  // the barrier alone
  return 0;
}
