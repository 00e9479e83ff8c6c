// membench.hip -- HBM streaming micro-benchmark that defines the "measured roofline" denominators of
// DESIGN.md / bench.py (BASELINE north star: "fraction of the measured HBM-read roofline").
// Streams far larger than the 256 MiB Infinity Cache with 16-byte accesses per lane:
//   read   : sum-reduce            copy : 1 read + 1 write        write : pure store
//   r1w2   : 1 read + 2 writes (the traversal kernel's steady-state mix when a child is forwarded)
// Each in two cache policies -- plain, and NON-TEMPORAL loads / stores (what the traversal kernels use since round 3:
// PHYHIP_STORE_AUX / PHYHIP_LOAD_AUX = 2) -- and, with "sweep", over the launch geometry: 1..16 workgroups per CU of 256 or 64
// lanes.  The headline keys (read_GBps, write_GBps, copy_GBps, r1w2_GBps) stay what they were in rounds 1-3 (plain, 8 x 256 per
// CU); *_nt_GBps are the non-temporal rates in the same geometry, best_* the best geometry of the sweep.
// usage: membench [GiB per stream = 2] [reps = 20] [sweep]   -> one JSON line
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ d2 ld(const d2 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(d2 *p, d2 v)
{
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

template <bool NT> __global__ void k_read(const d2 *__restrict__ a, size_t n, double *out)
{
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
  {
    const d2 v = ld<NT>(a + i);
    acc += v.x + v.y;
  }
  if (acc == 1.2345e300) out[0] = acc; // never true; keeps the loads alive
}
template <bool NT> __global__ void k_write(d2 *__restrict__ a, size_t n)
{
  const d2 v = {1.0, 2.0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) st<NT>(a + i, v);
}
template <bool NT> __global__ void k_copy(const d2 *__restrict__ a, d2 *__restrict__ b, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) st<NT>(b + i, ld<NT>(a + i));
}
template <bool NT> __global__ void k_r1w2(const d2 *__restrict__ a, d2 *__restrict__ b, d2 *__restrict__ c, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
  {
    const d2 v = ld<NT>(a + i);
    st<NT>(b + i, v);
    const d2 w = {v.y, v.x};
    st<NT>(c + i, w);
  }
}

template <typename F> static double time_ms(F &&launch, int reps)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); launch();
  CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int r = 0; r < reps; ++r)
  {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    t.push_back(ms);
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

struct Rates { double rd, wr, cp, mx; };

int main(int argc, char **argv)
{
  const double gib  = argc > 1 ? atof(argv[1]) : 2.0;
  const int    reps = argc > 2 ? atoi(argv[2]) : 20;
  const bool   sweep = argc > 3 && !strcmp(argv[3], "sweep");
  const size_t bytes = (size_t)(gib * (1ull << 30)), n = bytes / sizeof(d2);
  d2 *a, *b, *c; double *out;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&out, 8));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const double gb = bytes / 1e9;
  auto run = [&](bool nt, int per_cu, int lanes) {
    const int grid = prop.multiProcessorCount * per_cu;
    Rates r;
    if (nt)
    {
      r.rd = time_ms([&] { hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(lanes), 0, 0, a, n, out); }, reps);
      r.wr = time_ms([&] { hipLaunchKernelGGL(k_write<true>, dim3(grid), dim3(lanes), 0, 0, b, n); }, reps);
      r.cp = time_ms([&] { hipLaunchKernelGGL(k_copy<true>, dim3(grid), dim3(lanes), 0, 0, a, b, n); }, reps);
      r.mx = time_ms([&] { hipLaunchKernelGGL(k_r1w2<true>, dim3(grid), dim3(lanes), 0, 0, a, b, c, n); }, reps);
    }
    else
    {
      r.rd = time_ms([&] { hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(lanes), 0, 0, a, n, out); }, reps);
      r.wr = time_ms([&] { hipLaunchKernelGGL(k_write<false>, dim3(grid), dim3(lanes), 0, 0, b, n); }, reps);
      r.cp = time_ms([&] { hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(lanes), 0, 0, a, b, n); }, reps);
      r.mx = time_ms([&] { hipLaunchKernelGGL(k_r1w2<false>, dim3(grid), dim3(lanes), 0, 0, a, b, c, n); }, reps);
    }
    r.rd = gb / (r.rd * 1e-3); r.wr = gb / (r.wr * 1e-3); r.cp = 2 * gb / (r.cp * 1e-3); r.mx = 3 * gb / (r.mx * 1e-3);
    return r;
  };
  const Rates plain = run(false, 8, 256), nt = run(true, 8, 256);
  printf("{\"device\": \"%s\", \"cus\": %d, \"gib_per_stream\": %.2f, \"read_GBps\": %.1f, \"write_GBps\": %.1f, "
         "\"copy_GBps\": %.1f, \"r1w2_GBps\": %.1f, \"read_nt_GBps\": %.1f, \"write_nt_GBps\": %.1f, \"copy_nt_GBps\": %.1f, "
         "\"r1w2_nt_GBps\": %.1f",
         prop.name, prop.multiProcessorCount, gib, plain.rd, plain.wr, plain.cp, plain.mx, nt.rd, nt.wr, nt.cp, nt.mx);
  if (sweep)
  {
    Rates best = {0, 0, 0, 0};
    std::string rows;
    for (int lanes : {256, 64})
      for (int per_cu : {1, 2, 4, 8, 16, 32})
        for (int policy = 0; policy < 2; ++policy)
        {
          if (lanes == 256 && per_cu == 32) continue;
          const Rates r = run(policy != 0, per_cu, lanes);
          best.rd = std::max(best.rd, r.rd); best.wr = std::max(best.wr, r.wr); best.cp = std::max(best.cp, r.cp); best.mx = std::max(best.mx, r.mx);
          char buf[256];
          snprintf(buf, sizeof buf, "%s{\"lanes\": %d, \"wg_per_cu\": %d, \"nt\": %d, \"read\": %.0f, \"write\": %.0f, \"copy\": %.0f, \"r1w2\": %.0f}",
                   rows.empty() ? "" : ", ", lanes, per_cu, policy, r.rd, r.wr, r.cp, r.mx);
          rows += buf;
        }
    printf(", \"best_read_GBps\": %.1f, \"best_write_GBps\": %.1f, \"best_copy_GBps\": %.1f, \"best_r1w2_GBps\": %.1f, \"sweep\": [%s]",
           best.rd, best.wr, best.cp, best.mx, rows.c_str());
  }
  printf("}\n");
  return 0;
}
