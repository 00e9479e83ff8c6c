// membench.hip -- HBM streaming micro-benchmark that defines the "measured roofline" denominators of
// DESIGN.md / bench.py (BASELINE north star: "fraction of the measured HBM-read roofline").
// Streams far larger than the 256 MiB Infinity Cache with 16-byte accesses per lane:
//   read   : sum-reduce            copy : 1 read + 1 write        write : pure store
//   r1w2   : 1 read + 2 writes (the traversal kernel's steady-state mix when a child is forwarded)
// usage: membench [GiB per stream = 2] [reps = 20]   -> one JSON line
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_read(const double2 *__restrict__ a, size_t n, double *out)
{
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
  {
    const double2 v = a[i];
    acc += v.x + v.y;
  }
  if (acc == 1.2345e300) out[0] = acc; // never true; keeps the loads alive
}
__global__ __launch_bounds__(256) void k_write(double2 *__restrict__ a, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    a[i] = make_double2(1.0, 2.0);
}
__global__ __launch_bounds__(256) void k_copy(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_r1w2(const double2 *__restrict__ a, double2 *__restrict__ b, double2 *__restrict__ c, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
  {
    const double2 v = a[i];
    b[i] = v;
    c[i] = make_double2(v.y, v.x);
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
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main(int argc, char **argv)
{
  const double gib  = argc > 1 ? atof(argv[1]) : 2.0;
  const int    reps = argc > 2 ? atoi(argv[2]) : 20;
  const size_t bytes = (size_t)(gib * (1ull << 30)), n = bytes / sizeof(double2);
  double2 *a, *b, *c; double *out;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&out, 8));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int grid = prop.multiProcessorCount * 8;
  const double rd = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, out); }, reps);
  const double wr = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n); }, reps);
  const double cp = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); }, reps);
  const double mx = time_ms([&] { hipLaunchKernelGGL(k_r1w2, dim3(grid), dim3(256), 0, 0, a, b, c, n); }, reps);
  const double gb = bytes / 1e9;
  printf("{\"device\": \"%s\", \"cus\": %d, \"gib_per_stream\": %.2f, \"read_GBps\": %.1f, \"write_GBps\": %.1f, "
         "\"copy_GBps\": %.1f, \"r1w2_GBps\": %.1f}\n",
         prop.name, prop.multiProcessorCount, gib, gb / (rd * 1e-3), gb / (wr * 1e-3), 2 * gb / (cp * 1e-3), 3 * gb / (mx * 1e-3));
  return 0;
}
