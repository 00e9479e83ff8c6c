// How fast does a workgroup that stays on the device see a word the host changes in host-mapped memory, and how fast does
// its answer come back?  (Round trip of the resident evaluator's command path, without any work in between.)
// Also: does a kernel launched on a second stream start while the first one is still polling?
// build: hipcc --offload-arch=gfx950 -O3 -o resident_probe resident_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Cmd { unsigned long long seq, gen; unsigned long long ack[64]; unsigned long long polls[64]; };
__global__ void poller(Cmd *c, unsigned long long gen, unsigned long long idle_ticks)
{
  __shared__ unsigned long long sh[2];
  unsigned long long last = 0, t_last = wall_clock64(), polls = 0;
  for (;;)
  {
    if (threadIdx.x == 0)
    {
      const unsigned long long s = __hip_atomic_load(&c->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const unsigned long long g = __hip_atomic_load(&c->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      ++polls;
      unsigned long long act = 0;
      if (s == ~0ull || g != gen) act = 2;
      else if (s > last) act = 1;
      else if (wall_clock64() - t_last > idle_ticks) act = 2;
      sh[0] = act; sh[1] = s;
    }
    __syncthreads();
    const unsigned long long act = sh[0], s = sh[1];
    if (act == 2) { if (threadIdx.x == 0) c->polls[blockIdx.x] = polls; return; }
    if (act == 1)
    {
      if (threadIdx.x == 0) __hip_atomic_store(&c->ack[blockIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      last = s; t_last = wall_clock64();
    }
    __syncthreads();
  }
}
static double now_us() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }
int main(int argc, char **argv)
{
  const int nb = argc > 1 ? atoi(argv[1]) : 6;
  const unsigned flags = argc > 2 ? (unsigned)atoi(argv[2]) : (hipHostMallocMapped | hipHostMallocCoherent);
  Cmd *c;
  CHK(hipHostMalloc((void **)&c, sizeof(Cmd), flags));
  memset(c, 0, sizeof(Cmd));
  c->gen = 1;
  hipStream_t s1, s2;
  CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  double t0 = now_us();
  hipLaunchKernelGGL(poller, dim3(nb), dim3(256), 0, s1, c, 1ull, 100ull * 20000ull); // 20 ms idle
  CHK(hipGetLastError());
  printf("launch call: %.1f us\n", now_us() - t0);
  volatile Cmd *vc = c;
  double sum = 0, mx = 0, mn = 1e9;
  for (int i = 1; i <= 200; ++i)
  {
    double w = now_us();
    while (now_us() - w < 20.0) {}
    t0 = now_us();
    __atomic_store_n(&c->seq, (unsigned long long)i, __ATOMIC_RELEASE);
    for (int b = 0; b < nb; ++b)
      while (vc->ack[b] != (unsigned long long)i)
        if (now_us() - t0 > 2e6) { printf("no answer to %d from block %d\n", i, b); goto out; }
    {
      const double dt = now_us() - t0;
      if (i > 1) { sum += dt; if (dt > mx) mx = dt; if (dt < mn) mn = dt; }
      else printf("first round trip (incl. kernel start): %.1f us\n", dt);
    }
  }
  printf("round trip: avg %.2f us, min %.2f, max %.2f (%d workgroups)\n", sum / 199, mn, mx, nb);
  // a second generation on another stream while the first is alive
  t0 = now_us();
  __atomic_store_n(&c->gen, 2ull, __ATOMIC_RELEASE);
  hipLaunchKernelGGL(poller, dim3(nb), dim3(256), 0, s2, c, 2ull, 100ull * 20000ull);
  CHK(hipGetLastError());
  printf("second launch call: %.1f us\n", now_us() - t0);
  __atomic_store_n(&c->seq, 1000ull, __ATOMIC_RELEASE);
  for (int b = 0; b < nb; ++b)
    while (vc->ack[b] != 1000ull)
      if (now_us() - t0 > 2e6) { printf("second generation: no answer from block %d\n", b); goto out; }
  printf("second generation answered %.1f us after its launch began\n", now_us() - t0);
out:
  __atomic_store_n(&c->seq, ~0ull, __ATOMIC_RELEASE);
  t0 = now_us();
  CHK(hipStreamSynchronize(s1));
  CHK(hipStreamSynchronize(s2));
  printf("left %.1f us after being told to; polls of block 0: %llu\n", now_us() - t0, c->polls[0]);
  return 0;
}
