// tools/probes/bar_probe.hip -- can the host store straight into device memory (large BAR), and what does a device-side poll of
// such a word cost against a poll of host memory?  A resident workgroup finds a command by reading it: over the link when the
// record lives in host memory (a read round trip per poll), locally when the host pushed it into device memory.
// usage: bar_probe        -> one JSON line
#include <hip/hip_runtime.h>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <immintrin.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

static sigjmp_buf g_jmp;
static void on_segv(int) { siglongjmp(g_jmp, 1); }

// one wave: waits for word[0] == round, answers by writing round into host-mapped ack; `rounds` times
__global__ void pingpong(const volatile unsigned long long *word, volatile unsigned long long *ack, int rounds)
{
  for (int r = 1; r <= rounds; ++r)
  {
    while (__hip_atomic_load(const_cast<const unsigned long long *>(word), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (unsigned long long)r) __builtin_amdgcn_s_sleep(1);
    __hip_atomic_store(const_cast<unsigned long long *>(ack), (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// the evaluators' real poll: one wave reads 512 bytes = 16 sectors of {3 payload words, number}; a command is there when the
// first `nsec` sectors carry the expected number
__global__ void pingpong512(const unsigned long long *rec, volatile unsigned long long *ack, int rounds, int nsec)
{
  const int lane = threadIdx.x;
  for (int r = 1; r <= rounds; ++r)
  {
    for (;;)
    {
      const unsigned long long v = __hip_atomic_load(rec + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const bool good = !((lane & 3) == 3 && (lane >> 2) < nsec) || v == (unsigned long long)r;
      if (__builtin_amdgcn_ballot_w64(!good) == 0) break;
      __builtin_amdgcn_s_sleep(2);
    }
    if (lane == 0) __hip_atomic_store(const_cast<unsigned long long *>(ack), (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static double now_us() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }

static double run(volatile unsigned long long *word_host_view, const unsigned long long *word_dev_view, bool wc)
{
  unsigned long long *ack;
  CK(hipHostMalloc((void **)&ack, 64, hipHostMallocMapped | hipHostMallocCoherent));
  *ack = 0;
  const int rounds = 2000;
  hipLaunchKernelGGL(pingpong, dim3(1), dim3(1), 0, 0, word_dev_view, ack, rounds);
  const double t0 = now_us();
  for (int r = 1; r <= rounds; ++r)
  {
    *word_host_view = (unsigned long long)r;
    if (wc) _mm_sfence();
    while (*(volatile unsigned long long *)ack != (unsigned long long)r) __builtin_ia32_pause();
  }
  const double dt = (now_us() - t0) / rounds;
  CK(hipDeviceSynchronize());
  CK(hipHostFree(ack));
  return dt;
}

static double run512(unsigned long long *rec_host_view, const unsigned long long *rec_dev_view, bool wc, int nsec)
{
  unsigned long long *ack;
  CK(hipHostMalloc((void **)&ack, 64, hipHostMallocMapped | hipHostMallocCoherent));
  *ack = 0;
  const int rounds = 2000;
  hipLaunchKernelGGL(pingpong512, dim3(1), dim3(64), 0, 0, rec_dev_view, ack, rounds, nsec);
  const double t0 = now_us();
  for (int r = 1; r <= rounds; ++r)
  {
    for (int sct = 0; sct < nsec; ++sct)
    { // payload, then the sector's number (resident_send)
      volatile unsigned long long *q = rec_host_view + 4 * sct;
      q[0] = r; q[1] = r + 1; q[2] = r + 2;
      if (!wc) __atomic_thread_fence(__ATOMIC_RELEASE);
      q[3] = (unsigned long long)r;
    }
    if (wc) _mm_sfence();
    while (*(volatile unsigned long long *)ack != (unsigned long long)r) __builtin_ia32_pause();
  }
  const double dt = (now_us() - t0) / rounds;
  CK(hipDeviceSynchronize());
  CK(hipHostFree(ack));
  return dt;
}

int main()
{
  unsigned long long *hostw;
  CK(hipHostMalloc((void **)&hostw, 4096, hipHostMallocMapped | hipHostMallocCoherent));
  memset(hostw, 0, 4096);
  const double rt_host = run(hostw, hostw, false);
  printf("{\"round_trip_us_command_in_host_memory\": %.2f, \"same_15_sectors_512_byte_poll\": %.2f", rt_host, run512(hostw, hostw, false, 15));
  struct { const char *name; int kind; } kinds[] = {{"hipMalloc", 0}, {"fine_grained", 1}, {"uncached", 2}};
  for (auto &k : kinds)
  {
    unsigned long long *dev = nullptr;
    hipError_t e = k.kind == 0 ? hipMalloc((void **)&dev, 4096)
                               : hipExtMallocWithFlags((void **)&dev, 4096, k.kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
    if (e != hipSuccess) { printf(", \"%s\": \"alloc failed: %s\"", k.name, hipGetErrorString(e)); continue; }
    CK(hipMemset(dev, 0, 4096));
    CK(hipDeviceSynchronize());
    struct sigaction sa, old1, old2;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_segv;
    sigaction(SIGSEGV, &sa, &old1); sigaction(SIGBUS, &sa, &old2);
    volatile int ok = 0;
    if (sigsetjmp(g_jmp, 1) == 0)
    {
      volatile unsigned long long *p = dev;
      p[8] = 0x1234ull; // a host store into device memory
      _mm_sfence();
      ok = (p[8] == 0x1234ull) ? 1 : 2;
    }
    sigaction(SIGSEGV, &old1, nullptr); sigaction(SIGBUS, &old2, nullptr);
    if (!ok) { printf(", \"%s\": \"host store faults\"", k.name); continue; }
    const double rt = run(dev, dev, true);
    CK(hipMemset(dev, 0, 4096));
    CK(hipDeviceSynchronize());
    const double rt512 = run512(dev, dev, true, 15);
    printf(", \"%s\": {\"host_store_ok\": %d, \"round_trip_us_command_in_device_memory\": %.2f, \"same_15_sectors_512_byte_poll\": %.2f}", k.name, (int)ok, rt, rt512);
    CK(hipFree(dev));
  }
  printf("}\n");
  return 0;
}
