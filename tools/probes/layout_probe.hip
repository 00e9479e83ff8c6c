// tools/probes/layout_probe.hip -- does the ORDER in which a traversal launch touches memory limit its write stream?
// The lane-per-pattern kernel keeps a partials buffer as [row = category x state pair][pattern][2 doubles]: per operation a
// wave of 32 (G = 2) or 64 (G = 1) patterns stores 8 pieces of 512 B / 1 KiB that lie rowb = Ppad x 16 B (0.8 MB at 50 000
// patterns) apart, and the ~1 500 waves of a launch work on different operations, i.e. different buffers, at any moment.
// membench's pure write stream runs at 6.4 TB/s with few writers and 4.4-4.6 with many; the traversal writes at 3.9.
// This probe replays only the STORES of a cfg2-shaped launch (98 operations, 8 rows, nt stores) in three layouts:
//   A  row-major (today):        buffer + row * rowb + pattern * 16
//   B  block-major:              buffer + (pattern / 64) * 8 KiB + row * 1 KiB + (pattern % 64) * 16   (a wave's 8 rows contiguous)
//   C  block-major, 2 KiB rows:  blocks of 128 patterns
// and prints GB/s for each, for G = 1 (64 patterns per wave) and G = 2 (32 patterns per wave, two lanes per pattern).
// usage: layout_probe [patterns = 50000] [ops = 98] [reps = 20]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

// LAYOUT 0: row-major, 1: 64-pattern blocks, 2: 128-pattern blocks.  G lanes per pattern (rows split over the groups).
template <int LAYOUT, int G, bool READ> __global__ __launch_bounds__(64) void k(d2 *base, const d2 *src, size_t buf_elems, int Ppad, int n_ops)
{
  constexpr int R = 8, RL = R / G, PW = 64 / G;
  const int lane = threadIdx.x, grp = lane / PW, pl = lane % PW;
  const int p = blockIdx.x * PW + pl;
  d2 acc = {1.0, 2.0};
  for (int op = 0; op < n_ops; ++op)
  {
    d2 *buf = base + (size_t)op * buf_elems;
    const d2 *sb = src + (size_t)((op * 7 + 3) % n_ops) * buf_elems;
#pragma unroll
    for (int e = 0; e < RL; ++e)
    {
      const int row = grp * RL + e;
      size_t off;
      if (LAYOUT == 0) off = (size_t)row * Ppad + p;
      else if (LAYOUT == 1) off = (size_t)(p / 64) * (64 * R) + (size_t)row * 64 + (p % 64);
      else off = (size_t)(p / 128) * (128 * R) + (size_t)row * 128 + (p % 128);
      if (READ) { const d2 v = __builtin_nontemporal_load(sb + off); acc += v; }
      __builtin_nontemporal_store(acc, buf + off);
    }
  }
}

template <typename F> static double time_ms(F &&launch, int reps)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); launch(); CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int r = 0; r < reps; ++r)
  {
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main(int argc, char **argv)
{
  const int P = argc > 1 ? atoi(argv[1]) : 50000, n_ops = argc > 2 ? atoi(argv[2]) : 98, reps = argc > 3 ? atoi(argv[3]) : 20;
  const int Ppad = (P + 127) / 128 * 128;
  const size_t buf_elems = (size_t)Ppad * 8; // d2 elements per buffer
  d2 *a, *b;
  CK(hipMalloc(&a, buf_elems * 16 * n_ops)); CK(hipMalloc(&b, buf_elems * 16 * n_ops));
  CK(hipMemset(a, 0, buf_elems * 16 * n_ops)); CK(hipMemset(b, 0, buf_elems * 16 * n_ops));
  const double gb = (double)buf_elems * 16 * n_ops / 1e9;
  printf("{\"patterns\": %d, \"ops\": %d, \"GB_written\": %.3f", P, n_ops, gb);
#define RUN(L, G_, RD, name)                                                                                       \
  {                                                                                                                \
    const int grid = Ppad / (64 / G_);                                                                             \
    const double ms = time_ms([&] { hipLaunchKernelGGL((k<L, G_, RD>), dim3(grid), dim3(64), 0, 0, a, b, buf_elems, Ppad, n_ops); }, reps); \
    printf(", \"%s\": {\"us\": %.1f, \"write_GBps\": %.0f}", name, ms * 1e3, gb / (ms * 1e-3));                    \
  }
  RUN(0, 1, false, "w_rowmajor_g1") RUN(1, 1, false, "w_block64_g1") RUN(2, 1, false, "w_block128_g1")
  RUN(0, 2, false, "w_rowmajor_g2") RUN(1, 2, false, "w_block64_g2") RUN(2, 2, false, "w_block128_g2")
  RUN(0, 2, true, "rw_rowmajor_g2") RUN(1, 2, true, "rw_block64_g2") RUN(2, 2, true, "rw_block128_g2")
  RUN(0, 1, true, "rw_rowmajor_g1") RUN(1, 1, true, "rw_block64_g1")
  printf("}\n");
  return 0;
}
