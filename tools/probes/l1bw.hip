// Per-CU vector-memory path probe (gfx950): every wave re-reads the same 10 KiB (L1/L2 resident) with a chosen
// access width / active-lane pattern and reports bytes per shader cycle per CU.  Used to size the A-fragment
// traffic of the 20-state kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int MODE> // 0: 10 x dwordx4 (all lanes)  1: 20 x dwordx2  2: 10 x dwordx4 with 16 of 64 lanes active  3: 10 x ds_read_b128
__global__ __launch_bounds__(256) void probe(const unsigned *src, unsigned *out, int iters, unsigned long long *cyc)
{
  __shared__ unsigned lds[2560 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2560 * 4; i += 256) lds[i] = src[i];
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(src), 0, 1 << 20, 0x00020000);
  unsigned acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it)
  {
    const unsigned base = wave * 10240; // each wave its own 10 KiB (as the four categories)
    asm volatile("" ::: "memory");      // keep the loads inside the loop
    if (MODE == 0)
    {
#pragma unroll
      for (int j = 0; j < 10; ++j)
      {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, base + j * 1024 + lane * 16, 0, 0);
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
    }
    else if (MODE == 1)
    {
#pragma unroll
      for (int j = 0; j < 20; ++j)
      {
        u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, base + j * 512 + lane * 8, 0, 0);
        acc += v.x ^ v.y;
      }
    }
    else if (MODE == 2)
    {
      if ((lane & 12) == 0)
      {
#pragma unroll
        for (int j = 0; j < 10; ++j)
        {
          u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, base + j * 1024 + lane * 16, 0, 0);
          acc += v.x ^ v.y ^ v.z ^ v.w;
        }
      }
    }
    else
    {
#pragma unroll
      for (int j = 0; j < 10; ++j)
      {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(&lds[wave * 2560 + j * 256 + lane * 4]);
        acc += v.x ^ v.y ^ v.z ^ v.w;
      }
      asm volatile("" ::: "memory");
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char *name, int blocks_per_cu, unsigned *src, unsigned *out, unsigned long long *cyc)
{
  const int iters = 2000, grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, src, out, 10, cyc);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, src, out, iters, cyc);
  hipDeviceSynchronize();
  double avg = 0;
  for (int i = 0; i < grid; ++i) avg += (double)cyc[i];
  avg /= grid;
  const double bytes_cu = (double)iters * 10240.0 * 4 * blocks_per_cu * (MODE == 2 ? 0.25 : 1.0);
  printf("%-28s blocks/CU=%d  cycles=%.0f  B/clk/CU=%.1f  cycles per wave-instruction per CU=%.1f\n", name, blocks_per_cu, avg,
         bytes_cu / avg, avg / ((double)iters * (MODE == 1 ? 20 : 10) * 4 * blocks_per_cu));
}
int main()
{
  unsigned *src, *out; unsigned long long *cyc;
  hipMalloc(&src, 1 << 20); hipMemset(src, 1, 1 << 20);
  hipMalloc(&out, 256 * 4 * 256 * 4);
  hipMallocManaged(&cyc, 1024 * 8);
  for (int b = 1; b <= 3; ++b)
  {
    run<0>("buffer_load_dwordx4", b, src, out, cyc);
    run<1>("buffer_load_dwordx2", b, src, out, cyc);
    run<2>("dwordx4, 16/64 lanes active", b, src, out, cyc);
    run<3>("ds_read_b128", b, src, out, cyc);
  }
  return 0;
}
