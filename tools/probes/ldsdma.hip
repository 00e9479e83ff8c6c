// Probe: LDS-DMA (global_load_lds_dwordx4 / buffer_load_dwordx4 ... lds) on gfx950 with LDS destinations beyond 64 KiB:
// where does a piece land for a given wave-uniform LDS pointer and instruction offset?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kLds = 40 * 1024; // dwords = 160 KiB - a little
__global__ __launch_bounds__(64) void probe(const unsigned *src, unsigned *out, int lds_dword, int mode)
{
  __shared__ unsigned lds[38 * 1024];
  for (int i = threadIdx.x; i < 38 * 1024; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const int lane = threadIdx.x;
  if (mode == 0)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + lane * 4),
                                     (__attribute__((address_space(3))) void *)(lds + lds_dword), 16, 0, 0);
  else if (mode == 1)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + lane * 4),
                                     (__attribute__((address_space(3))) void *)(lds + lds_dword), 16, 1024, 0);
  else
  {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(src), 0, 1 << 20, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(lds + lds_dword), 16, lane * 16, 2048, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 38 * 1024; i += 64) out[i] = lds[i];
}
int main()
{
  unsigned *src, *out;
  hipMalloc(&src, 1 << 20); hipMalloc(&out, 38 * 1024 * 4);
  std::vector<unsigned> h(1 << 18);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)i;
  hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice);
  std::vector<unsigned> o(38 * 1024);
  for (int mode = 0; mode < 3; ++mode)
    for (int ld : {0, 256, 16384, 20000, 30000, 37000})
    {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out, ld, mode);
      hipDeviceSynchronize();
      hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
      int first = -1, n = 0; unsigned v0 = 0;
      for (int i = 0; i < (int)o.size(); ++i)
        if (o[i] != 0xdeadbeefu) { if (first < 0) { first = i; v0 = o[i]; } ++n; }
      printf("mode %d lds_dword %6d: %d dwords landed, first at dword %d (value %u), contiguous-ok %d\n", mode, ld, n, first, v0,
             first >= 0 && n == 256 && o[first + 255] == v0 + 255);
    }
  return 0;
}
