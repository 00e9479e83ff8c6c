// Layout probe for v_mfma_f64_4x4x4_4b_f64 on gfx950: prints which (block, row, k) / (block, k, col) / (block,row,col)
// each lane holds, by running the instruction on one-hot inputs.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const double *a, const double *b, double *d)
{
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
int main()
{
  double *a, *b, *d;
  hipMallocManaged(&a, 64 * 8); hipMallocManaged(&b, 64 * 8); hipMallocManaged(&d, 64 * 8);
  // for every (la, lb): which output lanes light up, and with one-hot values we learn the pairing
  for (int la = 0; la < 64; ++la)
  {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
    {
      for (int i = 0; i < 64; ++i) { a[i] = 0; b[i] = 0; d[i] = 0; }
      a[la] = 1; b[lb] = 1;
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
      hipDeviceSynchronize();
      for (int i = 0; i < 64; ++i) if (d[i] != 0) printf(" (B%d->D%d)", lb, i);
    }
    printf("\n");
  }
  return 0;
}
