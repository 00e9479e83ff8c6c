// phyhip_big.hip -- the large-grid resident evaluator's kernel (phyhip_big.hpp), in a translation unit of its own because it
// is compiled differently: -mllvm -disable-machine-licm.  The kernel is one loop around the bodies of three launched kernels;
// the machine-level loop-invariant code motion hoists every constant those bodies materialise (the polynomial coefficients of
// log / exp, lane masks, LDS offsets) in front of that loop and keeps them in vector registers across it -- at the two waves
// per SIMD the kernel must run at that meant ~130 registers spilled to scratch (tools/kres.py on the listing); without the
// pass: none.  Everything else of the library keeps the pass.
#include "phyhip_big.hpp"
#include "phyhip_aa.hpp"

namespace phyhip
{

// Launch the resident workgroups of an instance with C categories in G lanes per pattern; returns 0, or -1 when there is no
// kernel for that shape.
int big_waves_per_workgroup(int C, int G)
{
  switch (C * 8 + G)
  {
    case 1 * 8 + 1: return 16;
    case 2 * 8 + 1: return 12;
    case 2 * 8 + 2: return 16;
    case 3 * 8 + 1: return 8;
    case 4 * 8 + 1: return 8;
    case 4 * 8 + 2: return 12;
    default: return 0;
  }
}

int launch_resident_big(int C, int G, int workgroups, hipStream_t stream, const BigArgs &a)
{
#define BIGRES(c_, g_, nw_)                                                                                                 \
  hipLaunchKernelGGL((resident_big_kernel<c_, g_, nw_>), dim3(workgroups), dim3(64 * nw_), 0, stream, a);                     \
  return 0;
  // waves per workgroup (one workgroup per CU): what the kernel's registers allow -- 158 / 112 (4 / 2 categories in two lanes
  // per pattern), 252 / 204 / 156 / 106 (4 / 3 / 2 / 1 categories, one lane per pattern); tools/kres.py on the listing
  switch (C * 8 + G)
  {
    case 1 * 8 + 1: BIGRES(1, 1, 16)
    case 2 * 8 + 1: BIGRES(2, 1, 12)
    case 2 * 8 + 2: BIGRES(2, 2, 16)
    case 3 * 8 + 1: BIGRES(3, 1, 8)
    case 4 * 8 + 1: BIGRES(4, 1, 8)
    case 4 * 8 + 2: BIGRES(4, 2, 12)
    default: return -1;
  }
#undef BIGRES
}

// The resident form of the 20-state kernel (phyhip_aa.hpp: traverse_aa_kernel<..., RES>) is such a loop as well -- compiled with
// the pass it spilled ~120 registers at its two waves per SIMD -- so its four instantiations live here too.  Returns 0, or -1 for
// a category count without a kernel.
int launch_resident_aa(int C, int workgroups, int consumers, hipStream_t stream, const TreeParams &sq, const double *afrag, int n_frag_mats,
                       const uint32_t *tip_masks, const AaResident &rs)
{
  const dim3 blk(64 * (consumers + 1));
#define AARES(c_)                                                                                                           \
  case c_:                                                                                                                  \
    hipLaunchKernelGGL((traverse_aa_kernel<c_, false, 0, true, false, 1, false, true>), dim3(workgroups), blk, 0, stream, sq, (const IssueRec *)nullptr, \
                       (const ExecRec *)nullptr, afrag, n_frag_mats, tip_masks, (unsigned long long *)nullptr, rs);          \
    return 0;
  switch (C)
  {
    AARES(1) AARES(2) AARES(3) AARES(4)
    default: return -1;
  }
#undef AARES
}

} // namespace phyhip
