// phyhip_aa2.hpp -- amino-acid (20-state) traversal, second generation: one WAVE owns a tile of 16 patterns with ALL its
// rate categories and walks them one after the other on the FP64 matrix cores.
//
// Why (profiles/r01_pmc_cfg3*): the first generation (phyhip_aa.hpp: one wave per (tile, category), the C
// waves of a workgroup meeting at one barrier per operation for the cross-category maximum of the rescaling rule) ran
// cfg3 at 0.54 of the HBM roofline with the matrix cores 28 % busy: nothing was saturated -- every wave spent most of an
// operation waiting, for the single-buffered A fragments (the matrix operand, 10 KB per wave-operation, re-read by
// every tile), for its three partner waves at the barrier, for the LDS round trip of the maximum.  Here
//   * the cross-category quantities (rescaling maximum src/avx.c:498-510, category mixture src/lk.c:816-818) are
//     lane-local: no barrier, no LDS exchange, no workgroup at all (one-wave workgroups);
//   * one operation is a chain of C x 20 MFMAs (C x 800 matrix-core cycles) issued by ONE wave, long enough to hide,
//     by software pipelining inside the wave, every load it needs: the A fragments of category c+1 (double-buffered)
//     and the children of operation k+1 (category c's raw registers are refilled as soon as they have been unpacked)
//     are fetched behind the MFMAs of category c;
//   * a SIMD runs one such wave (<= 512 VGPRs): the matrix pipe sees back-to-back MFMAs instead of the interleaving of
//     three stalled waves.
// Arithmetic per (pattern, category, state) is unchanged -- same MFMA shapes, same ascending k-chunks, same product,
// same maximum, same power-of-two rescale -- so every buffer stays bit-identical to the first generation and to the
// oracle.  Buffer layout, A-operand tables and operation records are those of phyhip_aa.hpp.
#pragma once

#include "../phyhip_aa.hpp"

#include <type_traits>

namespace phyhip
{

template <int C, bool DBG = false>
__global__ __launch_bounds__(64, 1) void traverse_aa2_kernel(const TreeParams q, const IssueRec *__restrict__ irec,
                                                             const ExecRec *__restrict__ xrec,
                                                             const double *__restrict__ afrag, int n_frag_mats,
                                                             const uint8_t *__restrict__ tip_codes,
                                                             const uint32_t *__restrict__ code_masks, int n_masks,
                                                             unsigned long long *dbg = nullptr)
{
  constexpr int   T    = kAaT;
  // DBG (-DPHYHIP_DIAG, PHYHIP_ABLATE=8): cycle stamps of the first 64 operations of workgroup `dbg_block`
  __shared__ unsigned long long stamps[DBG ? 64 * 8 : 1];
  const bool stamper = DBG && blockIdx.x == (gridDim.x > 300 ? 300 : 0) && threadIdx.x == 0;
#define PHY_STAMP(k_, i_)                                                                                              \
  if (DBG)                                                                                                             \
  {                                                                                                                    \
    const unsigned long long t_ = __builtin_readcyclecounter();                                                        \
    if (stamper && (k_) < 64) stamps[(k_) * 8 + (i_)] = t_;                                                            \
  }
  const int       lane = threadIdx.x;
  const long long tile = blockIdx.x;
  const int       pp = lane & 15, kk = lane >> 4;
  const long long p0   = tile * 16 + pp; // < Ppad
  const bool      pact = p0 < q.P;
  const int       tips = q.tip_count;
  const size_t    ntiles     = (size_t)((q.P + 15) >> 4);
  const size_t    tile_elems = (size_t)C * kAaBlock;
  const size_t    buf_elems  = ntiles * tile_elems;
  const size_t    frag_mat   = (size_t)C * 2 * kAaBlock; // doubles per matrix in afrag
  const unsigned  blk_bytes  = (unsigned)((size_t)tile * tile_elems * 8);
  const unsigned  voff_d16 = blk_bytes + lane * 16, voff_d8 = blk_bytes + 2048 + lane * 8; // + c * 2560: category c
  const unsigned  voff_s = (unsigned)p0 * 4u, voff_t = (unsigned)p0;
  const unsigned  voff_a16 = lane * 16;                                                    // + c * 5120: category c

  __shared__ unsigned lmask[256]; // allowed-state masks of the tip codes
  for (int i = threadIdx.x; i < n_masks && i < 256; i += blockDim.x) lmask[i] = code_masks[i];
  __syncthreads();

  struct Frag
  { // a lane's five k-chunk values as they come from memory
    u32x4 p01, p23;
    u32x2 p4;
  };
  struct Raw
  {
    Frag a, b;
  };
  struct AFrag
  { // one child's matrix for one category: rows 0..15 (five k-chunks) then rows 16..19 (five k-chunks), ten values
    u32x4 q[5];
  };
  struct APair
  {
    AFrag m1, m2;
  };
  const __amdgpu_buffer_rsrc_t af_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<double *>(afrag), 0, (int)((size_t)n_frag_mats * frag_mat * 8), 0x00020000);
  auto rsrc = [](const Desc &d) {
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(d.base), 0, (int)d.bytes, 0x00020000);
  };
  auto load_frag = [](Frag &f, const __amdgpu_buffer_rsrc_t r, unsigned v16, unsigned v8) {
    f.p01 = __builtin_amdgcn_raw_buffer_load_b128(r, v16, 0, 0);
    f.p23 = __builtin_amdgcn_raw_buffer_load_b128(r, v16 + 1024, 0, 0);
    f.p4  = __builtin_amdgcn_raw_buffer_load_b64(r, v8, 0, 0);
  };
  auto unpack = [](const Frag &f, double (&x)[T]) {
    __builtin_memcpy(&x[0], &f.p01, 16);
    __builtin_memcpy(&x[2], &f.p23, 16);
    __builtin_memcpy(&x[4], &f.p4, 8);
  };
  auto pack = [](const double (&x)[T], Frag &f) {
    __builtin_memcpy(&f.p01, &x[0], 16);
    __builtin_memcpy(&f.p23, &x[2], 16);
    __builtin_memcpy(&f.p4, &x[4], 8);
  };
  auto load_afrag = [&](AFrag &A, unsigned mat_off, int c) {
#pragma unroll
    for (int g = 0; g < 5; ++g)
      A.q[g] = __builtin_amdgcn_raw_buffer_load_b128(af_rsrc, voff_a16 + (unsigned)c * (2 * kAaBlock * 8) + g * 1024, mat_off, 0);
  };
  auto unpack_afrag = [](const AFrag &A, double (&lo)[T], double (&hi)[T]) {
    double v[2 * T];
#pragma unroll
    for (int g = 0; g < 5; ++g) __builtin_memcpy(&v[2 * g], &A.q[g], 16);
#pragma unroll
    for (int t = 0; t < T; ++t) { lo[t] = v[t]; hi[t] = v[T + t]; }
  };
  auto and4 = [&](int v) { // AND over the four lanes (kk = 0..3) that share a pattern
    v &= __shfl_xor(v, 16, 64);
    v &= __shfl_xor(v, 32, 64);
    return v;
  };
  auto maxu4 = [&](unsigned v) {
    v = max(v, (unsigned)__shfl_xor((int)v, 16, 64));
    v = max(v, (unsigned)__shfl_xor((int)v, 32, 64));
    return v;
  };
  auto sum4 = [&](double v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
  };
  auto tip_vec = [&](unsigned code, double (&x)[T]) {
    const unsigned m = lmask[code & 255u] >> kk;
#pragma unroll
    for (int t = 0; t < T; ++t) x[t] = ((m >> (4 * t)) & 1u) ? 1.0 : 0.0;
  };

  double   prev[C][T]; // result of the previous operation (this lane's D fragments), all categories
  unsigned prev_sc = 0;
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int t = 0; t < T; ++t) prev[c][t] = 0.0;

  if (q.n_ops > 0)
  {
    const int last = q.n_ops - 1; // the host pads the list to an even length
    Raw       R[C];               // children of the operation about to run, one register set per category
    unsigned  sa, sb, ca, cb;     // ... their scale words and tip bytes
    APair     AB[C];              // A fragments of the operation about to run, one register set per category
    ExecRec   cur = xrec[0];
    IssueRec  nx1 = irec[(1 < last) ? 1 : last];
    {
      const IssueRec first = irec[0];
      const __amdgpu_buffer_rsrc_t r1 = rsrc(first.c1_data), r2 = rsrc(first.c2_data);
#pragma unroll
      for (int c = 0; c < C; ++c)
      {
        load_frag(R[c].a, r1, voff_d16 + c * 2560, voff_d8 + c * 2560);
        load_frag(R[c].b, r2, voff_d16 + c * 2560, voff_d8 + c * 2560);
      }
      sa = __builtin_amdgcn_raw_buffer_load_b32(rsrc(first.c1_scale), voff_s, 0, 0);
      sb = __builtin_amdgcn_raw_buffer_load_b32(rsrc(first.c2_scale), voff_s, 0, 0);
      ca = __builtin_amdgcn_raw_buffer_load_b8(rsrc(first.c1_tip), voff_t, 0, 0);
      cb = __builtin_amdgcn_raw_buffer_load_b8(rsrc(first.c2_tip), voff_t, 0, 0);
#pragma unroll
      for (int c = 0; c < C; ++c)
      {
        load_afrag(AB[c].m1, first.c1_data.x, c);
        load_afrag(AB[c].m2, first.c2_data.x, c);
      }
    }

    // One operation.  EVERY load is issued exactly one operation before its use, category by category: in phase c the
    // registers of category c (children R[c], matrices AB[c]) are unpacked and at once refilled for operation k+1.  A
    // wave's loads return in issue order, so a short-latency load (A fragments: L2 hits) issued behind a long-latency one
    // (children: HBM) inherits its latency; with the uniform one-operation distance (>= C x 800 matrix-core cycles) no
    // wait ever catches a load younger than that.  (Measured first with the A fragments only one category ahead: every
    // category phase then cost one HBM latency, 800 us at cfg3 instead of 578 for the first generation.)
    auto step = [&](const int k) {
      const unsigned fl  = cur.dst_data.x;
      const unsigned nx_off1 = nx1.c1_data.x, nx_off2 = nx1.c2_data.x;
      const __amdgpu_buffer_rsrc_t n1 = rsrc(nx1.c1_data), n2 = rsrc(nx1.c2_data);
      double   o[C][T];
      unsigned mxh = 0;
      unsigned s1 = (fl & kOpTip1) ? 0u : ((fl & kOpF11) ? prev_sc : sa);
      unsigned s2 = (fl & kOpTip2) ? 0u : ((fl & kOpF21) ? prev_sc : sb);
      double   t1[T], t2[T];
      tip_vec(ca, t1);
      tip_vec(cb, t2);
      unsigned nsa = 0, nsb = 0, nca = 0, ncb = 0;
#pragma unroll
      for (int c = 0; c < C; ++c)
      {
        PHY_STAMP(k, c)
        APair &A = AB[c];
        double x1[T], x2[T];
        if (fl & kOpTip1)
        {
#pragma unroll
          for (int t = 0; t < T; ++t) x1[t] = t1[t];
        }
        else if (fl & kOpF11)
        {
#pragma unroll
          for (int t = 0; t < T; ++t) x1[t] = prev[c][t];
        }
        else unpack(R[c].a, x1);
        if (fl & kOpTip2)
        {
#pragma unroll
          for (int t = 0; t < T; ++t) x2[t] = t2[t];
        }
        else if (fl & kOpF21)
        {
#pragma unroll
          for (int t = 0; t < T; ++t) x2[t] = prev[c][t];
        }
        else unpack(R[c].b, x2);
        // all-ones shortcut of the Inin kernel, per (pattern, category): src/avx.c:575-587
        int ones = 1;
#pragma unroll
        for (int t = 0; t < T; ++t) ones &= (x1[t] == 1.0) & (x2[t] == 1.0);
        ones = and4(ones);

        double a1lo[T], a1hi[T], a2lo[T], a2hi[T];
        unpack_afrag(A.m1, a1lo, a1hi);
        unpack_afrag(A.m2, a2lo, a2hi);
        // Behind this category's MFMAs (the wave issues in order and an MFMA holds the issue port until the pipe takes
        // it, so every load is slotted BETWEEN two MFMAs): category c of operation k+1 -- its A fragments and its
        // children -- into the registers just unpacked.
        load_afrag(A.m1, nx_off1, c);
        load_afrag(A.m2, nx_off2, c);
        load_frag(R[c].a, n1, voff_d16 + c * 2560, voff_d8 + c * 2560);
        load_frag(R[c].b, n2, voff_d16 + c * 2560, voff_d8 + c * 2560);
        if (c == 0)
        {
          nsa = __builtin_amdgcn_raw_buffer_load_b32(rsrc(nx1.c1_scale), voff_s, 0, 0);
          nsb = __builtin_amdgcn_raw_buffer_load_b32(rsrc(nx1.c2_scale), voff_s, 0, 0);
          nca = __builtin_amdgcn_raw_buffer_load_b8(rsrc(nx1.c1_tip), voff_t, 0, 0);
          ncb = __builtin_amdgcn_raw_buffer_load_b8(rsrc(nx1.c2_tip), voff_t, 0, 0);
        }
        v4d    lo1 = {0., 0., 0., 0.}, lo2 = {0., 0., 0., 0.};
        double hi1 = 0., hi2 = 0.;
#pragma unroll
        for (int t = 0; t < T; ++t)
        {
          lo1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1lo[t], x1[t], lo1, 0, 0, 0);
          lo2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2lo[t], x2[t], lo2, 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < T; ++t)
        {
          hi1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a1hi[t], x1[t], hi1, 0, 0, 0);
          hi2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2hi[t], x2[t], hi2, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4 * T; ++i)
        {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); // one vector-memory read
        }
        const double u1[T] = {lo1[0], lo1[1], lo1[2], lo1[3], hi1};
        const double u2[T] = {lo2[0], lo2[1], lo2[2], lo2[3], hi2};
#pragma unroll
        for (int t = 0; t < T; ++t)
        {
          o[c][t] = ones ? 1.0 : u1[t] * u2[t];
          mxh     = max(mxh, hi32(o[c][t]));
        }
      }
      const IssueRec nx2 = irec[(k + 2 < last) ? k + 2 : last];
      const ExecRec  nxe = xrec[(k + 1 < last) ? k + 1 : last];
      PHY_STAMP(k, 4)
      mxh = maxu4(mxh); // maximum over the pattern's C x 20 entries: lane-local over (c, t), then over the four kk lanes
      unsigned sc = s1 + s2; // src/avx.c:462-464
      if (mxh < kHiInvTwoToLarge && q.apply_scaling)
      { // src/avx.c:504-510
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
          for (int t = 0; t < T; ++t) o[c][t] *= kTwoToLarge;
        sc += kLarge;
      }
      {
        const __amdgpu_buffer_rsrc_t dr = rsrc(cur.dst_data), gr = rsrc(cur.dst_scale);
#pragma unroll
        for (int c = 0; c < C; ++c)
        {
          Frag w;
          pack(o[c], w);
          __builtin_amdgcn_raw_buffer_store_b128(w.p01, dr, voff_d16 + c * 2560, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(w.p23, dr, voff_d16 + c * 2560 + 1024, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(w.p4, dr, voff_d8 + c * 2560, 0, 0);
        }
        __builtin_amdgcn_raw_buffer_store_b32(sc, gr, voff_s, 0, 0); // every lane of the pattern stores the same word
      }
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int t = 0; t < T; ++t) prev[c][t] = o[c][t];
      prev_sc = sc;
      PHY_STAMP(k, 5)
      cur = nxe;
      nx1 = nx2;
      sa = nsa; sb = nsb; ca = nca; cb = ncb;
    };
    for (int k = 0; k < q.n_ops; k += 2)
    {
      step(k);
      step(k + 1);
    }
  }

  if (DBG && stamper && dbg)
    for (int i = 0; i < 64 * 8; ++i) dbg[i] = stamps[i];
#undef PHY_STAMP
  if (!q.edge_eval) return;

  // ---- K2: site likelihood at the evaluation edge (src/lk.c:608-645, 767-861) ---------------------
  double contrib = 0.0;
  {
    unsigned sl = 0, sr = 0;
    double   site = 0.0;
#pragma unroll
    for (int c = 0; c < C; ++c)
    {
      double x[T], y[T];
      auto side = [&](int idx, double (&v)[T], unsigned &sc) {
        if (idx < tips)
        {
          tip_vec(tip_codes[(size_t)idx * q.Ppad + p0], v);
          sc = 0;
        }
        else if (idx == q.last_dest)
        {
#pragma unroll
          for (int t = 0; t < T; ++t) v[t] = prev[c][t];
          sc = prev_sc;
        }
        else
        {
          const double *src = q.partials + (size_t)(idx - tips) * buf_elems + (size_t)tile * tile_elems + (size_t)c * kAaBlock;
#pragma unroll
          for (int t = 0; t < T; ++t) v[t] = src[aa_slot(t, lane)];
          sc = (unsigned)q.scales[(size_t)(idx - tips) * q.Ppad + p0];
        }
      };
      side(q.e_parent, x, sl);
      side(q.e_child, y, sr);
      AFrag A;
      load_afrag(A, (unsigned)((size_t)q.e_pm * frag_mat * 8), c);
      double alo[T], ahi[T];
      unpack_afrag(A, alo, ahi);
      v4d    lo = {0., 0., 0., 0.};
      double hi = 0.;
#pragma unroll
      for (int t = 0; t < T; ++t)
      {
        lo = __builtin_amdgcn_mfma_f64_16x16x4f64(alo[t], x[t], lo, 0, 0, 0);
        hi = __builtin_amdgcn_mfma_f64_4x4x4f64(ahi[t], x[t], hi, 0, 0, 0);
      }
      const double u[T] = {lo[0], lo[1], lo[2], lo[3], hi}; // rows: right-side state
      double part = 0.0;
#pragma unroll
      for (int t = 0; t < T; ++t) part += u[t] * (y[t] * q.pi[4 * t + kk]);
      const double lkc = sum4(part);
      if (pact && kk == 0 && q.site_cat) q.site_cat[(size_t)p0 * C + c] = lkc;
      site += lkc * q.cat_w[c]; // src/lk.c:816-818
    }
    if (kk == 0 && pact)
    {
      const double w = q.wght[p0];
      int          f = q.apply_scaling ? (int)(sl + sr) : 0;
      if (w > kSmall)
      {
        if (q.invar_model)
        { // src/lk.c:820-842, 1226-1273
          const int iv  = q.invar[p0];
          double    inv = 0.0;
          bool      issue_ = false;
          if (iv >= 0)
          {
            inv = q.pi[iv];
            if (q.apply_scaling)
            {
              int e = f;
              do
              {
                const int piece = e < 63 ? e : 63;
                inv *= (double)(1ull << piece);
                e -= piece;
              } while (e != 0);
            }
            issue_ = isinf(inv);
          }
          if (issue_) { f = 0; site = q.pi[iv] * q.pinvar; }
          else site = site * (1. - q.pinvar) + inv * q.pinvar;
        }
        if (site < kSmall) { site = kSmall; raise_warn(q); }
        const double lsl = log(site) - kLog2 * (double)f;
        if (q.site_lnl) q.site_lnl[p0] = lsl;
        if (q.site_lk) q.site_lk[p0] = exp(lsl);
        contrib = w * lsl;
      }
      q.fact[p0] = f;
    }
  }
  // only lanes kk == 0 carry contributions; fixed shuffle tree -> deterministic
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) contrib += __shfl_down(contrib, off, 64);
  publish_block_sum(q, contrib, lane);
}

} // namespace phyhip
