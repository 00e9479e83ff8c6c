// phyhip_aa3.hpp -- amino-acid (20-state) traversal with NT pattern tiles per wave.
//
// Same decomposition as phyhip_aa.hpp (one wave per rate category, the C waves of a workgroup exchange the rescaling
// maximum through LDS once per operation, MFMA shapes / layouts / operation records unchanged), but a wave multiplies
// NT tiles of 16 patterns against ONE load of the operation's A fragments.
//
// Why: the first-generation kernel is bound by the per-CU vector-memory INSTRUCTION rate, not by bytes: the texture
// addresser takes ~16 cycles per wave-instruction whatever its width (tools/probes/l1bw.hip), a (tile, category,
// operation) costs 28 of them -- 10 for the A fragments, which every tile re-reads --, and ablations that zero-size
// every load but keep the instructions change the run time by < 25 % (tools/aa_sweep.sh: 595 -> 452 us at cfg3,
// 4.95 -> 3.87 ms at 100 000 patterns; 10 waves x 28 instructions x 16 cycles = the measured ~4500 cycles per CU and
// operation).  Per (tile, category, operation): NT = 1: 28, NT = 2: 23 -> 19 with the shared A fragments, NT = 4: 16.5;
// the LDS exchange and the barrier are paid once per NT tiles as well.  Two independent tiles also give a wave a
// second MFMA chain to issue while the first one's epilogue runs.
#pragma once

#include "../phyhip_aa.hpp"

namespace phyhip
{

template <int CP, int NT>
__global__ __launch_bounds__(64 * CP, NT >= 4 ? 1 : 2) void traverse_aa3_kernel(
    const TreeParams q, const IssueRec *__restrict__ irec, const ExecRec *__restrict__ xrec, const double *__restrict__ afrag,
    int n_frag_mats, const uint8_t *__restrict__ tip_codes, const uint32_t *__restrict__ code_masks, int n_masks)
{
  constexpr int   T    = kAaT;
  const int       lane = threadIdx.x & 63;
  const int       c    = threadIdx.x >> 6;   // this wave's rate category (blockDim = 64 * C)
  const int       pp = lane & 15, kk = lane >> 4;
  const int       C    = q.C;
  const int       tips = q.tip_count;
  const size_t    ntiles     = (size_t)((q.P + 15) >> 4);
  const size_t    tile_elems = (size_t)C * kAaBlock;
  const size_t    buf_elems  = ntiles * tile_elems;
  const size_t    frag_mat   = (size_t)C * 2 * kAaBlock; // doubles per matrix in afrag
  long long       tile[NT], p0[NT];
  unsigned        voff_d16[NT], voff_d8[NT], voff_s[NT], voff_t[NT];
  bool            pact[NT], tile_ok[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
  { // a tile beyond the alignment (NT does not divide the tile count) lies outside every buffer descriptor: its loads
    // return zeros and its stores are dropped by the bounds check
    tile[j]    = (long long)blockIdx.x * NT + j;
    p0[j]      = tile[j] * 16 + pp;
    pact[j]    = p0[j] < q.P;
    tile_ok[j] = (size_t)tile[j] < ntiles;
    const unsigned blk = (unsigned)(((size_t)tile[j] * tile_elems + (size_t)c * kAaBlock) * 8);
    voff_d16[j] = blk + lane * 16; voff_d8[j] = blk + 2048 + lane * 8;
    voff_s[j] = (unsigned)p0[j] * 4u; voff_t[j] = (unsigned)p0[j];
  }
  const unsigned voff_a16 = (unsigned)c * 2 * kAaBlock * 8 + lane * 16; // this category's A table, this lane's pairs

  __shared__ unsigned xchm[2][CP][16 * NT]; // per-pattern maxima (high words), double-buffered by step parity
  __shared__ double   xchl[CP][16 * NT];    // category likelihoods of the edge evaluation
  __shared__ unsigned lmask[256];           // allowed-state masks of the tip codes
  for (int i = threadIdx.x; i < n_masks && i < 256; i += blockDim.x) lmask[i] = code_masks[i];
  __syncthreads();

  struct Frag
  {
    u32x4 p01, p23;
    u32x2 p4;
  };
  struct Raw
  {
    Frag     a, b;
    unsigned sa, sb, ca, cb;
  };
  struct AFrag
  {
    u32x4 q[5];
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
  auto issue_children = [&](const IssueRec &o, Raw (&r)[NT]) {
    const __amdgpu_buffer_rsrc_t d1 = rsrc(o.c1_data), d2 = rsrc(o.c2_data), s1 = rsrc(o.c1_scale), s2 = rsrc(o.c2_scale),
                                 t1 = rsrc(o.c1_tip), t2 = rsrc(o.c2_tip);
#pragma unroll
    for (int j = 0; j < NT; ++j)
    {
      load_frag(r[j].a, d1, voff_d16[j], voff_d8[j]);
      load_frag(r[j].b, d2, voff_d16[j], voff_d8[j]);
      r[j].sa = __builtin_amdgcn_raw_buffer_load_b32(s1, voff_s[j], 0, 0);
      r[j].sb = __builtin_amdgcn_raw_buffer_load_b32(s2, voff_s[j], 0, 0);
      r[j].ca = __builtin_amdgcn_raw_buffer_load_b8(t1, voff_t[j], 0, 0);
      r[j].cb = __builtin_amdgcn_raw_buffer_load_b8(t2, voff_t[j], 0, 0);
    }
  };
  auto load_afrag = [&](AFrag &A, unsigned off) {
#pragma unroll
    for (int g = 0; g < 5; ++g) A.q[g] = __builtin_amdgcn_raw_buffer_load_b128(af_rsrc, voff_a16 + g * 1024, off, 0);
  };
  auto unpack_afrag = [](const AFrag &A, double (&lo)[T], double (&hi)[T]) {
    double v[2 * T];
#pragma unroll
    for (int g = 0; g < 5; ++g) __builtin_memcpy(&v[2 * g], &A.q[g], 16);
#pragma unroll
    for (int t = 0; t < T; ++t) { lo[t] = v[t]; hi[t] = v[T + t]; }
  };
  auto and4 = [&](int v) {
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

  double   prev[NT][T]; // results of the previous operation (this lane's D fragments), one per tile
  unsigned prev_sc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
  {
    prev_sc[j] = 0;
#pragma unroll
    for (int t = 0; t < T; ++t) prev[j][t] = 0.0;
  }

  if (q.n_ops > 0)
  {
    const int last = q.n_ops - 1; // host pads the list to an even length
    Raw       RA[NT], RB[NT];
    AFrag     A1, A2;
    ExecRec   cur = xrec[0];
    IssueRec  nx1 = irec[(1 < last) ? 1 : last];
    {
      const IssueRec first = irec[0];
      issue_children(first, RA);
      load_afrag(A1, first.c1_data.x);
      load_afrag(A2, first.c2_data.x);
    }

    auto step = [&](const int k, const int parity, Raw (&R)[NT], Raw (&Rn)[NT]) {
      const unsigned fl = cur.dst_data.x;
      double         x1[NT][T], x2[NT][T], o[NT][T];
      unsigned       s1[NT], s2[NT];
      int            ones[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j)
      {
        if (fl & kOpTip1) { tip_vec(R[j].ca, x1[j]); s1[j] = 0; }
        else if (fl & kOpF11)
        {
#pragma unroll
          for (int t = 0; t < T; ++t) x1[j][t] = prev[j][t];
          s1[j] = prev_sc[j];
        }
        else { unpack(R[j].a, x1[j]); s1[j] = R[j].sa; }
        if (fl & kOpTip2) { tip_vec(R[j].cb, x2[j]); s2[j] = 0; }
        else if (fl & kOpF21)
        {
#pragma unroll
          for (int t = 0; t < T; ++t) x2[j][t] = prev[j][t];
          s2[j] = prev_sc[j];
        }
        else { unpack(R[j].b, x2[j]); s2[j] = R[j].sb; }
        // all-ones shortcut of the Inin kernel, per (pattern, category): src/avx.c:575-587
        int on = 1;
#pragma unroll
        for (int t = 0; t < T; ++t) on &= (x1[j][t] == 1.0) & (x2[j][t] == 1.0);
        ones[j] = and4(on);
      }
      const unsigned nx_off1 = nx1.c1_data.x, nx_off2 = nx1.c2_data.x;
      {
        // Matrix-core phase: NT x 20 MFMAs against one set of A fragments; the 10 x NT vector-memory instructions of
        // operation k+1's children go one per MFMA behind the matrix cores, the A fragments of operation k+1 follow
        // the last MFMA that reads this operation's, into the same registers.
        double a1lo[T], a1hi[T], a2lo[T], a2hi[T];
        unpack_afrag(A1, a1lo, a1hi);
        unpack_afrag(A2, a2lo, a2hi);
        issue_children(nx1, Rn);
        double u1[NT][T], u2[NT][T];
#pragma unroll
        for (int j = 0; j < NT; ++j)
        {
          v4d    lo1 = {0., 0., 0., 0.}, lo2 = {0., 0., 0., 0.};
          double hi1 = 0., hi2 = 0.;
#pragma unroll
          for (int t = 0; t < T; ++t)
          {
            lo1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1lo[t], x1[j][t], lo1, 0, 0, 0);
            lo2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2lo[t], x2[j][t], lo2, 0, 0, 0);
          }
#pragma unroll
          for (int t = 0; t < T; ++t)
          {
            hi1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a1hi[t], x1[j][t], hi1, 0, 0, 0);
            hi2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2hi[t], x2[j][t], hi2, 0, 0, 0);
          }
          u1[j][0] = lo1[0]; u1[j][1] = lo1[1]; u1[j][2] = lo1[2]; u1[j][3] = lo1[3]; u1[j][4] = hi1;
          u2[j][0] = lo2[0]; u2[j][1] = lo2[1]; u2[j][2] = lo2[2]; u2[j][3] = lo2[3]; u2[j][4] = hi2;
        }
        load_afrag(A1, nx_off1);
        load_afrag(A2, nx_off2);
#pragma unroll
        for (int i = 0; i < 2 * T * NT; ++i)
        {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); // one vector-memory read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * T * NT, 0); // the remaining MFMAs
        __builtin_amdgcn_sched_group_barrier(0x020, 10, 0);         // A fragments of operation k+1
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int t = 0; t < T; ++t) o[j][t] = ones[j] ? 1.0 : u1[j][t] * u2[j][t];
      }
      const IssueRec nx2 = irec[(k + 2 < last) ? k + 2 : last];
      const ExecRec  nxe = xrec[(k + 1 < last) ? k + 1 : last];

      unsigned mxh[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j)
      {
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < T; ++t) m = max(m, hi32(o[j][t]));
        mxh[j] = maxu4(m);
      }
      if (CP > 1)
      { // maximum over the categories of each pattern: one LDS round trip, ONE barrier for the NT tiles
        if (kk == 0)
#pragma unroll
          for (int j = 0; j < NT; ++j) xchm[parity][c][j * 16 + pp] = mxh[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int cc = 0; cc < CP; ++cc)
            if (cc < C) mxh[j] = max(mxh[j], xchm[parity][cc][j * 16 + pp]);
      }
      const __amdgpu_buffer_rsrc_t dr = rsrc(cur.dst_data), gr = rsrc(cur.dst_scale);
#pragma unroll
      for (int j = 0; j < NT; ++j)
      {
        unsigned sc = s1[j] + s2[j]; // src/avx.c:462-464
        if (mxh[j] < kHiInvTwoToLarge && q.apply_scaling)
        { // src/avx.c:504-510
#pragma unroll
          for (int t = 0; t < T; ++t) o[j][t] *= kTwoToLarge;
          sc += kLarge;
        }
        Frag w;
        pack(o[j], w);
        __builtin_amdgcn_raw_buffer_store_b128(w.p01, dr, voff_d16[j], 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(w.p23, dr, voff_d16[j] + 1024, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(w.p4, dr, voff_d8[j], 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(sc, gr, voff_s[j], 0, 0); // every lane of the pattern stores the same word
#pragma unroll
        for (int t = 0; t < T; ++t) prev[j][t] = o[j][t];
        prev_sc[j] = sc;
      }
      cur = nxe;
      nx1 = nx2;
    };
    for (int k = 0; k < q.n_ops; k += 2)
    {
      step(k, 0, RA, RB);
      step(k + 1, 1, RB, RA);
    }
  }

  if (!q.edge_eval) return;

  // ---- K2: site likelihood at the evaluation edge (src/lk.c:608-645, 767-861) ---------------------
  double contrib = 0.0;
  {
    unsigned sl[NT], sr[NT];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j)
    {
      double x[T], y[T];
      auto side = [&](int idx, double (&v)[T], unsigned &sc) {
        sc = 0;
        if (idx < tips)
          tip_vec(tile_ok[j] ? tip_codes[(size_t)idx * q.Ppad + p0[j]] : 0u, v);
        else if (idx == q.last_dest)
        {
#pragma unroll
          for (int t = 0; t < T; ++t) v[t] = prev[j][t];
          sc = prev_sc[j];
        }
        else if (tile_ok[j])
        {
          const double *src = q.partials + (size_t)(idx - tips) * buf_elems + (size_t)tile[j] * tile_elems + (size_t)c * kAaBlock;
#pragma unroll
          for (int t = 0; t < T; ++t) v[t] = src[aa_slot(t, lane)];
          sc = (unsigned)q.scales[(size_t)(idx - tips) * q.Ppad + p0[j]];
        }
        else
        {
#pragma unroll
          for (int t = 0; t < T; ++t) v[t] = 0.0;
        }
      };
      side(q.e_parent, x, sl[j]);
      side(q.e_child, y, sr[j]);
      AFrag A;
      load_afrag(A, (unsigned)((size_t)q.e_pm * frag_mat * 8));
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
      if (pact[j] && kk == 0 && q.site_cat) q.site_cat[(size_t)p0[j] * C + c] = lkc;
      if (kk == 0) xchl[c][j * 16 + pp] = lkc;
    }
    __syncthreads();
    if (c == 0 && kk == 0)
    {
#pragma unroll
      for (int j = 0; j < NT; ++j)
      {
        if (!pact[j]) continue;
        double site = 0.0;
#pragma unroll
        for (int cc = 0; cc < CP; ++cc)
          if (cc < C) site += xchl[cc][j * 16 + pp] * q.cat_w[cc]; // src/lk.c:816-818
        const double w = q.wght[p0[j]];
        int          f = q.apply_scaling ? (int)(sl[j] + sr[j]) : 0;
        if (w > kSmall)
        {
          if (q.invar_model)
          { // src/lk.c:820-842, 1226-1273
            const int iv  = q.invar[p0[j]];
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
          if (q.site_lnl) q.site_lnl[p0[j]] = lsl;
          if (q.site_lk) q.site_lk[p0[j]] = exp(lsl);
          contrib += w * lsl;
        }
        q.fact[p0[j]] = f;
      }
    }
  }
  // only wave 0 (category 0) carries contributions; fixed shuffle tree -> deterministic
  if (c == 0)
  {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) contrib += __shfl_down(contrib, off, 64);
    publish_block_sum(q, contrib, lane);
  }
}

} // namespace phyhip
