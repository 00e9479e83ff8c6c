// phyhip_aa.hpp -- amino-acid (20-state) traversal on the FP64 matrix cores of gfx950.
//
// Why MFMA here and only here (BASELINE north star): per site-update the 20-state path does
// 2 x 4 x (20x20) matrix-vector products = 6480 flop against 1289 B of traffic; as a
// batched product over 16 patterns per wave it is a dense [20 x 20] x [20 x 16] contraction per
// (child, rate class), which `v_mfma_f64_16x16x4_f64` executes without the per-FMA operand broadcast that
// limits the VALU form (one LDS/SGPR operand fetch per fused multiply-add).  The 4-state path has a
// 4x4 contraction per lane and stays on the VALU.
//
// MFMA shape and data layout
//   D[16 x 16] += A[16 x 4] * B[4 x 16]          (one instruction, one wave)
//   rows of D   = output states (two row tiles: states 0..15, and 16..19 padded with zero rows)
//   columns     = 16 patterns (one wave owns one tile of 16 patterns)
//   k           = 4 consecutive input states (five k-chunks cover the 20 states, ascending, so the
//                 accumulation order over input states is the reference's: src/avx.c:593-616)
//   lane l = (kk = l >> 4, pp = l & 15):  A operand  P[c][tile*16 + pp][4t + kk]
//                                         B operand  x[pattern pp][c][4t + kk]
//                                         D regs r   rows kk + 4r  -> output states kk + 4r (r = 0..3), 16 + kk
// The D fragment of an update is therefore already the B fragment its parent needs: a lane owns states
// {kk, kk+4, kk+8, kk+12, kk+16} of pattern pp on input and on output, results are forwarded in
// registers without any shuffle, and the elementwise product of the two children is lane-local.
//
// Device layout of an amino-acid partials buffer ("fragment-major"):
//   [pattern tile of 16][category c][k-chunk t][lane]   one double each
// so every fragment load/store is one fully coalesced 512-byte access.  The host-facing layout
// ([pattern][category][state], t_edge::p_lk_*) is restored by phyhip_get_partials / accepted by
// phyhip_set_partials.
#pragma once

#include "phyhip_kernels.hpp"

namespace phyhip
{

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kAaT = 5; // k-chunks of 4 states

// element offset of (pattern p, category c, state s) inside a fragment-major buffer
__host__ __device__ inline size_t aa_off(long long p, int C, int c, int s)
{
  const long long tile = p >> 4;
  const int       pp = (int)(p & 15), t = s >> 2, kk = s & 3;
  return ((size_t)(tile * C + c) * kAaT + t) * 64 + (size_t)kk * 16 + pp;
}

// A-operand fragments of a set of transition matrices: afrag[m][c][tile][t][lane]
struct FragParams
{
  const int    *indices; // nullptr: use small_idx
  int           small_idx[kSmallPm];
  int           count;
  int           C;
  const double *pmats; // natural [m][c][i][j]
  double       *afrag;
};

__global__ __launch_bounds__(256) void aa_frag_kernel(const FragParams q)
{
  int m;
  if (q.indices) m = q.indices[blockIdx.x];
  else
  {
    m = q.small_idx[0];
#pragma unroll
    for (int k = 1; k < kSmallPm; ++k)
      if ((int)blockIdx.x == k) m = q.small_idx[k];
  }
  const double *src = q.pmats + (size_t)m * q.C * 400;
  double       *dst = q.afrag + (size_t)m * q.C * (2 * kAaT * 64);
  for (int e = threadIdx.x; e < q.C * 2 * kAaT * 64; e += blockDim.x)
  {
    const int lane = e & 63, t = (e >> 6) % kAaT, tile = ((e >> 6) / kAaT) & 1, c = (e >> 6) / (2 * kAaT);
    const int i = tile * 16 + (lane & 15), j = 4 * t + (lane >> 4);
    dst[e] = (i < 20) ? src[(size_t)c * 400 + i * 20 + j] : 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// K1 + K2 for 20 states.  One WAVE owns (tile of 16 patterns, one rate category); the C waves of a
// workgroup own the C categories of the same tile.  10 000 patterns give only 625 tiles -- fewer than the
// chip's 1024 SIMDs -- so the category axis is spread over waves (2500 waves for C = 4) to keep every
// matrix core busy and to have other waves to switch to while one waits for memory.  The only
// cross-category quantities, the per-pattern maximum of the rescaling rule (src/avx.c:498-510) and the
// category mixture of Lk_Core (src/lk.c:816-818), go through a few bytes of LDS and one barrier per
// operation.  The A fragments (transition matrices) of operation k+1 are fetched while operation k's
// MFMAs run (two alternating register sets, as in the nucleotide kernel).
// ---------------------------------------------------------------------------------------------
template <int CP>
__global__ __launch_bounds__(64 * CP) void traverse_aa_kernel(const TreeParams q, const DevOp *__restrict__ ops,
                                                              const double *__restrict__ afrag,
                                                              const uint8_t *__restrict__ tip_codes,
                                                              const uint32_t *__restrict__ code_masks)
{
  constexpr int   T    = kAaT;
  const int       lane = threadIdx.x & 63;
  const int       c    = threadIdx.x >> 6;                 // this wave's rate category (blockDim = 64 * C)
  const long long tile = blockIdx.x;                       // grid = number of tiles: every wave is live
  const int       pp = lane & 15, kk = lane >> 4;
  const long long p0   = tile * 16 + pp;
  const bool      pact = p0 < q.P;                         // this lane's pattern exists
  const long long p    = pact ? p0 : q.P - 1;              // clamp for tip / scale / weight reads
  const int       C    = q.C;
  const int       tips = q.tip_count;
  const size_t    ntiles     = (size_t)((q.P + 15) >> 4);
  const size_t    tile_elems = (size_t)C * T * 64;
  const size_t    buf_elems  = ntiles * tile_elems;
  const size_t    ppad       = ntiles * 16;                // scale vectors are padded to whole tiles
  const size_t    lane_off   = (size_t)tile * tile_elems + (size_t)c * T * 64 + lane;
  const size_t    frag_mat   = (size_t)C * 2 * T * 64;     // doubles per matrix in afrag
  const size_t    frag_c     = (size_t)c * 2 * T * 64;

  __shared__ double xch[2][CP][16]; // per-pattern maxima / category likelihoods, double-buffered by parity

  double prev[T] = {0., 0., 0., 0., 0.};
  int    prev_sc = 0, prev_dest = -1;

  // B-operand fragment of one child: tip -> 0/1 from its state set, forwarded -> registers, else memory
  auto fetch = [&](int idx, double (&x)[T], int &sc) {
    if (idx < tips)
    {
      const uint32_t m = code_masks[tip_codes[(size_t)idx * q.Ppad + p]];
#pragma unroll
      for (int t = 0; t < T; ++t) x[t] = ((m >> (4 * t + kk)) & 1u) ? 1.0 : 0.0;
      sc = 0;
    }
    else if (idx == prev_dest)
    {
#pragma unroll
      for (int t = 0; t < T; ++t) x[t] = prev[t];
      sc = prev_sc;
    }
    else
    {
      const double *src = q.partials + (size_t)(idx - tips) * buf_elems + lane_off;
#pragma unroll
      for (int t = 0; t < T; ++t) x[t] = src[(size_t)t * 64];
      sc = q.scales[(size_t)(idx - tips) * ppad + (size_t)tile * 16 + pp];
    }
  };
  // A-operand fragments of one matrix for this category: [row tile][k-chunk]
  auto load_a = [&](int pm, double (&a)[2 * T]) {
    const double *A = afrag + (size_t)pm * frag_mat + frag_c + lane;
#pragma unroll
    for (int t = 0; t < 2 * T; ++t) a[t] = A[(size_t)t * 64];
  };
  // u[t] = sum over input states of P[c][state kk+4t][.] * x[.]   (two row tiles, five k-chunks)
  auto matvec = [&](const double (&a)[2 * T], const double (&x)[T], double (&u)[T]) {
    v4d lo = {0., 0., 0., 0.}, hi = {0., 0., 0., 0.};
#pragma unroll
    for (int t = 0; t < T; ++t)
    {
      lo = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], x[t], lo, 0, 0, 0);
      hi = __builtin_amdgcn_mfma_f64_16x16x4f64(a[T + t], x[t], hi, 0, 0, 0);
    }
    u[0] = lo[0]; u[1] = lo[1]; u[2] = lo[2]; u[3] = lo[3]; u[4] = hi[0];
  };
  auto and4 = [&](int v) { // AND over the four lanes (kk = 0..3) that share a pattern
    v &= __shfl_xor(v, 16, 64);
    v &= __shfl_xor(v, 32, 64);
    return v;
  };
  auto max4 = [&](double v) {
    v = fmax(v, __shfl_xor(v, 16, 64));
    v = fmax(v, __shfl_xor(v, 32, 64));
    return v;
  };
  auto sum4 = [&](double v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
  };

  if (q.n_ops > 0)
  {
    const int last = q.n_ops - 1;
    double    A1a[2 * T], A2a[2 * T], A1b[2 * T], A2b[2 * T];
    load_a(ops[0].pm1, A1a);
    load_a(ops[0].pm2, A2a);

    auto step = [&](const int k, const int parity, double (&A1)[2 * T], double (&A2)[2 * T], double (&A1n)[2 * T],
                    double (&A2n)[2 * T]) {
      const DevOp op = ops[(k < last) ? k : last];
      const DevOp nx = ops[(k + 1 < last) ? k + 1 : last];
      double      x1[T], x2[T], u1[T], u2[T], o[T];
      int         s1, s2;
      fetch(op.c1, x1, s1);
      fetch(op.c2, x2, s2);
      load_a(nx.pm1, A1n); // next operation's matrices: in flight during this operation's MFMAs
      load_a(nx.pm2, A2n);
      // all-ones shortcut of the Inin kernel, per (pattern, category): src/avx.c:575-587
      int ones = 1;
#pragma unroll
      for (int t = 0; t < T; ++t) ones &= (x1[t] == 1.0) & (x2[t] == 1.0);
      ones = and4(ones);
      matvec(A1, x1, u1);
      matvec(A2, x2, u2);
      double mx = -__builtin_huge_val();
#pragma unroll
      for (int t = 0; t < T; ++t)
      {
        o[t] = ones ? 1.0 : u1[t] * u2[t];
        mx   = (o[t] > mx) ? o[t] : mx;
      }
      mx = max4(mx);
      if (CP > 1)
      { // maximum over the categories of the pattern: one LDS round trip, one barrier
        if (kk == 0) xch[parity][c][pp] = mx;
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < CP; ++cc)
          if (cc < C) mx = fmax(mx, xch[parity][cc][pp]);
      }
      int sc = s1 + s2; // src/avx.c:462-464
      if (mx < kInvTwoToLarge && q.apply_scaling)
      { // src/avx.c:504-510
#pragma unroll
        for (int t = 0; t < T; ++t) o[t] *= kTwoToLarge;
        sc += kLarge;
      }
      {
        double *dst = q.partials + (size_t)(op.dest - tips) * buf_elems + lane_off;
#pragma unroll
        for (int t = 0; t < T; ++t) dst[(size_t)t * 64] = o[t];
        if (kk == 0 && c == 0) q.scales[(size_t)(op.dest - tips) * ppad + (size_t)tile * 16 + pp] = sc;
      }
#pragma unroll
      for (int t = 0; t < T; ++t) prev[t] = o[t];
      prev_sc   = sc;
      prev_dest = op.dest;
    };
    // an odd operation count re-executes the last operation once more (idempotent; keeps the two-set
    // alternation and the barrier count uniform across the workgroup)
    for (int k = 0; k < q.n_ops; k += 2)
    {
      step(k, 0, A1a, A2a, A1b, A2b);
      step(k + 1, 1, A1b, A2b, A1a, A2a);
    }
  }

  if (!q.edge_eval) return;

  // ---- K2: site likelihood at the evaluation edge (src/lk.c:608-645, 767-861) ---------------------
  double contrib = 0.0;
  {
    double x[T], y[T], u[T], a[2 * T];
    int    sl, sr;
    __syncthreads();
    fetch(q.e_parent, x, sl);
    fetch(q.e_child, y, sr);
    load_a(q.e_pm, a);
    matvec(a, x, u); // rows: right-side state
    double part = 0.0;
#pragma unroll
    for (int t = 0; t < T; ++t) part += u[t] * (y[t] * q.pi[4 * t + kk]);
    const double lkc = sum4(part);
    if (pact && kk == 0 && q.site_cat) q.site_cat[(size_t)p0 * C + c] = lkc;
    if (kk == 0) xch[0][c][pp] = lkc;
    __syncthreads();
    if (c == 0 && kk == 0 && pact)
    {
      double site = 0.0;
#pragma unroll
      for (int cc = 0; cc < CP; ++cc)
        if (cc < C) site += xch[0][cc][pp] * q.cat_w[cc]; // src/lk.c:816-818
      const double w = q.wght[p0];
      int          f = q.apply_scaling ? (sl + sr) : 0;
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
        if (site < kSmall) { site = kSmall; *q.warn = 1; }
        const double lsl = log(site) - kLog2 * (double)f;
        if (q.site_lnl) q.site_lnl[p0] = lsl;
        if (q.site_lk) q.site_lk[p0] = exp(lsl);
        contrib = w * lsl;
      }
      q.fact[p0] = f;
    }
  }
  // only wave 0 (category 0) carries contributions; fixed shuffle tree -> deterministic
  if (c == 0)
  {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) contrib += __shfl_down(contrib, off, 64);
    if (lane == 0) q.block_sums[blockIdx.x] = contrib;
  }
}

} // namespace phyhip
