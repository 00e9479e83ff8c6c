// phyhip_aa.hpp -- amino-acid (20-state) traversal on the FP64 matrix cores of gfx950.
//
// Why MFMA here and only here (BASELINE north star): per site-update the 20-state path does
// 2 x 4 x (20x20) matrix-vector products = 6480 flop against 1289 B of traffic; as a
// batched product over 16 patterns per wave it is a dense [20 x 20] x [20 x 16] contraction per
// (child, rate class), which `v_mfma_f64_16x16x4_f64` executes without the per-FMA operand broadcast that
// limits the VALU form (one LDS/SGPR operand fetch per fused multiply-add).  The 4-state path has a
// 4x4 contraction per lane and stays on the VALU.
//
// MFMA shapes and data layout
//   rows 0..15 of the result:  D[16 x 16] += A[16 x 4] * B[4 x 16]      v_mfma_f64_16x16x4_f64
//   rows 16..19             :  4 blocks of D[4 x 4] += A[4 x 4] * B[4 x 4]   v_mfma_f64_4x4x4_4b_f64
//   columns     = 16 patterns (one wave owns one tile of 16 patterns; block b of the 4x4x4 form = patterns 4b..4b+3)
//   k           = 4 consecutive input states (five k-chunks cover the 20 states, ascending, so the
//                 accumulation order over input states is the reference's: src/avx.c:593-616)
//   lane l = (kk = l >> 4, pp = l & 15):  B operand (both shapes)  x[pattern pp][c][4t + kk]
//                                         A operand 16x16x4        P[c][pp][4t + kk]
//                                         A operand 4x4x4_4b       P[c][16 + (l & 3)][4t + kk]   (same for every block)
//                                         D 16x16x4 regs r         output states kk + 4r  (r = 0..3)
//                                         D 4x4x4_4b               output state 16 + kk
//   (4x4x4_4b lane maps measured with tools/probes/mfma4b.hip: A = 16k + 4b + i, B = 16k + 4b + j, D = 16i + 4b + j.)
// The 4x4x4 form does the last four rows in a quarter of the matrix-core time of a zero-padded second
// 16-row tile.  The D fragments of an update are already the B fragment its parent needs: a lane owns states
// {kk, kk+4, kk+8, kk+12, kk+16} of pattern pp on input and on output, results are forwarded in
// registers without any shuffle, and the elementwise product of the two children is lane-local.
//
// Device layout of an amino-acid partials buffer ("fragment-major"):
//   [pattern tile of 16][category c][320 doubles: aa_slot(k-chunk t, lane)]
// i.e. chunk pairs (0|1), (2|3) as 16 bytes per lane and chunk 4 as 8 bytes per lane: a lane's five values
// move as 2 x dwordx4 + 1 x dwordx2, each instruction one contiguous 1 KiB / 512 B access.  The per-CU
// vector-memory path issues ~1 wave-instruction per 12 cycles whatever its width (measured with the cycle
// stamps of PHYHIP_ABLATE=8), so instruction count, not bytes, is what the layout minimises.  The host-facing
// layout ([pattern][category][state], t_edge::p_lk_*) is restored by phyhip_get_partials / accepted by
// phyhip_set_partials.
#pragma once

#include "phyhip_kernels.hpp"
#include "phyhip_nt2.hpp"

namespace phyhip
{

typedef double       v4d __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kAaT     = 5;   // k-chunks of 4 states
constexpr int kAaBlock = 320; // doubles per (tile, category) block, and per half (rows 0..15 | 16..19) of an A table

// element offset of (pattern p, category c, state s) inside a fragment-major buffer
__host__ __device__ inline size_t aa_off(long long p, int C, int c, int s)
{
  const long long tile = p >> 4;
  return (size_t)(tile * C + c) * kAaBlock + (size_t)aa_slot(s >> 2, (s & 3) * 16 + (int)(p & 15));
}

// A-operand fragments of a set of transition matrices: afrag[m][c][half][aa_slot(t, lane)]
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
  double       *dst = q.afrag + (size_t)m * q.C * (2 * kAaBlock);
  for (int e = threadIdx.x; e < q.C * 2 * kAaT * 64; e += blockDim.x)
  {
    const int lane = e & 63, t = (e >> 6) % kAaT, half = ((e >> 6) / kAaT) & 1, c = (e >> 6) / (2 * kAaT);
    const int i = half ? 16 + (lane & 3) : (lane & 15), j = 4 * t + (lane >> 4);
    dst[(size_t)c * 2 * kAaBlock + aa_afrag_slot(half, t, lane)] = src[(size_t)c * 400 + i * 20 + j];
  }
}

// Host-computed transition matrices (phyhip_set_transition_matrix, the bit-exact route of src/lk.c:2360): up to
// kUploadBatch matrices per launch are read straight from the pinned staging memory (a few KB over the host link),
// written to their slots and -- for 20 states -- once more in MFMA A-operand order.  One launch replaces one copy
// command (plus one aa_frag_kernel launch) per matrix.
constexpr int kUploadBatch = 16;
struct MatUploadParams
{
  int           count, S, C;
  int           idx[kUploadBatch];
  const double *src[kUploadBatch]; // host-pinned, device-accessible
  double       *pmats;
  double       *afrag;             // nullptr unless 20 states
};

__global__ __launch_bounds__(256) void upload_matrices_kernel(const MatUploadParams q)
{
  extern __shared__ __attribute__((aligned(16))) double mat[]; // [C][S][S]
  int           m   = q.idx[0];
  const double *src = q.src[0];
#pragma unroll
  for (int k = 1; k < kUploadBatch; ++k)
    if ((int)blockIdx.x == k) { m = q.idx[k]; src = q.src[k]; }
  const int n = q.C * q.S * q.S;
  for (int e = threadIdx.x; e < n; e += blockDim.x) mat[e] = src[e];
  __syncthreads();
  double *out = q.pmats + (size_t)m * n;
  for (int e = threadIdx.x; e < n; e += blockDim.x) out[e] = mat[e];
  if (q.afrag)
  {
    double *dst = q.afrag + (size_t)m * q.C * (2 * kAaBlock);
    for (int e = threadIdx.x; e < q.C * 2 * kAaT * 64; e += blockDim.x)
    {
      const int lane = e & 63, t = (e >> 6) % kAaT, half = ((e >> 6) / kAaT) & 1, c = (e >> 6) / (2 * kAaT);
      const int i = half ? 16 + (lane & 3) : (lane & 15), j = 4 * t + (lane >> 4);
      dst[(size_t)c * 2 * kAaBlock + aa_afrag_slot(half, t, lane)] = mat[(size_t)c * 400 + i * 20 + j];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K1 + K2 for 20 states.  One WAVE owns (tile of 16 patterns, one rate category); the C waves of a
// workgroup own the C categories of the same tile.  10 000 patterns give only 625 tiles -- fewer than the
// chip's 1024 SIMDs -- so the category axis is spread over waves (2500 waves for C = 4) to keep every
// matrix core busy and to have other waves to switch to while one waits for memory.  The only
// cross-category quantities, the per-pattern maximum of the rescaling rule (src/avx.c:498-510) and the
// category mixture of Lk_Core (src/lk.c:816-818), go through a few bytes of LDS and one barrier per
// operation.
// Pipeline: the host hands every operation over as ready-made buffer descriptors (size 0 = load disabled)
// plus forwarding flags.  Within step k the children / scale words / tip bytes of operation k+1 are issued
// before the matrix-core work into the second raw register set, and the A fragments of operation k+1 are
// issued right after the MFMAs of operation k have consumed theirs, into the same registers (three waves per
// SIMD need <= 168 VGPRs); the previous result is forwarded in registers.
// ---------------------------------------------------------------------------------------------
template <int CP, bool DBG = false>
__global__ __launch_bounds__(64 * CP, 3) void traverse_aa_kernel(const TreeParams q, const IssueRec *__restrict__ irec,
                                                                 const ExecRec *__restrict__ xrec,
                                                                 const double *__restrict__ afrag, int n_frag_mats,
                                                                 const uint8_t *__restrict__ tip_codes,
                                                                 const uint32_t *__restrict__ code_masks, int n_masks,
                                                                 int ablate, unsigned long long *dbg = nullptr)
{
  constexpr int   T    = kAaT;
  // DBG: cycle stamps of the first 64 steps of wave 0 of block 0 (PHYHIP_ABLATE=8), kept in LDS until the end
  __shared__ unsigned long long stamps[DBG ? 64 * 8 : 1];
  const bool stamper = DBG && blockIdx.x == 0 && threadIdx.x == 0;
#define PHY_STAMP(k_, i_)                                                                                              \
  if (DBG)                                                                                                             \
  {                                                                                                                    \
    const unsigned long long t_ = __builtin_readcyclecounter();                                                        \
    if (stamper && (k_) < 64) stamps[(k_) * 8 + (i_)] = t_;                                                            \
  }
  const int       lane = threadIdx.x & 63;
  const int       c    = threadIdx.x >> 6;                 // this wave's rate category (blockDim = 64 * C)
  const long long tile = blockIdx.x;                       // grid = number of tiles: every wave is live
  const int       pp = lane & 15, kk = lane >> 4;
  const long long p0   = tile * 16 + pp;                   // < Ppad
  const bool      pact = p0 < q.P;
  const int       C    = q.C;
  const int       tips = q.tip_count;
  const size_t    ntiles     = (size_t)((q.P + 15) >> 4);
  const size_t    tile_elems = (size_t)C * kAaBlock;
  const size_t    buf_elems  = ntiles * tile_elems;
  const size_t    frag_mat   = (size_t)C * 2 * kAaBlock;   // doubles per matrix in afrag
  const unsigned  blk_bytes  = (unsigned)(((size_t)tile * tile_elems + (size_t)c * kAaBlock) * 8);
  const unsigned  voff_d16 = blk_bytes + lane * 16, voff_d8 = blk_bytes + 2048 + lane * 8; // chunk pairs | chunk 4
  // class axis (mixture classes as categories): every class has its own scale vector, [buffer][class][pattern]
  const bool      cls    = q.class_axis != 0;
  const unsigned  voff_s = ((cls ? (unsigned)c * (unsigned)q.Ppad : 0u) + (unsigned)p0) * 4u, voff_t = (unsigned)p0;
  const unsigned  voff_a16 = (unsigned)c * 2 * kAaBlock * 8 + lane * 16; // this category's A table, this lane's pairs

  __shared__ unsigned xchm[2][CP][16]; // per-pattern maxima (high words), double-buffered by step parity
  __shared__ double   xchl[CP][16];    // category likelihoods of the edge evaluation
  __shared__ unsigned lmask[256];      // allowed-state masks of the tip codes
  for (int i = threadIdx.x; i < n_masks && i < 256; i += blockDim.x) lmask[i] = code_masks[i];
  __syncthreads();

  struct Frag
  { // a lane's five k-chunk values as they come from memory
    u32x4 p01, p23;
    u32x2 p4;
  };
  struct Raw
  {
    Frag     a, b;
    unsigned sa, sb, ca, cb;
  };
  struct AFrag
  { // one child's matrix for this category: rows 0..15 (five k-chunks) then rows 16..19 (five k-chunks), ten values
    u32x4 q[5];
  };
  const __amdgpu_buffer_rsrc_t af_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<double *>(afrag), 0, (int)((size_t)n_frag_mats * frag_mat * 8), 0x00020000);
  auto rsrc = [](const Desc &d) {
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(d.base), 0, (int)d.bytes, 0x00020000);
  };
  auto load_frag = [](Frag &f, const __amdgpu_buffer_rsrc_t r, unsigned v16, unsigned v8, unsigned soff) {
    f.p01 = __builtin_amdgcn_raw_buffer_load_b128(r, v16, soff, 0);
    f.p23 = __builtin_amdgcn_raw_buffer_load_b128(r, v16 + 1024, soff, 0);
    f.p4  = __builtin_amdgcn_raw_buffer_load_b64(r, v8, soff, 0);
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

  auto issue_children = [&](const IssueRec &o, Raw &r) {
    load_frag(r.a, rsrc(o.c1_data), voff_d16, voff_d8, 0);
    load_frag(r.b, rsrc(o.c2_data), voff_d16, voff_d8, 0);
    r.sa = __builtin_amdgcn_raw_buffer_load_b32(rsrc(o.c1_scale), voff_s, 0, 0);
    r.sb = __builtin_amdgcn_raw_buffer_load_b32(rsrc(o.c2_scale), voff_s, 0, 0);
    r.ca = __builtin_amdgcn_raw_buffer_load_b8(rsrc(o.c1_tip), voff_t, 0, 0);
    r.cb = __builtin_amdgcn_raw_buffer_load_b8(rsrc(o.c2_tip), voff_t, 0, 0);
  };
  auto load_afrag = [&](AFrag &A, unsigned off) {
#pragma unroll
    for (int g = 0; g < 5; ++g) A.q[g] = __builtin_amdgcn_raw_buffer_load_b128(af_rsrc, voff_a16 + g * 1024, off, 0);
  };
  auto issue_matrices = [&](unsigned off1, unsigned off2, AFrag &A1, AFrag &A2) {
    load_afrag(A1, off1);
    load_afrag(A2, off2);
  };
  auto unpack_afrag = [](const AFrag &A, double (&lo)[T], double (&hi)[T]) {
    double v[2 * T];
#pragma unroll
    for (int g = 0; g < 5; ++g) __builtin_memcpy(&v[2 * g], &A.q[g], 16);
#pragma unroll
    for (int t = 0; t < T; ++t) { lo[t] = v[t]; hi[t] = v[T + t]; }
  };
  // u[t] = sum over input states of P[c][state kk + 4t][.] * x[.]: rows 0..15 on the 16x16x4 shape, rows 16..19 on
  // the four-block 4x4x4 shape, five k-chunks each, ascending
  auto matvec = [&](const AFrag &A, const double (&x)[T], double (&u)[T]) {
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
    u[0] = lo[0]; u[1] = lo[1]; u[2] = lo[2]; u[3] = lo[3]; u[4] = hi;
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

  double   prev[T] = {0., 0., 0., 0., 0.}; // result of the previous operation (this lane's D fragments)
  unsigned prev_sc = 0;

  if (q.n_ops > 0)
  {
    const int last = q.n_ops - 1; // host pads the list to an even length
    Raw       RA, RB;
    AFrag     A1, A2;
    ExecRec   cur = xrec[0];
    IssueRec  nx1 = irec[(1 < last) ? 1 : last];
    {
      const IssueRec first = irec[0];
      issue_children(first, RA);
      issue_matrices(first.c1_data.x, first.c2_data.x, A1, A2);
      // The loop body sees [A loads][4 result stores][children loads] between an A load and its use.  Four
      // stores through a zero-sized descriptor (dropped by the hardware, but counted) give the loop entry the
      // same in-flight shape, so the compiler's merged s_waitcnt counts never include the previous step's stores.
      const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc(nullptr, 0, 0, 0x00020000);
      const u32x4                  z4   = {0u, 0u, 0u, 0u};
      const u32x2                  z2   = {0u, 0u};
      __builtin_amdgcn_raw_buffer_store_b128(z4, none, 0, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(z4, none, 1024, 0, 0); // distinct offsets: identical stores would be merged
      __builtin_amdgcn_raw_buffer_store_b64(z2, none, 2048, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b32(0u, none, 4096, 0, 0);
    }

    auto step = [&](const int k, const int parity, Raw &R, Raw &Rn) {
      const unsigned fl = cur.dst_data.x;
      double         x1[T], x2[T], u1[T], u2[T], o[T];
      unsigned       s1, s2;
      PHY_STAMP(k, 0)
      if (fl & kOpTip1) { tip_vec(R.ca, x1); s1 = 0; }
      else if (fl & kOpF11)
      {
#pragma unroll
        for (int t = 0; t < T; ++t) x1[t] = prev[t];
        s1 = prev_sc;
      }
      else { unpack(R.a, x1); s1 = R.sa; }
      if (fl & kOpTip2) { tip_vec(R.cb, x2); s2 = 0; }
      else if (fl & kOpF21)
      {
#pragma unroll
        for (int t = 0; t < T; ++t) x2[t] = prev[t];
        s2 = prev_sc;
      }
      else { unpack(R.b, x2); s2 = R.sb; }

      PHY_STAMP(k, 1)
      // all-ones shortcut of the Inin kernel, per (pattern, category): src/avx.c:575-587.  Evaluated without a
      // branch and before the matrix-core phase so that x1 / x2 die with the last MFMA that reads them.
      int ones = 1;
#pragma unroll
      for (int t = 0; t < T; ++t) ones &= (x1[t] == 1.0) & (x2[t] == 1.0);
      ones = and4(ones);
      PHY_STAMP(k, 2)
      const unsigned nx_off1 = nx1.c1_data.x, nx_off2 = nx1.c2_data.x;
      {
        // Matrix-core phase.  The ten 16x16x4 MFMAs occupy the pipe for 64 cycles each; the ten vector-memory
        // instructions of operation k+1 (children, scale words, tip bytes -> the other raw set) are slotted one
        // per MFMA so that their issue cost disappears behind the matrix cores (sched_group_barrier pins the
        // interleave).  Measured alternatives, both slower at 10 000 patterns: all of them before the operand
        // select (628 vs 589 us), all of them after the MFMA chain (627 vs 576 us).  The A fragments of operation k+1 follow the last MFMA that reads this operation's, into
        // the same registers.
        double a1lo[T], a1hi[T], a2lo[T], a2hi[T];
        unpack_afrag(A1, a1lo, a1hi);
        unpack_afrag(A2, a2lo, a2hi);
        v4d    lo1 = {0., 0., 0., 0.}, lo2 = {0., 0., 0., 0.};
        double hi1 = 0., hi2 = 0.;
        issue_children(nx1, Rn);
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
        issue_matrices(nx_off1, nx_off2, A1, A2);
#pragma unroll
        for (int i = 0; i < 2 * T; ++i)
        {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); // one vector-memory read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * T, 0);  // the 4x4x4 chain
        __builtin_amdgcn_sched_group_barrier(0x020, 10, 0);     // A fragments of operation k+1
        u1[0] = lo1[0]; u1[1] = lo1[1]; u1[2] = lo1[2]; u1[3] = lo1[3]; u1[4] = hi1;
        u2[0] = lo2[0]; u2[1] = lo2[1]; u2[2] = lo2[2]; u2[3] = lo2[3]; u2[4] = hi2;
      }
      const IssueRec nx2 = irec[(k + 2 < last) ? k + 2 : last];
      const ExecRec  nxe = xrec[(k + 1 < last) ? k + 1 : last];

      PHY_STAMP(k, 3)
      unsigned mxh = 0;
#pragma unroll
      for (int t = 0; t < T; ++t)
      {
        o[t] = ones ? 1.0 : u1[t] * u2[t];
        mxh  = max(mxh, hi32(o[t]));
      }
      mxh = maxu4(mxh);
      PHY_STAMP(k, 4)
      if (CP > 1 && !(ablate & 2) && !cls)
      { // maximum over the categories of the pattern: one LDS round trip, one barrier (a mixture class rescales alone)
        if (kk == 0) xchm[parity][c][pp] = mxh;
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < CP; ++cc)
          if (cc < C) mxh = max(mxh, xchm[parity][cc][pp]);
      }
      PHY_STAMP(k, 5)
      unsigned sc = s1 + s2; // src/avx.c:462-464
      if (mxh < kHiInvTwoToLarge && q.apply_scaling)
      { // src/avx.c:504-510
#pragma unroll
        for (int t = 0; t < T; ++t) o[t] *= kTwoToLarge;
        sc += kLarge;
      }
      {
        const __amdgpu_buffer_rsrc_t dr = rsrc(cur.dst_data), gr = rsrc(cur.dst_scale);
        Frag w;
        pack(o, w);
        __builtin_amdgcn_raw_buffer_store_b128(w.p01, dr, voff_d16, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(w.p23, dr, voff_d16 + 1024, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(w.p4, dr, voff_d8, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(sc, gr, voff_s, 0, 0); // every lane of the pattern stores the same word
      }
      PHY_STAMP(k, 6)
#pragma unroll
      for (int t = 0; t < T; ++t) prev[t] = o[t];
      prev_sc = sc;
      cur     = nxe;
      nx1     = nx2;
    };
    for (int k = 0; k < q.n_ops; k += 2)
    {
      step(k, 0, RA, RB);
      step(k + 1, 1, RB, RA);
    }
  }

  if (DBG && stamper && dbg)
    for (int i = 0; i < 64 * 8; ++i) dbg[i] = stamps[i];
#undef PHY_STAMP
  if (!q.edge_eval) return;

  // ---- K2: site likelihood at the evaluation edge (src/lk.c:608-645, 767-861) ---------------------
  double contrib = 0.0;
  {
    double   x[T], y[T], u[T];
    unsigned sl, sr;
    auto side = [&](int idx, double (&v)[T], unsigned &sc) {
      if (idx < tips)
      {
        tip_vec(tip_codes[(size_t)idx * q.Ppad + p0], v);
        sc = 0;
      }
      else if (idx == q.last_dest)
      {
#pragma unroll
        for (int t = 0; t < T; ++t) v[t] = prev[t];
        sc = prev_sc;
      }
      else
      {
        const double *src = q.partials + (size_t)(idx - tips) * buf_elems + (size_t)tile * tile_elems + (size_t)c * kAaBlock;
#pragma unroll
        for (int t = 0; t < T; ++t) v[t] = src[aa_slot(t, lane)];
        sc = (unsigned)q.scales[((size_t)(idx - tips) * (cls ? C : 1) + (cls ? c : 0)) * q.Ppad + p0];
      }
    };
    __syncthreads();
    side(q.e_parent, x, sl);
    side(q.e_child, y, sr);
    {
      AFrag A;
      load_afrag(A, (unsigned)((size_t)q.e_pm * frag_mat * 8));
      matvec(A, x, u); // rows: right-side state
    }
    const double *pi_c = q.pi + (cls ? c * 20 : 0);
    double part = 0.0;
#pragma unroll
    for (int t = 0; t < T; ++t) part += u[t] * (y[t] * pi_c[4 * t + kk]);
    const double lkc = sum4(part);
    if (pact && kk == 0 && q.site_cat) q.site_cat[(size_t)p0 * C + c] = lkc;
    if (cls)
    { // per class: its likelihood (above) and its scale exponent; the mixture is combined by class_combine_kernel
      if (pact && kk == 0) q.fact[(size_t)c * q.P + p0] = q.apply_scaling ? (int)(sl + sr) : 0;
      return;
    }
    if (kk == 0) xchl[c][pp] = lkc;
    if (q.fence_post) __threadfence(); // every wave's stores are in memory before wave 0 posts the workgroup's sum
    __syncthreads();
    if (c == 0 && kk == 0 && pact)
    {
      double site = 0.0;
#pragma unroll
      for (int cc = 0; cc < CP; ++cc)
        if (cc < C) site += xchl[cc][pp] * q.cat_w[cc]; // src/lk.c:816-818
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
  // only wave 0 (category 0) carries contributions; fixed shuffle tree -> deterministic
  if (c == 0)
  {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) contrib += __shfl_down(contrib, off, 64);
    publish_block_sum(q, contrib, lane);
  }
}

} // namespace phyhip
