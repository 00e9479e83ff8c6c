// phyhip_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for PhyML's likelihood hot path.
//
// Work decomposition (DESIGN.md "Kernels"): the unit of parallelism is one (pattern, rate category) pair
// per lane.  Partial vectors keep the reference's [pattern][category][state] layout
// (src/lk.c:1474,1520), so consecutive lanes read consecutive 8*S-byte records: a wave's load of one
// child is ONE contiguous span (2 KiB for S=4), every byte of every fetched line is used, and the
// device buffers are byte-identical to the host's t_edge::p_lk_* buffers.
// The CP lanes of one pattern (CP = categories rounded up to a power of two) sit in adjacent lanes, so
// the per-pattern max-scan of the rescaling rule (src/avx.c:498-510) and the category mixture of
// Lk_Core (src/lk.c:816-818) are CP-lane cross-lane operations, no LDS, no barrier.
//
// One launch executes a whole *list* of update operations (the post-order traversal recorded by the
// host) for its pattern tile: patterns are independent, so a thread that owns (pattern, category) can
// walk the entire tree alone and the launch needs no inter-workgroup synchronisation at all.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "phyhip_exp.hpp"

namespace phyhip
{

// exp() wherever the reference calls its libm's on this path (transition matrices, the eigen-basis tables, cur_site_lk): the
// reference's doubles (phyhip_exp.hpp)
// (resident kernels keep a copy of the 2 KB table in LDS, copied once per launch: thousands of waves gathering from the global
// table in their matrix rebuild cost an SPR candidate 1.5 us at 500 x 100 000, measured; exp_tab_to_lds / dev_exp(x, lds table))
typedef const __attribute__((address_space(3))) unsigned long long *ExpLdsTab;
#ifdef PHYHIP_OCML_EXP // (A/B builds only, tools/build_variant.sh: the device library's exp -- a few ulp from the reference's)
__device__ __forceinline__ double dev_exp(const double x, const unsigned long long * = nullptr) { return exp(x); }
#else
__device__ __forceinline__ double dev_exp(const double x) { return phyhip_exp_ref(x, phyhip_exp_tab); }
// lds: a copy of the table in LDS (generic pointer to it), or nullptr
__device__ __forceinline__ double dev_exp(const double x, const unsigned long long *lds)
{
  return lds ? phyhip_exp_ref(x, (ExpLdsTab)lds) : phyhip_exp_ref(x, phyhip_exp_tab);
}
#endif
__device__ __forceinline__ void exp_tab_to_lds(unsigned long long *lds, const int tid, const int nth)
{
  for (int i = tid; i < 256; i += nth) lds[i] = phyhip_exp_tab[i];
}

constexpr int    kLarge         = 256;                         // src/utilities.h:507
constexpr double kTwoToLarge    = 0x1p256;                     // TWO_TO_THE_LARGE
constexpr double kInvTwoToLarge = 0x1p-256;                    // INV_TWO_TO_THE_LARGE
constexpr double kSmall         = 2.2250738585072014e-308;     // SMALL = DBL_MIN, src/utilities.h:476
constexpr double kLog2          = 0.69314718055994528623;      // LOG2, src/utilities.h:267
constexpr double kSmallPij      = 1.E-100;                     // SMALL_PIJ, src/utilities.h:478
constexpr int    kArgUp         = 3;                           // host-computed matrices per launch in the kernel arguments
constexpr int    kMaxExpl       = 2 * 8 * 20;                  // dLk's expl table in the kernel arguments: [C<=8][2][S<=20]
constexpr int    kMaxCategories = 64;                          // a pattern's categories sit in the lanes of one wave
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct HostBlock
{ // one 16-byte store per workgroup: the sum, then the tag the host polls for
  double             sum;
  unsigned long long tag;
};

struct DevOp
{
  int dest, c1, c2, pm1, pm2, pad;
};

// Host-prepared operation records for the pipelined nucleotide kernel.  All per-operation address
// arithmetic is done ONCE on the host at flush time (the scalar unit is shared by the four SIMDs of a CU:
// ~200 SALU instructions per operation were the kernel's bottleneck when descriptors were derived on
// the device).  A `Desc` is (base address, size in bytes); size 0 disables the load through the
// hardware bounds check.
struct Desc
{
  unsigned long long base;
  unsigned           bytes;
  unsigned           x; // spare word, meaning depends on the slot
};
struct IssueRec          // what the load stage of operation k needs (consumed two steps before execution)
{
  Desc c1_data;  // x: byte offset of transition matrix 1 inside the matrix table
  Desc c2_data;  // x: byte offset of transition matrix 2
  Desc c1_scale;
  Desc c2_scale;
  Desc c1_tip;
  Desc c2_tip;
};
struct ExecRec           // what the compute / store stage of operation k needs
{
  Desc dst_data; // x: flags
  Desc dst_scale;
};
enum : unsigned
{
  kOpTip1 = 1u, kOpTip2 = 2u,   // child is a tip: expand its state byte
  kOpF11 = 4u, kOpF12 = 8u,     // child 1 is the result of the previous / second-previous operation
  kOpF21 = 16u, kOpF22 = 32u,   // child 2 likewise
  // child 1 / child 2 is a VIRTUAL tip x tip result (phyhip_host.hpp) computed inside this operation's step from its two tips
  // and matrices (lane-per-pattern nucleotide kernel, list form; at most one such child per operation).  Its record slots:
  // data.base = byte offsets of its two matrices (low | high word), data.bytes = 0, scale = its first tip's row, tip = its
  // second tip's row with x = 1
  kOpCh1 = 64u, kOpCh2 = 128u
};

// Everything a traversal launch needs.  Passed by value (fits the 4 KiB kernarg segment easily).
struct TreeParams
{
  double         *partials;   // internal buffer b at (b - tip_count) * P * C * S
  int            *scales;     // internal buffer b at (b - tip_count) * P
  const double   *wght;       // [P]
  long long       P;
  long long       Ppad;       // patterns per scale vector (P, or P rounded up to 16 for the fragment-major layout)
  int             perm;       // 0 host layout; 1: 20-state fragment-major (phyhip_aa.hpp); 2: 4-state pattern-minor (phyhip_nt2.hpp)
  int             C;
  int             tip_count;
  int             apply_scaling;
  int             n_ops;
  int             last_dest;  // destination buffer of the last queued operation (-1: none)
  // fused root-edge evaluation (K2); 2: the eigen-basis products of Update_Eigen_Lr (K3) for the edge (e_parent = left side,
  // e_child = right side) instead -- lane-per-pattern nucleotide kernel only: the partial update Br_Len_Opt queues in front of
  // Update_Eigen_Lr and the products are ONE launch (dot_out, eigen system in m_evec / m_ivec; completion: an empty record per workgroup)
  int             edge_eval;
  double             *dot_out;
  int             e_parent, e_child, e_pm;
  const double   *pi;
  const double   *cat_w;
  int             invar_model;
  double          pinvar;
  const short    *invar;
  double         *site_lnl;   // c_lnL_sorted      (may be null)
  double         *site_lk;    // cur_site_lk       (may be null)
  double         *site_cat;   // unscaled_site_lk_cat [P][C] (may be null)
  int            *fact;       // fact_sum_scale [P] (always written: dLk needs it)
  double         *block_sums; // [gridDim.x]
  // fused final reduction (result != nullptr): the last workgroup to finish adds the block sums in the fixed order
  // of final_reduce_kernel and hands lnL, the warning flag and the sequence number to the host
  unsigned           *tickets;     // one counter, zero between launches
  double             *result;      // device lnL (or the caller's device pointer)
  double             *result_host; // host-mapped {lnL, -, seq} or nullptr
  int                *warn_host;
  unsigned long long  seq;
  int            *warn;
  double             *warn_out;    // sharded evaluation: the flag as a double next to the shard's sum (rides in the all-reduce)
  // Short lists of transition matrices to rebuild (SPR: three per candidate, src/spr.c:643-646) ride here and are built
  // by every workgroup of the lane-per-pattern nucleotide kernel in its prologue: no separate pmat_kernel launch, no
  // dependent-launch gap.  (pmat_kernel's arithmetic, src/models.c:257-326.)
  int             n_fresh;
  int             fresh_idx[8];
  double          fresh_len[8];
  double          m_evec[16], m_ivec[16], m_eval[4], m_rates[4]; // U, U^-1, eigenvalues, category rates (4 states, <= 4 categories)
  double          br_len_mult, l_min, l_max;
  double         *pmats_rw;
  // Host-computed matrices (phyhip_set_transition_matrix, the bit-exact route of src/lk.c:2360) queued since the last launch,
  // up to kArgUp of them (SPR: three per candidate): they ride here, every workgroup of the lane-per-pattern nucleotide
  // kernel writes them to their slots in its prologue (same values everywhere) and reads them back in program order -- no
  // upload kernel in front of the traversal, nothing to wait for.
  int             n_up;
  int             up_idx[3];
  double          up_val[3][64]; // [matrix][c * 16 + i * 4 + j], C <= 4
  // One- and two-operation launches (an SPR regraft candidate is ONE update + the edge evaluation, src/spr.c:643-646)
  // carry their operation records here, in the kernel arguments, instead of a staged copy into the device slot ring: one
  // copy command less in front of the launch.
  int             recs_in_args;
  int             n_real_ops;   // operations before padding (records in the arguments: an odd list is NOT re-executed)
  int             fence_post;   // a workgroup's stores are complete and written back before it posts its sum (see flush_impl)
  int             e_prefetch;   // bit 0 / 1: the evaluation edge's parent / child side is an internal buffer no queued operation writes
  IssueRec        arg_ir[2];
  ExecRec         arg_xr[2];
  // Large grids, scalar wanted on the host: every workgroup posts {block sum, tag} straight into host-mapped memory and
  // retires; the HOST adds the block sums (same grouping and order as final_reduce_kernel).  No second launch, no ticket
  // draw, no waiting for the workgroup's result stores (see fuse_reduce() in phyhip.hip for what those cost).
  HostBlock      *host_blocks;   // [gridDim.x] or nullptr
  unsigned long long host_tag;   // sequence number of this evaluation
  // Class axis (mixtures of class models on ONE instance, src/mixt.c): "category" c is class c of the mixture -- its own
  // eigen system / frequencies (pi[c][S], matrices built per class), its own scale vector per partials buffer
  // ([buffer][class][pattern]: every class tree rescales on its own, src/mixt.c:2603-2640), no mixing over categories in
  // the edge evaluation (per-class likelihoods and scale exponents go out, class_combine_kernel repeats MIXT_Lk's site loop).
  int             class_axis;
  // Large-grid resident evaluator (phyhip_big.hpp) with the final sum on the device: a tile's sum goes here (device memory,
  // [tile]) instead of to the host; the workgroup that finishes last adds them in final_reduce_kernel's order and posts ONE record.
  // (tile_sums_in_wave(): the same evaluator adding per workgroup, kBigGroupSum -- the sum stays in the wave's LDS area)
  double         *tile_sums;
  // PHYHIP_FLAG_GENERIC_LOOP: the arithmetic of Update_Partial_Lk_Generic under mod->use_m4mod (`phyml --cov`): no all-ones
  // shortcut (src/lk.c:1463-1528 has none; src/avx.c:575-587 does)
  int             generic_loop;
  // 20-state kernel, lists with in-step tip x tip children (traverse_aa_kernel<..., INL>): where the evaluation edge's matrix
  // tables sit in the LDS ring (first table slot) and how many items must have been released before they may be written
  int             aa_e_slot, aa_e_need;
  int             aa_tpw;   // wave-tiles per workgroup (traverse_aa_kernel with two tiles per consumer wave)
};

constexpr int kBigGroupWgs = 256; // workgroups of the large-grid resident evaluator when it adds per workgroup (phyhip_big.hpp)
__device__ __forceinline__ double *tile_sums_in_wave() { return reinterpret_cast<double *>(8ull); } // (no buffer's address)

// ---------------------------------------------------------------------------------------------
// cross-lane helpers over the CP adjacent lanes that share a pattern
// ---------------------------------------------------------------------------------------------
template <int CP> __device__ __forceinline__ double group_max(double v)
{
#pragma unroll
  for (int off = 1; off < CP; off <<= 1) v = fmax(v, __shfl_xor(v, off, CP));
  return v;
}
// The rescaling test `max < 2^-256` (src/avx.c:498-506) on non-negative doubles only needs the HIGH 32 bits:
// for x >= 0, x < 2^-256  <=>  hi32(x) < 0x2FF00000 (2^-256 has biased exponent 767 = 0x2FF and a zero
// mantissa), and unsigned order of hi32 is the order of the values' magnitudes.  Partial likelihoods are
// products of probabilities (never negative; a NaN has hi32 >= 0x7FF00000 and so never triggers a rescale, as
// the reference's `>` scan never selects it).  One v_max3_u32 + two DPP quad permutes replace four FP64
// compare/select pairs, two canonicalising v_max_f64 and two LDS ds_bpermute round trips.
constexpr unsigned kHiInvTwoToLarge = 0x2FF00000u;
__device__ __forceinline__ unsigned hi32(double x) { return (unsigned)(__double_as_longlong(x) >> 32); }
template <int CP> __device__ __forceinline__ unsigned group_max_u32(unsigned v)
{
  if (CP >= 2) v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
  if (CP >= 4) v = max(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
#pragma unroll
  for (int off = 4; off < CP; off <<= 1) v = max(v, (unsigned)__shfl_xor((int)v, off, CP));
  return v;
}
template <int CP> __device__ __forceinline__ int group_bcast0(int v) { return CP == 1 ? v : __shfl(v, 0, CP); }
template <int CP> __device__ __forceinline__ int group_and(int v)
{
#pragma unroll
  for (int off = 1; off < CP; off <<= 1) v &= __shfl_xor(v, off, CP);
  return v;
}

// ---------------------------------------------------------------------------------------------
// Read-only inputs travel as separate __restrict__ kernel parameters (not inside TreeParams): that
// gives them the `noalias` attribute, so the compiler may use scalar loads for wave-uniform data (the
// operation list) and need not order these loads against the partial-vector stores.
// ---------------------------------------------------------------------------------------------
struct RO
{
  const DevOp    *__restrict__ ops;        // operation list
  const double   *__restrict__ pmats;      // matrix m at m * C * S * S, [c][from][to]
  const uint8_t  *__restrict__ tip_codes;  // tip t at t * P : index into code_masks
  const uint32_t *__restrict__ code_masks; // allowed-state bit mask per code
};

// 20-state fragment-major layout (phyhip_aa.hpp).  A wave of the amino-acid kernel owns a "wave-tile": the MFMA
// v_mfma_f64_4x4x4_4b_f64 has four blocks of four columns; a pattern's categories sit in aa_cb(C) consecutive blocks
// (C = 3 leaves one idle), so a wave-tile is 16 / aa_cb(C) patterns x all categories = one block of 320 doubles.
// Lane = 16 (state & 3) + 4 block + (pattern & 3) owns the five k-chunk values (states 4t + (state & 3)) of its
// (pattern, category): stored as two 16-byte pairs (chunks 0|1 and 2|3) and one single (chunk 4), each group contiguous
// over the 64 lanes, so a fragment set moves as 2 x dwordx4 + 1 x dwordx2.
__host__ __device__ inline int aa_cb(int C) { return C == 1 ? 1 : (C == 2 ? 2 : 4); }
__host__ __device__ inline int aa_slot(int t, int lane) { return t < 4 ? (t >> 1) * 128 + lane * 2 + (t & 1) : 256 + lane; }
// element offset of (pattern p, category c, state s) inside a fragment-major buffer
__host__ __device__ inline size_t aa_off(long long p, int C, int c, int s)
{
  const int       cb = aa_cb(C), npw = 16 / cb;
  const long long w  = p / npw;
  const int       r  = (int)(p % npw);
  const int       lane = 16 * (s & 3) + 4 * ((r >> 2) * cb + c) + (r & 3);
  return (size_t)w * 320 + (size_t)aa_slot(s >> 2, lane);
}
// doubles per partials buffer of Ppad patterns (Ppad a multiple of 16)
__host__ __device__ inline size_t aa_buf_elems(long long Ppad, int C) { return (size_t)Ppad * aa_cb(C) * 20; }
// A-operand table of one matrix (ALL its categories): 25 values per lane -- row group r (output states 4r + i) x k-chunk t
// (input states 4t + k), lane = 16k + 4 block + i -- stored per k-chunk as the five row groups in aa_slot order, so the
// consumer's LDS reads per chunk are 2 x ds_read_b128 + 1 x ds_read_b64, conflict-free
constexpr int kAaMat = 1600;
__host__ __device__ inline int aa_a_slot(int t, int r, int lane) { return t * 320 + aa_slot(r, lane); }

// child fetch in two phases so that ALL loads of an operation are in flight before the first wait:
//   issue_side  -- only issues the loads (tip: one code byte; internal: S doubles + the scale word)
//   finish_side -- turns the raw values into the S-vector and the pattern's scale exponent
template <int S> struct RawSide
{
  double2  v[S / 2];
  int      sc;
  unsigned code;
  bool     tip;
};

template <int S, int CP>
__device__ __forceinline__ void issue_side(const TreeParams &q, const RO &ro, int idx, long long p, int c, RawSide<S> &r)
{
  r.tip = idx < q.tip_count; // uniform: idx is the same for the whole grid
  r.sc = 0;
  r.code = 0;
  if (r.tip)
    r.code = ro.tip_codes[(size_t)idx * q.Ppad + p];
  else if (S == 4 && q.perm == 2)
  { // pattern-minor layout of the lane-per-pattern nucleotide kernel (phyhip_nt2.hpp): the state pair (2h, 2h+1) of
    // category c of pattern p is one 16-byte element at ((c*2 + h) * Ppad + p)
    const size_t   b    = (size_t)(idx - q.tip_count);
    const double2 *base = reinterpret_cast<const double2 *>(q.partials + b * (size_t)q.Ppad * (q.C * S)) + (size_t)(c * 2) * q.Ppad + p;
#pragma unroll
    for (int j = 0; j < S / 2; ++j) r.v[j] = base[(size_t)j * q.Ppad];
    if (c == 0) r.sc = q.scales[b * q.Ppad + p];
  }
  else if (S == 20 && q.perm)
  { // fragment-major layout (aa_off)
    const size_t  b    = (size_t)(idx - q.tip_count);
    const double *base = q.partials + b * aa_buf_elems(q.Ppad, q.C);
#pragma unroll
    for (int j = 0; j < S / 2; ++j)
    {
      r.v[j].x = base[aa_off(p, q.C, c, 2 * j)];
      r.v[j].y = base[aa_off(p, q.C, c, 2 * j + 1)];
    }
    if (c == 0) r.sc = q.scales[b * q.Ppad + p];
  }
  else
  {
    const size_t   b  = (size_t)(idx - q.tip_count);
    const double2 *s2 = reinterpret_cast<const double2 *>(q.partials + (b * q.P + p) * (size_t)(q.C * S) + (size_t)c * S);
    static_assert(S % 2 == 0, "state count must be even for 16-byte loads");
#pragma unroll
    for (int j = 0; j < S / 2; ++j) r.v[j] = s2[j];
    // the scale word is written by the c==0 lane of the pattern; read it back through the same lane
    if (c == 0) r.sc = q.scales[b * q.Ppad + p];
  }
}

template <int S, int CP>
__device__ __forceinline__ void finish_side(const RO &ro, const RawSide<S> &r, double (&x)[S], int &sc)
{
  if (r.tip)
  {
    // S <= 8: the stored byte IS the allowed-state mask; wider alphabets index the mask table
    const uint32_t m = (S <= 8) ? r.code : ro.code_masks[r.code];
#pragma unroll
    for (int j = 0; j < S; ++j) x[j] = ((m >> j) & 1u) ? 1.0 : 0.0;
    sc = 0;
  }
  else
  {
#pragma unroll
    for (int j = 0; j < S / 2; ++j)
    {
      x[2 * j]     = r.v[j].x;
      x[2 * j + 1] = r.v[j].y;
    }
    sc = group_bcast0<CP>(r.sc);
  }
}

template <int S, int CP>
__device__ __forceinline__ void load_side(const TreeParams &q, const RO &ro, int idx, long long p, int c, double (&x)[S], int &sc)
{
  RawSide<S> r;
  issue_side<S, CP>(q, ro, idx, p, c, r);
  finish_side<S, CP>(ro, r, x, sc);
}

// u[i] = sum_j M[i*S+j] * x[j] in the reference's AVX order: first product, then an FMA chain over j
// (src/avx.c:593-616).  Bit-identical to the AVX kernel given the same matrix.
template <int S>
__device__ __forceinline__ void matvec_rows(const double *__restrict__ M, const double (&x)[S], double (&u)[S])
{
  constexpr int kOuter = (S <= 4) ? S : 1; // wide alphabets: keep one row of the matrix live at a time
#pragma unroll kOuter
  for (int i = 0; i < S; ++i)
  {
    double a = M[i * S] * x[0];
#pragma unroll
    for (int j = 1; j < S; ++j) a = __builtin_fma(M[i * S + j], x[j], a);
    u[i] = a;
  }
}

// ---------------------------------------------------------------------------------------------
// K1 + K2: traversal kernel
// ---------------------------------------------------------------------------------------------
// ---- end of an evaluation: per-workgroup sums -> one scalar (or two) on the host ---------------------------------
constexpr unsigned kTicketGroups = 32;
struct FinishParams
{
  unsigned           *tickets;     // [1 + kTicketGroups] counters (top | groups), zero between launches
  double             *block_sums;  // [NS][stride]
  int                 stride;
  double             *result;      // device scalars [NS] (or the caller's device pointer); nullptr: final_reduce_kernel follows
  double             *result_host; // host-mapped {sum 0, sum 1, seq} or nullptr
  int                *warn, *warn_host;
  unsigned long long  seq;
  double             *warn_out;    // device double that receives the warning flag (sharded evaluation) or nullptr
  // Scalar wanted on the host: every workgroup posts {sum k, tag} for each of its NS sums straight into host-mapped memory
  // (record k * stride + workgroup) and retires; the host polls the tags and adds the sums in final_reduce_kernel's order.
  // No ticket draw, no second launch: the shortest path from a workgroup's sum to the caller, at every grid size.
  HostBlock          *host_blocks;
  unsigned long long  host_tag;
};

__device__ __forceinline__ void raise_warn(int *warn)
{ // visible to whichever workgroup ends up doing the final reduction (other XCD, other L2) -- or to the host, when the
  // flag lives in host-mapped memory (host-side final sum)
  __hip_atomic_store(warn, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // ... and performed before this wave goes on: the workgroup's {sum, tag} record, which tells the host that the evaluation
  // is complete, is posted later (program order, or behind the workgroup's barrier) and must not overtake it.  Rare path.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
}

__device__ __forceinline__ void post_host_block(HostBlock *dst, double sum, unsigned long long tag)
{
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  u64x2 rec;
  __builtin_memcpy(&rec, &sum, 8);
  rec.y = tag;
  // one 16-byte write: sum and tag arrive together.  System scope (sc0 sc1): written through at once -- a plain or
  // non-temporal store may sit in the L2 until the kernel ends, which a resident workgroup never does (measured: the host
  // saw the records of resident_dlk_kernel only when it left, an idle time-out later).
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(rec) : "memory");
}

// Called by every lane of ONE wave per workgroup with the workgroup's sums in lane 0.  With f.host_blocks each sum is posted to the host and the workgroup retires.  Without f.result the sums are
// only recorded (final_reduce_kernel follows).  With it, they are stored through to memory (agent scope), a ticket is
// drawn, and the workgroup that draws the last ticket adds all block sums -- same grouping and order as
// final_reduce_kernel, so the result does not depend on which workgroup finishes last -- and publishes it: one kernel
// launch and one inter-kernel gap less per scalar-returning call.
template <int NS> __device__ __forceinline__ void finish_sums(const FinishParams &f, const double (&s)[NS], int lane)
{
  if (f.host_blocks)
  {
    if (lane == 0)
#pragma unroll
      for (int k = 0; k < NS; ++k) post_host_block(f.host_blocks + (size_t)k * f.stride + blockIdx.x, s[k], f.host_tag);
    return;
  }
  if (!f.result)
  {
    if (lane == 0)
#pragma unroll
      for (int k = 0; k < NS; ++k) f.block_sums[(size_t)k * f.stride + blockIdx.x] = s[k];
    return;
  }
  unsigned ticket = 0;
  if (lane == 0)
  {
#pragma unroll
    for (int k = 0; k < NS; ++k)
    {
      unsigned long long bits;
      __builtin_memcpy(&bits, &s[k], 8);
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(f.block_sums) + (size_t)k * f.stride + blockIdx.x, bits,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(0); // the write-through stores are acknowledged before the ticket is drawn
    // Two-level ticket draw.  Atomics on ONE address serialise (~15 ns each): with one counter the ~1500 one-wave
    // workgroups of a 50 000-pattern nucleotide launch, which all finish within a few microseconds of each other, queued
    // for 23 us (measured: 209 vs 186 us kernel time).  Workgroups draw from one of kTicketGroups counters; the last one
    // of each group draws from the top counter: at most gridDim / kTicketGroups + kTicketGroups draws deep.
    const unsigned g = blockIdx.x % kTicketGroups, ng = gridDim.x < kTicketGroups ? gridDim.x : kTicketGroups;
    const unsigned members = (gridDim.x - g + kTicketGroups - 1) / kTicketGroups;
    ticket = 0;
    if (__hip_atomic_fetch_add(f.tickets + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1)
      ticket = __hip_atomic_fetch_add(f.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1 ? 1u : 0u;
  }
  ticket = __shfl(ticket, 0, 64);
  if (!ticket) return;
  const int n = (int)gridDim.x;
  // every block sum was written through to memory before its ticket was drawn: read them with atomic loads at agent scope
  // (they bypass whatever stale copies this CU's L1 / this XCD's L2 may hold).  NOT an acquire fence and ordinary loads: the
  // fence invalidates the L2, which first writes back every dirty line in it -- with a traversal's results there, tens of
  // microseconds (measured in round 4 on the resident evaluators: one such fence per command cost 20 us at 100 000 patterns)
  double tot[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k)
  {
    const double *in = f.block_sums + (size_t)k * f.stride;
    double        acc[4]; // the four "threads" lane, lane+64, lane+128, lane+192 of the 256-thread reduction
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = 0.0;
    // (eight rounds of loads in flight at once: one after the other -- a dependent trip to memory each -- the 13 rounds of a
    // 3 126-workgroup evaluation took 27 us, measured in round 4; each accumulator still adds its terms in ascending order)
    for (int i0 = lane; i0 < n; i0 += 8 * 256)
    {
      double v[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
          unsigned long long bits = 0ull;
          if (i0 + u * 256 + 64 * j < n)
            bits = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(in) + i0 + u * 256 + 64 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __builtin_memcpy(&v[u][j], &bits, 8);
        }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (i0 + u * 256 + 64 * j < n) acc[j] += v[u][j];
    }
    double t = (acc[0] + acc[2]) + (acc[1] + acc[3]); // tree levels 128 and 64
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
    tot[k] = t;
  }
  if (lane == 0)
  {
#pragma unroll
    for (int k = 0; k < NS; ++k) f.result[k] = tot[k];
    const int w = __hip_atomic_load(f.warn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(f.warn, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 1
    for (int k = 0; k <= kTicketGroups; ++k) __hip_atomic_store(f.tickets + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (f.warn_host) __hip_atomic_store(f.warn_host, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (f.warn_out) *f.warn_out = (double)w;
    if (f.result_host)
    { // results and flag written through at system scope and ACKNOWLEDGED, then the sequence number: the same order as a
      // release fence gives, without the fence's write-back of this XCD's whole L2 (megabytes of a traversal's results: tens of
      // microseconds on large alignments)
#pragma unroll
      for (int k = 0; k < NS; ++k)
      {
        unsigned long long bits;
        __builtin_memcpy(&bits, &tot[k], 8);
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(f.result_host) + k, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __builtin_amdgcn_s_waitcnt(0);
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(f.result_host + 2), f.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__device__ __forceinline__ void raise_warn(const TreeParams &q) { raise_warn(q.warn); }

// `tile`: which of the evaluation's per-tile sums this is -- the workgroup's number in a launched kernel, the tile a wave of
// the large-grid resident evaluator has just finished
__device__ __forceinline__ void publish_block_sum(const TreeParams &q, double s, int lane, unsigned tile)
{
  if (q.tile_sums)
  { // (written through: the workgroup that adds the tile sums may sit on another XCD)
    if (lane == 0)
    {
      unsigned long long bits;
      __builtin_memcpy(&bits, &s, 8);
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(q.tile_sums) + tile, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (q.host_blocks)
  {
    if (q.fence_post) __threadfence(); // (this wave's stores; kernels with several waves per workgroup fence before their last barrier)
    if (lane == 0) post_host_block(q.host_blocks + tile, s, q.host_tag);
    return;
  }
  FinishParams f;
  f.host_blocks = nullptr; f.host_tag = 0;
  f.tickets = q.tickets; f.block_sums = q.block_sums; f.stride = 0; f.result = q.result; f.result_host = q.result_host;
  f.warn = q.warn; f.warn_host = q.warn_host; f.seq = q.seq; f.warn_out = q.warn_out;
  const double v[1] = {s};
  finish_sums<1>(f, v, lane);
}
__device__ __forceinline__ void publish_block_sum(const TreeParams &q, double s, int lane) { publish_block_sum(q, s, lane, blockIdx.x); }

template <int S, int CP>
__global__ __launch_bounds__(256) void traverse_kernel(const TreeParams q, const DevOp *__restrict__ ops_,
                                                       const double *__restrict__ pmats_,
                                                       const uint8_t *__restrict__ tip_codes_,
                                                       const uint32_t *__restrict__ code_masks_)
{
  const RO ro{ops_, pmats_, tip_codes_, code_masks_};
  const long long gl  = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long p0  = gl / CP;
  const int       c0  = (int)(gl % CP);
  const bool      act = (p0 < q.P) && (c0 < q.C);
  const long long p   = (p0 < q.P) ? p0 : (q.P - 1); // clamp so idle lanes read valid memory
  const int       c   = (c0 < q.C) ? c0 : 0;
  const int       CS  = q.C * S;
  const int       MS  = q.C * S * S;

  for (int k = 0; k < q.n_ops; ++k)
  {
    const DevOp op = ro.ops[k];
    double      x1[S], x2[S], u1[S], u2[S];
    int         s1, s2;
    const double *__restrict__ M1 = ro.pmats + (size_t)op.pm1 * MS + (size_t)c * S * S;
    const double *__restrict__ M2 = ro.pmats + (size_t)op.pm2 * MS + (size_t)c * S * S;
    RawSide<S> r1, r2;
    issue_side<S, CP>(q, ro, op.c1, p, c, r1);
    issue_side<S, CP>(q, ro, op.c2, p, c, r2);
    // small alphabets: pull this lane's two 4x4 blocks into registers while the children are in flight
    double m1[(S <= 4) ? S * S : 1], m2[(S <= 4) ? S * S : 1];
    if (S <= 4)
    {
      const double2 *a = reinterpret_cast<const double2 *>(M1), *b = reinterpret_cast<const double2 *>(M2);
#pragma unroll
      for (int j = 0; j < (S * S) / 2; ++j)
      {
        const double2 va = a[j], vb = b[j];
        m1[(2 * j) % (S * S)] = va.x; m1[(2 * j + 1) % (S * S)] = va.y;
        m2[(2 * j) % (S * S)] = vb.x; m2[(2 * j + 1) % (S * S)] = vb.y;
      }
    }
    finish_side<S, CP>(ro, r1, x1, s1);
    finish_side<S, CP>(ro, r2, x2, s2);

    // all-ones shortcut of the Inin kernel (src/avx.c:575-587): exact 1.0 when both children are 1.0
    bool ones = true;
#pragma unroll
    for (int j = 0; j < S; ++j) ones = ones && (x1[j] == 1.0) && (x2[j] == 1.0);
    ones = ones && !q.generic_loop;

    if (S <= 4)
    {
      matvec_rows<S>(m1, x1, u1);
      matvec_rows<S>(m2, x2, u2);
    }
    else
    {
      matvec_rows<S>(M1, x1, u1);
      matvec_rows<S>(M2, x2, u2);
    }

    double o[S];
    double mx = -__builtin_huge_val();
#pragma unroll
    for (int i = 0; i < S; ++i)
    {
      o[i] = ones ? 1.0 : u1[i] * u2[i];
      mx   = (o[i] > mx) ? o[i] : mx; // `>` like src/avx.c:500-502: a NaN never becomes the maximum
    }
    if (c0 >= q.C) mx = -__builtin_huge_val();
    mx = group_max<CP>(mx);

    int sc = s1 + s2; // src/avx.c:462-464
    if (mx < kInvTwoToLarge && q.apply_scaling)
    { // src/avx.c:504-510; multiplication by 2^256 is exact
#pragma unroll
      for (int i = 0; i < S; ++i) o[i] *= kTwoToLarge;
      sc += kLarge;
    }

    if (act)
    {
      const size_t b   = (size_t)(op.dest - q.tip_count);
      double2     *dst = reinterpret_cast<double2 *>(q.partials + (b * q.P + p) * (size_t)CS + (size_t)c * S);
      // the reference's generic loop writes zeros for a pattern without weight and leaves its scale exponent alone
      // (src/lk.c:1405,1581-1584); the SIMD kernels leave such a pattern untouched (src/avx.c:399) -- its values are never used
      const bool zero_w = q.generic_loop && !(q.wght[p] > kSmall);
#pragma unroll
      for (int j = 0; j < S / 2; ++j) dst[j] = zero_w ? make_double2(0.0, 0.0) : make_double2(o[2 * j], o[2 * j + 1]);
      if (c == 0 && !zero_w) q.scales[b * q.Ppad + p] = sc;
    }
  }

  if (!q.edge_eval) return;

  // ---- K2: site likelihood at the evaluation edge (src/lk.c:608-645, 767-861) -----------------
  double contrib = 0.0;
  {
    double x[S], y[S];
    int    sl, sr;
    RawSide<S> rl, rr;
    issue_side<S, CP>(q, ro, q.e_parent, p, c, rl);
    issue_side<S, CP>(q, ro, q.e_child, p, c, rr);
    finish_side<S, CP>(ro, rl, x, sl);
    finish_side<S, CP>(ro, rr, y, sr);
    const double *__restrict__ M = ro.pmats + (size_t)q.e_pm * MS + (size_t)c * S * S; // rows: right-side state
    // acc[k] = sum_i P[k][i] x[i] as an FMA chain from zero (src/avx.c:130-145, 191-206);
    // lk_c = sum_k acc[k] * (y[k]*pi[k]) with the 4-wide horizontal order of AVX_Vect_Norm.
    double lkc = 0.0;
    constexpr int kOuter = (S <= 4) ? 1 : 1;
#pragma unroll kOuter
    for (int b4 = 0; b4 < S / 4; ++b4)
    {
      double t[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
      {
        const int k = b4 * 4 + kk;
        double    a = 0.0;
#pragma unroll
        for (int i = 0; i < S; ++i) a = __builtin_fma(M[k * S + i], x[i], a);
        t[kk] = a * (y[k] * q.pi[k]);
      }
      const double nrm = (t[0] + t[2]) + (t[1] + t[3]);
      lkc              = (b4 == 0) ? nrm : lkc + nrm;
    }
    if (act && q.site_cat) q.site_cat[(size_t)p * q.C + c] = lkc; // Pull_Scaling_Factors copy, src/lk.c:2801

    // mixture over categories in category order (src/lk.c:816-818)
    const double t    = (c0 < q.C) ? lkc * q.cat_w[c] : 0.0;
    double       site = 0.0;
#pragma unroll
    for (int cc = 0; cc < CP; ++cc)
    {
      const double tc = (CP == 1) ? t : __shfl(t, cc, CP);
      if (cc < q.C) site += tc;
    }

    if (act && c == 0)
    {
      const double w = q.wght[p];
      int          f = q.apply_scaling ? (sl + sr) : 0; // SCALE_FAST, src/lk.c:2777-2794 / :2701-2705
      if (w > kSmall)
      {
        if (q.invar_model)
        { // src/lk.c:820-842 with Invariant_Lk :1226-1273
          const int iv  = q.invar[p];
          double    inv = 0.0;
          bool      issue = false;
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
            issue = isinf(inv);
          }
          if (issue)
          {
            f    = 0;
            site = q.pi[iv] * q.pinvar;
          }
          else
            site = site * (1. - q.pinvar) + inv * q.pinvar;
        }
        if (site < kSmall)
        { // src/lk.c:847-851
          site = kSmall;
          raise_warn(q);
        }
        const double lsl = log(site) - kLog2 * (double)f; // src/lk.c:854
        if (q.site_lnl) q.site_lnl[p] = lsl;
        if (q.site_lk) q.site_lk[p] = dev_exp(lsl);
        contrib = w * lsl; // src/lk.c:856
      }
      q.fact[p] = f;
    }
  }

  // deterministic block reduction: fixed shuffle tree per wave, then wave order
  __shared__ double wsum[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) contrib += __shfl_down(contrib, off, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) wsum[wid] = contrib;
  __syncthreads();
  if (wid == 0)
  {
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += wsum[w];
    publish_block_sum(q, s, lane);
  }
}

// ---------------------------------------------------------------------------------------------
// K1 + K2, nucleotide specialisation (S = 4): software-pipelined traversal.
//
// The generic kernel above pays one full memory round trip per operation (the chain is latency-,
// not bandwidth-bound at 50k patterns: ~3 waves per SIMD).  This version
//   * is branch-free inside the operation loop: tip / internal / forwarded children differ only in
//     which address the (always issued) loads use and in a wave-uniform select afterwards, so the
//     compiler keeps every load of an operation in flight together;
//   * forwards the previous operation's result in registers when it is a child of the next one (in a
//     post-order walk the parent usually follows its last child immediately), which removes that
//     read and its store->load round trip;
//   * issues the loads of operation k+1 (children that are not forwarded, tip bytes, and this lane's
//     16-byte piece of the two transition matrices) BEFORE computing operation k;
//   * stages the two matrices of an operation through a per-wave LDS double buffer: one 16-byte
//     global load per lane instead of sixteen, then broadcast ds_read_b128s.
// ---------------------------------------------------------------------------------------------
// ABL != 0 builds timing-only ablation variants (PHYHIP_ABLATE): bit0 no stores, bit1 no matrix-vector work,
// bit2 no LDS staging.  Results are wrong by construction; used to attribute kernel time.
template <int CP, int ABL = 0, int DIST = 2>
__global__ __launch_bounds__(256, (DIST == 2 ? 3 : 4)) void traverse_nt_kernel(const TreeParams q, const IssueRec *__restrict__ irec,
                                                             const ExecRec *__restrict__ xrec,
                                                             const double *__restrict__ pmats,
                                                             const uint8_t *__restrict__ tip_codes)
{
  constexpr int S   = 4;
  constexpr int NCH = (16 * CP + 63) / 64;           // 16-byte matrix pieces per lane and operation
  __shared__ __attribute__((aligned(16))) double2 lds_p[4][2][16 * CP]; // [wave][buffer][piece]

  const int       lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long long gl   = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long p0   = gl / CP;
  const int       c0   = (int)(gl % CP);
  const bool      act  = (p0 < q.P) && (c0 < q.C);
  // Idle lanes (tail of the last block, categories beyond C) are clamped onto a real (pattern, category):
  // they compute and STORE exactly what that real lane does -- same value to the same address -- which
  // keeps every memory instruction in the loop unconditional (exact s_waitcnt counts, no exec branches).
  const long long p    = (p0 < q.P) ? p0 : (q.P - 1);
  const int       c    = (c0 < q.C) ? c0 : 0;
  const int       C    = q.C;
  const int       CS   = C * S, MS = C * S * S;
  const int       tips = q.tip_count;
  // Addressing: every buffer is reached through a 128-bit buffer descriptor held in SGPRs (base and size
  // come ready-made from the host's operation record) plus ONE 32-bit per-lane byte offset that is the
  // same for every partials buffer.  A load that is not needed (tip or forwarded child, tip byte of an
  // internal child) has size 0: the hardware bounds check returns 0 without touching memory, so the loop
  // body has neither branches nor conditional memory instructions.
  const unsigned poffb = (unsigned)(((size_t)p * CS + (size_t)c * S) * sizeof(double)); // bytes
  const unsigned pidx  = (unsigned)p;
  const size_t   bufsz = (size_t)q.P * CS;                                              // doubles per buffer

  struct Raw
  {
    u32x4    a0, a1, b0, b1; // child 1 / child 2 records (2 doubles each)
    unsigned sa, sb;         // scale words (every lane of the pattern reads and writes its own copy)
    unsigned ca, cb;         // tip bytes
  };
  const __amdgpu_buffer_rsrc_t pm_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(pmats), 0, 0x7fffffff, 0x00020000);
  auto rsrc = [](const Desc &d) {
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(d.base), 0, (int)d.bytes, 0x00020000);
  };

  // issue every load an operation needs (which ones are live was decided by the host)
  auto issue = [&](const IssueRec &o, Raw &r, u32x4 (&pc)[NCH]) {
    const __amdgpu_buffer_rsrc_t d1r = rsrc(o.c1_data), d2r = rsrc(o.c2_data), g1r = rsrc(o.c1_scale),
                                 g2r = rsrc(o.c2_scale), y1r = rsrc(o.c1_tip), y2r = rsrc(o.c2_tip);
    r.a0 = __builtin_amdgcn_raw_buffer_load_b128(d1r, poffb, 0, 0);
    r.a1 = __builtin_amdgcn_raw_buffer_load_b128(d1r, poffb + 16, 0, 0);
    r.b0 = __builtin_amdgcn_raw_buffer_load_b128(d2r, poffb, 0, 0);
    r.b1 = __builtin_amdgcn_raw_buffer_load_b128(d2r, poffb + 16, 0, 0);
    r.sa = __builtin_amdgcn_raw_buffer_load_b32(g1r, pidx * 4, 0, 0);
    r.sb = __builtin_amdgcn_raw_buffer_load_b32(g2r, pidx * 4, 0, 0);
    r.ca = __builtin_amdgcn_raw_buffer_load_b8(y1r, pidx, 0, 0);
    r.cb = __builtin_amdgcn_raw_buffer_load_b8(y2r, pidx, 0, 0);
#pragma unroll
    for (int h = 0; h < NCH; ++h)
    {
      int ch = lane + 64 * h;                    // piece index over [matrix 1 | matrix 2]
      ch     = (ch < 16 * C) ? ch : 0;
      const int      mat = ch / (8 * C), within = ch - mat * 8 * C;
      const unsigned off = (mat ? o.c2_data.x : o.c1_data.x) + (unsigned)within * 16u;
      pc[h] = __builtin_amdgcn_raw_buffer_load_b128(pm_rsrc, off, 0, 0);
    }
  };
  auto as_d2 = [](const u32x4 &v) { double2 d; __builtin_memcpy(&d, &v, 16); return d; };

  // results of the last two operations stay in registers: F1 = operation k-1 (buffer d1), F2 = k-2 (d2)
  double F1[S] = {0., 0., 0., 0.}, F2[S] = {0., 0., 0., 0.};
  int    sc1 = 0, sc2 = 0;

  if (q.n_ops > 0)
  {
    const int last = q.n_ops - 1;
    Raw       RA, RB;
    u32x4     PA[NCH], PB[NCH];
    issue(irec[0], RA, PA);
    if (DIST == 2) issue(irec[(1 < last) ? 1 : last], RB, PB);
    ExecRec  cur = xrec[0];
    IssueRec nx2 = irec[(DIST < last) ? DIST : last]; // load-stage record of the operation DIST steps ahead

    // One pipeline step.  (R, PC) hold the loads of operation k, issued two steps ago; after its operands
    // are extracted the same registers receive the loads of operation k+2.  The loop below alternates two
    // register sets, so no loaded value is ever copied (a copy would force a wait on loads still in flight).
    auto step = [&](const int k, const int parity, Raw &R, u32x4 (&PC)[NCH], Raw &Rn, u32x4 (&PCn)[NCH]) {
      // stage this operation's matrices into the wave's LDS buffer (same-wave write -> read, in order);
      // lanes beyond the last piece rewrite piece 0 with the identical bytes they loaded for it
      double2 *buf = &lds_p[wid][parity][0];
#pragma unroll
      for (int h = 0; h < NCH; ++h)
      {
        const int ch = lane + 64 * h;
        if (!(ABL & 4)) buf[(ch < 16 * C) ? ch : 0] = as_d2(PC[h]);
      }
      __builtin_amdgcn_wave_barrier();

      // ---- operands of operation k ----
      const unsigned fl = cur.dst_data.x;
      const bool     t1 = fl & kOpTip1, t2 = fl & kOpTip2;
      const bool     f11 = fl & kOpF11, f12 = (DIST == 2) && (fl & kOpF12), f21 = fl & kOpF21,
                     f22 = (DIST == 2) && (fl & kOpF22);
      double         x1[S], x2[S];
      int        s1, s2;
      {
        const unsigned m1 = R.ca, m2 = R.cb; // S <= 8: the byte is the allowed-state mask
        const double2  a0 = as_d2(R.a0), a1 = as_d2(R.a1), b0 = as_d2(R.b0), b1 = as_d2(R.b1);
        const double   l1[S] = {a0.x, a0.y, a1.x, a1.y}, l2[S] = {b0.x, b0.y, b1.x, b1.y};
#pragma unroll
        for (int j = 0; j < S; ++j)
        {
          x1[j] = t1 ? (((m1 >> j) & 1u) ? 1.0 : 0.0) : (f11 ? F1[j] : (f12 ? F2[j] : l1[j]));
          x2[j] = t2 ? (((m2 >> j) & 1u) ? 1.0 : 0.0) : (f21 ? F1[j] : (f22 ? F2[j] : l2[j]));
        }
        s1 = t1 ? 0 : (f11 ? sc1 : (f12 ? sc2 : (int)R.sa));
        s2 = t2 ? 0 : (f21 ? sc1 : (f22 ? sc2 : (int)R.sb));
      }

      // prefetch operation k+2 into the registers just freed (the host left out of its record whatever
      // operations k and k+1 are still to produce: those are forwarded from F1 / F2)
      if (DIST == 2) issue(nx2, R, PC);   // operation k+2 into the registers just freed
      else issue(nx2, Rn, PCn);           // operation k+1 into the other register set

      // scalar loads for the NEXT step, issued only now (scalar loads return out of order, so any wait on an
      // older one would also wait on these): load-stage record of operation k+DIST+1, compute-stage record of k+1
      const IssueRec nx3 = irec[(k + DIST + 1 < last) ? k + DIST + 1 : last];
      const ExecRec  nx1 = xrec[(k + 1 < last) ? k + 1 : last];

      // all-ones shortcut, src/avx.c:575-587.  Exact 1.0 in all eight entries only happens under fully
      // ambiguous subtrees; test two entries first and pay for the rest only when a lane of the wave passes
      bool ones = (x1[0] == 1.0) && (x2[0] == 1.0);
      if (__builtin_amdgcn_ballot_w64(ones))
      {
#pragma unroll
        for (int j = 1; j < S; ++j) ones = ones && (x1[j] == 1.0) && (x2[j] == 1.0);
      }

      // ---- this lane's two 4x4 blocks from LDS, then the AVX-ordered products (src/avx.c:593-616) ----
      double u1[S], u2[S];
      if (ABL & 2)
      {
        const double2 pv = as_d2(PC[0]);
#pragma unroll
        for (int i = 0; i < S; ++i) { u1[i] = x1[i] + pv.x; u2[i] = x2[i] + pv.y; }
      }
      else
      {
        // row i of a 4x4 block = two 16-byte pieces; first product, then the FMA chain (src/avx.c:593-616)
        auto rows = [&](const double2 *m, const double (&x)[S], double (&u)[S]) {
#pragma unroll
          for (int i = 0; i < S; ++i)
          {
            const double2 lo = m[2 * i], hi = m[2 * i + 1];
            double        a  = lo.x * x[0];
            a    = __builtin_fma(lo.y, x[1], a);
            a    = __builtin_fma(hi.x, x[2], a);
            a    = __builtin_fma(hi.y, x[3], a);
            u[i] = a;
          }
        };
        rows(buf + c * 8, x1, u1);          // matrix 1, category c: 8 pieces of 16 bytes
        rows(buf + C * 8 + c * 8, x2, u2);  // matrix 2
      }
      __builtin_amdgcn_wave_barrier();

      double o_[S];
#pragma unroll
      for (int i = 0; i < S; ++i) o_[i] = ones ? 1.0 : u1[i] * u2[i];
      // clamped lanes hold a copy of a real lane's values: the maximum is unaffected
      const unsigned mxh = group_max_u32<CP>(max(max(hi32(o_[0]), hi32(o_[1])), max(hi32(o_[2]), hi32(o_[3]))));
      int            osc = s1 + s2; // src/avx.c:462-464
      if (mxh < kHiInvTwoToLarge && q.apply_scaling)
      { // src/avx.c:504-510
#pragma unroll
        for (int i = 0; i < S; ++i) o_[i] *= kTwoToLarge;
        osc += kLarge;
      }
      {
        const __amdgpu_buffer_rsrc_t dr = rsrc(cur.dst_data), gr = rsrc(cur.dst_scale);
        u32x4 w0, w1;
        const double2 v0 = make_double2(o_[0], o_[1]), v1 = make_double2(o_[2], o_[3]);
        __builtin_memcpy(&w0, &v0, 16);
        __builtin_memcpy(&w1, &v1, 16);
        if (!(ABL & 1))
        {
          __builtin_amdgcn_raw_buffer_store_b128(w0, dr, poffb, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(w1, dr, poffb + 16, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32((unsigned)osc, gr, pidx * 4, 0, 0);
        }
      }
      // rotate results and descriptors (computed values and SGPRs only)
#pragma unroll
      for (int i = 0; i < S; ++i) { if (DIST == 2) F2[i] = F1[i]; F1[i] = o_[i]; }
      sc2 = sc1; sc1 = osc;
      cur = nx1;
      nx2 = nx3;
    };

    // the host pads the record list to an even length (see flush()), so the two-set alternation is exact
    for (int k = 0; k < q.n_ops; k += 2)
    {
      step(k, 0, RA, PA, RB, PB);
      step(k + 1, 1, RB, PB, RA, PA);
    }
  }
  const int    prev_dest = q.last_dest; // buffer whose value is still live in F1
  const int    osc = sc1;
  const double o_[S] = {F1[0], F1[1], F1[2], F1[3]};

  if (!q.edge_eval) return;

  // ---- K2 (same arithmetic as the generic kernel) ------------------------------------------------
  double contrib = 0.0;
  {
    double x[S], y[S];
    int    sl, sr;
    auto side = [&](int idx, double (&v)[S], int &sc) {
      if (idx < tips)
      {
        const unsigned m = tip_codes[(size_t)idx * q.Ppad + p];
#pragma unroll
        for (int j = 0; j < S; ++j) v[j] = ((m >> j) & 1u) ? 1.0 : 0.0;
        sc = 0;
      }
      else if (idx == prev_dest)
      {
#pragma unroll
        for (int j = 0; j < S; ++j) v[j] = o_[j];
        sc = osc;
      }
      else
      {
        const size_t   b  = (size_t)(idx - tips);
        const double2 *s2 = reinterpret_cast<const double2 *>(q.partials + b * bufsz + (size_t)p * CS + (size_t)c * S);
        const double2  v0 = s2[0], v1 = s2[1];
        v[0] = v0.x; v[1] = v0.y; v[2] = v1.x; v[3] = v1.y;
        sc = (q.scales + b * q.P)[pidx];
      }
    };
    side(q.e_parent, x, sl);
    side(q.e_child, y, sr);
    const double *__restrict__ M = pmats + (size_t)q.e_pm * MS + (size_t)c * S * S;
    double t[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
    {
      double a = 0.0;
#pragma unroll
      for (int i = 0; i < S; ++i) a = __builtin_fma(M[kk * S + i], x[i], a);
      t[kk] = a * (y[kk] * q.pi[kk]);
    }
    const double lkc = (t[0] + t[2]) + (t[1] + t[3]);
    if (act && q.site_cat) q.site_cat[(size_t)p * C + c] = lkc;
    const double tw   = (c0 < C) ? lkc * q.cat_w[c] : 0.0;
    double       site = 0.0;
#pragma unroll
    for (int cc = 0; cc < CP; ++cc)
    {
      const double tc = (CP == 1) ? tw : __shfl(tw, cc, CP);
      if (cc < C) site += tc;
    }
    if (act && c == 0)
    {
      const double w = q.wght[p];
      int          f = q.apply_scaling ? (sl + sr) : 0;
      if (w > kSmall)
      {
        if (q.invar_model)
        {
          const int iv = q.invar[p];
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
        if (q.site_lnl) q.site_lnl[p] = lsl;
        if (q.site_lk) q.site_lk[p] = dev_exp(lsl);
        contrib = w * lsl;
      }
      q.fact[p] = f;
    }
  }
  __shared__ double wsum[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) contrib += __shfl_down(contrib, off, 64);
  if (lane == 0) wsum[wid] = contrib;
  __syncthreads();
  if (wid == 0)
  {
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += wsum[w];
    publish_block_sum(q, s, lane);
  }
}

// Second stage: one block sums `n` per-block values of up to two interleaved streams in a fixed order.
// out[k] = sum_i in[k*stride + i].  Results, the numerical-warning flag and finally a sequence number go to
// host-mapped memory: the host spins on the sequence number instead of paying a stream-synchronise wake-up
// (SPR enters the surface ~10^5 times per search, each time waiting for one scalar).
static __global__ __launch_bounds__(256) void final_reduce_kernel(const double *__restrict__ in, int n, int nstreams, int stride,
                                                          double *__restrict__ out, double *__restrict__ out_host,
                                                          int *warn, int *warn_host, unsigned long long seq,
                                                          double *warn_out)
{
  __shared__ double sh[256];
  for (int k = 0; k < nstreams; ++k)
  {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += in[(size_t)k * stride + i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1)
    {
      if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0)
    {
      out[k] = sh[0];
      if (out_host) out_host[k] = sh[0];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
  {
    // hand the numerical-warning flag of this evaluation to the host and re-arm it for the next one
    if (warn)
    {
      const int w = *warn;
      if (warn_host) *warn_host = w;
      if (warn_out) *warn_out = (double)w;
      *warn = 0;
    }
    if (out_host)
    {
      __threadfence_system(); // results before the sequence number, system scope (host reads over PCIe)
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(out_host + 2), seq, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K3: eigen-basis products (src/lk.c:1038-1114, src/avx.c:21-105)
// ---------------------------------------------------------------------------------------------
struct EigenParams
{
  TreeParams    t;
  RO            ro;
  int           left, rght;
  const double *r_e_vect, *l_e_vect;
  double       *dot_prod;
  // small grids: the workgroup that finishes last reports to the host that the products (and everything queued on the
  // stream before this kernel) are in memory -- what the resident evaluator waits for before it reads them
  unsigned           *tickets;    // one counter, zero between launches (nullptr: no report)
  unsigned long long *stamp_host; // host-mapped
  unsigned long long  stamp;
};

// CLS: class-axis instance (per-class frequencies and eigenvectors); a separate instantiation, because only with the offset a
// compile-time zero are the eigenvector reads provably the same for every lane (scalar loads)
// 20 states: the S products of a (pattern, category) are split over kEigenSplit lanes (S / kEigenSplit of them each: every
// product is its own two FMA chains, so the arithmetic does not change): small alignments -- Br_Len_Opt on a few thousand
// patterns -- get four times the lanes and a quarter of the serial chain (11 -> ~5 us at 2000 patterns x 4 categories).
template <int S> constexpr int kEigenSplit = (S == 20) ? 4 : 1;
template <int S, int CP, bool CLS = false>
__global__ __launch_bounds__(256) void eigen_lr_kernel(const EigenParams e)
{
  constexpr int     KS = kEigenSplit<S>, KN = S / KS; // lanes per (pattern, category), products per lane
  const TreeParams &q  = e.t;
  const long long   g0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int         ks = (int)(g0 % KS);
  const long long   gl = g0 / KS;
  const long long   p0 = gl / CP;
  const int         c0 = (int)(gl % CP);
  const bool        act = (p0 < q.P) && (c0 < q.C);
  const long long   p   = (p0 < q.P) ? p0 : (q.P - 1);
  const int         c   = (c0 < q.C) ? c0 : 0;
  double            x[S], y[S], lp[S];
  int               sl, sr;
  extern __shared__ double sh_ev[]; // [2][classes or 1][S][S]
  load_side<S, CP>(q, e.ro, e.left, p, c, x, sl);
  load_side<S, CP>(q, e.ro, e.rght, p, c, y, sr);
  double d[KN];
  // a[k] = sum_i R[i][k] lp[i];  b[k] = sum_i L[k][i] y[i]  (column-wise FMA chains, src/avx.c:81-82).  Every lane reads the
  // same 2 S^2 eigenvector entries (per class): straight from memory that was 2 S^2 dependent vector loads per lane (27 us
  // per call at 20 states, whatever the pattern count; as scalar loads 37 us) -- the workgroup stages them in LDS once and
  // the lanes read them back as broadcasts.
  {
    const int     nsys = CLS ? q.C : 1;
    const double *sh_r = sh_ev, *sh_l = sh_ev + nsys * S * S;
    for (int t = threadIdx.x; t < nsys * S * S; t += blockDim.x)
    {
      sh_ev[t] = e.r_e_vect[t];
      sh_ev[nsys * S * S + t] = e.l_e_vect[t];
    }
    __syncthreads();
    const int     co = CLS ? c : 0;
    const double *rev = sh_r + co * S * S, *lev = sh_l + co * S * S;
    const double *__restrict__ pi = q.pi + co * S;
#pragma unroll
    for (int i = 0; i < S; ++i) lp[i] = x[i] * pi[i]; // src/avx.c:79
#pragma unroll
    for (int kk = 0; kk < KN; ++kk)
    {
      const int k = ks * KN + kk;
      double a = rev[k] * lp[0];
      double b = lev[k * S] * y[0];
#pragma unroll
      for (int i = 1; i < S; ++i)
      {
        a = __builtin_fma(rev[i * S + k], lp[i], a);
        b = __builtin_fma(lev[k * S + i], y[i], b);
      }
      d[kk] = a * b;
    }
  }
  if (act)
  {
    double *dst = e.dot_prod + (size_t)p * (q.C * S) + (size_t)c * S + ks * KN;
    if constexpr (KN % 2 == 0)
    {
#pragma unroll
      for (int j = 0; j < KN / 2; ++j) reinterpret_cast<double2 *>(dst)[j] = make_double2(d[2 * j], d[2 * j + 1]);
    }
    else
    {
#pragma unroll
      for (int j = 0; j < KN; ++j) dst[j] = d[j];
    }
  }
  if (e.tickets)
  {
    __threadfence(); // this thread's stores are complete and written back (agent scope) ...
    __syncthreads(); // ... for every thread of the workgroup, before its ticket is drawn
    if (threadIdx.x == 0 &&
        __hip_atomic_fetch_add(e.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1)
    {
      __hip_atomic_store(e.tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(e.stamp_host, e.stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K4: lnL and dlnL/dl in the eigen basis (src/lk.c:655-753, 955-1032; src/avx.c:250-276),
//     and Lk(b) with use_eigen_lr (src/lk.c:866-950, src/avx.c:220-245)
// ---------------------------------------------------------------------------------------------
struct DlkParams
{
  const double *dot_prod;
  const double *wght;
  const int    *fact;
  const double *cat_w;
  const double *pi;
  const short  *invar;
  long long     P;
  int           C;
  int           invar_model;
  int           apply_scaling;
  int           with_derivative; // 1: expl is [c][2*S] interleaved (value, derivative); 0: [c][S]
  double        pinvar;
  FinishParams  fin;             // block_sums [2][stride], warn, and (optionally) the fused final sum
  double        expl[kMaxExpl];
  const double *expl_dev;        // more than 8 categories: the table in device memory instead (kernel arguments hold 4 KiB)
  // 20 states, one eigen system: the evaluation is described by the edge LENGTH and the workgroups build the table themselves
  // (dlk_expl_from_len: the arithmetic of src/lk.c:594-602 / :688-726 with the device's exp) -- a resident command of four words
  // instead of 163 (one 512-byte read per poll instead of four: what made the 20-state resident dLk lose to the launch in rounds 3
  // and 4), and the launched form takes the same route so that both give the same doubles.
  int           from_len;        // != 0: build the table from `len`
  double        len;
  const double *eval_dev, *rates_dev; // eigenvalues [S], category rates [C]
  double        br_len_mult, l_min, l_max;
};

// The expl table of an eigen-basis evaluation from the edge length (src/lk.c:594-602 without, :688-726 with the derivative;
// the host's version: eigen_eval, phyhip_eigen.hip), entry by entry over the calling threads.
template <int S>
__device__ __forceinline__ void dlk_expl_from_len(double *tab, const double l, const bool deriv, const int C, const double *ev, const double *rates,
                                                  const double mult, const double l_min, const double l_max, const int tid, const int nth, const unsigned long long *exp_lds = nullptr)
{
  for (int e = tid; e < C * S; e += nth)
  {
    const int c = e / S, s = e % S;
    if (deriv)
    {
      const double rr  = rates[c] * mult;
      double       len = l * rr;
      if (len < l_min) len = l_min;
      else if (len > l_max) len = l_max;
      const double v = ev[s], ex = dev_exp(v * len, exp_lds);
      tab[c * 2 * S + 2 * s]     = ex;
      tab[c * 2 * S + 2 * s + 1] = ex * v * rr;
    }
    else
    {
      double len = (l > 0.0 ? l : 0.0) * rates[c];
      len *= mult;
      if (len < l_min) len = l_min;
      else if (len > l_max) len = l_max;
      tab[c * S + s] = dev_exp(ev[s] * len, exp_lds);
    }
  }
}

// What varies from one evaluation to the next on one edge (the rest of DlkParams is fixed while the instance lives)
struct DlkCall
{
  int    with_derivative, invar_model, apply_scaling;
  double pinvar;
};

// What one lane of a dLk evaluation has fetched for its (pattern, category): the S products and the pattern's weight, scale
// exponent and invariant state (fetched together, before any arithmetic).
template <int S> struct DlkIn
{
  double2 d[S / 2];
  double  wt;
  int     f, iv;
};

template <int S, int CP>
__device__ __forceinline__ void dlk_fetch(const DlkParams &q, const DlkCall &k, const long long gl, DlkIn<S> &in)
{
  const long long p0 = gl / CP;
  const int       c0 = (int)(gl % CP);
  const long long p  = (p0 < q.P) ? p0 : (q.P - 1);
  const int       c  = (c0 < q.C) ? c0 : 0;
  in.wt = q.wght[p];
  in.f  = q.fact[p];
  in.iv = k.invar_model ? (int)q.invar[p] : -1;
  const double2 *s2 = reinterpret_cast<const double2 *>(q.dot_prod + (size_t)p * (q.C * S) + (size_t)c * S);
#pragma unroll
  for (int j = 0; j < S / 2; ++j) in.d[j] = s2[j];
}

// The arithmetic of one lane = one (pattern, category) of a dLk evaluation; the CP lanes of a pattern are adjacent and all
// active.  Shared by every shape the evaluation runs in (dlk_kernel's 256-thread workgroups, the 64-lane virtual blocks of
// dlk64_kernel and of the large-grid resident evaluator), so that the per-pattern terms are the same doubles everywhere.
template <int S, int CP>
__device__ __forceinline__ void dlk_lane(const DlkParams &q, const DlkCall &k, const double *expl, int *warn, const long long gl,
                                         const DlkIn<S> &in, double &c_lnl, double &c_dlnl)
{
  const long long p0  = gl / CP;
  const int       c0  = (int)(gl % CP);
  const bool      act = (p0 < q.P) && (c0 < q.C);
  const int       c   = (c0 < q.C) ? c0 : 0;
  double          dp[S];
#pragma unroll
  for (int j = 0; j < S / 2; ++j)
  {
    dp[2 * j] = in.d[j].x;
    dp[2 * j + 1] = in.d[j].y;
  }
  double lkc, dlkc = 0.0;
  if (k.with_derivative)
  { // four lanes (lk,dlk,lk,dlk) over pairs of states, then lane0+lane2 / lane1+lane3 (src/avx.c:257-274)
    const double *ex = expl + c * 2 * S;
    double        z0 = 0., z1 = 0., z2 = 0., z3 = 0.;
#pragma unroll
    for (int i = 0; i < S / 2; ++i)
    {
      z0 = __builtin_fma(dp[2 * i], ex[4 * i], z0);
      z1 = __builtin_fma(dp[2 * i], ex[4 * i + 1], z1);
      z2 = __builtin_fma(dp[2 * i + 1], ex[4 * i + 2], z2);
      z3 = __builtin_fma(dp[2 * i + 1], ex[4 * i + 3], z3);
    }
    lkc  = z0 + z2;
    dlkc = z1 + z3;
  }
  else
  { // elementwise product, blockwise lane sums, horizontal norm (src/avx.c:227-244)
    const double *ex = expl + c * S;
    double        l4[4] = {0., 0., 0., 0.};
#pragma unroll
    for (int b4 = 0; b4 < S / 4; ++b4)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) l4[kk] = l4[kk] + dp[b4 * 4 + kk] * ex[b4 * 4 + kk];
    lkc = (l4[0] + l4[2]) + (l4[1] + l4[3]);
  }
  const double w  = (c0 < q.C) ? q.cat_w[c] : 0.0;
  const double t1 = (c0 < q.C) ? lkc * w : 0.0, t2 = (c0 < q.C) ? dlkc * w : 0.0;
  double       lk = 0.0, dlk = 0.0;
#pragma unroll
  for (int cc = 0; cc < CP; ++cc)
  {
    const double a = (CP == 1) ? t1 : __shfl(t1, cc, CP);
    const double b = (CP == 1) ? t2 : __shfl(t2, cc, CP);
    if (cc < q.C)
    {
      lk += a;
      dlk += b;
    }
  }
  c_lnl = c_dlnl = 0.0;
  if (act && c == 0)
  {
    const double wt = in.wt;
    if (wt > kSmall)
    {
      int f = in.f;
      if (k.invar_model)
      { // src/lk.c:1005-1025 (dLk) / :910-931 (Lk in the eigen basis)
        const int iv  = in.iv;
        double    inv = 0.0;
        bool      issue = false;
        if (iv >= 0)
        {
          inv = q.pi[iv];
          if (k.apply_scaling)
          {
            int e = f;
            do
            {
              const int piece = e < 63 ? e : 63;
              inv *= (double)(1ull << piece);
              e -= piece;
            } while (e != 0);
          }
          issue = isinf(inv);
        }
        if (issue)
        {
          if (k.with_derivative) { lk = inv * k.pinvar; dlk = 0.0; }
          else { f = 0; lk = q.pi[iv] * k.pinvar; }
        }
        else
        {
          lk  = lk * (1. - k.pinvar) + inv * k.pinvar;
          dlk = dlk * (1. - k.pinvar);
        }
      }
      if (lk < kSmall)
      {
        lk = kSmall;
        raise_warn(warn);
      }
      c_dlnl = wt * (dlk / lk);                          // src/lk.c:742-744
      c_lnl  = wt * (log(lk) - kLog2 * (double)f);       // src/lk.c:745
    }
  }
}

// One workgroup's share of an evaluation: on return wave 0 holds the workgroup's two sums (every lane of it).
template <int S, int CP>
__device__ __forceinline__ bool dlk_block(const DlkParams &q, const DlkCall k, const double *expl, int *warn, double (&v)[2])
{
  // grid-stride over (pattern, category) lanes: a bounded number of workgroups streams dot_prod, each thread
  // accumulating its patterns, so that the per-workgroup sums stay few enough for the fused final sum
  const long long total = (((long long)q.P * CP + 255) / 256) * 256;
  double          tot_lnl = 0.0, tot_dlnl = 0.0;
  for (long long gl = (long long)blockIdx.x * blockDim.x + threadIdx.x; gl < total; gl += (long long)gridDim.x * blockDim.x)
  {
    DlkIn<S> in;
    dlk_fetch<S, CP>(q, k, gl, in);
    double c_lnl, c_dlnl;
    dlk_lane<S, CP>(q, k, expl, warn, gl, in, c_lnl, c_dlnl);
    tot_lnl += c_lnl;
    tot_dlnl += c_dlnl;
  }
  double c_lnl = tot_lnl, c_dlnl = tot_dlnl;
  __shared__ double ws[2][4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
  {
    c_lnl += __shfl_down(c_lnl, off, 64);
    c_dlnl += __shfl_down(c_dlnl, off, 64);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0)
  {
    ws[0][wid] = c_lnl;
    ws[1][wid] = c_dlnl;
  }
  __syncthreads();
  v[0] = v[1] = 0.0;
  if (wid == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w)
    {
      v[0] += ws[0][w];
      v[1] += ws[1][w];
    }
  __syncthreads(); // (ws may be rewritten by the next evaluation of a resident workgroup)
  return wid == 0;
}

// The same evaluation cut along the TILES of the lane-per-pattern traversal kernel: tile t holds the patterns
// [t * 64 / G, (t + 1) * 64 / G) -- IT = CP / G rounds of 64 (pattern, category) lanes, each lane adding its rounds in order, then
// the wave's shuffle tree.  A tile's sums do not depend on who computes them -- a one-wave workgroup of dlk64_kernel or a wave
// of the large-grid resident evaluator (phyhip_big.hpp) -- so the launched and the resident form return the same doubles; and
// the wave that evaluates a tile is the one that wrote its products and scale exponents (Update_Eigen_Lr and the edge
// evaluation run on the same tiles): the resident workgroups need no write-back between commands.  The loads of all rounds
// are in flight at once (a round is a dependent trip to memory; one after the other they were most of dlk_kernel's 10 us at
// 100 000 patterns).  On return every lane of the wave holds the two sums in v.
template <int S, int CP, int IT>
__device__ __forceinline__ void dlk_tile(const DlkParams &q, const DlkCall k, const double *expl, int *warn, const unsigned tile,
                                         const int lane, double (&v)[2])
{
  const long long g0 = (long long)tile * (IT * 64) + lane;
  DlkIn<S>        in[IT];
#pragma unroll
  for (int u = 0; u < IT; ++u) dlk_fetch<S, CP>(q, k, g0 + u * 64, in[u]);
  double tot_lnl = 0.0, tot_dlnl = 0.0;
#pragma unroll
  for (int u = 0; u < IT; ++u)
  {
    double c_lnl, c_dlnl;
    dlk_lane<S, CP>(q, k, expl, warn, g0 + u * 64, in[u], c_lnl, c_dlnl);
    tot_lnl += c_lnl;
    tot_dlnl += c_dlnl;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
  {
    tot_lnl += __shfl_down(tot_lnl, off, 64);
    tot_dlnl += __shfl_down(tot_dlnl, off, 64);
  }
  v[0] = __shfl(tot_lnl, 0, 64);
  v[1] = __shfl(tot_dlnl, 0, 64);
}

template <int S, int CP>
__global__ __launch_bounds__(256) void dlk_kernel(const DlkParams q)
{
  const DlkCall k = {q.with_derivative, q.invar_model, q.apply_scaling, q.pinvar};
  double        v[2];
  if constexpr (S == 20)
  {
    if (q.from_len)
    { // (every workgroup builds the table: 160 exponentials over 256 threads)
      __shared__ double sh_tab[kMaxExpl];
      dlk_expl_from_len<S>(sh_tab, q.len, q.with_derivative != 0, q.C, q.eval_dev, q.rates_dev, q.br_len_mult, q.l_min, q.l_max, (int)threadIdx.x, (int)blockDim.x);
      __syncthreads();
      if (dlk_block<S, CP>(q, k, sh_tab, q.fin.warn, v)) finish_sums<2>(q.fin, v, (int)(threadIdx.x & 63));
      return;
    }
  }
  if (dlk_block<S, CP>(q, k, q.expl_dev ? q.expl_dev : q.expl, q.fin.warn, v)) finish_sums<2>(q.fin, v, (int)(threadIdx.x & 63));
}

// One tile (dlk_tile) per one-wave workgroup: the launched form of what the large-grid resident evaluator serves
template <int S, int CP, int IT>
__global__ __launch_bounds__(64) void dlk64_kernel(const DlkParams q)
{
  const DlkCall k = {q.with_derivative, q.invar_model, q.apply_scaling, q.pinvar};
  double        v[2];
  dlk_tile<S, CP, IT>(q, k, q.expl, q.fin.warn, blockIdx.x, (int)threadIdx.x, v);
  finish_sums<2>(q.fin, v, (int)threadIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Resident evaluator.  A branch-length Newton step (src/optimiz.c: Br_Len_Opt) is a chain of dLk calls on ONE edge, each
// waiting for the previous scalar: per call the host pays a kernel launch (~3 us of CPU time), the command processor's
// dispatch and the kernel's start and end -- more than the ~2 us of work in it.  While such a chain runs, the workgroups of
// this kernel stay on the device and take their evaluations from a host-mapped command record instead: the host writes
// {expl table, tag} and bumps the sequence word; every workgroup polls it, evaluates its share exactly as dlk_kernel does
// (same grid, same block sums, so the host-side final sum gives the same double) and posts {sum, tag}.  A workgroup leaves
// when told to (seq = ~0), when a newer launch has superseded it (gen), or after idle_ticks without a command -- so a
// device-wide synchronisation never waits longer than that, and a host that died leaves nothing behind.
// ---------------------------------------------------------------------------------------------
// The command record is made of 32-byte sectors, each carrying three payload words and, LAST, the number of the command it
// belongs to: the host fills a sector's payload and then its number, the device reads whole sectors (aligned 32-byte
// pieces of one 512-byte wave access -- the smallest unit a read of host memory may be split into, so a sector is never
// seen half old, half new) and takes a command when every sector carries the number it expects: payload and "go" arrive
// in ONE trip over the link instead of two (a poll of a sequence word, then a dependent read of the body: ~2 us each).
// Only the first wave of a workgroup reads host memory (every reader costs the others: 24 polling workgroups answered in
// 8.5 us, 6 in 4.3 -- tools/probes/resident_probe.hip); the rest get the payload through LDS.
// dLk payload words: 0 tag of the {sum, tag} records, 1 flags (bit 0 derivative, 1 invariant-site model, 2 scaling, 3 device
// data changed since the last command), 2 pinvar, 3.. the expl table.  Sector 0 is control: word 0 generation in charge,
// word 1 != 0: leave.
constexpr int kResidentPay  = 3;                // payload words per sector
constexpr int kResidentUnit = kResidentPay + 1; // words per sector
struct ResidentSector
{
  unsigned long long w[kResidentPay];
  unsigned long long seq;
};
constexpr int kResidentWords   = 3 + kMaxExpl;
constexpr int kResidentSectors = (kResidentWords + kResidentPay - 1) / kResidentPay;
struct ResidentCmd
{
  ResidentSector ctl;
  ResidentSector sec[kResidentSectors];
  ResidentSector spare;  // (formerly the device's report; that is ResidentCtl::report now)
};
// index of payload word k among the record's words (sector 0 is control)
__host__ __device__ constexpr int resident_slot(int k) { return (1 + k / kResidentPay) * kResidentUnit + k % kResidentPay; }
struct ResidentCtl
{
  const ResidentCmd *cmd;        // host-mapped
  unsigned long long gen;        // this launch
  unsigned long long start_seq;  // commands up to here have been served
  unsigned long long idle_ticks; // wall_clock64 ticks
  int                n_sectors;  // command sectors in use
  // More than a handful of workgroups: only workgroup 0 polls the host (every reader of host memory slows the others down);
  // it copies each command into a mailbox in device memory -- same line format, payload before numbers -- which the others poll.
  unsigned long long *mail;      // device, a copy of the record's layout
  int                 relay;     // != 0: the other workgroups take their commands from the mailbox instead of the host
  // host-mapped: the generation whose workgroups have left (not in the record: the record may live in DEVICE memory, where the
  // host pushes the commands through the BAR -- every workgroup then polls it locally, no relay, phyhip_resident.hip)
  unsigned long long *report;
  // Host-computed transition matrices of a command (the bit-exact route: the host layer's PMat() + phyhip_set_transition_matrix,
  // src/lk.c:2360): device memory the host stores into through the BAR BEFORE it pushes the command -- posted writes stay in
  // order, so a workgroup that sees the command finds them there -- [kArgUp][64] doubles, read past the caches (the same
  // lines carried the previous command's matrices).  nullptr: the record is host memory, such commands are launched instead.
  const double       *up_area;
  // Workgroup 0 alone decides when a generation leaves (idle time-out): it says so in the mailbox's word 0, which the others
  // watch, and reports to the host -- which therefore KNOWS whether anybody is there instead of guessing from elapsed time.
};

// One poll by one (whole) wave: the record's lines go to sh_raw ([n_loads * 64] words); returns 0: nothing yet, 1: command
// last + 1 is complete in sh_raw, 2: leave.  Workgroup 0 of a relaying launch also keeps the mailbox up to date.
// LIGHT: the relay orders payload before sector numbers by waiting for the (written-through) payload stores instead of a
// release fence -- which writes back the whole L2 of the XCD, megabytes of results when the evaluator serves a large alignment
template <bool LIGHT = false>
__device__ __forceinline__ int resident_poll_wave(const ResidentCtl &r, const unsigned long long last, const unsigned long long t_last,
                                                  bool &mail_open, unsigned long long *sh_raw, const int n_loads, const int lane,
                                                  const bool decider)
{
  // one aligned 8-byte word per lane and load: 512 contiguous bytes = 16 sectors per instruction
  const bool                from_host = decider || !r.relay;
  const unsigned long long *base = from_host ? reinterpret_cast<const unsigned long long *>(r.cmd) : r.mail;
  bool                      good = true;
  unsigned long long        first = 0;
  for (int j = 0; j < n_loads; ++j)
  {
    const unsigned long long v = from_host ? __hip_atomic_load(base + j * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                           : __hip_atomic_load(base + j * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh_raw[j * 64 + lane] = v;
    if (j == 0) first = v;
    const int sector = j * 16 + (lane >> 2);
    if ((lane & 3) == 3 && sector >= 1 && sector <= r.n_sectors) good = good && (v == last + 1);
  }
  // the mailbox's word 0 (generation * 2 + leave): what workgroup 0 has decided
  unsigned long long m0 = 0;
  if (!decider && from_host) m0 = __hip_atomic_load(r.mail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool all_good = __builtin_amdgcn_ballot_w64(!good) == 0;
  // control: the host's line 0 holds {generation, leave}; the mailbox packs both into one word so that a leaving workgroup 0
  // can say so with ONE compare-and-swap that fails if a newer generation took over
  const unsigned long long w0 = __shfl(first, 0, 64), w1 = __shfl(first, 1, 64);
  const unsigned long long gen = from_host ? w0 : (w0 >> 1), stop = from_host ? w1 : (w0 & 1);
  if (!from_host) m0 = w0;
  int act = 0;
  if (from_host && (stop != 0 || gen != r.gen)) act = 2;                                   // the host says so
  // (a complete command of this generation is served even when "leave" already stands behind it: a workgroup that starts
  // late -- hundreds of them take microseconds to start -- would otherwise leave the command it was launched for unanswered)
  else if (all_good && gen == r.gen) act = 1;
  else if (!decider && ((m0 >> 1) > r.gen || ((m0 >> 1) == r.gen && (m0 & 1)))) act = 2; // workgroup 0 (or a newer one) says so
  else if (wall_clock64() - t_last > (decider ? r.idle_ticks : 16 * r.idle_ticks)) act = 2; // (the others: a safety net only)
  if (decider)
  {
    if (act == 2)
    { // (a workgroup of an older generation that is still around sees the newer number and leaves by itself)
      if (lane == 0)
      { // (also when this generation never got to open the mailbox: told to leave at its very first poll)
        unsigned long long cur = __hip_atomic_load(r.mail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((cur >> 1) <= r.gen)
          __hip_atomic_compare_exchange_strong(r.mail, &cur, (r.gen << 1) | 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(r.report, r.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    else if (r.relay ? (act == 1 || !mail_open) : !mail_open)
    { // pass it on (relay), or just open the mailbox for this generation
      for (int j = 0; j < (r.relay ? n_loads : 1); ++j)
        if ((lane & 3) != 3 && !(j == 0 && lane < 4 && lane > 0) && (r.relay || (j == 0 && lane == 0)))
          __hip_atomic_store(r.mail + j * 64 + lane, (j == 0 && lane == 0) ? (r.gen << 1) : sh_raw[j * 64 + lane], __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      if (r.relay)
      {
        if (LIGHT) __builtin_amdgcn_s_waitcnt(0); // (the payload stores are atomic at agent scope: acknowledged = visible)
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // payload (and the control word) before the sector numbers
        if (act == 1)
          for (int j = 0; j < n_loads; ++j)
            if ((lane & 3) == 3 && j * 16 + (lane >> 2) != 0)
              __hip_atomic_store(r.mail + j * 64 + lane, sh_raw[j * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      mail_open = true;
    }
  }
  return act;
}

template <int S, int CP>
__global__ __launch_bounds__(256) void resident_dlk_kernel(const DlkParams q, const ResidentCtl r)
{
  __shared__ unsigned long long sh_raw[(1 + kResidentSectors + 15) / 16 * 64];
  __shared__ double             sh_expl[kMaxExpl];
  __shared__ unsigned long long sh_ctl[1];
  __shared__ unsigned long long sh_exp[256]; // the exp table (dev_exp), copied once per launch
  exp_tab_to_lds(sh_exp, (int)threadIdx.x, (int)blockDim.x);
  unsigned long long last = r.start_seq, t_last = wall_clock64();
  const int          n_loads = (1 + r.n_sectors + 15) / 16;
  bool               mail_open = false; // workgroup 0: the mailbox carries this generation's control line
  for (;;)
  {
    if (threadIdx.x < 64)
    {
      const int a = resident_poll_wave(r, last, t_last, mail_open, sh_raw, n_loads, (int)threadIdx.x, blockIdx.x == 0);
      if (threadIdx.x == 0) sh_ctl[0] = (unsigned long long)a;
    }
    __syncthreads();
    const unsigned long long act = sh_ctl[0];
    if (act == 2) return;
    if (act == 0)
    {
      __syncthreads();
      continue;
    }
    auto word = [&](int k) { return sh_raw[resident_slot(k)]; };
    const unsigned long long tag = word(0), flags = word(1);
    double                   pinvar;
    {
      const unsigned long long b = word(2);
      __builtin_memcpy(&pinvar, &b, 8);
    }
    const DlkCall k = {(int)(flags & 1), (int)((flags >> 1) & 1), (int)((flags >> 2) & 1), pinvar};
    const int     ne = q.C * (k.with_derivative ? 2 : 1) * S;
    if (flags & 16)
    { // the table from the edge length (word 3): DlkParams::from_len
      double                   l;
      const unsigned long long b = word(3);
      __builtin_memcpy(&l, &b, 8);
      dlk_expl_from_len<S>(sh_expl, l, k.with_derivative != 0, q.C, q.eval_dev, q.rates_dev, q.br_len_mult, q.l_min, q.l_max, (int)threadIdx.x, (int)blockDim.x, sh_exp);
    }
    else
    for (int e = threadIdx.x; e < ne; e += blockDim.x)
    {
      const unsigned long long b = word(3 + e);
      __builtin_memcpy(&sh_expl[e], &b, 8);
    }
    // what other kernels wrote since the last command (the host says whether anything was) is re-read from memory
    if (flags & 8) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
    double v[2];
    if (dlk_block<S, CP>(q, k, sh_expl, q.fin.warn, v))
    {
      FinishParams f = q.fin;
      f.host_tag = tag;
      finish_sums<2>(f, v, (int)(threadIdx.x & 63));
    }
    last = last + 1; t_last = wall_clock64();
  }
}

// ---------------------------------------------------------------------------------------------
// Mixture of class models: the site loop of MIXT_Lk after the per-class Lk_Core calls (src/mixt.c:1027-1135).
// Every class is its own instance (one rate category, own rate matrix / frequencies, the class rate as category rate);
// their edge evaluations left the per-site class likelihood (unscaled_site_lk_cat) and the scale exponent
// (fact_sum_scale) in device memory.  Here: common scale 2^-sum (sum capped at 1023 above 1024 with a warning, :1027-1034),
// weights in the reference's operation order (:1048-1053), DBL_MIN floor (:1114-1117), log, weighted sum.
// ---------------------------------------------------------------------------------------------
// (per-class instances: profile mixtures of the C10-C60 kind have up to 60 classes; the per-class tables below ride in the
// kernel arguments, whose segment holds 4 KB)
constexpr int kMaxMixClasses = 64;
struct MixParams
{
  int           count;
  long long     P;
  const double *site_cat[kMaxMixClasses]; // per class: [P] (C = 1: cat_stride 1), or base + class with cat_stride = C (a class of a class-axis instance)
  const int    *fact[kMaxMixClasses];
  unsigned short cat_stride[kMaxMixClasses]; // (per class: a mixture may be spread over several instances of different shapes)
  // +I mixture (src/mixt.c:1079-1112): the invariant class is not a class tree of the device; its share enters here
  int           invar_model;
  double        pinvar;
  const short  *invar;   // data->invar[pattern]: the constant state or -1
  double        pi_inv[20]; // frequencies of the invariant class tree's model
  double        proba[kMaxMixClasses], r_w[kMaxMixClasses], e_w[kMaxMixClasses];
  double        r_sum, e_sum, sum_probas;
  const double *wght;
  double       *site_lnl; // mixture c_lnL_sorted (optional)
  FinishParams  fin;
};

static __global__ __launch_bounds__(256) void mixture_combine_kernel(const MixParams q)
{
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  double          contrib = 0.0;
  if (p < q.P)
  {
    double site_lk = 0.0;
    for (int k = 0; k < q.count; ++k)
    {
      int s = q.fact[k][p];
      if (s > 1024) { s = 1023; raise_warn(q.fin.warn); }
      const double x = ldexp(q.site_cat[k][(size_t)p * q.cat_stride[k]], -s); // == site_lk_cat / pow(2, sum): exact power-of-two scaling
      site_lk += x * q.proba[k] * q.r_w[k] / q.r_sum * q.e_w[k] / q.e_sum / q.sum_probas;
    }
    if (q.invar_model)
    { // Invariant_Lk(0, site, ..) of the invariant class: pi[state] at scale 2^0 (src/lk.c:1226-1273), then the mixing
      const int    iv  = q.invar[p];
      const double inv = iv >= 0 ? q.pi_inv[iv] : 0.0;
      site_lk = site_lk * (1. - q.pinvar) + inv * q.pinvar;
    }
    if (site_lk < kSmall) { site_lk = kSmall; raise_warn(q.fin.warn); }
    const double lsl = log(site_lk);
    if (q.site_lnl) q.site_lnl[p] = lsl;
    contrib = q.wght[p] * lsl; // src/mixt.c:1133 (zero-weight patterns contribute 0)
  }
  __shared__ double ws[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) contrib += __shfl_down(contrib, off, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) ws[wid] = contrib;
  __syncthreads();
  if (wid == 0)
  {
    double v[1] = {0.0};
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) v[0] += ws[k];
    finish_sums<1>(q.fin, v, lane);
  }
}

// MIXT_dLk (src/mixt.c:2962-3340): lnL and dlnL/dl of a mixture in the eigen basis.  Per class and pattern lk and dlk come
// from that class's dot_prod with its own expl pairs in the AVX order (src/avx.c:250-276), are brought to a common scale
// with 2^-sum of the class edge's scale exponents (1023 cap above 1024, :3176-3194), weighted like MIXT_Lk (:3214-3226),
// and enter lnL += w log(site_lk), dlnL += w site_dlk / site_lk (:3308-3309).  No +I.
template <int S> struct MixDlkParams
{
  int           count;
  long long     P;
  const double *dot[kMaxMixClasses];                 // per class: dot_prod [P][S] (dot_stride = S), or base + class * S with dot_stride = C * S
  unsigned short dot_stride[kMaxMixClasses];
  const int    *scale_l[kMaxMixClasses], *scale_r[kMaxMixClasses]; // scale exponents of the two edge sides (nullptr: tip)
  double        proba[kMaxMixClasses], r_w[kMaxMixClasses], e_w[kMaxMixClasses];
  double        r_sum, e_sum, sum_probas;
  const double *expl;                                // device: [count][2*S] (value, derivative) pairs
  const double *wght;
  int           invar_model;                         // +I mixture: src/mixt.c:3212-3275
  double        pinvar;
  const short  *invar;
  double        pi_inv[20];
  FinishParams  fin;
};

static_assert(sizeof(MixParams) <= 4096 && sizeof(MixDlkParams<20>) <= 4096, "the per-class tables ride in the kernel arguments");

template <int S> __global__ __launch_bounds__(256) void mixture_dlk_kernel(const MixDlkParams<S> q)
{
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  double          c_lnl = 0.0, c_dlnl = 0.0;
  if (p < q.P)
  {
    const double wt = q.wght[p];
    const double one_m_pinv = q.invar_model ? 1. - q.pinvar : 1.; // src/mixt.c:3212-3215
    double       site_lk = 0.0, site_dlk = 0.0;
    for (int k = 0; k < q.count; ++k)
    {
      const double2 *s2 = reinterpret_cast<const double2 *>(q.dot[k] + (size_t)p * q.dot_stride[k]);
      const double  *ex = q.expl + (size_t)k * 2 * S;
      double         z0 = 0., z1 = 0., z2 = 0., z3 = 0.;
#pragma unroll
      for (int i = 0; i < S / 2; ++i)
      {
        const double2 v = s2[i];
        z0 = __builtin_fma(v.x, ex[4 * i], z0);
        z1 = __builtin_fma(v.x, ex[4 * i + 1], z1);
        z2 = __builtin_fma(v.y, ex[4 * i + 2], z2);
        z3 = __builtin_fma(v.y, ex[4 * i + 3], z3);
      }
      int sum = (q.scale_l[k] ? q.scale_l[k][p] : 0) + (q.scale_r[k] ? q.scale_r[k][p] : 0);
      if (sum > 1024) { sum = 1023; raise_warn(q.fin.warn); }
      const double lk  = ldexp(z0 + z2, -sum); // == / pow(2, sum)
      const double dlk = ldexp(z1 + z3, -sum);
      if (wt > kSmall)
      {
        site_lk  += lk * q.proba[k] * q.r_w[k] / q.r_sum * q.e_w[k] / q.e_sum / q.sum_probas;
        site_dlk += dlk * one_m_pinv * q.proba[k] * q.r_w[k] / q.r_sum * q.e_w[k] / q.e_sum / q.sum_probas;
      }
    }
    if (q.invar_model)
    { // src/mixt.c:3250-3275
      const int    iv  = q.invar[p];
      const double inv = iv >= 0 ? q.pi_inv[iv] : 0.0;
      site_lk = site_lk * (1. - q.pinvar) + inv * q.pinvar;
    }
    if (wt > kSmall)
    {
      c_lnl  = wt * log(site_lk);
      c_dlnl = wt * (site_dlk / site_lk);
    }
  }
  __shared__ double ws[2][4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
  {
    c_lnl += __shfl_down(c_lnl, off, 64);
    c_dlnl += __shfl_down(c_dlnl, off, 64);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { ws[0][wid] = c_lnl; ws[1][wid] = c_dlnl; }
  __syncthreads();
  if (wid == 0)
  {
    double v[2] = {0.0, 0.0};
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { v[0] += ws[0][k]; v[1] += ws[1][k]; }
    finish_sums<2>(q.fin, v, lane);
  }
}

// ---------------------------------------------------------------------------------------------
// K5: transition matrices on the device (src/models.c:257-326 behind src/lk.c:2280-2316)
//     one block per matrix; thread (c,i) builds row i of category c
// ---------------------------------------------------------------------------------------------
constexpr int kSmallPm = 8; // up to this many (index, length) pairs travel inside the kernel arguments
constexpr int kEagerPmBatch = 32; // a single call queueing at least this many matrices launches their rebuild at once
struct PmatParams
{
  const int    *indices; // [count]  (nullptr: use small_idx / small_len)
  const double *lengths; // [count] raw edge lengths b->l->v
  int           small_idx[kSmallPm];
  double        small_len[kSmallPm];
  const int    *shadow;  // [count] or nullptr: slot that receives the matrix's OLD value before it is overwritten (-1: none) --
  int           small_shadow[kSmallPm]; // a virtual buffer defined on the old matrix keeps reading it there (phyhip_host.hpp)
  int           count;
  int           S, C;
  const double *U, *V, *R;   // r_e_vect, l_e_vect, e_val
  const double *rates;       // gamma_rr
  double        br_len_mult, l_min, l_max;
  double       *pmats;
  double       *afrag;       // 20-state MFMA A-operand copy (nullptr otherwise)
  int           class_axis;  // U, V, R are [C] eigen systems: category c is class c of a mixture
};

// ST: the state count at compile time (4 | 20): unrolled 20-term chains, and for 20 states the products U[i][k] e[c][k] formed
// once per (c, i, k) in LDS instead of once per entry (a third of the LDS reads of the chain), 1024 threads (two entries each).
// The arithmetic is unchanged: (U[i][k] * e[c][k]) first, then the FMA into the sum over ascending k (src/models.c:278-292).
// PRE: the U e products formed first (short lists: latency; a whole tree's batch runs faster without that extra phase:
// 12 vs 15-17 us for 397 matrices, measured)
template <int ST, bool PRE = true>
__global__ __launch_bounds__(ST == 4 ? 64 : 1024) void pmat_kernel(const PmatParams q)
{
  extern __shared__ __attribute__((aligned(16))) double expt[]; // [C][S] | tmp [C][S][S] | row sums [C][S] | U [S][S] | V [S][S] | W [C][S][S]
  const int    m  = blockIdx.x;
  constexpr int S = ST;
  const int    C = q.C;
  // SPR refreshes three matrices per candidate: such short lists ride in the kernel arguments (no H2D copy)
  double l;
  int    mat, shd;
  if (q.indices) { l = q.lengths[m]; mat = q.indices[m]; shd = q.shadow ? q.shadow[m] : -1; }
  else
  {
    l = q.small_len[0]; mat = q.small_idx[0]; shd = q.small_shadow[0];
#pragma unroll
    for (int k = 1; k < kSmallPm; ++k)
      if (m == k) { l = q.small_len[k]; mat = q.small_idx[k]; shd = q.small_shadow[k]; }
  }
  if (shd >= 0)
  { // the old value first (the loads complete before this thread's stores issue, the barriers below come before any overwrite)
    const double *src = q.pmats + (size_t)mat * q.C * S * S;
    double       *dst = q.pmats + (size_t)shd * q.C * S * S;
    for (int e = threadIdx.x; e < q.C * S * S; e += blockDim.x) dst[e] = src[e];
    if (ST == 20 && q.afrag)
      for (int e = threadIdx.x; e < kAaMat; e += blockDim.x) q.afrag[(size_t)shd * kAaMat + e] = q.afrag[(size_t)mat * kAaMat + e];
  }
  double *tmp  = expt + C * S;         // [C][S][S] floored, un-normalised entries
  double *rsum = tmp + C * S * S;      // [C][S]
  const int NE = q.class_axis ? C : 1; // eigen systems
  double *Us   = rsum + C * S;         // eigenvectors staged once per block
  double *Vs   = Us + NE * S * S;
  double *Ws   = Vs + NE * S * S;      // [C][S][S]: U[i][k] * e[c][k]
  for (int t = threadIdx.x; t < NE * S * S; t += blockDim.x)
  {
    Us[t] = q.U[t];
    Vs[t] = q.V[t];
  }
  for (int t = threadIdx.x; t < C * S; t += blockDim.x)
  {
    const int c   = t / S, k = t % S;
    double    len = (l > 0.0 ? l : 0.0) * q.rates[c]; // src/lk.c:2296
    len *= q.br_len_mult;                             // :2297
    if (len < q.l_min) len = q.l_min;                 // :2299-2300
    else if (len > q.l_max) len = q.l_max;
    expt[t] = dev_exp(q.R[(q.class_axis ? c * S : 0) + k] * len); // src/models.c:275
  }
  __syncthreads();
  if (PRE)
  {
    for (int e = threadIdx.x; e < C * S * S; e += blockDim.x)
    {
      const int c = e / (S * S), ik = e % (S * S), k = e % S;
      Ws[e] = Us[(q.class_axis ? c * S * S : 0) + ik] * expt[c * S + k];
    }
    __syncthreads();
  }
  // one thread per entry: acc = sum_k (U[i][k]*expt[c][k]) * V[k][j], ascending k with FMA (src/models.c:278-292)
  for (int e = threadIdx.x; e < C * S * S; e += blockDim.x)
  {
    const int c = e / (S * S), i = (e / S) % S, j = e % S;
    const double *Wc = Ws + (c * S + i) * S, *Vc = Vs + (q.class_axis ? c * S * S : 0) + j;
    const double *Uc = Us + (q.class_axis ? c * S * S : 0) + i * S, *ec = expt + c * S;
    double    acc = 0.0;
#pragma unroll
    for (int k = 0; k < S; ++k) acc = __builtin_fma(PRE ? Wc[k] : Uc[k] * ec[k], Vc[k * S], acc);
    tmp[e] = (acc < kSmallPij) ? kSmallPij : acc; // :293
  }
  __syncthreads();
  // row sums in ascending j (src/models.c:296-297)
  for (int t = threadIdx.x; t < C * S; t += blockDim.x)
  {
    const double *row = tmp + (size_t)t * S;
    double        sum = 0.0;
#pragma unroll
    for (int j = 0; j < S; ++j) sum += row[j];
    rsum[t] = sum;
  }
  __syncthreads();
  // the division (src/models.c:298), one coalesced pass over the matrix
  double *out = q.pmats + (size_t)mat * C * S * S;
  for (int e = threadIdx.x; e < C * S * S; e += blockDim.x) out[e] = tmp[e] / rsum[e / S];
  if (ST == 20 && q.afrag)
  { // 20 states: the same entries once more in MFMA A-operand order (phyhip_aa.hpp: aa_a_slot), all categories in one table
    double   *dst = q.afrag + (size_t)mat * kAaMat;
    const int cb  = aa_cb(C);
    for (int e = threadIdx.x; e < kAaMat; e += blockDim.x)
    {
      const int lane = e & 63, rt = e >> 6, r = rt % 5, t = rt / 5;
      const int k = lane >> 4, b = (lane >> 2) & 3, i = 4 * r + (lane & 3), j = 4 * t + k, c = b % cb;
      dst[aa_a_slot(t, r, lane)] = (c < C) ? tmp[(c * S + i) * S + j] / rsum[c * S + i] : 0.0;
    }
  }
}

// 20 states, register-blocked (round 5).  pmat_kernel<20> above takes every entry's twenty-term chain through LDS term by term
// (two reads per term), the matrix through LDS twice more for the row sums and the division, and four workgroup barriers.  Here
// a lane owns a 2 x 4 block of entries (rows i0, i0 + 1 of category c, columns 4 jq .. 4 jq + 3): per term one e[c][k], two U
// words and four V words (two 16-byte reads) feed eight FMAs, the products U[i][k] e[c][k] are formed in registers, the five
// lanes of a row pair sit side by side and pass the row sums along by shuffle in ascending j, the division happens in registers
// and the natural-layout matrix is stored from them; the A-operand table is scattered into LDS and leaves in 16-byte pieces.
// One workgroup of 256 threads per matrix, one barrier before the arithmetic (eigen system and exponentials staged in LDS), one
// behind it for the table.  Every entry is the same chain of operations in the same order as above (src/models.c:275-298): (U e)
// first, FMA over ascending k, the floor, the row sum over ascending j, the division -- the same bits
// (tests/test_gpu_switches.py, PHYHIP_PMAT20).  Measured against pmat_kernel<20>: the three matrices of an SPR candidate 7.0
// against 8.0 us (a short list is a chain of latencies either way: profiles/r05_aa_candidate.md), the 397 matrices of a
// 200-taxon tree 15.2 against 16.8 us.
// The exponentials of one matrix: expt[c][k] = exp(eigenvalue k x the clamped, rate-scaled length of category c), by the threads
// tid, tid + nth, ... of whoever builds it (src/lk.c:2296-2300, src/models.c:275)
__device__ __forceinline__ void pmat20_exponentials(double *expt, const double l, const int C, const bool cls, const double *R, const double *rates,
                                                    const double br_len_mult, const double l_min, const double l_max, const int tid, const int nth,
                                                    const unsigned long long *exp_lds = nullptr)
{
  constexpr int S = 20;
  for (int t = tid; t < C * S; t += nth)
  {
    const int c   = t / S, k = t % S;
    double    len = (l > 0.0 ? l : 0.0) * rates[c]; // src/lk.c:2296
    len *= br_len_mult;                             // :2297
    if (len < l_min) len = l_min;                   // :2299-2300
    else if (len > l_max) len = l_max;
    expt[t] = dev_exp(R[(cls ? c * S : 0) + k] * len, exp_lds);  // src/models.c:275
  }
}

// The entries of one matrix from its staged exponentials and the staged eigen system, by the nwv waves wv = 0 .. nwv - 1 of
// whoever builds it (every lane of each of them calls this): natural layout to `out` (or nowhere: nullptr), MFMA A-operand
// order to those of the tables tabs[] (LDS) whose bit is set in `mask`.  A lane owns a 2 x 4 block of entries; see pmat20_kernel.
template <int MAXTABS>
__device__ __forceinline__ void pmat20_entries(const double *expt, const double *Us, const double *Vs, const int C, const bool cls, const int lane,
                                               const int wv, const int nwv, double *out, double *const (&tabs)[MAXTABS], const unsigned mask)
{
  constexpr int S = 20;
  const int jq = lane % 5, cb = aa_cb(C);
  for (int g0 = wv * 12; g0 < C * 10; g0 += nwv * 12) // (uniform per wave: 12 row pairs of 5 lanes each, lanes 60-63 idle)
  {
    const int  grp  = g0 + lane / 5;
    const bool live = lane < 60 && grp < C * 10;
    const int  g = live ? grp : 0, c = g / 10, i0 = (g % 10) * 2;
    const double *U0 = Us + (cls ? c * S * S : 0) + i0 * S, *U1 = U0 + S, *ec = expt + c * S;
    const double *Vq = Vs + (cls ? c * S * S : 0) + 4 * jq;
    double a0[4] = {0., 0., 0., 0.}, a1[4] = {0., 0., 0., 0.};
#pragma unroll
    for (int k = 0; k < S; ++k)
    {
      const double e = ec[k], w0 = U0[k] * e, w1 = U1[k] * e; // src/models.c:278-292
      const double2 va = *reinterpret_cast<const double2 *>(Vq + k * S), vb = *reinterpret_cast<const double2 *>(Vq + k * S + 2);
      a0[0] = __builtin_fma(w0, va.x, a0[0]); a0[1] = __builtin_fma(w0, va.y, a0[1]);
      a0[2] = __builtin_fma(w0, vb.x, a0[2]); a0[3] = __builtin_fma(w0, vb.y, a0[3]);
      a1[0] = __builtin_fma(w1, va.x, a1[0]); a1[1] = __builtin_fma(w1, va.y, a1[1]);
      a1[2] = __builtin_fma(w1, vb.x, a1[2]); a1[3] = __builtin_fma(w1, vb.y, a1[3]);
    }
#pragma unroll
    for (int x = 0; x < 4; ++x)
    { // :293
      a0[x] = (a0[x] < kSmallPij) ? kSmallPij : a0[x];
      a1[x] = (a1[x] < kSmallPij) ? kSmallPij : a1[x];
    }
    // row sums in ascending j (src/models.c:296-297): the lane of column quad jq continues the sum of the lane of jq - 1
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int step = 0; step < 5; ++step)
    {
      const double p0 = __shfl_up(s0, 1, 64), p1 = __shfl_up(s1, 1, 64);
      const double b0 = step ? p0 : 0.0, b1 = step ? p1 : 0.0;
      const double n0 = (((b0 + a0[0]) + a0[1]) + a0[2]) + a0[3], n1 = (((b1 + a1[0]) + a1[1]) + a1[2]) + a1[3];
      if (jq == step) { s0 = n0; s1 = n1; }
    }
    const int    last = (lane / 5) * 5 + 4;
    const double t0 = __shfl(s0, last < 64 ? last : 63, 64), t1 = __shfl(s1, last < 64 ? last : 63, 64);
    if (live)
    {
      double r0[4], r1[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) { r0[x] = a0[x] / t0; r1[x] = a1[x] / t1; } // :298
      if (out)
      {
        double2 *o0 = reinterpret_cast<double2 *>(out + (size_t)(c * S + i0) * S + 4 * jq), *o1 = o0 + S / 2;
        o0[0] = double2{r0[0], r0[1]}; o0[1] = double2{r0[2], r0[3]};
        o1[0] = double2{r1[0], r1[1]}; o1[1] = double2{r1[2], r1[3]};
      }
      // A-operand order (phyhip_aa.hpp): lane (k, b, i) of row group r, k-chunk t holds P[category of block b][4r + i][4t + k]
      const int r = i0 >> 2, il = i0 & 3;
#pragma unroll
      for (int y = 0; y < MAXTABS; ++y)
        if ((mask >> y) & 1u)
        {
          double *const tab = tabs[y];
          for (int b = c; b < 4; b += cb)
#pragma unroll
            for (int x = 0; x < 4; ++x)
            {
              tab[aa_a_slot(jq, r, 16 * x + 4 * b + il)]     = r0[x];
              tab[aa_a_slot(jq, r, 16 * x + 4 * b + il + 1)] = r1[x];
            }
        }
    }
  }
}

static __global__ __launch_bounds__(256) void pmat20_kernel(const PmatParams q)
{
  constexpr int S = 20;
  extern __shared__ __attribute__((aligned(16))) double expt[]; // [C][S] | U [NE][S][S] | V [NE][S][S] | A-operand table [kAaMat]
  const int m = blockIdx.x, C = q.C;
  double l;
  int    mat, shd;
  if (q.indices) { l = q.lengths[m]; mat = q.indices[m]; shd = q.shadow ? q.shadow[m] : -1; }
  else
  {
    l = q.small_len[0]; mat = q.small_idx[0]; shd = q.small_shadow[0];
#pragma unroll
    for (int k = 1; k < kSmallPm; ++k)
      if (m == k) { l = q.small_len[k]; mat = q.small_idx[k]; shd = q.small_shadow[k]; }
  }
  if (shd >= 0)
  { // the old value first (as in pmat_kernel: the barriers below come before any overwrite)
    const double *src = q.pmats + (size_t)mat * C * S * S;
    double       *dst = q.pmats + (size_t)shd * C * S * S;
    for (int e = threadIdx.x; e < C * S * S; e += blockDim.x) dst[e] = src[e];
    if (q.afrag)
      for (int e = threadIdx.x; e < kAaMat; e += blockDim.x) q.afrag[(size_t)shd * kAaMat + e] = q.afrag[(size_t)mat * kAaMat + e];
  }
  const bool cls = q.class_axis != 0;
  const int  NE  = cls ? C : 1;
  double *Us = expt + C * S, *Vs = Us + NE * S * S, *tab = q.afrag ? Vs + NE * S * S : nullptr;
  for (int t = threadIdx.x; t < NE * S * S; t += blockDim.x)
  {
    Us[t] = q.U[t];
    Vs[t] = q.V[t];
  }
  pmat20_exponentials(expt, l, C, cls, q.R, q.rates, q.br_len_mult, q.l_min, q.l_max, (int)threadIdx.x, (int)blockDim.x);
  if (tab && C == 3) // (blocks beyond the category count -- C = 3: block 3 -- stay zero)
    for (int e = threadIdx.x; e < kAaMat; e += blockDim.x) tab[e] = 0.0;
  __syncthreads();
  double *const tabs[1] = {tab};
  pmat20_entries<1>(expt, Us, Vs, C, cls, (int)(threadIdx.x & 63), (int)(threadIdx.x >> 6), (int)(blockDim.x >> 6), q.pmats + (size_t)mat * C * S * S, tabs,
                    tab ? 1u : 0u);
  if (tab)
  {
    __syncthreads();
    double2       *dst = reinterpret_cast<double2 *>(q.afrag + (size_t)mat * kAaMat);
    const double2 *src = reinterpret_cast<const double2 *>(tab);
    for (int e = threadIdx.x; e < kAaMat / 2; e += blockDim.x) dst[e] = src[e];
  }
}

} // namespace phyhip
