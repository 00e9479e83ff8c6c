// phyhip_aa.hpp -- amino-acid (20-state) traversal on the FP64 matrix cores of gfx950, second generation.
//
// Why MFMA here and only here (BASELINE north star): per site-update the 20-state path does
// 2 x 4 x (20x20) matrix-vector products = 6480 flop against 1289 B of traffic; batched over patterns it is
// a dense [20 x 20] x [20 x n] contraction per (child, rate class).  The 4-state path has a 4x4 contraction per lane
// and stays on the VALU.
//
// Shape: everything runs on the four-block form  v_mfma_f64_4x4x4_4b_f64  (4 blocks of D[4x4] += A[4x4] * B[4x4],
// 16 cycles, the same 16 FMA per cycle as the 16x16x4 form), with
//     block b      = one RATE CATEGORY (C = 4; C = 2: two categories x two pattern groups; C = 1: four pattern groups)
//     columns j    = 4 patterns
//     rows i       = 4 output states of row group r (states 4r + i, r = 0..4)
//     k            = 4 input states of k-chunk t (states 4t + k, t = 0..4, ascending: the accumulation order over
//                    input states is the reference's FMA chain, src/avx.c:593-616 -- bit-identical results)
//   lane maps (measured, tools/probes/mfma4b.hip):  A lane = 16k + 4b + i,  B lane = 16k + 4b + j,  D lane = 16i + 4b + j.
// One wave therefore owns a "wave-tile" of 4 patterns x ALL categories (16 patterns x 1 category for C = 1): 25 MFMAs per
// child matrix, 20 rows = 5 row groups exactly (no padding), and
//   * the D fragments of an update ARE the B fragments its parent needs (lane 16i + 4b + j holds states i + 4r of
//     pattern j, category b, on input and on output): results are forwarded in registers, the product of the two
//     children is lane-local;
//   * the maximum over the categories of a pattern (the rescaling rule, src/avx.c:498-510) and the category mixture
//     of Lk_Core (src/lk.c:816-818) are cross-LANE operations of one wave (DPP row rotations by 4 and 8 lanes): no LDS
//     exchange, no barrier -- the first generation (one wave per (16-pattern tile, category), 16x16x4 + 4x4x4 MFMAs)
//     spent a workgroup barrier per operation on them and ran its four waves in lock-step;
//   * waves are fully independent and small: 10 000 patterns are 2500 wave-tiles, 9.8 per CU (the first generation's
//     625 four-wave tiles filled 256 CUs as "2 or 3");
//   * the A operand (the transition matrix: 12.8 KB per matrix with all four categories, natural size, nothing
//     replicated) is the same for EVERY wave of the launch.  The first generation pulled 10 KB per wave-operation
//     through the per-CU vector-memory path (5 GB per launch at cfg3, 57 % of that path's bytes); here ONE loader
//     wave per workgroup stages the two matrices of each operation into an LDS ring (kAaRing operations deep) and
//     every other wave of the workgroup feeds its MFMAs straight from LDS (ds_read_b128: 256 B/clk/CU): the
//     vector-memory path carries only children and results.
// Synchronisation inside a workgroup is two kinds of LDS words and no barrier in the loop: `ready` (operations whose
// matrices are in the ring; written by the loader) and `done[w]` (operations wave w has finished reading; read by the
// loader before it overwrites a slot).  LDS executes a wave's instructions in order, so "data, then flag" needs no fence.
//
// Device layout of an amino-acid partials buffer ("fragment-major"):
//   [wave-tile][320 doubles: aa_slot(k-chunk t, lane)],  lane = 16 (state & 3) + 4 block + (pattern & 3)
// i.e. chunk pairs (0|1), (2|3) as 16 bytes per lane and chunk 4 as 8 bytes per lane: a lane's five values move as
// 2 x dwordx4 + 1 x dwordx2, each instruction one contiguous 1 KiB / 512 B access.  The host-facing layout
// ([pattern][category][state], t_edge::p_lk_*) is restored by phyhip_get_partials / accepted by phyhip_set_partials
// (aa_off below is the only place that knows the mapping).
#pragma once

#include "phyhip_kernels.hpp"
#include "phyhip_nt2.hpp"

namespace phyhip
{

typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int kAaT      = 5;    // k-chunks of 4 input states = row groups of 4 output states
constexpr int kAaBlock  = 320;  // doubles per wave-tile block of a partials buffer
#ifndef AA_RING
#define AA_RING 4
#endif
constexpr int kAaRing   = AA_RING;    // operations whose matrices the LDS ring holds (2 x 12.8 KB each)
constexpr int kAaMaxCons = 15;  // consumer waves per workgroup (+ 1 loader = 16 waves = 4 per SIMD at <= 128 VGPRs)


// Write matrix `mat`'s A-operand table (kAaMat doubles) from its natural [c][i][j] entries: `get(c, i, j)`.
// Lane (k, b, i) of the A operand holds P[category of block b][4r + i][4t + k]; blocks beyond the category count
// (C = 3: block 3) get zeros, so their lanes produce zeros.
template <typename F> __device__ __forceinline__ void aa_fill_atable(double *dst, int C, F get)
{
  const int cb = aa_cb(C);
  for (int e = threadIdx.x; e < kAaMat; e += blockDim.x)
  {
    const int lane = e & 63, rt = e >> 6, r = rt % kAaT, t = rt / kAaT;
    const int k = lane >> 4, b = (lane >> 2) & 3, i = lane & 3, c = b % cb;
    dst[aa_a_slot(t, r, lane)] = (c < C) ? get(c, 4 * r + i, 4 * t + k) : 0.0;
  }
}

struct FragParams
{
  const int    *indices; // nullptr: use small_idx
  int           small_idx[kSmallPm];
  int           count;
  int           C;
  const double *pmats; // natural [m][c][i][j]
  double       *afrag;
};

static __global__ __launch_bounds__(256) void aa_frag_kernel(const FragParams q)
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
  aa_fill_atable(q.afrag + (size_t)m * kAaMat, q.C, [&](int c, int i, int j) { return src[(size_t)c * 400 + i * 20 + j]; });
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
  int           shadow[kUploadBatch]; // slot that receives the matrix's OLD value first, or -1 (virtual buffers, phyhip_host.hpp)
  const double *src[kUploadBatch]; // host-pinned, device-accessible
  double       *pmats;
  double       *afrag;             // nullptr unless 20 states
};

static __global__ __launch_bounds__(256) void upload_matrices_kernel(const MatUploadParams q)
{
  extern __shared__ __attribute__((aligned(16))) double mat[]; // [C][S][S]
  int           m   = q.idx[0], shd = q.shadow[0];
  const double *src = q.src[0];
#pragma unroll
  for (int k = 1; k < kUploadBatch; ++k)
    if ((int)blockIdx.x == k) { m = q.idx[k]; src = q.src[k]; shd = q.shadow[k]; }
  const int n = q.C * q.S * q.S;
  if (shd >= 0)
  { // (as in pmat_kernel: the old value moves to the snapshot slot before anything of the matrix is overwritten)
    for (int e = threadIdx.x; e < n; e += blockDim.x) q.pmats[(size_t)shd * n + e] = q.pmats[(size_t)m * n + e];
    if (q.afrag)
      for (int e = threadIdx.x; e < kAaMat; e += blockDim.x) q.afrag[(size_t)shd * kAaMat + e] = q.afrag[(size_t)m * kAaMat + e];
  }
  double *out = q.pmats + (size_t)m * n;
  if (!q.afrag)
  { // (no A-operand table to derive: straight from the staging memory to the slot, element e by the thread that moved its old
    // value -- no LDS, so any category count fits)
    for (int e = threadIdx.x; e < n; e += blockDim.x) out[e] = src[e];
    return;
  }
  for (int e = threadIdx.x; e < n; e += blockDim.x) mat[e] = src[e];
  __syncthreads();
  for (int e = threadIdx.x; e < n; e += blockDim.x) out[e] = mat[e];
  if (q.afrag)
    aa_fill_atable(q.afrag + (size_t)m * kAaMat, q.C, [&](int c, int i, int j) { return mat[(size_t)c * 400 + i * 20 + j]; });
}

// ---------------------------------------------------------------------------------------------
// K1 + K2 for 20 states.  Workgroup = 1 loader wave (wave 0) + up to kAaMaxCons consumer waves, one wave-tile each.
// Pipeline of a consumer: the host hands every operation over as ready-made buffer descriptors (size 0 = load disabled)
// plus forwarding flags (tips: one 32-bit allowed-state mask per pattern); within step k the children and the auxiliary words of operation k+1 are issued before the
// matrix-core phase into the second raw register set; the previous result is forwarded in registers.  One auxiliary
// dword per child: the scale word of an internal child or the aligned four tip codes of a tip child (the host points the
// descriptor at whichever row exists).
// ---------------------------------------------------------------------------------------------
// ABL (diag build only, results INVALID): 1 no matrix phase, 2 no cross-lane maximum, 4 children loads and result stores
// zero-sized, 8 no ring hand-shake (nothing is loaded), 16 no all-ones test
// ARGS: one- and two-operation launches (an SPR regraft candidate) carry their records in the kernel arguments
// (TreeParams::arg_ir / arg_xr, as the nucleotide kernel's short launches do): no staged copy in front of the launch.  A
// separate instantiation: the records of the list form must stay provably uniform (scalar loads through the noalias
// parameters) -- selecting between two pointers at run time turned every record load into a vector load (+35 % kernel time).
// INL (list form only): an operation may carry ONE child that is a virtual tip x tip result (kOpCh1 / kOpCh2, phyhip_host.hpp):
// its two matrix tables ride in the operation's ring item (four tables instead of two; the host hands out the table slots of
// the 8-table ring and says what must have been released before an item is written), its tips' mask words come with the
// operation's children, and the consumer computes it inside the step -- exactly the step its defining operation would have
// been (matrix columns for one-state tips, products through the matrix cores otherwise, the all-ones rule, the rescaling
// rule) -- instead of that operation occupying a pipeline step of its own.
template <int C_, bool DBG = false, int ABL = 0, bool ARGS = false, bool INL = false>
__global__ __launch_bounds__(64 * (kAaMaxCons + 1)) void traverse_aa_kernel(const TreeParams q, const IssueRec *__restrict__ irec,
                                                                            const ExecRec *__restrict__ xrec,
                                                                            const double *__restrict__ afrag, int n_frag_mats,
                                                                            const uint32_t *__restrict__ tip_masks,
                                                                            unsigned long long *dbg = nullptr)
{
  constexpr int T   = kAaT;
  auto IR = [&](int i) -> IssueRec {
    if constexpr (ARGS) return i ? q.arg_ir[1] : q.arg_ir[0];
    else return irec[i];
  };
  auto XR = [&](int i) -> ExecRec {
    if constexpr (ARGS) return i ? q.arg_xr[1] : q.arg_xr[0];
    else return xrec[i];
  };
  constexpr int CB  = C_ == 1 ? 1 : (C_ == 2 ? 2 : 4); // blocks (categories) per pattern
  constexpr int NPW = 16 / CB;                          // patterns per wave-tile
  static_assert(C_ >= 1 && C_ <= 4, "one MFMA block per category: at most four");
  static_assert(!INL || (!ARGS && !DBG && ABL == 0), "in-step tip x tip children: list form only");

  __shared__ __attribute__((aligned(16))) double ring[kAaRing][2][kAaMat];
  // (read and written with relaxed workgroup-scope atomics: those stay plain ds_read / ds_write instructions, whereas a
  // volatile access to LDS is compiled as a flat access behind a vmcnt(0) wait)
  __shared__ int      s_ready;             // operations (+ the evaluation edge) whose matrices are in the ring
  __shared__ int      s_done[16];          // per consumer wave: operations whose matrices it has finished reading
  __shared__ double   s_wsum[16];          // per consumer wave: its share of the workgroup's sum
  __shared__ unsigned long long stamps[DBG ? 64 * 8 : 1];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw   = (int)(blockDim.x >> 6) - 1; // consumer waves
  const int C    = q.C;
  const size_t ntiles = (size_t)(q.Ppad / NPW);
  // (the list form is padded to an even length by the host -- the nucleotide kernel's convention; the argument form is not)
  const int n_ops   = ARGS ? q.n_real_ops : q.n_ops;
  const int n_items = n_ops + (q.edge_eval ? 1 : 0);

  if (threadIdx.x < 16)
  {
    const size_t tl = (size_t)blockIdx.x * nw + threadIdx.x;
    s_done[threadIdx.x] = ((int)threadIdx.x < nw && tl < ntiles) ? 0 : 0x7fffffff; // absent waves never hold a slot
  }
  if (threadIdx.x == 0) s_ready = 0;
  __syncthreads();
  const bool stamper = DBG && blockIdx.x == gridDim.x / 2 && threadIdx.x == 64;
#define PHY_STAMP(k_, i_)                                                                                              \
  if (DBG)                                                                                                             \
  {                                                                                                                    \
    const unsigned long long t_ = __builtin_readcyclecounter();                                                        \
    if (stamper && (k_) < 64) stamps[(k_) * 8 + (i_)] = t_;                                                            \
  }

  if (wave == 0)
  {
    __builtin_amdgcn_s_setprio(3); // a late matrix stalls every consumer of the workgroup: the loader issues first
    // ---- loader: the two A tables of item j (operation j, or the evaluation edge's matrix) -> ring[j % kAaRing] -------
    const __amdgpu_buffer_rsrc_t af_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(afrag), 0, (int)((size_t)n_frag_mats * kAaMat * 8), 0x00020000);
    constexpr int      kPieces = 2 * kAaMat * 8 / 1024; // 25 x 1 KiB per item
    constexpr unsigned kMatB   = kAaMat * 8;            // 12 800 B: piece 12 straddles the two tables
    static_assert(2 * kAaMat * 8 == kPieces * 1024 && kPieces == 25, "an item is 25 pieces of 1 KiB (the vmcnt(25) below)");
    // LDS-DMA (buffer_load_dwordx4 ... lds): the pieces go from L2 straight into the ring, no registers, no ds_write; a
    // wave's pieces land in issue order, so "item j has landed" is vmcnt <= kPieces once item j+1 has been issued -- two
    // items in flight (the 6-bit vmcnt counter allows 63 pieces).
    typedef __attribute__((address_space(3))) void *lds_ptr;
    int  done_seen = 0, flagged = 0; // items known to be released by every consumer / items published in s_ready
    // The loader's own flag accesses are written as instructions: the compiler orders every LDS access it can see behind
    // ALL pending LDS-DMA of the wave (s_waitcnt vmcnt(0)), which would allow only one item in flight.
    const unsigned a_ready = (unsigned)(uintptr_t)(lds_ptr)&s_ready, a_done = (unsigned)(uintptr_t)(lds_ptr)&s_done[lane & 15];
    auto publish   = [&](int n) {
      if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(a_ready), "v"(n) : "memory");
      flagged = n;
    };
    auto slot_free = [&](int need) { // every consumer has finished item need - 1
      if (done_seen >= need) return true;
      int d;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(a_done) : "memory");
      if (__builtin_amdgcn_ballot_w64(d < need) != 0) return false;
      done_seen = need;
      return true;
    };
    for (int j = 0; j < ((ABL & 8) ? 0 : n_items); ++j)
    {
      unsigned off1, off2, off3 = 0, off4 = 0;
      int      need = j - kAaRing + 1, tslot = (j % kAaRing) * 2; // (two tables per item: the slot is free once every consumer has finished item j - kAaRing)
      bool     four = false;
      if (j < n_ops)
      {
        const IssueRec rj = IR(j);
        off1 = rj.c1_data.x; off2 = rj.c2_data.x;
        if constexpr (INL)
        { // the host's plan for this item (flush_impl): table slot, what must be released first, and -- with an in-step child --
          // the two more tables
          four = rj.c1_tip.x != 0;
          off3 = (unsigned)rj.c1_tip.base; off4 = (unsigned)(rj.c1_tip.base >> 32);
          need = (int)rj.c1_tip.bytes; tslot = (int)rj.c2_tip.x;
        }
      }
      else
      {
        off1 = off2 = (unsigned)q.e_pm * kMatB;
        if constexpr (INL) { need = q.aa_e_need; tslot = q.aa_e_slot; }
      }
      if (!slot_free(need))
      { // the consumers may be waiting for what is still in flight: publish it before waiting for them
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
        asm volatile("" ::: "memory");
        publish(j);
        while (!slot_free(need)) __builtin_amdgcn_s_sleep(2);
      }
      asm volatile("" ::: "memory");
      char *slot = reinterpret_cast<char *>(&ring[0][0][0]) + (size_t)tslot * kMatB;
      auto pair = [&](char *to, const unsigned oa, const unsigned ob) { // two tables = 25 pieces of 1 KiB
#pragma unroll
        for (int g = 0; g < kPieces; ++g)
        {
          const unsigned b = (unsigned)g * 1024u + (unsigned)lane * 16u; // byte inside the pair
          lds_ptr dst = (lds_ptr)(to + g * 1024);
          if (g * 1024 + 1024 <= (int)kMatB) __builtin_amdgcn_raw_ptr_buffer_load_lds(af_rsrc, dst, 16, b, oa, 0, 0);
          else if (g * 1024 >= (int)kMatB) __builtin_amdgcn_raw_ptr_buffer_load_lds(af_rsrc, dst, 16, b - kMatB, ob, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(af_rsrc, dst, 16, b < kMatB ? b + oa : b - kMatB + ob, 0, 0, 0);
        }
      };
      pair(slot, off1, off2);
      if (INL && four) pair(slot + 2 * kMatB, off3, off4);
      if (flagged < j)
      { // everything but the item just issued has landed
        if (INL && four) __builtin_amdgcn_s_waitcnt(0xCF72); // vmcnt(50)
        else __builtin_amdgcn_s_waitcnt(0x4F79);             // vmcnt(25)
        asm volatile("" ::: "memory");
        publish(j);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    asm volatile("" ::: "memory");
    publish(n_items);
  }
  else
  {
    // ---- consumers -------------------------------------------------------------------------------------------------
    const int       w    = wave - 1;
    const long long tile = (long long)blockIdx.x * nw + w;
    if (tile < (long long)ntiles)
    {
      const int kk = lane >> 4, b = (lane >> 2) & 3, jj = lane & 3; // state residue / MFMA block / pattern within the group
      const int c  = b % CB;                                        // this lane's rate category
      const bool idle = c >= C_;                                    // C = 3: block 3 carries nothing
      const long long p0 = tile * NPW + (b / CB) * 4 + jj;          // < Ppad
      const bool      pact = p0 < q.P && !idle;
      const int       tips = q.tip_count;
      const bool      cls  = q.class_axis != 0;
      const unsigned  blk_bytes = (unsigned)((size_t)tile * kAaBlock * 8);
      const unsigned  voff_d16 = blk_bytes + lane * 16, voff_d8 = blk_bytes + 2048 + lane * 8; // chunk pairs | chunk 4
      // class axis (mixture classes as categories): every class has its own scale vector, [buffer][class][pattern]
      const unsigned  voff_s = ((cls ? (unsigned)(idle ? 0 : c) * (unsigned)q.Ppad : 0u) + (unsigned)p0) * 4u;
      const unsigned  voff_t = (unsigned)p0 * 4u; // tip rows: one allowed-state mask (32 bits) per pattern
      // the scale word of a pattern (of a (class, pattern)) is stored by ONE lane; the others aim past the end of the buffer
      const unsigned  voff_sst = (kk == 0 && !idle && (cls || c == 0)) ? voff_s : 0x7ffffff0u;

      struct Frag
      { // a lane's five k-chunk values as they come from memory
        u32x4 p01, p23;
        u32x2 p4;
      };
      struct Raw
      {
        Frag     a, b;
        unsigned xa, xb; // scale word, or the dword holding the tip code
        unsigned xt;     // INL: the mask word of the in-step child's SECOND tip
      };
      auto rsrc = [](const Desc &d) {
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(d.base), 0, (ABL & 4) ? 0 : (int)d.bytes, 0x00020000);
      };
      auto load_frag = [](Frag &f, const __amdgpu_buffer_rsrc_t r, unsigned v16, unsigned v8) {
        f.p01 = __builtin_amdgcn_raw_buffer_load_b128(r, v16, 0, PHYHIP_LOAD_AUX);
        f.p23 = __builtin_amdgcn_raw_buffer_load_b128(r, v16 + 1024, 0, PHYHIP_LOAD_AUX);
        f.p4  = __builtin_amdgcn_raw_buffer_load_b64(r, v8, 0, PHYHIP_LOAD_AUX);
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
        load_frag(r.a, rsrc(o.c1_data), voff_d16, voff_d8);
        load_frag(r.b, rsrc(o.c2_data), voff_d16, voff_d8);
        r.xa = __builtin_amdgcn_raw_buffer_load_b32(rsrc(o.c1_scale), o.c1_scale.x ? voff_t : voff_s, 0, 0);
        r.xb = __builtin_amdgcn_raw_buffer_load_b32(rsrc(o.c2_scale), o.c2_scale.x ? voff_t : voff_s, 0, 0);
        if constexpr (INL) r.xt = __builtin_amdgcn_raw_buffer_load_b32(rsrc(o.c2_tip), voff_t, 0, 0); // (size 0 without such a child)
      };
      // cross-lane helpers.  Lanes that share a (pattern, category) differ in bits 4-5 (the state residue): the two
      // half-exchange instructions of gfx950 put both partners' values side by side without an LDS round trip.
      auto xor16_pair = [](unsigned v, unsigned &a, unsigned &bb) {
        auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        a = r[0]; bb = r[1];
      };
      auto xor32_pair = [](unsigned v, unsigned &a, unsigned &bb) {
        auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        a = r[0]; bb = r[1];
      };
      auto and_states = [&](unsigned v) { // AND over the four lanes (kk = 0..3) of a (pattern, category)
        unsigned a, bb;
        xor16_pair(v, a, bb); v = a & bb;
        xor32_pair(v, a, bb); return a & bb;
      };
      auto max_states = [&](unsigned v) {
        unsigned a, bb;
        xor16_pair(v, a, bb); v = max(a, bb);
        xor32_pair(v, a, bb); return max(a, bb);
      };
      auto max_cats = [&](unsigned v) { // maximum over the categories of a pattern: lanes that differ in the block bits
        if (CB == 4)
        {
          v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false)); // row_ror:4
          v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false)); // row_ror:8
        }
        else if (CB == 2) v = max(v, (unsigned)__shfl_xor((int)v, 4, 64));
        return v;
      };
      auto sum_states = [&](double v) { // (kk0 + kk1) + (kk2 + kk3), whichever lane asks
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        return v;
      };
      auto tip_vec = [&](unsigned word, double (&x)[T]) {
        const unsigned m = (word ? word : 1u) >> kk;
#pragma unroll
        for (int t = 0; t < T; ++t) x[t] = ((m >> (4 * t)) & 1u) ? 1.0 : 0.0;
      };
      // wait until the ring holds item k; returns the item's first A table
      int ready_seen = 0;
      auto wait_item = [&](int k, int tslot) -> const double * { // tslot (INL): the item's first table slot
        while (!(ABL & 8) && ready_seen <= k)
        {
          ready_seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
          if (ready_seen <= k) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        if constexpr (INL) return &ring[0][0][0] + (size_t)tslot * kAaMat;
        else return &ring[k % kAaRing][0][0];
      };
      auto release_item = [&](int k) {
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_store(&s_done[w], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      };
      // u[r] = sum over input states of P[c][4r + i][.] * x[.]: five row groups x five k-chunks, k ascending
      auto mfma_chunk = [&](const double *A, int t, const double xt, double (&u)[T]) {
        const v2d   *pr = reinterpret_cast<const v2d *>(A + t * kAaBlock) + lane;
        const v2d    a01 = pr[0], a23 = pr[64];
        const double a4  = A[t * kAaBlock + 256 + lane];
        u[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a01.x, xt, u[0], 0, 0, 0);
        u[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a01.y, xt, u[1], 0, 0, 0);
        u[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a23.x, xt, u[2], 0, 0, 0);
        u[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a23.y, xt, u[3], 0, 0, 0);
        u[4] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4, xt, u[4], 0, 0, 0);
      };

      // column `state` of a table, rows 4r + kk of this lane's category: the A lane that holds P[c][4r + i][4t + k] is
      // 16k + 4b + i, so this (D) lane kk reads lane 16 (state & 3) + 4b + kk of k-chunk state >> 2
      auto tip_column = [&](const double *A, unsigned mask, double (&u)[T]) {
        const int st = __builtin_ctz(mask), la = 16 * (st & 3) + 4 * b + kk;
        const double *At = A + (st >> 2) * kAaBlock;
        const v2d    *pr = reinterpret_cast<const v2d *>(At) + la;
        const v2d     a01 = pr[0], a23 = pr[64];
        u[0] = a01.x; u[1] = a01.y; u[2] = a23.x; u[3] = a23.y; u[4] = At[256 + la];
      };

      // results of the last two operations (this lane's D fragments), alternating: step k finds the result of k-1 in one set and
      // writes its own over the result of k-2 in the other (a result that stays virtual -- phyhip_host.hpp -- exists only here)
      double   FA[T] = {0., 0., 0., 0., 0.}, FB[T] = {0., 0., 0., 0., 0.};
      unsigned scA = 0, scB = 0;

      if (n_ops > 0)
      {
        const int last = n_ops - 1;
        Raw       RA, RB;
        ExecRec   cur = XR(0);
        IssueRec  nx1 = IR((1 < last) ? 1 : last);
        issue_children(IR(0), RA);
        {
          // The loop body sees [children loads of k+1][4 result stores of k] in flight when step k+1 starts.  Four stores
          // through a zero-sized descriptor (dropped by the hardware, but counted) give the loop entry the same shape, so
          // the compiler's merged s_waitcnt counts never make a step wait for the previous step's stores.
          const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc(nullptr, 0, 0, 0x00020000);
          const u32x4                  z4   = {0u, 0u, 0u, 0u};
          const u32x2                  z2   = {0u, 0u};
          __builtin_amdgcn_raw_buffer_store_b128(z4, none, 0, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(z4, none, 1024, 0, 0); // distinct offsets: identical stores would be merged
          __builtin_amdgcn_raw_buffer_store_b64(z2, none, 2048, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(0u, none, 4096, 0, 0);
        }

        auto step = [&](const int k, Raw &R, Raw &Rn, double (&Fout)[T], unsigned &scout, const double (&Fprev)[T], const unsigned scprev) {
          const unsigned fl = cur.dst_data.x;
          double         x1[T], x2[T], o[T];
          unsigned       s1, s2;
          PHY_STAMP(k, 0)
          // The loads of operation k+1 go out first (their registers were consumed by step k-1), then the records of
          // k+2 / k+1 are requested into the scalar registers the issue just freed: both have the whole step to arrive.
          if (!ARGS || k < last) issue_children(nx1, Rn);
          const IssueRec nx2 = IR((k + 2 < last) ? k + 2 : last);
          const ExecRec  nxe = XR((k + 1 < last) ? k + 1 : last);
          // A tip child whose patterns all carry ONE state contributes a column of its matrix (the reference's Exex / Exin
          // kernels, src/avx.c:527-564): five values per lane straight from the ring, no product.  (Through the matrix
          // cores the result would be the same doubles -- the other 19 products are exact zeros -- at 25 MFMAs.)
          unsigned m1 = 0, m2 = 0;
          bool     hot1 = false, hot2 = false;
          const bool in1 = INL && (fl & kOpCh1), in2 = INL && (fl & kOpCh2); // (this child is computed below, from the ring)
          if (fl & kOpTip1)
          {
            m1   = R.xa ? R.xa : 1u; // (padding patterns carry no state: any column will do)
            hot1 = __builtin_amdgcn_ballot_w64((m1 & (m1 - 1u)) != 0u) == 0;
          }
          if (fl & kOpTip2)
          {
            m2   = R.xb ? R.xb : 1u;
            hot2 = __builtin_amdgcn_ballot_w64((m2 & (m2 - 1u)) != 0u) == 0;
          }
          if (in1) s1 = 0;
          else if (fl & kOpTip1) { if (!hot1) tip_vec(m1, x1); s1 = 0; }
          else if (fl & kOpF11)
          {
#pragma unroll
            for (int t = 0; t < T; ++t) x1[t] = Fprev[t];
            s1 = scprev;
          }
          else if (fl & kOpF12)
          {
#pragma unroll
            for (int t = 0; t < T; ++t) x1[t] = Fout[t];
            s1 = scout;
          }
          else { unpack(R.a, x1); s1 = R.xa; }
          if (in2) s2 = 0;
          else if (fl & kOpTip2) { if (!hot2) tip_vec(m2, x2); s2 = 0; }
          else if (fl & kOpF21)
          {
#pragma unroll
            for (int t = 0; t < T; ++t) x2[t] = Fprev[t];
            s2 = scprev;
          }
          else if (fl & kOpF22)
          {
#pragma unroll
            for (int t = 0; t < T; ++t) x2[t] = Fout[t];
            s2 = scout;
          }
          else { unpack(R.b, x2); s2 = R.xb; }
          PHY_STAMP(k, 1)
          // all-ones shortcut of the Inin kernel, per (pattern, category): src/avx.c:575-587 (never with a one-state tip)
          unsigned ones = 0;
          auto all_ones = [&]() {
            ones = 1;
#pragma unroll
            for (int t = 0; t < T; ++t) ones &= (unsigned)((x1[t] == 1.0) & (x2[t] == 1.0));
            if (!(ABL & 16)) ones = and_states(ones);
          };
          if (!hot1 && !hot2 && !in1 && !in2) all_ones();
          PHY_STAMP(k, 2)
          double u1[T] = {0., 0., 0., 0., 0.}, u2[T] = {0., 0., 0., 0., 0.};
          {
            const double *A = wait_item(k, (int)cur.dst_scale.x);
            PHY_STAMP(k, 3)
            if constexpr (INL)
            {
              if (in1 || in2)
              { // the in-step child: the step of its defining operation (tips a, b through tables 2 and 3 of this item)
                const unsigned ma = (in1 ? R.xa : R.xb) ? (in1 ? R.xa : R.xb) : 1u, mb = R.xt ? R.xt : 1u;
                const bool     hota = __builtin_amdgcn_ballot_w64((ma & (ma - 1u)) != 0u) == 0;
                const bool     hotb = __builtin_amdgcn_ballot_w64((mb & (mb - 1u)) != 0u) == 0;
                double         ua[T] = {0., 0., 0., 0., 0.}, ub[T] = {0., 0., 0., 0., 0.}, va[T], vb[T];
                if (hota) tip_column(A + 2 * kAaMat, ma, ua);
                else
                {
                  tip_vec(ma, va);
#pragma unroll
                  for (int t = 0; t < T; ++t) mfma_chunk(A + 2 * kAaMat, t, va[t], ua);
                }
                if (hotb) tip_column(A + 3 * kAaMat, mb, ub);
                else
                {
                  tip_vec(mb, vb);
#pragma unroll
                  for (int t = 0; t < T; ++t) mfma_chunk(A + 3 * kAaMat, t, vb[t], ub);
                }
                unsigned cones = 0;
                if (!hota && !hotb)
                {
                  cones = 1;
#pragma unroll
                  for (int t = 0; t < T; ++t) cones &= (unsigned)((va[t] == 1.0) & (vb[t] == 1.0));
                  cones = and_states(cones);
                }
                double  (&xc)[T] = in1 ? x1 : x2;
                unsigned cm = 0;
#pragma unroll
                for (int t = 0; t < T; ++t)
                {
                  xc[t] = cones ? 1.0 : ua[t] * ub[t];
                  cm    = max(cm, hi32(xc[t]));
                }
                if (C_ == 3 && idle) cm = 0;
                cm = max_states(cm);
                if (!cls) cm = max_cats(cm);
                unsigned csc = 0;
                if (cm < kHiInvTwoToLarge && q.apply_scaling)
                {
#pragma unroll
                  for (int t = 0; t < T; ++t) xc[t] *= kTwoToLarge;
                  csc = kLarge;
                }
                if (in1) s1 = csc; else s2 = csc;
                if (!hot1 && !hot2) all_ones(); // (the operation's own all-ones test, now that both children are there)
              }
            }
            if (ABL & 1)
            {
#pragma unroll
              for (int t = 0; t < T; ++t) { u1[t] = x1[t]; u2[t] = x2[t]; }
            }
            else
            {
              if (hot1) tip_column(A, m1, u1);
              else
              {
#pragma unroll
                for (int t = 0; t < T; ++t) mfma_chunk(A, t, x1[t], u1);
              }
              if (hot2) tip_column(A + kAaMat, m2, u2);
              else
              {
#pragma unroll
                for (int t = 0; t < T; ++t) mfma_chunk(A + kAaMat, t, x2[t], u2);
              }
            }
            release_item(k);
          }
          PHY_STAMP(k, 4)
          unsigned mxh = 0;
#pragma unroll
          for (int t = 0; t < T; ++t)
          {
            o[t] = ones ? 1.0 : u1[t] * u2[t];
            mxh  = max(mxh, hi32(o[t]));
          }
          if (C_ == 3 && idle) mxh = 0;
          if (!(ABL & 2))
          {
            mxh = max_states(mxh);
            if (!cls) mxh = max_cats(mxh); // a mixture class rescales alone
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
            Frag wv;
            pack(o, wv);
            __builtin_amdgcn_raw_buffer_store_b128(wv.p01, dr, voff_d16, 0, PHYHIP_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(wv.p23, dr, voff_d16 + 1024, 0, PHYHIP_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b64(wv.p4, dr, voff_d8, 0, PHYHIP_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b32(sc, gr, voff_sst, 0, 0);
          }
          PHY_STAMP(k, 6)
#pragma unroll
          for (int t = 0; t < T; ++t) Fout[t] = o[t];
          scout = sc;
          cur   = nxe;
          nx1   = nx2;
        };
        if constexpr (ARGS)
        {
          step(0, RA, RB, FA, scA, FB, scB);
          if (n_ops > 1) step(1, RB, RA, FB, scB, FA, scA);
          else
          { // the evaluation below expects the last result in the second set
#pragma unroll
            for (int t = 0; t < T; ++t) FB[t] = FA[t];
            scB = scA;
          }
        }
        else
        { // (the list form is padded to an even length: the last result ends up in the second set)
          for (int k = 0; k < n_ops; k += 2)
          {
            step(k, RA, RB, FA, scA, FB, scB);
            step(k + 1, RB, RA, FB, scB, FA, scA);
          }
        }
      }

      if (q.edge_eval)
      {
        // ---- K2: site likelihood at the evaluation edge (src/lk.c:608-645, 767-861) -------------------------------
        double contrib = 0.0;
        double   x[T], y[T], u[T] = {0., 0., 0., 0., 0.};
        unsigned sl, sr;
        auto side = [&](int idx, double (&v)[T], unsigned &sc) {
          if (idx < tips)
          {
            tip_vec(tip_masks[(size_t)idx * q.Ppad + p0], v);
            sc = 0;
          }
          else if (idx == q.last_dest)
          {
#pragma unroll
            for (int t = 0; t < T; ++t) v[t] = FB[t];
            sc = scB;
          }
          else
          {
            const double *src = q.partials + (size_t)(idx - tips) * (ntiles * kAaBlock) + (size_t)tile * kAaBlock;
#pragma unroll
            for (int t = 0; t < T; ++t) v[t] = src[aa_slot(t, lane)];
            sc = (unsigned)q.scales[(size_t)(idx - tips) * (cls ? C : 1) * q.Ppad + voff_s / 4];
          }
        };
        side(q.e_parent, x, sl);
        side(q.e_child, y, sr);
        {
          const double *A = wait_item(n_ops, q.aa_e_slot);
#pragma unroll
          for (int t = 0; t < T; ++t) mfma_chunk(A, t, x[t], u); // rows: right-side state
          release_item(n_ops);
        }
        const double *pi_c = q.pi + ((cls && !idle) ? c * 20 : 0);
        double part = 0.0;
#pragma unroll
        for (int t = 0; t < T; ++t) part += u[t] * (y[t] * pi_c[4 * t + kk]);
        const double lkc = sum_states(part);
        if (pact && kk == 0 && q.site_cat) q.site_cat[(size_t)p0 * C + c] = lkc;
        if (cls)
        { // per class: its likelihood (above) and its scale exponent; the mixture is combined by class_combine_kernel
          if (pact && kk == 0) q.fact[(size_t)c * q.P + p0] = q.apply_scaling ? (int)(sl + sr) : 0;
        }
        else
        {
          // the categories of this lane's pattern, in category order (src/lk.c:816-818)
          double site = 0.0;
#pragma unroll
          for (int cc = 0; cc < C_; ++cc) site += __shfl(lkc, (lane & ~(3 << 2)) | (((b / CB) * CB + cc) << 2), 64) * q.cat_w[cc];
          if (pact && kk == 0 && c == 0)
          {
            const double wt = q.wght[p0];
            int          f  = q.apply_scaling ? (int)(sl + sr) : 0;
            if (wt > kSmall)
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
              contrib = wt * lsl;
            }
            q.fact[p0] = f;
          }
          // this wave's share: fixed shuffle tree -> deterministic
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) contrib += __shfl_down(contrib, off, 64);
          if (lane == 0) s_wsum[w] = contrib;
        }
        if (q.fence_post) __threadfence(); // every wave's stores are in memory before the workgroup's sum is posted
      }
    }
  }

  if (DBG && dbg)
  {
    __syncthreads();
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0)
      for (int i = 0; i < 64 * 8; ++i) dbg[i] = stamps[i];
  }
#undef PHY_STAMP
  if (!q.edge_eval || q.class_axis) return;
  __syncthreads();
  if (wave == 0)
  { // the consumers' shares in wave order
    double tot = 0.0;
    for (int i = 0; i < nw; ++i)
      if ((size_t)blockIdx.x * nw + i < ntiles) tot += s_wsum[i];
    publish_block_sum(q, tot, lane);
  }
}

} // namespace phyhip
