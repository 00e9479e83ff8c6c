// phyhip_nt2.hpp -- nucleotide traversal, second generation: one lane = one PATTERN (all categories).
//
// Why: rocprofv3 showed the (pattern, category)-per-lane kernel to be bound by instruction issue, not by
// memory (SQ_WAIT_INST_ANY ~45 % of wave-cycles, ~270 wave-instructions per operation for only 16 patterns:
// every SIMD issues about one instruction per 4 cycles whatever its type, so 3125 waves x 98 operations x
// 270 instructions is ~150 us of pure issue time on 1024 SIMDs).  Giving a lane the whole pattern
//   * removes every cross-lane step (the max-scan of the rescaling rule, the category mixture and the scale
//     exponent are lane-local), the replicated per-pattern work of the four category lanes and the LDS
//     shuffles, and amortises the per-operation scalar work over 64 patterns instead of 16:
//     ~270 instructions per operation per wave for 64 patterns (4x fewer per pattern);
//   * lets tips use the reference's own shortcut: an unambiguous tip contributes the COLUMN P[.][s] of its
//     matrix (Exex / Exin, src/avx.c:527-564) -- a 32-byte LDS lookup per category instead of a 4x4 product.
//
// Layout ("pattern-minor"): buffer[c*2 + s/2][pattern][s%2], patterns padded to 64: a lane's state pair is one 16-byte
// element, a wave's access to one (category, state pair) row is one contiguous 1 KiB transaction (dwordx4 per lane:
// half the vector-memory instructions of an 8-byte form -- the per-CU vector-memory path is instruction-rate bound).
// The host-facing layout ([pattern][category][state]) is restored by phyhip_get_partials.
//
// Everything else is the pipeline of traverse_nt_kernel: host-prepared buffer descriptors (size 0 = load
// disabled, answered by the bounds check), loads of operation k+2 in flight while k is computed, results of
// the last two operations forwarded in registers (two alternating register files, no copies), both
// transition matrices of an operation staged once per wave into LDS and read back as broadcasts.
#pragma once
#ifndef PHYHIP_LOAD_AUX
#define PHYHIP_LOAD_AUX 2 // cache policy bits of the children loads: non-temporal too (read once: -1 ... -3 % on the large cases)
#endif
#ifndef PHYHIP_STORE_AUX
#define PHYHIP_STORE_AUX 2 // cache policy bits of the result stores: 2 = non-temporal (a result is read back at most once, much later: cfg2 kernel 186 -> 166 us, 1 M patterns -2.5 %; tools/gpu_store_ab.sh)
#endif

#include "phyhip_kernels.hpp"

namespace phyhip
{

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// G > 1 ("category groups"): a pattern is spread over G lanes, lane (g, pl) = g * (64/G) + pl owning the C/G
// categories g*C/G ... of pattern pl.  Mid-sized alignments (50 000 patterns = 782 waves of 64 patterns on 1024
// SIMDs) leave every SIMD with at most one wave, so nothing overlaps a wave's FMA chain (256 FP64 FMAs per
// operation = ~1300 cycles) with its own memory waits; with G = 2 there are twice as many waves of half the
// arithmetic and half the registers, two or three per SIMD.  The only cross-lane traffic is one 32-bit exchange per
// operation (the rescaling maximum) and the category mixture of the edge evaluation.  G = 1 stays the choice for
// large alignments, where the kernel is HBM-bound and fewer, fatter waves issue fewer instructions per pattern.
// ARGS != 0 (short launches): the operation records are read from the kernel arguments (TreeParams::arg_ir / arg_xr) --
// ARGS = 1: one operation, 2: two, 3: none (evaluation only) -- with the count known at compile time, instead of
// the device slot ring -- a separate instantiation, so that the long-list kernel carries no trace of it.
// Where the prologue finds what it needs to rebuild transition matrices (lane-indexed reads: plain loads through these
// pointers).  traverse_nt2_kernel points them at its own argument segment (indexing the by-value struct would make the
// compiler copy it to scratch first), the resident kernel at device arrays and the command it received.
struct NtFresh
{
  const int    *idx;                         // [n_fresh] matrix indices
  const double *len;                         // [n_fresh] edge lengths
  const double *evec, *ivec, *eval, *rates;  // U, U^-1, eigenvalues, category rates
  const int    *up_idx = nullptr;            // [n_up] host-computed matrices of this launch (TreeParams::up_idx / up_val)
  const double *up_val = nullptr;            // [n_up][64]
  bool          up_sys = false;              // up_val is device memory the HOST wrote (resident commands, ResidentCtl::up_area): read past the caches
  const unsigned long long *exp_lds = nullptr; // resident kernels: their LDS copy of the exp table (dev_exp)
};

// One workgroup's (= one wave's) share of a launch: the whole kernel body, callable from a kernel that stays resident.
// DIST: how many operations ahead the loads run (2: two register sets in flight; 1: one -- 34 registers less per lane,
// a third wave per SIMD at G = 2; the host then flags forwarding from the previous operation only).
// NW: waves per workgroup (the large-grid resident evaluator runs several independent waves per workgroup, each with its own
// LDS staging area); tile: which PW patterns this wave works on -- the workgroup's number in a launched kernel.
// LREC (with ARGS != 0): the one or two records are read through irec / xrec -- LDS, where the resident evaluator put the
// command's -- each time they are needed, instead of living in scalar registers from the command's arrival to their use
// (the records of a launched kernel sit in its argument segment and cost no register until they are loaded).
// INL (list form only): operations may carry ONE child that is a virtual tip x tip result (kOpCh1 / kOpCh2): the step computes it
// from its two tips and two matrices -- exactly the step the defining operation would have been (column look-ups or row sums,
// the all-ones rule, the product, the rescaling rule) -- instead of that operation occupying a pipeline step of its own.  Costs
// every step one more auxiliary dword and one more 16-byte matrix piece per lane; launched only for lists that have such children.
template <int C, int G, bool DBG, int ARGS, int DIST = 2, int NW = 1, bool LREC = false, bool INL = false>
__device__ __forceinline__ void nt2_run(const TreeParams &q, const IssueRec *__restrict__ irec, const ExecRec *__restrict__ xrec,
                                        const double *pmats, // (not restrict: the prologue may rewrite entries)
                                        const uint8_t *__restrict__ tip_codes, unsigned long long *dbg, const NtFresh fr,
                                        const unsigned tile, const unsigned tid, double *lds_dot, const unsigned rec_no = 0xffffffffu)
{ // rec_no: which record the wave's sum is posted to (default: its tile's) -- traverse_nt2_mixed_kernel's four-lane waves // lds_dot: this wave's LDS staging area for the eigen products of its tile, (64 / G) * C * 4 doubles (edge_eval == 2) // (tid: threadIdx.x -- handed in, so that a kernel that runs this body inside a loop can keep what derives from it inside too)
  // DBG: cycle stamps of the first 64 steps of one wave (PHYHIP_ABLATE=8), kept in LDS until the end
  __shared__ unsigned long long stamps[DBG ? 64 * 8 : 1];
  const bool stamper = DBG && tile == gridDim.x / 2 && tid == 0;
#define PHY_STAMP(k_, i_)                                                                                              \
  if (DBG)                                                                                                             \
  {                                                                                                                    \
    const unsigned long long t_ = __builtin_readcyclecounter();                                                        \
    if (stamper && (k_) < 64) stamps[(k_) * 8 + (i_)] = t_;                                                            \
  }
  static_assert(C % G == 0 && 64 % G == 0, "category groups must divide the categories and the wave");
  constexpr int S = 4, CL = C / G, CS = CL * S, PW = 64 / G; // categories per lane, entries per lane, patterns per wave
  static_assert(!INL || (ARGS == 0 && DIST == 2 && NW == 1 && !LREC), "in-step tip x tip children: list form only");
  constexpr int NM = INL ? 4 : 2; // matrices staged per step: the operation's two (+ the two of its in-step child)
  __shared__ __attribute__((aligned(16))) double lds_all[NW][2][NM * C * 16]; // [wave][buffer][matrix][c][i][j]
  double (&lds_p)[2][NM * C * 16] = lds_all[NW > 1 ? (tid >> 6) : 0];

  const int      lane = NW > 1 ? (int)(tid & 63) : (int)tid;
  const int      grp  = lane / PW, pl = lane % PW;        // category group, pattern within the wave
  const int      c0   = grp * CL;                         // first category of this lane
  const unsigned p    = tile * (unsigned)PW + pl;         // < Ppad by construction of the grid
  constexpr int  HP   = CS / 2;                           // 16-byte state pairs per lane
  const unsigned rowb = (unsigned)(q.Ppad * 16);          // bytes between consecutive (c, state pair) rows
  // Class axis (mixture classes as categories, TreeParams::class_axis; the host then runs G = C: one class per lane): every
  // class has its own scale vector, [buffer][class][pattern], rescales alone and goes out of the edge evaluation alone
  const bool     cls    = (CL == 1) && q.class_axis != 0;
  const unsigned voff16 = p * 16u + (unsigned)(c0 * 2) * rowb, voff4 = (p + (cls ? (unsigned)c0 * (unsigned)q.Ppad : 0u)) * 4u;

  struct Raw
  {
    u32x4    a[HP], b[HP]; // child 1 / child 2 entries, as state pairs
    unsigned sa, sb;       // internal child: scale exponent; tip child: the aligned 4 bytes that hold its state byte
    unsigned tx;           // INL: the aligned 4 bytes that hold the state byte of the in-step child's SECOND tip
  };
  const __amdgpu_buffer_rsrc_t pm_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(pmats), 0, 0x7fffffff, 0x00020000);
  // (a descriptor is the same for every lane: said explicitly, so that records read from LDS end up in scalar registers --
  // for records that already are, the compiler drops the instruction)
  auto uni = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
  auto rsrc = [&](const Desc &d) {
    const unsigned long long base = ((unsigned long long)uni((unsigned)(d.base >> 32)) << 32) | uni((unsigned)d.base);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(base), 0, (int)uni(d.bytes), 0x00020000);
  };
  auto first_d = [](const u32x4 &v) { double d; __builtin_memcpy(&d, &v, 8); return d; };

  // issue every load an operation needs (which ones are live was decided by the host)
  auto issue_data = [&](const IssueRec &o, Raw &r) {
    // one auxiliary dword per child: a tip child has no scale vector and an internal child no tip byte, so the host
    // points the same descriptor at whichever row exists (spare word x = 1: tip row, addressed by aligned dword)
    const __amdgpu_buffer_rsrc_t d1r = rsrc(o.c1_data), d2r = rsrc(o.c2_data), g1r = rsrc(o.c1_scale), g2r = rsrc(o.c2_scale);
#pragma unroll
    for (int e = 0; e < HP; ++e) r.a[e] = __builtin_amdgcn_raw_buffer_load_b128(d1r, voff16, (unsigned)e * rowb, PHYHIP_LOAD_AUX);
#pragma unroll
    for (int e = 0; e < HP; ++e) r.b[e] = __builtin_amdgcn_raw_buffer_load_b128(d2r, voff16, (unsigned)e * rowb, PHYHIP_LOAD_AUX);
    r.sa = __builtin_amdgcn_raw_buffer_load_b32(g1r, uni(o.c1_scale.x) ? (p & ~3u) : voff4, 0, 0);
    r.sb = __builtin_amdgcn_raw_buffer_load_b32(g2r, uni(o.c2_scale.x) ? (p & ~3u) : voff4, 0, 0);
    if constexpr (INL)
    { // the in-step child's second tip row (size 0 when the operation has no such child: the load is dropped)
      const bool second = uni(o.c2_tip.x) != 0;
      Desc       ct;
      ct.base = second ? o.c2_tip.base : o.c1_tip.base; ct.bytes = second ? o.c2_tip.bytes : o.c1_tip.bytes; ct.x = 0;
      r.tx = __builtin_amdgcn_raw_buffer_load_b32(rsrc(ct), p & ~3u, 0, 0);
    }
  };
  auto issue_pm = [&](const IssueRec &o, u32x4 &pc, u32x4 &pc2) {
    // this lane's 16-byte piece of [matrix 1 | matrix 2] (C*16 doubles each)
    int ch = (lane < 16 * C) ? lane : 0;
    const int      mat = ch / (8 * C), within = ch - mat * 8 * C;
    const unsigned off = (mat ? uni(o.c2_data.x) : uni(o.c1_data.x)) + (unsigned)within * 16u;
    pc = __builtin_amdgcn_raw_buffer_load_b128(pm_rsrc, off, 0, 0);
    if constexpr (INL)
    { // ... and of the in-step child's two matrices (matrix 0 when there is none: read, staged and never used)
      const bool               c1 = uni(o.c1_tip.x) != 0, c2 = uni(o.c2_tip.x) != 0;
      const unsigned long long ab = c2 ? o.c2_data.base : (c1 ? o.c1_data.base : 0ull);
      const unsigned           off2 = (mat ? uni((unsigned)(ab >> 32)) : uni((unsigned)ab)) + (unsigned)within * 16u;
      pc2 = __builtin_amdgcn_raw_buffer_load_b128(pm_rsrc, off2, 0, 0);
    }
  };
  auto issue = [&](const IssueRec &o, Raw &r, u32x4 &pc, u32x4 &pc2) {
    issue_data(o, r);
    issue_pm(o, pc, pc2);
  };

  // u[c*4+i] = sum_j P[c][i][j] * x[c*4+j]: first product, then the FMA chain (src/avx.c:593-616);
  // the matrix rows come from LDS as wave-wide broadcasts
  auto matvec_x = [&](const double2 *M, const double (&x)[CS], double (&u)[CS]) {
#pragma unroll
    for (int c = 0; c < CL; ++c)
#pragma unroll
      for (int i = 0; i < S; ++i)
      {
        const double2 lo = M[(c0 + c) * 8 + 2 * i], hi = M[(c0 + c) * 8 + 2 * i + 1];
        double        a  = lo.x * x[c * 4];
        a = __builtin_fma(lo.y, x[c * 4 + 1], a);
        a = __builtin_fma(hi.x, x[c * 4 + 2], a);
        a = __builtin_fma(hi.y, x[c * 4 + 3], a);
        u[c * 4 + i] = a;
      }
  };
  auto unpack = [](const u32x4 (&xr)[HP], double (&x)[CS]) {
#pragma unroll
    for (int e = 0; e < HP; ++e) __builtin_memcpy(&x[2 * e], &xr[e], 16);
  };
  auto matvec_r = [&](const double2 *M, const u32x4 (&xr)[HP], double (&u)[CS]) {
    double x[CS];
    unpack(xr, x);
    matvec_x(M, x, u);
  };
  // tip child with allowed-state mask m: u[c*4+i] = sum_{j in m} P[c][i][j] (ascending j).  An unambiguous tip
  // (one state s) is the column lookup of the reference's Exex / Exin kernels (src/avx.c:527-564).
  auto tip_u = [&](const double *Md, const double2 *M, unsigned m, double (&u)[CS]) {
    const bool onehot = (m == 1u) || (m == 2u) || (m == 4u) || (m == 8u);
    if (__builtin_amdgcn_ballot_w64(!onehot) == 0)
    {
      const int s = (m >> 1) - (m >> 3); // 1,2,4,8 -> 0,1,2,3
#pragma unroll
      for (int e = 0; e < CS; ++e) u[e] = Md[(c0 * S + e) * 4 + s];
    }
    else
    {
#pragma unroll
      for (int c = 0; c < CL; ++c)
#pragma unroll
        for (int i = 0; i < S; ++i)
        {
          const double2 lo = M[(c0 + c) * 8 + 2 * i], hi = M[(c0 + c) * 8 + 2 * i + 1];
          double        a  = (m & 1u) ? lo.x : 0.0;
          a = (m & 2u) ? a + lo.y : a;
          a = (m & 4u) ? a + hi.x : a;
          a = (m & 8u) ? a + hi.y : a;
          u[c * 4 + i] = a;
        }
    }
  };

  // records: device slot ring, or the kernel arguments for launches of one or two operations (last <= 1)
  auto IR = [&](int i) -> IssueRec {
    if constexpr (ARGS != 0 && !LREC) return i ? q.arg_ir[1] : q.arg_ir[0];
    else return irec[i];
  };
  auto XR = [&](int i) -> ExecRec {
    if constexpr (ARGS != 0 && !LREC) return i ? q.arg_xr[1] : q.arg_xr[0];
    else return xrec[i];
  };
  const int last = ARGS ? 1 : q.n_ops - 1; // host pads the list to an even length
  Raw       RA, RB;
  u32x4     PA, PB, PA2, PB2; // (PA2 / PB2: INL only)
  // The children of the first two operations do not depend on the matrices rebuilt below: their loads go out first and
  // travel while the prologue computes (a launch of one or two operations is a chain of dependent round trips --
  // kernel arguments, matrices, children, evaluation edge -- and every one taken off the chain is ~1 us of ~9).
  constexpr bool single  = ARGS == 1; // one operation, records in the arguments: no padded second step
  const bool     has_ops = ARGS == 0 ? q.n_ops > 0 : ARGS != 3;
  if (has_ops)
  {
    issue_data(IR(0), RA);
    if constexpr (!single && DIST == 2) issue_data(IR((1 < last) ? 1 : last), RB);
  }
  // Short launches: the evaluation edge's far side(s) too, when no queued operation writes them (host's call)
  u32x4    EX[HP], EY[HP];
  unsigned esl = 0, esr = 0;
  if constexpr (ARGS)
  {
    if (q.edge_eval)
    {
      const size_t bufsz = (size_t)q.Ppad * C * S;
      auto pre = [&](int idx, u32x4 (&v)[HP], unsigned &sc) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(q.partials + (size_t)(idx - q.tip_count) * bufsz) + (size_t)(c0 * 2) * q.Ppad + p;
#pragma unroll
        for (int e = 0; e < HP; ++e) v[e] = src[(size_t)e * q.Ppad];
        sc = (unsigned)q.scales[(size_t)(idx - q.tip_count) * (cls ? C : 1) * q.Ppad + voff4 / 4];
      };
      if (q.e_prefetch & 1) pre(q.e_parent, EX, esl);
      if (q.e_prefetch & 2) pre(q.e_child, EY, esr);
    }
  }

  if (q.n_up > 0)
  { // host-computed matrices from the arguments to their slots (see TreeParams::n_up)
    for (int m = 0; m < q.n_up; ++m)
      if (lane < C * 16)
      {
        double v;
        if (fr.up_sys)
        {
          const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(fr.up_val) + m * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __builtin_memcpy(&v, &b, 8);
        }
        else v = fr.up_val[m * 64 + lane];
        q.pmats_rw[(size_t)fr.up_idx[m] * (C * 16) + lane] = v;
      }
    asm volatile("" ::: "memory"); // (as below: this wave's later loads see these stores)
    __builtin_amdgcn_wave_barrier();
  }
  if (q.n_fresh > 0)
  { // rebuild the queued matrices (see TreeParams::n_fresh) with pmat_kernel's arithmetic; every workgroup writes the same
    // values.  The double-precision exp() is the expensive part (four per entry when every lane builds its own entry: 2 us
    // of a 8 us launch, measured): a matrix needs only 4 C distinct ones, so lane (mm, c, k) computes exp(eval[k] * len(m0 +
    // mm, c)) for 64 / (4 C) matrices at once and the entries pick theirs up by shuffle -- one exp deep instead of 4 n_fresh.
    constexpr int PER = 4 * C, MPR = 64 / PER; // exps per matrix, matrices per round
    for (int m0 = 0; m0 < q.n_fresh; m0 += MPR)
    {
      const int mm = lane / PER, r = lane % PER, ec = r >> 2, ek = r & 3;
      double    ex = 0.0;
      if (mm < MPR && m0 + mm < q.n_fresh)
      {
        const double fl = fr.len[m0 + mm];
        double       len = (fl > 0.0 ? fl : 0.0) * fr.rates[ec]; // src/lk.c:2296
        len *= q.br_len_mult;
        if (len < q.l_min) len = q.l_min;
        else if (len > q.l_max) len = q.l_max;
        ex = dev_exp(fr.eval[ek] * len, fr.exp_lds);
      }
      // entry (c, i, j) of a matrix = lane (C <= 4: one pass); idle lanes compute entry 0 and store nothing, so that every
      // shuffle below runs with the whole wave active
      const bool live = lane < C * 16;
      const int  e = live ? lane : 0, c = e >> 4, i = (e >> 2) & 3, j = e & 3;
      double     u[4], v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { u[k] = fr.evec[i * 4 + k]; v[k] = fr.ivec[k * 4 + j]; }
      for (int m2 = 0; m2 < MPR && m0 + m2 < q.n_fresh; ++m2)
      {
        double *out = q.pmats_rw + (size_t)fr.idx[m0 + m2] * (C * 16);
        double  acc = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = __builtin_fma(u[k] * __shfl(ex, m2 * PER + c * 4 + k, 64), v[k], acc);
        acc = (acc < kSmallPij) ? kSmallPij : acc; // src/models.c:293
        // row sum in ascending j over the four lanes of the row (src/models.c:296-297)
        const double t0 = __shfl(acc, (lane & ~3) + 0, 64), t1 = __shfl(acc, (lane & ~3) + 1, 64),
                     t2 = __shfl(acc, (lane & ~3) + 2, 64), t3 = __shfl(acc, (lane & ~3) + 3, 64);
        const double sum = (((0.0 + t0) + t1) + t2) + t3;
        if (live) out[e] = acc / sum;
      }
    }
    // No wait for the stores: the loads below come from this same wave, and a wave's vector-memory instructions reach
    // the cache in program order -- a load issued after a store to the same address returns the stored value.
    asm volatile("" ::: "memory"); // nothing that reads the matrices moves above this point
    __builtin_amdgcn_wave_barrier();
  }

  double   FA[CS], FB[CS]; // results of the last two operations (alternating)
  unsigned scA = 0, scB = 0;
#pragma unroll
  for (int e = 0; e < CS; ++e) FA[e] = FB[e] = 0.0;

  if (has_ops)
  {
    issue_pm(IR(0), PA, PA2);
    if constexpr (!single && DIST == 2) issue_pm(IR((1 < last) ? 1 : last), PB, PB2);
    ExecRec  cur = XR(0);
    IssueRec nx2 = IR((DIST < last) ? DIST : last); // the next operation whose loads go out

    // One pipeline step: operation k; its loads are in (R, PC); Fprev = result of k-1, Fout = result of k-2
    // on entry and the result of k on exit.
    auto step = [&](const int k, const int parity, Raw &R, u32x4 &PC, u32x4 &PC2, double (&Fout)[CS], unsigned &scout,
                    const double (&Fprev)[CS], const unsigned scprev) {
      PHY_STAMP(k, 0)
      // the execute record of operation k+1 is the first thing the next step needs (its flags steer the operand
      // selection): its scalar load goes out first so that it has the whole step to come back from L2
      const ExecRec nx1 = XR((k + 1 < last) ? k + 1 : last);
      double2 *buf = reinterpret_cast<double2 *>(&lds_p[parity][0]);
      {
        double2 v;
        __builtin_memcpy(&v, &PC, 16);
        buf[(lane < 16 * C) ? lane : 0] = v;
        if constexpr (INL)
        {
          __builtin_memcpy(&v, &PC2, 16);
          buf[16 * C + ((lane < 16 * C) ? lane : 0)] = v;
        }
      }
      __builtin_amdgcn_wave_barrier();
      const double *bufd = &lds_p[parity][0];

      const unsigned fl = uni(cur.dst_data.x);
      double         u1[CS], u2[CS];
      unsigned       s1, s2;
      bool           one1, one2; // first-entry test of the all-ones shortcut
      const unsigned tsh = (p & 3u) * 8u; // position of this pattern's byte inside a tip row's dword
      const unsigned tca = (R.sa >> tsh) & 255u, tcb = (R.sb >> tsh) & 255u;
      // in-step child (INL): the step its defining operation would have been -- both children tips (tip_u), the all-ones rule
      // (both tips fully ambiguous: exactly 1.0), the product, the rescaling rule over all categories of the pattern
      double   XC[INL ? CS : 1];
      unsigned scx = 0;
      auto     cherry = [&](const unsigned t1, const unsigned t2) {
        if constexpr (INL)
        {
          double ua[CS], ub[CS];
          tip_u(bufd + 2 * C * 16, buf + 2 * C * 8, t1, ua);
          tip_u(bufd + 3 * C * 16, buf + 3 * C * 8, t2, ub);
          const bool all1 = (t1 == 15u) && (t2 == 15u);
          unsigned   mx = 0;
#pragma unroll
          for (int e = 0; e < CS; ++e)
          {
            const double v = all1 ? 1.0 : ua[e] * ub[e];
            XC[e] = v;
            mx    = max(mx, hi32(v));
          }
          if constexpr (G > 1)
          {
            if (!cls)
            {
#pragma unroll
              for (int d = PW; d < 64; d <<= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, d, 64));
            }
          }
          if (mx < kHiInvTwoToLarge && q.apply_scaling)
          {
#pragma unroll
            for (int e = 0; e < CS; ++e) XC[e] *= kTwoToLarge;
            scx = kLarge;
          }
        }
      };
      const unsigned tcx = INL ? ((R.tx >> tsh) & 255u) : 0u;
      bool in1 = false, in2 = false;
      if constexpr (INL) { in1 = (fl & kOpCh1) != 0; in2 = (fl & kOpCh2) != 0; }
      // ---- child 1 ----
      if constexpr (INL)
      {
        if (in1)
        {
          cherry(tca, tcx);
          matvec_x(buf, XC, u1);
          s1 = scx; one1 = (XC[0] == 1.0);
        }
      }
      if (in1) {}
      else if (fl & kOpTip1)
      {
        tip_u(bufd, buf, tca, u1);
        s1 = 0; one1 = (tca == 15u);
      }
      else if (fl & kOpF11) { matvec_x(buf, Fprev, u1); s1 = scprev; one1 = (Fprev[0] == 1.0); }
      else if (fl & kOpF12) { matvec_x(buf, Fout, u1); s1 = scout; one1 = (Fout[0] == 1.0); }
      else { matvec_r(buf, R.a, u1); s1 = R.sa; one1 = (first_d(R.a[0]) == 1.0); }
      PHY_STAMP(k, 1)
      // ---- child 2 ----
      if constexpr (INL)
      {
        if (in2)
        {
          cherry(tcb, tcx);
          matvec_x(buf + C * 8, XC, u2);
          s2 = scx; one2 = (XC[0] == 1.0);
        }
      }
      if (in2) {}
      else if (fl & kOpTip2)
      {
        tip_u(bufd + C * 16, buf + C * 8, tcb, u2);
        s2 = 0; one2 = (tcb == 15u);
      }
      else if (fl & kOpF21) { matvec_x(buf + C * 8, Fprev, u2); s2 = scprev; one2 = (Fprev[0] == 1.0); }
      else if (fl & kOpF22) { matvec_x(buf + C * 8, Fout, u2); s2 = scout; one2 = (Fout[0] == 1.0); }
      else { matvec_r(buf + C * 8, R.b, u2); s2 = R.sb; one2 = (first_d(R.b[0]) == 1.0); }

      PHY_STAMP(k, 2)
      // all-ones shortcut (src/avx.c:575-587): a category whose eight child entries are exactly 1.0 yields 1.0.
      // Only fully ambiguous subtrees get here; the full test runs when some lane passes the first-entry test.
      unsigned ones_mask = 0; // bit c: category c is all ones in both children
      if (__builtin_amdgcn_ballot_w64(one1 && one2))
      {
        double ra[CS], rb[CS];
        unpack(R.a, ra);
        unpack(R.b, rb);
#pragma unroll
        for (int c = 0; c < CL; ++c)
        {
          bool a1, a2;
          if (fl & kOpTip1) a1 = (tca == 15u);
          else
          {
            a1 = true;
#pragma unroll
            for (int j = 0; j < S; ++j)
              a1 = a1 && ((in1 ? XC[INL ? c * 4 + j : 0] : (fl & kOpF11) ? Fprev[c * 4 + j] : (fl & kOpF12) ? Fout[c * 4 + j] : ra[c * 4 + j]) == 1.0);
          }
          if (fl & kOpTip2) a2 = (tcb == 15u);
          else
          {
            a2 = true;
#pragma unroll
            for (int j = 0; j < S; ++j)
              a2 = a2 && ((in2 ? XC[INL ? c * 4 + j : 0] : (fl & kOpF21) ? Fprev[c * 4 + j] : (fl & kOpF22) ? Fout[c * 4 + j] : rb[c * 4 + j]) == 1.0);
          }
          ones_mask |= (a1 && a2) ? (1u << c) : 0u;
        }
      }

      PHY_STAMP(k, 3)
      // prefetch operation k+2 into the registers just freed; then the scalar records of the next step
      if constexpr (!ARGS) issue(nx2, R, PC, PC2); // (records in the arguments: at most two operations, nothing to prefetch)
      PHY_STAMP(k, 7)
      const IssueRec nx3 = IR((k + 1 + DIST < last) ? k + 1 + DIST : last);
      __builtin_amdgcn_wave_barrier();
      PHY_STAMP(k, 4)

      unsigned mxh = 0;
#pragma unroll
      for (int e = 0; e < CS; ++e)
      {
        const double v = ((ones_mask >> (e / 4)) & 1u) ? 1.0 : u1[e] * u2[e];
        Fout[e] = v;
        mxh     = max(mxh, hi32(v));
      }
      if constexpr (G > 1)
      { // the maximum runs over all categories of the pattern (src/avx.c:498-503): combine the category groups
        // (not for mixture classes: a class tree rescales on its own, src/mixt.c:2603-2640)
        if (!cls)
        {
#pragma unroll
          for (int d = PW; d < 64; d <<= 1) mxh = max(mxh, (unsigned)__shfl_xor((int)mxh, d, 64));
        }
      }
      unsigned sc = s1 + s2; // src/avx.c:462-464
      if (mxh < kHiInvTwoToLarge && q.apply_scaling)
      { // src/avx.c:504-510
#pragma unroll
        for (int e = 0; e < CS; ++e) Fout[e] *= kTwoToLarge;
        sc += kLarge;
      }
      scout = sc;
      PHY_STAMP(k, 5)
      {
        const __amdgpu_buffer_rsrc_t dr = rsrc(cur.dst_data), gr = rsrc(cur.dst_scale);
#pragma unroll
        for (int e = 0; e < HP; ++e)
        {
          u32x4 w;
          __builtin_memcpy(&w, &Fout[2 * e], 16);
          __builtin_amdgcn_raw_buffer_store_b128(w, dr, voff16, (unsigned)e * rowb, PHYHIP_STORE_AUX);
        }
        __builtin_amdgcn_raw_buffer_store_b32(sc, gr, voff4, 0, 0);
      }
      PHY_STAMP(k, 6)
      cur = nx1;
      nx2 = nx3;
    };

    if constexpr (ARGS != 0)
    {
      step(0, 0, RA, PA, PA2, FA, scA, FB, scB);
      if constexpr (single)
      { // the evaluation below expects the last result in the second register set
#pragma unroll
        for (int e = 0; e < CS; ++e) FB[e] = FA[e];
        scB = scA;
      }
      else step(1, 1, RB, PB, PB2, FB, scB, FA, scA);
    }
    else
      for (int k = 0; k < q.n_ops; k += 2)
      {
        step(k, 0, RA, PA, PA2, FA, scA, FB, scB);
        if constexpr (DIST == 2) step(k + 1, 1, RB, PB, PB2, FB, scB, FA, scA);
        else step(k + 1, 1, RA, PA, PA2, FB, scB, FA, scA);
      }
  }

  if (DBG && stamper && dbg)
    for (int i = 0; i < 64 * 8; ++i) dbg[i] = stamps[i];
#undef PHY_STAMP
  if (!q.edge_eval) return;

  // ---- K2: site likelihood at the evaluation edge (src/lk.c:608-645, 767-861), all lane-local ----------
  double contrib = 0.0;
  {
    const int    tips = q.tip_count;
    const size_t bufsz = (size_t)q.Ppad * C * S;
    double       x[CS], y[CS];
    unsigned     sl, sr;
    auto side = [&](int idx, double (&v)[CS], unsigned &sc, const int bit, const u32x4 (&ev)[HP], const unsigned esc) {
      if (ARGS && (q.e_prefetch & bit))
      {
        unpack(ev, v);
        sc = esc;
      }
      else if (idx < tips)
      {
        const unsigned m = tip_codes[(size_t)idx * q.Ppad + p];
#pragma unroll
        for (int e = 0; e < CS; ++e) v[e] = ((m >> (e & 3)) & 1u) ? 1.0 : 0.0;
        sc = 0;
      }
      else if (idx == q.last_dest)
      { // the last queued operation's result is still in registers (FB: the list length is even)
#pragma unroll
        for (int e = 0; e < CS; ++e) v[e] = FB[e];
        sc = scB;
      }
      else
      {
        const double2 *src = reinterpret_cast<const double2 *>(q.partials + (size_t)(idx - tips) * bufsz) + (size_t)(c0 * 2) * q.Ppad + p;
#pragma unroll
        for (int e = 0; e < HP; ++e)
        {
          const double2 t2 = src[(size_t)e * q.Ppad];
          v[2 * e] = t2.x; v[2 * e + 1] = t2.y;
        }
        sc = (unsigned)q.scales[(size_t)(idx - tips) * (cls ? C : 1) * q.Ppad + voff4 / 4];
      }
    };
    // this pattern's weight (and invariant state) are fetched with the partials, not after the arithmetic
    const unsigned pc_ = p < (unsigned)q.P ? p : 0u;
    const double   w_pre = q.wght[pc_];
    const int      iv_pre = q.invar_model ? (int)q.invar[pc_] : -1;
    side(q.e_parent, x, sl, 1, EX, esl);
    side(q.e_child, y, sr, 2, EY, esr);
    if (q.edge_eval == 2)
    { // K3 behind the partial update it needs (Update_Eigen_Lr, src/lk.c:1038-1114; arithmetic of src/avx.c:79-82 as in
      // eigen_lr_kernel): dot_prod[p][c][k] = (sum_i R[i][k] (x_i pi_i)) (sum_i L[k][i] y_i), first product, then the FMA chains
      // The products go out through LDS: a lane holds the CL x 32 bytes of its (pattern, categories) -- stored directly, every
      // instruction of the wave touched 32 different 128-byte lines, 16 bytes each (the fused form lost 2 us to eigen_lr_kernel
      // at small sizes and 4-7 us beyond 4 096 patterns for that).  The tile's PW x C x 32 bytes are one contiguous piece of
      // dot_prod ([pattern][category][state], padded patterns included): staged, then written 1 KiB per instruction.
      {
        double2 *stg = reinterpret_cast<double2 *>(lds_dot);
#pragma unroll
        for (int c = 0; c < CL; ++c)
        {
          double lp[4], d[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) lp[i] = x[c * 4 + i] * q.pi[i];
#pragma unroll
          for (int kq = 0; kq < 4; ++kq)
          {
            double a = fr.evec[kq] * lp[0];
            double b = fr.ivec[kq * 4] * y[c * 4];
#pragma unroll
            for (int i = 1; i < 4; ++i)
            {
              a = __builtin_fma(fr.evec[i * 4 + kq], lp[i], a);
              b = __builtin_fma(fr.ivec[kq * 4 + i], y[c * 4 + i], b);
            }
            d[kq] = a * b;
          }
          stg[(pl * C + c0 + c) * 2] = make_double2(d[0], d[1]);
          stg[(pl * C + c0 + c) * 2 + 1] = make_double2(d[2], d[3]);
        }
        __builtin_amdgcn_wave_barrier();
        double2 *dst = reinterpret_cast<double2 *>(q.dot_out + (size_t)tile * (PW * C * 4));
        constexpr int NCH = PW * C * 2 / 64; // 16-byte pieces per lane
#pragma unroll
        for (int j = 0; j < NCH; ++j) dst[j * 64 + lane] = stg[j * 64 + lane];
        __builtin_amdgcn_wave_barrier(); // (the next tile of a resident wave stages into the same area)
      }
      // completion as an evaluation's (launched or resident): stores fenced, then an empty record per workgroup
      if (LREC && q.tile_sums == tile_sums_in_wave()) return; // (adding per workgroup: the workgroup reports, phyhip_big.hpp)
      publish_block_sum(q, 0.0, lane, rec_no == 0xffffffffu ? tile : rec_no);
      return;
    }
    const double *__restrict__ M = pmats + (size_t)q.e_pm * (C * 16) + c0 * 16; // rows: right-side state
    double prod[CL];
#pragma unroll
    for (int c = 0; c < CL; ++c)
    {
      double t[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
      {
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < S; ++i) a = __builtin_fma(M[c * 16 + kk * 4 + i], x[c * 4 + i], a);
        t[kk] = a * (y[c * 4 + kk] * q.pi[(cls ? c0 * 4 : 0) + kk]);
      }
      const double lkc = (t[0] + t[2]) + (t[1] + t[3]);
      if (p < (unsigned)q.P && q.site_cat) q.site_cat[(size_t)p * C + c0 + c] = lkc;
      prod[c] = lkc * q.cat_w[c0 + c];
    }
    if (cls)
    { // per class: its likelihood (above) and its scale exponent; the mixture is combined by the caller's combination kernel
      if (p < (unsigned)q.P) q.fact[(size_t)c0 * q.P + p] = q.apply_scaling ? (int)(sl + sr) : 0;
      return;
    }
    double site = 0.0; // src/lk.c:816-818: categories in ascending order, whichever lane holds them
#pragma unroll
    for (int c = 0; c < C; ++c)
    {
      double t = prod[c % CL];
      if constexpr (G > 1) t = __shfl(t, (c / CL) * PW + pl, 64);
      site += t;
    }
    if ((G == 1 || grp == 0) && p < (unsigned)q.P) // the group-0 lane of a pattern reports
    {
      const double w = w_pre;
      int          f = q.apply_scaling ? (int)(sl + sr) : 0;
      if (w > kSmall)
      {
        if (q.invar_model)
        {
          const int iv = iv_pre;
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
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) contrib += __shfl_down(contrib, off, 64);
  if (LREC && q.tile_sums == tile_sums_in_wave())
  { // (large-grid resident evaluator adding per workgroup, phyhip_big.hpp: the sum of the wave's n-th tile stays in the wave's
    // staging area, which a sum command does not use otherwise; tile = (n * NW + wave) * kBigGroupWgs + workgroup)
    if (lane == 0) lds_dot[tile / (unsigned)(NW * kBigGroupWgs)] = contrib;
    return;
  }
  publish_block_sum(q, contrib, lane, rec_no == 0xffffffffu ? tile : rec_no);
}

template <int C, int G = 1, bool DBG = false, int ARGS = 0, int DIST = 2, bool INL = false>
__global__ __launch_bounds__(64, DIST == 1 ? G + 1 : G) void traverse_nt2_kernel(const TreeParams q, const IssueRec *__restrict__ irec,
                                                             const ExecRec *__restrict__ xrec, const double *pmats,
                                                             const uint8_t *__restrict__ tip_codes,
                                                             unsigned long long *dbg = nullptr)
{
  typedef const __attribute__((address_space(4))) char karg_char;
  const char *ka = (const char *)(karg_char *)__builtin_amdgcn_kernarg_segment_ptr(); // q is the first argument
  NtFresh     fr;
  fr.idx  = reinterpret_cast<const int *>(ka + offsetof(TreeParams, fresh_idx));
  fr.len  = reinterpret_cast<const double *>(ka + offsetof(TreeParams, fresh_len));
  fr.evec = reinterpret_cast<const double *>(ka + offsetof(TreeParams, m_evec));
  fr.ivec = reinterpret_cast<const double *>(ka + offsetof(TreeParams, m_ivec));
  fr.eval = reinterpret_cast<const double *>(ka + offsetof(TreeParams, m_eval));
  fr.rates = reinterpret_cast<const double *>(ka + offsetof(TreeParams, m_rates));
  fr.up_idx = reinterpret_cast<const int *>(ka + offsetof(TreeParams, up_idx));
  fr.up_val = reinterpret_cast<const double *>(ka + offsetof(TreeParams, up_val));
  __shared__ __attribute__((aligned(16))) double lds_dot[(64 / G) * C * 4];
  nt2_run<C, G, DBG, ARGS, DIST, 1, false, INL>(q, irec, xrec, pmats, tip_codes, dbg, fr, blockIdx.x, threadIdx.x, lds_dot);
}

// Whole-tree traversals of mid-sized alignments with TWO wave shapes in one launch (four categories).  The two-lanes-per-pattern
// wave covers 32 patterns; 50 000 patterns are 1 563 of them on 256 CUs: six per CU and a seventh on 27 CUs -- and the kernel ends
// with those 27 (measured: 49 152 patterns = six per CU 127.6 us, 50 000 patterns 150 us, eight per CU 157 us).  So the first
// n2 workgroups -- a multiple of the CU count -- are two-lane waves, and the remaining patterns go to four-lane waves of 16
// patterns (one category per lane: about half a two-lane wave's work), one more SMALL wave on up to 256 CUs instead of one more
// big wave on a few.  The buffer layout does not depend on the lane grouping; a four-lane wave's tile number is its first pattern
// over 16, its record is its workgroup's.  The block sums are added in workgroup order like any launch's.
template <int C, bool INL>
__global__ __launch_bounds__(64, 2) void traverse_nt2_mixed_kernel(const TreeParams q, const IssueRec *__restrict__ irec,
                                                                   const ExecRec *__restrict__ xrec, const double *pmats,
                                                                   const uint8_t *__restrict__ tip_codes, const int n2)
{
  static_assert(C == 4, "four-lane waves: one category per lane");
  typedef const __attribute__((address_space(4))) char karg_char;
  const char *ka = (const char *)(karg_char *)__builtin_amdgcn_kernarg_segment_ptr(); // q is the first argument
  NtFresh     fr;
  fr.idx  = reinterpret_cast<const int *>(ka + offsetof(TreeParams, fresh_idx));
  fr.len  = reinterpret_cast<const double *>(ka + offsetof(TreeParams, fresh_len));
  fr.evec = reinterpret_cast<const double *>(ka + offsetof(TreeParams, m_evec));
  fr.ivec = reinterpret_cast<const double *>(ka + offsetof(TreeParams, m_ivec));
  fr.eval = reinterpret_cast<const double *>(ka + offsetof(TreeParams, m_eval));
  fr.rates = reinterpret_cast<const double *>(ka + offsetof(TreeParams, m_rates));
  fr.up_idx = reinterpret_cast<const int *>(ka + offsetof(TreeParams, up_idx));
  fr.up_val = reinterpret_cast<const double *>(ka + offsetof(TreeParams, up_val));
  __shared__ __attribute__((aligned(16))) double lds_dot[32 * C * 4];
  if ((int)blockIdx.x < n2)
    nt2_run<C, 2, false, 0, 2, 1, false, INL>(q, irec, xrec, pmats, tip_codes, nullptr, fr, blockIdx.x, threadIdx.x, lds_dot);
  else
    nt2_run<C, 4, false, 0, 2, 1, false, INL>(q, irec, xrec, pmats, tip_codes, nullptr, fr, 2u * (unsigned)n2 + (blockIdx.x - (unsigned)n2),
                                              threadIdx.x, lds_dot, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Resident form of the short launches (see resident_dlk_kernel in phyhip_kernels.hpp for the mechanism).  An SPR regraft
// candidate -- three matrices rebuilt, one partial update, the edge evaluation (src/spr.c:643-646) -- or an Lk(b) is a
// ~5 us kernel behind a ~3 us launch call, the dispatch, and the kernel's start and end.  While nothing else of the instance
// is on its stream, the one-wave workgroups of this kernel stay on the device and take those evaluations from the command
// record: payload words 0 tag, 1 flags (bits 0-1 operations, bit 2 device data changed, bits 4-7 matrices to rebuild, bits
// 8-9 evaluation sides to fetch early, bit 10 the eigen products of the edge -- Update_Eigen_Lr -- instead of its sum), 2 evaluation edge (parent | child << 32), 3 its matrix | last destination << 32,
// 4-5 matrix indices (of the matrices to rebuild, or -- bits 17-18 of the flags: how many -- of the host-computed matrices waiting in
// ResidentCtl::up_area), 6-9 their edge lengths, then per operation 12 words: the four child descriptors and the two
// destination descriptors of the launch form's records.  Everything else is the launch's TreeParams, fixed at launch.
// A workgroup completes and writes back its stores before it posts its sum (fence_post), so kernels launched afterwards
// -- on whichever XCD -- find them in memory.
// ---------------------------------------------------------------------------------------------
constexpr int kResidentNtWords = 10 + 2 * 12;

template <int C, int G>
__global__ __launch_bounds__(64, 1) void resident_nt2_kernel(const TreeParams sq, const ResidentCtl r, const double *pmats,
                                                             const uint8_t *__restrict__ tip_codes, const double *evec,
                                                             const double *ivec, const double *eval, const double *rates)
{
  __shared__ unsigned long long sh_raw[64];
  __shared__ int                sh_idx[4];
  __shared__ double             sh_len[4];
  __shared__ __attribute__((aligned(16))) double sh_dot[(64 / G) * C * 4];
  __shared__ unsigned long long sh_exp[256];
  const int          lane = threadIdx.x;
  unsigned long long last = r.start_seq, t_last = wall_clock64();
  bool               mail_open = false;
  exp_tab_to_lds(sh_exp, lane, 64);
  __builtin_amdgcn_wave_barrier();
  for (;;)
  {
    const int act = resident_poll_wave(r, last, t_last, mail_open, sh_raw, 1, lane, blockIdx.x == 0);
    if (act == 2) return;
    if (act == 0) continue;
    __builtin_amdgcn_wave_barrier();
    // every word is the same for all lanes: make that known (descriptors and loop bounds belong in scalar registers)
    auto word = [&](int k) {
      const unsigned long long v = sh_raw[resident_slot(k)];
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
      return ((unsigned long long)hi << 32) | lo;
    };
    auto desc = [&](int k) {
      Desc               d;
      const unsigned long long b = word(k + 1);
      d.base = word(k); d.bytes = (unsigned)b; d.x = (unsigned)(b >> 32);
      return d;
    };
    const unsigned long long fl = word(1), ed = word(2), pm = word(3);
    TreeParams               q = sq;
    q.host_tag = word(0);
    q.n_fresh = (int)((fl >> 4) & 15); q.e_prefetch = (int)((fl >> 8) & 3); q.n_up = (int)((fl >> 17) & 3);
    q.edge_eval = (fl >> 10) & 1 ? 2 : 1; q.e_parent = (int)(unsigned)ed; q.e_child = (int)(unsigned)(ed >> 32);
    q.e_pm = (int)(unsigned)pm; q.last_dest = (int)(unsigned)(pm >> 32);
    if (lane < 4)
    {
      const unsigned long long ix = sh_raw[resident_slot(4 + lane / 2)];
      sh_idx[lane] = (int)(unsigned)(ix >> (32 * (lane & 1)));
      const unsigned long long lb = sh_raw[resident_slot(6 + lane)];
      __builtin_memcpy(&sh_len[lane], &lb, 8);
    }
    const int n_ops = (int)(fl & 3);
#pragma unroll
    for (int o = 0; o < 2; ++o)
    {
      q.arg_ir[o].c1_data = desc(10 + o * 12); q.arg_ir[o].c2_data = desc(12 + o * 12);
      q.arg_ir[o].c1_scale = desc(14 + o * 12); q.arg_ir[o].c2_scale = desc(16 + o * 12);
      q.arg_xr[o].dst_data = desc(18 + o * 12); q.arg_xr[o].dst_scale = desc(20 + o * 12);
    }
    // what kernels on the stream wrote since the last command (the host says whether any did) is re-read from memory
    if (fl & 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __builtin_amdgcn_wave_barrier();
    NtFresh fr;
    fr.idx = sh_idx; fr.len = sh_len; fr.evec = evec; fr.ivec = ivec; fr.eval = eval; fr.rates = rates;
    fr.up_idx = sh_idx; fr.up_val = r.up_area; fr.up_sys = true; fr.exp_lds = sh_exp;
    if (n_ops == 1) nt2_run<C, G, false, 1>(q, nullptr, nullptr, pmats, tip_codes, nullptr, fr, blockIdx.x, threadIdx.x, sh_dot);
    else if (n_ops == 2) nt2_run<C, G, false, 2>(q, nullptr, nullptr, pmats, tip_codes, nullptr, fr, blockIdx.x, threadIdx.x, sh_dot);
    else nt2_run<C, G, false, 3>(q, nullptr, nullptr, pmats, tip_codes, nullptr, fr, blockIdx.x, threadIdx.x, sh_dot);
    last = last + 1; t_last = wall_clock64();
  }
}

} // namespace phyhip
