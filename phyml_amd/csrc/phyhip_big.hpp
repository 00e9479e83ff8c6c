// phyhip_big.hpp -- the large-grid resident evaluator: launch-free SPR candidates, Lk(b), Update_Eigen_Lr and dLk for
// nucleotide alignments of any size (the reference's call patterns: src/spr.c:640-646, src/optimiz.c:607-663, src/lk.c:655-753).
//
// resident_nt2_kernel / resident_dlk_kernel (one tile per workgroup, up to 64 of them) serve alignments of up to ~2 000
// patterns.  Beyond that a scalar-returning call was a launch again: at 500 taxa x 100 000 patterns an SPR candidate cost
// 33.7 us for 11 us of data (pmat_kernel + a one-operation traversal: two launch calls, two dispatches, two kernel starts and
// ends, 3 126 block records handed to the host), a dLk 20 us for 3.  Here ONE kernel fills the device once -- a workgroup of
// NW waves per CU -- and stays:
//   * wave 0 of every workgroup polls (workgroup 0 the host-mapped command record, the others the device-memory mailbox it
//     relays each command through -- the mechanism of resident_poll_wave); the other waves sleep at the workgroup barrier
//     meanwhile: a parked wave issues nothing and reads nothing, so 2 048 resident waves cost 256 pollers;
//   * a wave then walks ITS tiles (tile = wave, wave + waves, ...): the launched kernels' per-tile bodies (nt2_run, dlk_tile)
//     unchanged, so every tile sum is the double the launched form computes; the transition matrices of the command are
//     rebuilt in the wave's prologue (nt2_run's n_fresh), once per command;
//   * one command kind per call of the surface: 0-2 partial updates + the edge evaluation (SPR candidate, Lk(b)), the same
//     + the eigen products instead of the sum (Update_Eigen_Lr), or a dLk / eigen-basis Lk;
//   * the sums leave either as one {sum, tag} record per tile (the host adds them, as after a launch) or -- flag bit 15 --
//     through device memory: the workgroup that draws the last ticket adds the tile sums in final_reduce_kernel's order and
//     posts ONE record per sum.  Same order, same double; which is faster depends on the number of tiles (measured).
// The resident workgroups are not ordered with the instance's stream: the host side (phyhip.hip, big_*) uses them only while
// the stream is known to be idle, exactly as for the small evaluators.
#pragma once
#include "phyhip_nt2.hpp"

namespace phyhip
{

// payload words: 0 tag, 1 flags, then
//   traversal command (resident_nt2_kernel's layout): 2 evaluation edge, 3 its matrix | last destination, 4-5 matrix indices,
//     6-9 their lengths, 10.. two operations x 12 descriptor words;
//   dLk command: 2 pinvar, 3.. the expl table (C x 2 x 4 doubles).
// flags: bits 0-1 operations, 2 device data changed since the last command, 4-7 matrices to rebuild, 8-9 evaluation sides to
// fetch early, 10 eigen products instead of the sum, 11 dLk command, 12 with derivative, 13 invariant-site model, 14 scaling,
// 15 final sum on the device, 16 (with 15) through one partial sum per workgroup, 17-18 host-computed matrices waiting in
// ResidentCtl::up_area (their indices in words 4-5; never together with matrices to rebuild).
constexpr int kBigWords = kResidentNtWords > 3 + 32 ? kResidentNtWords : 3 + 32;
static_assert((kBigWords + kResidentPay - 1) / kResidentPay <= 15, "a command must fit the one 512-byte read of a poll");
enum : unsigned long long
{
  kBigChanged = 1ull << 2, kBigEigen = 1ull << 10, kBigDlk = 1ull << 11, kBigDeriv = 1ull << 12, kBigInvar = 1ull << 13,
  kBigScaling = 1ull << 14, kBigDeviceSum = 1ull << 15, kBigGroupSum = 1ull << 16
};
// Final sum through one partial sum per workgroup: final_reduce_kernel adds the tile sums into 256 accumulators, tile i into
// accumulator i % 256 in the order of i, and then the accumulators in a binary tree.  With EXACTLY 256 resident workgroups the
// tiles of workgroup b are b, 256 + b, 512 + b, ... (tile = (round * NW + wave) * 256 + b): accumulator b, and nobody else's.
// So the waves leave their tile sums in LDS (in their staging areas), the workgroup adds them in that order and posts ONE {sum, tag} record to device
// memory, and workgroup 255 -- no tickets -- watches the 256 records until all carry the command's tag, then runs the tree:
// same additions, same double; no per-tile stores to memory, no wait for their acknowledgement, no dependent atomics (two
// round trips to memory), one load per thread instead of thirteen (3 126 tiles).  At most kBigGroupTiles tiles per workgroup.
constexpr int kBigGroupTiles = 8 * 16; // (16 rounds of the smallest workgroup fit every staging area twice over)
constexpr unsigned long long kBigGroupPatience = 20000; // wall-clock ticks (200 us) the watching workgroup waits for a record, + 8 per tile (a command is ~1 tick per tile)

struct BigCtl
{
  int           n_tiles;   // tiles of a traversal command (= the launched kernel's grid, grid_nt2)
  int           n_vdlk;    // tiles of a dLk command (= dlk64_kernel's grid = n_tiles)
  double       *tile_sums; // device [2][max(n_tiles, n_vdlk)]: the tile sums of a command whose final sum runs on the device
  unsigned     *tickets;   // device [1 + kTicketGroups], zero between commands
  const double *dot_prod;  // the eigen products (Update_Eigen_Lr's output, dLk's input)
  HostBlock    *wg_recs;   // device [2][kBigGroupWgs] {sum, tag}: a workgroup's partial sums (kBigGroupSum), tags never 0
  // PHYHIP_RESIDENT_STATS: wall-clock stamps per workgroup, [workgroup][16]: of the last command 0 command seen, 1 after the
  // workgroup's barrier, 2 wave 0 through with its tiles, 3 all waves through, 4 ticket drawn, 5 final sum posted (nullptr: none)
  unsigned long long *stamps;
};

// Everything the kernel is launched with, in ONE argument: each command re-reads what it needs from the argument segment
// (see the loop below), at fixed offsets.
struct BigArgs
{
  TreeParams     t;
  ResidentCtl    r;
  BigCtl         b;
  const double  *pmats;
  const uint8_t *tip_codes;
  // One-shot form: the same kernel LAUNCHED for one evaluation, the command in its arguments -- what an evaluation of a large
  // grid costs when the resident workgroups are not there (stream busy, first call of a streak, another instance holds the
  // device, profiling): one launch of 256 workgroups that rebuild the matrices themselves, walk the tiles and add per workgroup
  // (one record to the host), instead of pmat_kernel + a traversal of thousands of one-wave workgroups + thousands of records.
  // Same tile bodies, same order of additions: the same double.  n_one_shot = payload words used (0: resident form).
  int                n_one_shot;
  unsigned long long one_shot[kBigWords];
};

// A struct out of the argument segment (constant address space: scalar loads), dword by dword
template <typename T> __device__ __forceinline__ void karg_copy(T &dst, const __attribute__((address_space(4))) void *src)
{
  static_assert(sizeof(T) % 4 == 0, "whole dwords");
  const __attribute__((address_space(4))) unsigned *s = reinterpret_cast<const __attribute__((address_space(4))) unsigned *>(src);
  unsigned tmp[sizeof(T) / 4];
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; ++i) tmp[i] = s[i];
  __builtin_memcpy(&dst, tmp, sizeof(T));
}

// A pointer that came out of the re-read argument segment is, to the compiler, a pointer to anywhere (flat loads and stores,
// which count against the LDS counter too and wait for each other); those of a launched kernel's own arguments are known to
// be global.  Rebuilt as global pointers from their bits, they say the same.
template <typename T> __device__ __forceinline__ T *as_global(T *p)
{ // (through an integer: a pointer-to-pointer round trip between the address spaces is folded away before it can tell)
  const unsigned long long v = (unsigned long long)p;
  return (T *)(__attribute__((address_space(1))) T *)v;
}
__device__ __forceinline__ void globalize(TreeParams &q)
{
  q.partials = as_global(q.partials); q.scales = as_global(q.scales); q.wght = as_global(q.wght); q.pi = as_global(q.pi);
  q.cat_w = as_global(q.cat_w); q.invar = as_global(q.invar); q.site_lnl = as_global(q.site_lnl); q.site_lk = as_global(q.site_lk);
  q.site_cat = as_global(q.site_cat); q.fact = as_global(q.fact); q.dot_out = as_global(q.dot_out); q.warn = as_global(q.warn);
  q.pmats_rw = as_global(q.pmats_rw); q.host_blocks = as_global(q.host_blocks); q.tile_sums = as_global(q.tile_sums);
  q.block_sums = as_global(q.block_sums); q.tickets = as_global(q.tickets); q.result = as_global(q.result);
  q.result_host = as_global(q.result_host); q.warn_host = as_global(q.warn_host); q.warn_out = as_global(q.warn_out);
}

template <int C, int G, int NW>
__global__ __launch_bounds__(64 * NW, (NW + 3) / 4) void resident_big_kernel(const BigArgs args_)
{
  constexpr int                 CP = C == 3 ? 4 : C;
  __shared__ unsigned long long sh_raw[64];
  __shared__ int                sh_idx[4];
  __shared__ double             sh_len[4];
  __shared__ double             sh_expl[2 * 4 * 4];
  __shared__ int                sh_act;
  __shared__ unsigned           sh_last;
  __shared__ IssueRec           sh_ir[2]; // the command's operation records: nt2_run reads them from here when it needs them
  __shared__ ExecRec            sh_xr[2];
  __shared__ __attribute__((aligned(16))) double sh_dot[NW][(64 / G) * C * 4]; // per wave: staging of a tile's eigen products
  __shared__ double             sh_red[2][256];                                // the final sum's accumulators
  __shared__ int                sh_late;
  __shared__ unsigned long long sh_exp[256]; // the exp table (dev_exp): every wave rebuilds the command's matrices
  exp_tab_to_lds(sh_exp, (int)threadIdx.x, 64 * NW);
  unsigned long long last = args_.r.start_seq, t_last = wall_clock64();
  bool               mail_open = false, served = false;
  for (;;)
  {
    // Nothing that is the same for every command may be computed once in front of this loop and kept: left to itself the
    // compiler hoists all of it -- the launch arguments, every lane predicate, every address -- and holds it in registers
    // across the whole command (two hundred scalar registers spilled into vector registers, which at two waves per SIMD
    // spilled to scratch in turn; a launched kernel loads each argument next to its use).  So the roots of those
    // computations -- the argument segment's address, the thread's and the workgroup's number -- are laundered through an
    // empty asm at the top of every iteration, and everything derives from the laundered values.
    typedef const __attribute__((address_space(4))) char karg_char;
    unsigned long long kaddr = (unsigned long long)(karg_char *)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned           tid = threadIdx.x, bid = blockIdx.x, nwg = gridDim.x;
    asm volatile("" : "+s"(kaddr), "+v"(tid), "+s"(bid), "+s"(nwg));
    // (what comes out of an asm counts as different from lane to lane: say that these are not -- loads through the pointer
    // are scalar loads, the workgroup's number and count scalar registers)
    kaddr = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(kaddr >> 32)) << 32) |
            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)kaddr);
    bid = (unsigned)__builtin_amdgcn_readfirstlane((int)bid); nwg = (unsigned)__builtin_amdgcn_readfirstlane((int)nwg);
    karg_char *ka = (karg_char *)kaddr;
    // (the address space stays on the pointer: loads through it are scalar loads)
    typedef const __attribute__((address_space(4))) BigArgs karg_args;
    karg_args *A = reinterpret_cast<karg_args *>(ka);
    const int      lane = (int)(tid & 63), wid = (int)(tid >> 6);
    const int      gw = wid * (int)nwg + (int)bid, TW = NW * (int)nwg; // this wave among all of them
    const int      n_one_shot = A->n_one_shot;
    if (wid == 0)
    {
      int act;
      if (n_one_shot)
      { // the launched form: the command is in the arguments, and there is only one
        act = served ? 2 : 1;
        if (!served && lane < n_one_shot) sh_raw[resident_slot(lane)] = A->one_shot[lane];
        static_assert(kBigWords <= 64, "one lane per payload word");
      }
      else
      {
        ResidentCtl r;
        karg_copy(r, &A->r);
        r.cmd = as_global(r.cmd); r.mail = as_global(r.mail);
        for (;;)
        {
          act = resident_poll_wave<true>(r, last, t_last, mail_open, sh_raw, 1, lane, bid == 0);
          if (act) break;
          __builtin_amdgcn_s_sleep(2);
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) sh_act = act;
      if (act == 1)
      {
        if (lane < 4)
        {
          const unsigned long long ix = sh_raw[resident_slot(4 + lane / 2)];
          sh_idx[lane] = (int)(unsigned)(ix >> (32 * (lane & 1)));
          const unsigned long long lb = sh_raw[resident_slot(6 + lane)];
          __builtin_memcpy(&sh_len[lane], &lb, 8);
        }
        if (lane < 2 * 4 * 4)
        {
          const unsigned long long eb = sh_raw[resident_slot(3 + lane)];
          __builtin_memcpy(&sh_expl[lane], &eb, 8);
        }
        if (lane < 2 * 12)
        { // a descriptor is two payload words (base; bytes | spare << 32): per operation four child descriptors into the
          // issue record (its two tip descriptors are not used by this kernel), two destination descriptors into the other
          const int                o = lane / 12, j = lane % 12;
          const unsigned long long w = sh_raw[resident_slot(10 + lane)];
          unsigned long long      *dst = j < 8 ? reinterpret_cast<unsigned long long *>(&sh_ir[o]) + j
                                               : reinterpret_cast<unsigned long long *>(&sh_xr[o]) + (j - 8);
          *dst = w;
        }
      }
    }
    const unsigned long long st0 = wall_clock64();
    __syncthreads(); // (the other waves of the workgroup have been asleep here since they finished the previous command)
    if (sh_act == 2) return;
    // every word is the same for all lanes: make that known (loop bounds and flags belong in scalar registers)
    auto word = [&](int k) {
      const unsigned long long v = sh_raw[resident_slot(k)];
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
      return ((unsigned long long)hi << 32) | lo;
    };
    // The command's fields and the launch's control block are read where they are needed and AGAIN behind the tiles (from LDS
    // and from the argument segment: a few scalar loads) instead of living in registers across them -- scalar registers that
    // do not fit are kept in vector registers, and the four-category, one-lane-group kernel has none to spare.
    struct Cmd
    {
      unsigned long long tag, fl;
      BigCtl             b;
      HostBlock         *host_blocks;
    };
    auto fetch = [&](unsigned long long k0) {
      asm volatile("" : "+s"(k0) : : "memory");
      k0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(k0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)k0);
      karg_args *A0 = reinterpret_cast<karg_args *>((karg_char *)k0);
      Cmd        c;
      c.tag = word(0); c.fl = word(1);
      karg_copy(c.b, &A0->b);
      c.b.tile_sums = as_global(c.b.tile_sums); c.b.tickets = as_global(c.b.tickets); c.b.dot_prod = as_global(c.b.dot_prod);
      c.b.stamps = as_global(c.b.stamps); c.b.wg_recs = as_global(c.b.wg_recs);
      c.host_blocks = as_global(A0->t.host_blocks);
      return c;
    };
    int n_rec, ns;
    {
    const Cmd                cm = fetch(kaddr);
    const unsigned long long tag = cm.tag, fl = cm.fl;
    const bool               dsum = (fl & kBigDeviceSum) != 0, gsum = (fl & kBigGroupSum) != 0;
    const BigCtl            &b = cm.b;
    HostBlock *const         host_blocks = cm.host_blocks;
    auto stamp = [&](int i, unsigned long long t) {
      // (the sums as atomics whose result nobody waits for: a read-modify-write would hold wave 0 for a trip to memory per stamp)
      if (b.stamps && tid == 0) { b.stamps[(size_t)bid * 16 + i] = t; (void)__hip_atomic_fetch_add(&b.stamps[(size_t)bid * 16 + 8 + i], i ? t - st0 : 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    };
    stamp(0, st0);
    stamp(1, wall_clock64());
    // what kernels on the stream wrote since the last command (the host says whether any did) is re-read from memory
    if (fl & kBigChanged) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    constexpr int IT = CP / G > 0 ? CP / G : 1; // rounds of 64 (pattern, category) lanes per tile (dlk_tile)
    if (fl & kBigDlk)
    { // ---- dLk / Lk in the eigen basis: dlk64_kernel's virtual blocks ----
      DlkParams dq;
      dq.dot_prod = b.dot_prod; dq.wght = as_global(A->t.wght); dq.fact = as_global(A->t.fact); dq.cat_w = as_global(A->t.cat_w);
      dq.pi = as_global(A->t.pi); dq.invar = as_global(A->t.invar);
      dq.P = A->t.P; dq.C = C;
      double pinvar;
      {
        const unsigned long long pb = word(2);
        __builtin_memcpy(&pinvar, &pb, 8);
      }
      const DlkCall k = {(fl & kBigDeriv) ? 1 : 0, (fl & kBigInvar) ? 1 : 0, (fl & kBigScaling) ? 1 : 0, pinvar};
      n_rec = b.n_vdlk; ns = 2;
      for (int vb = gw, round = 0; vb < n_rec; vb += TW, ++round)
      {
        double v[2];
        dlk_tile<4, CP, IT>(dq, k, sh_expl, as_global(A->t.warn), (unsigned)vb, lane, v);
        if (lane == 0)
        {
          if (gsum)
          {
            sh_dot[wid][2 * round] = v[0];
            sh_dot[wid][2 * round + 1] = v[1];
          }
          else if (dsum)
          {
#pragma unroll
            for (int s = 0; s < 2; ++s)
            {
              unsigned long long bits;
              __builtin_memcpy(&bits, &v[s], 8);
              __hip_atomic_store(reinterpret_cast<unsigned long long *>(b.tile_sums) + (size_t)s * n_rec + vb, bits, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
            }
          }
          else
          {
            post_host_block(host_blocks + vb, v[0], tag);
            post_host_block(host_blocks + (size_t)n_rec + vb, v[1], tag);
          }
        }
      }
    }
    else
    { // ---- 0-2 partial updates + the edge evaluation (or the eigen products): traverse_nt2_kernel's tiles ----
      n_rec = b.n_tiles; ns = 1;
      // A command of two operations runs as two passes of the one-operation body per tile -- the first operation alone, then
      // the second with the evaluation; the host built its records without forwarding between the two, the second pass reads
      // what the first stored (same wave, program order) -- so that the kernel needs the registers of the one-operation form
      // only: three waves per SIMD instead of two at four categories in two lanes (158 against 206 registers), i.e. 3 072
      // resident waves: one round of tiles for alignments of up to 98 304 patterns instead of 65 536.
      const int n_tiles = n_rec, passes = ((int)(fl & 3) == 2) ? 2 : 1;
      bool      first = true;
      for (int it = 0, tile = gw; tile < n_tiles; ++it, tile = gw + (it / passes) * TW)
      {
        const int pass = it % passes;
        // (per tile what the outer loop does per command: nothing tile-invariant -- the launch's TreeParams, the command's
        // fields, every lane predicate and address -- may be computed in front of THIS loop and kept either: 50 vector and
        // 100 scalar registers, measured on the listing)
        unsigned long long kaddr2 = kaddr;
        unsigned           tid2 = tid;
        asm volatile("" : "+s"(kaddr2), "+v"(tid2));
        kaddr2 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(kaddr2 >> 32)) << 32) |
                 (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)kaddr2);
        karg_char *ka2 = (karg_char *)kaddr2;
        karg_args *A2 = reinterpret_cast<karg_args *>(ka2);
        const unsigned long long fl2 = word(1), ed = word(2), pm = word(3);
        TreeParams               q;
        karg_copy(q, &A2->t);
        q.host_tag = word(0);
        q.n_fresh = first ? (int)((fl2 >> 4) & 15) : 0; // (a wave rebuilds the command's matrices with its first tile)
        q.n_up = first ? (int)((fl2 >> 17) & 3) : 0;     // (... or takes the host-computed ones to their slots)
        const bool mid = passes == 2 && pass == 0;      // the first of two operations: no evaluation behind it
        q.e_prefetch = mid ? 0 : (int)((fl2 >> 8) & 3);
        q.edge_eval = mid ? 0 : ((fl2 & kBigEigen) ? 2 : 1); q.e_parent = (int)(unsigned)ed; q.e_child = (int)(unsigned)(ed >> 32);
        q.e_pm = (int)(unsigned)pm; q.last_dest = (int)(unsigned)(pm >> 32);
        q.tile_sums = (fl2 & kBigDeviceSum) ? A2->b.tile_sums : nullptr;
        globalize(q);
        if (fl2 & kBigGroupSum) q.tile_sums = tile_sums_in_wave();
        const int n_ops = (int)(fl2 & 3);
        NtFresh   fr;
        // (the eigen system rides in the launch's TreeParams, as in a launched kernel: scalar loads from the argument segment;
        // a model change makes the host launch a new generation)
        const char *kt = (const char *)ka2 + offsetof(BigArgs, t);
        fr.idx = sh_idx; fr.len = sh_len;
        fr.up_idx = sh_idx; fr.up_val = as_global(A2->r.up_area); fr.up_sys = true; fr.exp_lds = sh_exp;
        fr.evec = reinterpret_cast<const double *>(kt + offsetof(TreeParams, m_evec));
        fr.ivec = reinterpret_cast<const double *>(kt + offsetof(TreeParams, m_ivec));
        fr.eval = reinterpret_cast<const double *>(kt + offsetof(TreeParams, m_eval));
        fr.rates = reinterpret_cast<const double *>(kt + offsetof(TreeParams, m_rates));
        double *const my_dot = sh_dot[tid2 >> 6];
        const double  *pmats = as_global(A2->pmats);
        const uint8_t *tip_codes = as_global(A2->tip_codes);
        if (n_ops != 0) nt2_run<C, G, false, 1, 2, NW, true>(q, sh_ir + pass, sh_xr + pass, pmats, tip_codes, nullptr, fr, (unsigned)tile, tid2, my_dot);
        else nt2_run<C, G, false, 3, 2, NW, true>(q, sh_ir, sh_xr, pmats, tip_codes, nullptr, fr, (unsigned)tile, tid2, my_dot);
        first = false;
      }
    }
    stamp(2, wall_clock64());
    }
    const Cmd                cm = fetch(kaddr);
    const unsigned long long tag = cm.tag, fl = cm.fl;
    const bool               dsum = (fl & kBigDeviceSum) != 0, gsum = (fl & kBigGroupSum) != 0;
    const BigCtl            &b = cm.b;
    HostBlock *const         host_blocks = cm.host_blocks;
    auto stamp = [&](int i, unsigned long long t) {
      if (b.stamps && tid == 0) { b.stamps[(size_t)bid * 16 + i] = t; (void)__hip_atomic_fetch_add(&b.stamps[(size_t)bid * 16 + 8 + i], t - st0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    };
    if (gsum)
    { // ---- the final sum on the device, one partial sum per workgroup (kBigGroupSum above) ----
      __syncthreads(); // every wave's tile sums are in LDS
      stamp(3, wall_clock64());
      if (wid == 0 && lane < ns)
      {
        // (a wave's sums sit in its staging area: [round] of a traversal command, [round][2] of a dLk command)
        // (an Update_Eigen_Lr command has no sum: its record says "done", the staging areas hold eigen products)
        const int mine = (int)bid < n_rec && !(fl & kBigEigen) ? (n_rec - (int)bid + kBigGroupWgs - 1) / kBigGroupWgs : 0;
        double    acc = 0.0;
        for (int m = 0; m < mine; ++m) acc += sh_dot[m % NW][(m / NW) * ns + lane];
        post_host_block(b.wg_recs + lane * kBigGroupWgs + bid, acc, tag); // (written through, sum and tag in one piece)
      }
      stamp(4, wall_clock64());
      if (bid == kBigGroupWgs - 1)
      {
        if (tid == 0) sh_late = 0;
        __syncthreads();
        if (tid < kBigGroupWgs)
        {
          typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
          const unsigned long long t_poll = wall_clock64();
          for (int sidx = 0; sidx < ns; ++sidx)
          {
            u64x2 rec;
            for (;;)
            { // (like the atomic loads of the ticket path: from memory, where the other XCDs' workgroups wrote it through)
              asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(rec) : "v"(b.wg_recs + sidx * kBigGroupWgs + tid) : "memory");
              if (rec.y == tag) break;
              // (a resident workgroup that left: nobody answers, the host launches; the workgroups of a launch all come)
              if (!n_one_shot && wall_clock64() - t_poll > kBigGroupPatience + 8ull * (unsigned long long)b.n_tiles) { sh_late = 1; break; }
            }
            double d;
            const unsigned long long bits = rec.x;
            __builtin_memcpy(&d, &bits, 8);
            sh_red[sidx][tid] = d;
          }
          if (ns == 1) sh_red[1][tid] = 0.0;
        }
        __syncthreads();
        if (tid < 128) { sh_red[0][tid] += sh_red[0][tid + 128]; sh_red[1][tid] += sh_red[1][tid + 128]; }
        __syncthreads();
        if (wid == 0)
        {
          double t0 = sh_red[0][lane] + sh_red[0][lane + 64], t1 = sh_red[1][lane] + sh_red[1][lane + 64];
#pragma unroll
          for (int off = 32; off > 0; off >>= 1)
          {
            t0 += __shfl_down(t0, off, 64);
            t1 += __shfl_down(t1, off, 64);
          }
          if (lane == 0 && !sh_late)
          {
            post_host_block(host_blocks, t0, tag);
            if (ns == 2) post_host_block(host_blocks + 1, t1, tag);
            stamp(5, wall_clock64());
          }
        }
      }
    }
    else if (dsum)
    { // ---- the final sum on the device: tile sums written through, a ticket per workgroup, the last one adds and posts ----
      __builtin_amdgcn_s_waitcnt(0); // this wave's tile sums (atomic stores at agent scope: written through) are acknowledged ...
      __syncthreads();               // ... and every other wave's of the workgroup, before its ticket is drawn
      stamp(3, wall_clock64());
      if (tid == 0)
      { // two-level draw (see finish_sums: atomics on one address serialise)
        const unsigned g = bid % kTicketGroups, ng = nwg < kTicketGroups ? nwg : kTicketGroups;
        const unsigned members = (nwg - g + kTicketGroups - 1) / kTicketGroups;
        unsigned       lastwg = 0;
        if (__hip_atomic_fetch_add(b.tickets + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1)
          lastwg = __hip_atomic_fetch_add(b.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng - 1 ? 1u : 0u;
        sh_last = lastwg;
      }
      stamp(4, wall_clock64());
      __syncthreads();
      if (sh_last)
      { // final_reduce_kernel's order: 256 strided accumulators, then the binary tree.  Thread t < 256 owns accumulator t of
        // both sums; its loads go out together, up to sixteen per sum (one after the other -- a dependent trip to memory each
        // -- the 13 loads per thread of a 3 126-tile evaluation took 27 us, measured; the sums themselves were in memory
        // after 7), as atomic loads at agent scope: they see what the other XCDs' workgroups wrote through.
        double acc[2] = {0.0, 0.0};
        if (tid < 256)
          for (int i0 = (int)tid; i0 < n_rec; i0 += 16 * 256)
          {
            unsigned long long v[2][16];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
              for (int u = 0; u < 16; ++u)
                v[s][u] = (s < ns && i0 + u * 256 < n_rec)
                              ? __hip_atomic_load(reinterpret_cast<const unsigned long long *>(b.tile_sums) + (size_t)s * n_rec + i0 + u * 256,
                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                              : 0ull;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
              for (int u = 0; u < 16; ++u)
                if (s < ns && i0 + u * 256 < n_rec)
                {
                  double d;
                  __builtin_memcpy(&d, &v[s][u], 8);
                  acc[s] += d;
                }
          }
        if (tid < 256) { sh_red[0][tid] = acc[0]; sh_red[1][tid] = acc[1]; }
        __syncthreads();
        if (tid < 128) { sh_red[0][tid] += sh_red[0][tid + 128]; sh_red[1][tid] += sh_red[1][tid + 128]; }
        __syncthreads();
        if (wid == 0)
        {
          double t0 = sh_red[0][lane] + sh_red[0][lane + 64], t1 = sh_red[1][lane] + sh_red[1][lane + 64];
#pragma unroll
          for (int off = 32; off > 0; off >>= 1)
          {
            t0 += __shfl_down(t0, off, 64);
            t1 += __shfl_down(t1, off, 64);
          }
          // the counters are zero again before the host can send the next command (one store instruction of this wave,
          // acknowledged before lane 0 posts)
          if (lane <= (int)kTicketGroups) __hip_atomic_store(b.tickets + lane, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __builtin_amdgcn_s_waitcnt(0);
          if (lane == 0)
          {
            post_host_block(host_blocks, t0, tag);
            if (ns == 2) post_host_block(host_blocks + 1, t1, tag);
            stamp(5, wall_clock64());
          }
        }
      }
    }
    __syncthreads(); // (wave 0 rewrites the command's staging area with its next poll)
    if (!dsum && !gsum) stamp(3, wall_clock64());
    last = last + 1; t_last = wall_clock64(); served = true;
  }
}

// (defined in phyhip_big.hip, the only translation unit that instantiates the kernel)
int launch_resident_big(int C, int G, int workgroups, hipStream_t stream, const BigArgs &a);
int big_waves_per_workgroup(int C, int G); // (0: no kernel for that shape)

} // namespace phyhip
