// phyhip_queue.hip -- the deferred operation queue turned into launches, and the waits for their scalars
// (libphyhip.so, gfx950 only; the units and what they share: phyhip_host.hpp)
#include "phyhip_host.hpp"
#include <algorithm>

namespace phyhip_host
{

// ---- virtual buffers (Instance::virt) ---------------------------------------------------------------------------------------
constexpr int kOpNoStore = 1; // DevOp::pad bit 0: the operation's result is forwarded in registers only (descriptors of size 0)
constexpr int kOpInl1 = 2, kOpInl2 = 4; // DevOp::pad bits 1, 2: child 1 / child 2 is computed inside the operation's own step (Instance::pending_inl)

void devirtualise(Instance *I, int buf)
{
  // whoever asks is about to read the buffer from memory -- a kernel outside the traversal launch (eigen_lr_kernel, a mixture
  // combination), a copy to the host, or a setter that changes what the buffer was computed from: the launch in between must not
  // leave it virtual AGAIN (a long queue that writes it with a tip x tip operation and reads it later would: rewrite_pending)
  if (buf >= I->tips && buf < I->nbuf && I->virt_min_ops > 0 && !I->keep_real_flag[buf])
  { // (a flag per buffer + the list of the flagged ones: constant-time lookup in rewrite_pending, no duplicates)
    I->keep_real_flag[buf] = 1;
    I->keep_real.push_back(buf);
  }
  if (I->n_virtual == 0 || buf < I->tips || buf >= I->nbuf || !I->virt[buf]) return;
  DevOp op = I->vdef[buf];
  op.pad   = 0;
  // in FRONT of what is queued: it reads tips and matrices only (unchanged since the launch that left it virtual), and a queued
  // operation may read it
  I->pending.insert(I->pending.begin(), op);
  I->mat_in_queue[op.pm1] = 1; I->mat_in_queue[op.pm2] = 1;
  I->virt[buf] = 0;
  --I->n_virtual;
  ++I->n_virt_material;
}

void devirtualise_all(Instance *I)
{
  for (int b = I->tips; b < I->nbuf && I->n_virtual > 0; ++b)
    if (I->virt[b]) devirtualise(I, b);
}

void devirtualise_matrix(Instance *I, int m)
{
  for (int b = I->tips; b < I->nbuf && I->n_virtual > 0; ++b)
    if (I->virt[b] && (I->vdef[b].pm1 == m || I->vdef[b].pm2 == m)) devirtualise(I, b);
}

void devirtualise_tip(Instance *I, int tip)
{
  for (int b = I->tips; b < I->nbuf && I->n_virtual > 0; ++b)
    if (I->virt[b] && (I->vdef[b].c1 == tip || I->vdef[b].c2 == tip)) devirtualise(I, b);
}

// The queue as it will be launched.  Short lists (or kernels without two-deep forwarding): the first one that reads a virtual
// buffer gets the storing definitions of ALL virtual buffers in front of it.  Long lists (may_virtualise): a tip x tip operation
// whose result is written once in this list, read only later in it and not by the evaluation leaves its place; every reader of a
// virtual buffer -- one left virtual by this list or by an earlier one -- computes it inside its own step (one per operation:
// DevOp::pad bits 1-2, pending_inl) or gets the definition re-issued in front of it, not storing (the kernels forward the
// results of the previous two operations in registers); what the evaluation edge reads is stored.
void rewrite_pending(Instance *I, const EdgeEval *ee, bool may_virtualise)
{
  I->pending_inl.clear();
  const int n0 = (int)I->pending.size();
  // (only launches of the LIST form: one- and two-operation launches -- records in the kernel arguments, the resident evaluators --
  // run kernels without in-step children and without the second forwarding level's guarantees: a list must keep three
  // operations even if each of its tip x tip operations leaves its place)
  bool virtualise = may_virtualise && I->virt_min_ops > 0 && n0 >= I->virt_min_ops && n0 >= 3;
  if (virtualise)
  {
    int stay = 0;
    for (const DevOp &o : I->pending) stay += !(o.c1 < I->tips && o.c2 < I->tips);
    virtualise = stay >= 3;
  }
  if (!virtualise && I->n_virtual == 0) return;
  if (!virtualise)
  { // A short launch.  If it reads a virtual buffer, EVERY virtual buffer is stored now, in this one launch: the short launches
    // of a tree search (SPR candidates, Lk(b), Br_Len_Opt) come in long runs after a full traversal, and one list of tip x tip
    // operations in front of the first of them costs less than a launch of three operations -- instead of a resident command
    // -- at each buffer's first use.
    bool reads = false;
    for (const DevOp &o : I->pending) reads = reads || (o.c1 >= I->tips && I->virt[o.c1]) || (o.c2 >= I->tips && I->virt[o.c2]);
    if (ee) reads = reads || (ee->parent >= I->tips && I->virt[ee->parent]) || (ee->child >= I->tips && I->virt[ee->child]);
    if (reads) devirtualise_all(I);
    // (a queued operation that WRITES a virtual buffer stores it: real again -- its old definition must never be stored over it)
    for (const DevOp &o : I->pending)
      if (I->virt[o.dest]) { I->virt[o.dest] = 0; --I->n_virtual; }
    return;
  }
  std::vector<unsigned char> skip;
  {
    std::vector<int> n_dest(I->nbuf, 0), first_read(I->nbuf, -1);
    for (int k = 0; k < n0; ++k)
    {
      const DevOp &o = I->pending[k];
      ++n_dest[o.dest];
      for (int c : {o.c1, o.c2})
        if (c >= I->tips && first_read[c] < 0) first_read[c] = k;
    }
    skip.assign(n0, 0);
    for (int k = 0; k < n0; ++k)
    {
      const DevOp &o = I->pending[k];
      if (o.c1 < I->tips && o.c2 < I->tips && n_dest[o.dest] == 1 && first_read[o.dest] > k &&
          !(ee && (ee->parent == o.dest || ee->child == o.dest)) &&
          !I->keep_real_flag[o.dest])
        skip[k] = 1;
    }
  }
  std::vector<DevOp>     L;
  std::vector<InlineDef> LI; // (parallel to L)
  L.reserve((size_t)n0 + (size_t)I->n_virtual + 8);
  LI.reserve(L.capacity());
  const InlineDef none{-1, -1, -1, -1};
  // both pipelined kernels compute ONE virtual child inside its reader's step (phyhip_nt2.hpp / phyhip_aa.hpp, INL); a second
  // one is re-issued as a non-storing operation in front of the reader
  const bool in_step = I->virt_inline && ((I->soa && I->nt_groups <= 2) || I->perm); // (the instantiations that exist: flush_impl)
  bool       any_inl = false;
  auto before_read = [&](int c, DevOp &reader, InlineDef &rin, int bit) {
    if (c < I->tips || !I->virt[c]) return;
    DevOp d = I->vdef[c];
    I->mat_in_queue[d.pm1] = 1; I->mat_in_queue[d.pm2] = 1;
    if (in_step && rin.a < 0 && reader.c1 != reader.c2)
    {
      rin = InlineDef{d.c1, d.c2, d.pm1, d.pm2};
      reader.pad |= bit;
      any_inl = true;
      ++I->n_virt_recomputed;
      return;
    }
    d.pad = kOpNoStore; ++I->n_virt_recomputed;
    L.push_back(d); LI.push_back(none);
  };
  for (int k = 0; k < n0; ++k)
  {
    DevOp o = I->pending[k];
    if (skip[k])
    { // from here on the buffer is what this operation says; nothing is launched for it until somebody reads it
      if (!I->virt[o.dest]) { I->virt[o.dest] = 1; ++I->n_virtual; }
      o.pad = 0;
      I->vdef[o.dest] = o;
      ++I->n_virt_skipped;
      continue;
    }
    InlineDef in = none;
    before_read(o.c1, o, in, kOpInl1);
    if (o.c2 != o.c1) before_read(o.c2, o, in, kOpInl2);
    L.push_back(o); LI.push_back(in);
    if (I->virt[o.dest]) { I->virt[o.dest] = 0; --I->n_virtual; } // (a storing operation: the buffer is real again)
  }
  auto store_now = [&](int b) { // what is read from memory behind this launch (or by its evaluation): stored
    if (b < I->tips || !I->virt[b]) return;
    DevOp d = I->vdef[b];
    d.pad = 0;
    I->virt[b] = 0; --I->n_virtual; ++I->n_virt_material;
    I->mat_in_queue[d.pm1] = 1; I->mat_in_queue[d.pm2] = 1;
    L.push_back(d); LI.push_back(none);
  };
  for (int b : I->keep_real) store_now(b);
  if (ee)
    for (int side : {ee->parent, ee->child}) store_now(side);
  // The pipelined kernels pad an odd list by running its last operation once more (flush_impl), at a position where the result
  // of the operation THREE steps back is no longer in registers: a child of the last operation produced there is then read from
  // memory -- so it must have been stored (two non-storing re-issues in front of a reader; reachable without in-step children)
  if ((L.size() & 1) && L.size() >= 3)
  {
    DevOp       &d1 = L[L.size() - 3];
    const DevOp &rd = L.back();
    if ((d1.pad & kOpNoStore) && (rd.c1 == d1.dest || rd.c2 == d1.dest))
    {
      d1.pad &= ~kOpNoStore;
      if (I->virt[d1.dest]) { I->virt[d1.dest] = 0; --I->n_virtual; ++I->n_virt_material; }
    }
  }
  I->pending.swap(L);
  if (any_inl) I->pending_inl.swap(LI);
}

// Host-computed matrices queued by phyhip_set_transition_matrix: one launch per kUploadBatch of them.
int flush_uploads(Instance *I)
{
  if (!I->up_idx.empty())
  { // (the upload kernel writes the matrix table: behind the exit of the large-grid resident workgroups, whose own rebuilt
    // matrices sit in their L2s until they leave -- reachable from entry points that keep them, phyhip.hip)
    big_release(I);
    I->touched_call = true;
  }
  size_t done = 0;
  while (done < I->up_idx.size())
  {
    const int n = (int)std::min<size_t>(I->up_idx.size() - done, kUploadBatch);
    MatUploadParams q;
    memset(&q, 0, sizeof q);
    q.count = n; q.S = I->S; q.C = I->C; q.pmats = I->d_pmats; q.afrag = I->perm ? I->d_afrag : nullptr;
    for (int k = 0; k < kUploadBatch; ++k) q.shadow[k] = -1;
    for (int k = 0; k < n; ++k) { q.idx[k] = I->up_idx[done + k]; q.src[k] = I->up_src[done + k]; q.shadow[k] = I->up_shadow[done + k]; }
    hipLaunchKernelGGL(upload_matrices_kernel, dim3(n), dim3(256), q.afrag ? sizeof(double) * (size_t)I->C * I->S * I->S : 0, I->stream, q);
    HIPCHK(hipGetLastError());
    done += n;
  }
  for (int m : I->up_idx) I->up_slot[m] = -1;
  I->up_idx.clear();
  I->up_src.clear();
  I->up_shadow.clear();
  I->n_up_shadow = 0;
  return 0;
}

// Rebuild every queued transition matrix on the device: one staged copy of (index, length) pairs, one launch.
int flush_pmats(Instance *I)
{
  big_release(I);
  I->touched_call = true;
  int done = 0, rc = 0;
  if (!I->up_idx.empty() && (rc = flush_uploads(I))) return rc;
  const int count = (int)I->pm_idx.size();
  while (done < count)
  {
    const int  n     = std::min(count - done, I->pm_scratch_cap);
    const bool small = n <= kSmallPm; // short lists (SPR: 3 per candidate) ride in the kernel arguments
    PmatParams q;
    memset(&q, 0, sizeof q);
    if (small)
    {
      for (int k = 0; k < kSmallPm; ++k) q.small_shadow[k] = -1;
      for (int k = 0; k < n; ++k)
      {
        q.small_idx[k] = I->pm_idx[done + k];
        q.small_len[k] = I->pm_len[done + k];
        q.small_shadow[k] = I->pm_shadow[done + k];
      }
    }
    else
    {
      void        *st = nullptr;
      const bool   sh = I->n_pm_shadow > 0; // (snapshots of old values for virtual buffers: a third array)
      const size_t bi = (sizeof(int) * n + 15) & ~size_t(15), bl = sizeof(double) * n;
      rc = I->ring.alloc(bi + bl + (sh ? bi : 0), I->stream, &st);
      if (rc) return rc;
      memcpy(st, I->pm_idx.data() + done, sizeof(int) * n);
      memcpy((char *)st + bi, I->pm_len.data() + done, bl);
      if (sh) memcpy((char *)st + bi + bl, I->pm_shadow.data() + done, sizeof(int) * n);
      if (I->pm_copy)
      {
        HIPCHK(hipMemcpyAsync(I->d_pmscratch, st, bi + bl, hipMemcpyHostToDevice, I->stream));
        q.indices = (const int *)I->d_pmscratch;
        q.lengths = (const double *)((char *)I->d_pmscratch + bi);
      }
      else
      { // the kernels read the (index, length) pairs straight from the pinned staging chunk: a few hundred bytes over
        // the host link cost less than a copy command ahead of the launch
        q.indices = (const int *)st;
        q.lengths = (const double *)((char *)st + bi);
      }
      if (sh) q.shadow = (const int *)((char *)st + bi + bl); // (always read from the staging chunk)
    }
    q.count = n;
    q.S = I->S; q.C = I->C; q.U = I->d_evec; q.V = I->d_ivec; q.R = I->d_eval; q.rates = I->d_catr;
    q.br_len_mult = I->br_len_mult; q.l_min = I->l_min; q.l_max = I->l_max; q.pmats = I->d_pmats;
    // (20 states: a short list is latency -- 1024 threads, two entries each, products pre-formed: 7.9 vs 11.5 us for three
    // matrices; a whole tree's 397 matrices: 512 threads without the extra phase 14.7 us; 256 / 1024 threads 16.3-17.4 / 15.5)
    int threads = (I->S == 4) ? 64 : (n <= 16 ? 1024 : 512);
    if (const char *e = diag_env("PHYHIP_PMAT_THREADS"))
    { // (a multiple of 64 within the kernel's launch bounds, or ignored)
      const int v = atoi(e);
      if (v >= 64 && v % 64 == 0 && v <= (I->S == 4 ? 64 : 1024)) threads = v;
    }
    q.afrag = I->perm ? I->d_afrag : nullptr; // 20 states: the MFMA A-operand fragments come out of the same kernel
    q.class_axis = I->class_axis ? 1 : 0;
    const size_t lds = sizeof(double) * ((size_t)2 * I->C * I->S + (size_t)2 * I->C * I->S * I->S + (size_t)2 * I->NE * I->S * I->S);
    // 20 states: the register-blocked kernel (phyhip_kernels.hpp: pmat20_kernel).  PHYHIP_PMAT20 (diag) = 0: the LDS-staged
    // pmat_kernel<20> builds every list, 1: it builds the lists of more than 16 matrices
    static const int pmat20 = diag_env("PHYHIP_PMAT20") ? atoi(diag_env("PHYHIP_PMAT20")) : 2;
    if (I->S == 4) hipLaunchKernelGGL((pmat_kernel<4, true>), dim3(n), dim3(threads), lds, I->stream, q);
    else if (pmat20 == 2 || (pmat20 == 1 && n <= 16))
      hipLaunchKernelGGL(pmat20_kernel, dim3(n), dim3(256), sizeof(double) * ((size_t)I->C * 20 + (size_t)2 * I->NE * 400 + (q.afrag ? kAaMat : 0)),
                         I->stream, q);
    else if (n <= 16) hipLaunchKernelGGL((pmat_kernel<20, true>), dim3(n), dim3(threads), lds, I->stream, q);
    else hipLaunchKernelGGL((pmat_kernel<20, false>), dim3(n), dim3(threads), lds, I->stream, q);
    HIPCHK(hipGetLastError());
    done += n;
  }
  for (int m : I->pm_idx) I->pm_slot[m] = -1;
  I->pm_idx.clear();
  I->pm_len.clear();
  I->pm_shadow.clear();
  I->n_pm_shadow = 0;
  return 0;
}

// Final sum inside the producing kernel (last workgroup) or as a separate 1-block kernel?  Measured on MI355X (round 2,
// docs/history/tools/gpu_step_ab.sh, with PHYHIP_SPLIT_REDUCE actually honoured): fused saves the second launch (~3.4 us + gap)
// whenever the grid is small -- every SPR / Br_Len_Opt call on small and mid-sized alignments.  On large grids it costs
// the traversal kernel ~6-10 % (cfg2 198 vs 186 us, 125 000 patterns 469 vs 414, 1 M 3.54 vs 3.22 ms, cfg3 574 vs 553):
// a workgroup must see its block sum acknowledged by memory before it draws its ticket, i.e. it waits for ALL its
// outstanding result stores instead of retiring behind them, and holds its wave slot meanwhile.  PHYHIP_SPLIT_REDUCE=0/1
// forces either.
bool fuse_reduce(const Instance *I, int nblocks)
{
  if (I->split_reduce_forced) return !I->split_reduce;
  return nblocks <= 512;
}

// The host-computed matrices a resident command carries, kept until it is answered (see flush_and_wait)
static void keep_uploads(Instance *I, const TreeParams &q)
{
  I->rt_up_n = q.n_up;
  for (int k = 0; k < q.n_up; ++k)
  {
    I->rt_up_idx[k] = q.up_idx[k];
    memcpy(I->rt_up_val[k], q.up_val[k], sizeof(double) * 64);
  }
}

// Launch the queued operations (and optionally the fused edge evaluation) as one traversal kernel.
int flush_impl(Instance *I, const EdgeEval *ee)
{
  const unsigned long long hp0 = hp_now();
  const int n_queued = (int)I->pending.size(); // (what the caller asked for: the unit of phyhip_profile_read's update count)
  // virtual buffers: what the queue reads of them is (re)computed in front of its reader; a long list of the two pipelined
  // kernels with two-deep register forwarding leaves its own tip x tip results virtual (rewrite_pending)
  rewrite_pending(I, ee, (I->soa || I->perm) && !I->generic_nt && I->prefetch_dist == 2 && !I->class_axis && !I->generic_loop &&
                             !I->ablate && !I->no_loads);
  keep_real_clear(I);
  const int n_ops = (int)I->pending.size();
  int rc = 0;
  if (n_ops > 0 || ee || !I->pm_idx.empty() || !I->up_idx.empty()) I->stream_dirty = true;
  if (ee) I->fenced_eval = false;
  // a short list of device-built matrices is folded into the lane-per-pattern nucleotide kernel's prologue when the grid
  // is small (measured: 16.7 vs 17.8 us per scalar-returning call on a 382-pattern search prefix; at 100 000 patterns
  // the redundant per-workgroup rebuild costs more than the launch it saves: 45.1 vs 42.5 us per SPR candidate)
  static const int fold_grid_max = diag_env("PHYHIP_FOLD_GRID") ? atoi(diag_env("PHYHIP_FOLD_GRID")) : 512;
  // large grids: an evaluation the large-grid resident workgroups can take (phyhip_big.hpp) carries its matrices in the command
  // (host-computed matrices -- the bit-exact route -- stay with the one-wave launch that takes them in its arguments: the
  // large-grid kernel launched with them in ITS arguments was measured, 30-31 against 28.4 us per candidate at 500 x 100 000)
  // Host-computed matrices that can travel with the evaluation (kernel arguments of a launch: TreeParams::n_up; a resident
  // command: ResidentCtl::up_area)
  const bool up_ride = I->soa && I->arg_uploads && !I->up_idx.empty() && (int)I->up_idx.size() <= kArgUp && I->pm_idx.empty() &&
                       (n_ops > 0 || ee) && I->C <= 4 && !I->class_axis && !(I->ablate & 8) && I->n_up_shadow == 0;
  // ... in a resident command: where the host stores into device memory (a pushed record)
  auto up_resident = [&](const Resident &R) { return up_ride && I->push_cmds != 0 && (R.cmd ? R.pushed && R.up_area != nullptr : true); };
  const bool big_form0 = ee && (ee->eigen || (ee->to_host && !ee->dev_out)) && n_ops <= 2 && I->args_recs && I->fold_pmats &&
                         (int)I->pm_idx.size() <= 4 && !I->rt_skip; // (an evaluation that kernel can take)
  const bool big_form = big_form0 && I->up_idx.empty();
  const bool big_fit = big_form0 && (I->up_idx.empty() || up_resident(I->rb)) && big_eligible(I) && !I->prof;
  const bool big_try = big_fit && big_ready(I);
  // ... and they are there (or launched now): decided before the records are built -- a resident command of two operations runs
  // them one after the other per tile, without register forwarding between them (phyhip_big.hpp)
  bool big_take = false;
  if (big_try)
  {
    const int brc = big_ensure(I);
    if (brc < 0) return brc;
    big_take = brc == 0 && (I->up_idx.empty() || up_resident(I->rb)); // (a record that turned out to live in host memory: launch)
  }
  // ... or they are not, and the same kernel is LAUNCHED for this one evaluation (BigArgs::n_one_shot) instead of pmat_kernel +
  // a traversal of one-wave workgroups + a record per workgroup: where the final sum can run through one partial sum per
  // workgroup (256 workgroups, more tiles than the host adds itself)
  const bool one_shot = big_form && !big_take && I->big_oneshot && big_shape(I) && I->spin_wait && I->dev >= 0 && I->grid_nt2 > I->big_device_sum &&
                        big_sum_by_group(I, I->grid_nt2);
  const bool big_cmd = big_take || one_shot; // (the records below are built for that kernel: no forwarding between two operations)
  if (kDiag && ee && getenv("PHYHIP_RESIDENT_DEBUG") && big_shape(I))
    fprintf(stderr, "big: fit %d try %d | eligible %d ops %d pm %zu up %zu prof %d skip %d | dirty_prev %d touched %d clean_after %llu stamp %llu streak %d launched %d owner %p me %p\n",
            (int)big_fit, (int)big_try, (int)big_eligible(I), n_ops, I->pm_idx.size(), I->up_idx.size(), (int)I->prof, (int)I->rt_skip, (int)I->dirty_prev,
            (int)I->touched_call, I->clean_after, *reinterpret_cast<volatile unsigned long long *>(I->h_result + 3), I->big_streak, (int)I->rb.launched,
            (void *)g_big_owner[I->dev < 64 ? I->dev : 0].load(), (void *)I);
  // small 20-state alignments: an evaluation the resident workgroups of traverse_aa_kernel<..., RES> can take -- decided here, before
  // the matrix queue is dealt with, because they rebuild the queued matrices themselves (a launched 20-state kernel cannot: without
  // them the rebuild is pmat20_kernel's launch in front)
  bool aa_take = false;
  if (resident_aa_eligible(I) && ee && !ee->eigen && ee->to_host && !ee->dev_out && n_ops <= 2 && I->args_recs && I->fold_pmats &&
      (int)I->pm_idx.size() <= 4 && I->up_idx.empty() && I->n_pm_shadow == 0 && !I->prof && !I->rt_skip)
  {
    // (a rebuilt matrix the command does not read goes through the ring's spare item: two tables)
    int unread = 0;
    for (int m : I->pm_idx)
    {
      bool read = m == ee->pm;
      for (const DevOp &o : I->pending) read = read || o.pm1 == m || o.pm2 == m;
      unread += !read;
    }
    bool clean = unread <= 2 && !I->dirty_prev && !I->touched_call;
    if (clean && I->clean_after)
    { // the report of the last Update_Eigen_Lr (bounded wait, else the ordinary launch)
      volatile unsigned long long *stamp = reinterpret_cast<volatile unsigned long long *>(I->h_result + 3);
      struct timespec t0;
      clock_gettime(CLOCK_MONOTONIC, &t0);
      for (long it = 1; *stamp < I->clean_after && clean; ++it)
      {
        __builtin_ia32_pause();
        if ((it & 255) == 0 && ns_since(t0) > 200000.0) clean = false;
      }
      if (clean) { __atomic_thread_fence(__ATOMIC_ACQUIRE); I->clean_after = 0; ++I->clean_epoch; }
    }
    if (!clean) ++I->rt.n_busy;
    aa_take = clean;
  }
  const bool fold_pm = (aa_take && !I->pm_idx.empty()) ||
                       (I->soa && I->fold_pmats && (I->grid_nt2 <= fold_grid_max || big_try || one_shot) && !I->pm_idx.empty() && (int)I->pm_idx.size() <= 8 &&
                        I->up_idx.empty() && (n_ops > 0 || ee) && I->C <= 4 && !I->class_axis && !(I->ablate & 8) && I->n_pm_shadow == 0);
  // a short list of HOST-computed matrices rides in the arguments of the lane-per-pattern nucleotide kernel at every grid size
  // (TreeParams::n_up): no upload kernel in front of the traversal
  const bool arg_up = up_ride;
  if (!fold_pm && !arg_up && (!I->pm_idx.empty() || !I->up_idx.empty()) && (rc = flush_pmats(I))) return rc;
  if (n_ops == 0 && !ee) return 0;
  rc = upload_masks(I);
  if (rc) return rc;

  TreeParams q = base_params(I);
  RO         ro = base_ro(I, nullptr);
  bool       fused_sum = false;
  if (arg_up)
  {
    q.n_up = (int)I->up_idx.size();
    for (int k = 0; k < q.n_up; ++k)
    {
      q.up_idx[k] = I->up_idx[k];
      memcpy(q.up_val[k], I->up_src[k], sizeof(double) * 16 * I->C); // (pinned staging memory: an ordinary host read)
    }
    q.pmats_rw = I->d_pmats;
    for (int m : I->up_idx) I->up_slot[m] = -1;
    I->up_idx.clear();
    I->up_src.clear();
    I->up_shadow.clear();
  }
  if (fold_pm)
  {
    q.n_fresh = (int)I->pm_idx.size();
    for (int k = 0; k < q.n_fresh; ++k) { q.fresh_idx[k] = I->pm_idx[k]; q.fresh_len[k] = I->pm_len[k]; }
    // (4 states, one eigen system, <= 4 categories: see fold_pm) the eigen system rides in the arguments as well
    if (I->S == 4)
    {
      memcpy(q.m_evec, I->h_evec.data(), 16 * sizeof(double)); memcpy(q.m_ivec, I->h_ivec.data(), 16 * sizeof(double));
      memcpy(q.m_eval, I->h_eval.data(), 4 * sizeof(double));
      for (int c = 0; c < 4; ++c) q.m_rates[c] = c < I->C ? I->h_rates[c] : 0.0;
    }
    q.br_len_mult = I->br_len_mult; q.l_min = I->l_min; q.l_max = I->l_max; q.pmats_rw = I->d_pmats;
    // (the matrix queue is cleared only after the launch that rebuilds it has been issued, see below)
  }
  const bool fat = ((I->S == 4) && !I->generic_nt) || I->perm;
  const bool has_inl = !I->pending_inl.empty(); // (some operation of this list computes a virtual child inside its own step)
  // 20-state kernel, such a list: an item of its LDS ring is two matrix tables or -- with an in-step child -- four.  The host
  // hands out the table slots (8 tables; an item never wraps) and says, per item, how many items every consumer must have
  // released before the loader may overwrite them (phyhip_aa.hpp)
  std::vector<int> aa_slot, aa_need;
  if (I->perm && has_inl)
  {
    const int n_rec = n_ops + (n_ops & 1), n_items = n_rec + (ee ? 1 : 0);
    aa_slot.resize(n_items); aa_need.resize(n_items);
    int cur = 0, owner[2 * kAaRing];
    for (int &x : owner) x = -1;
    for (int k = 0; k < n_items; ++k)
    {
      const int nt = (k < n_rec && (I->pending[std::min(k, n_ops - 1)].pad & (kOpInl1 | kOpInl2))) ? 4 : 2;
      if (cur + nt > 2 * kAaRing) cur = 0;
      int need = 0;
      for (int t = cur; t < cur + nt; ++t) { need = std::max(need, owner[t] + 1); owner[t] = k; }
      aa_slot[k] = cur; aa_need[k] = need;
      cur = (cur + nt) % (2 * kAaRing);
    }
    if (ee) { q.aa_e_slot = aa_slot[n_rec]; q.aa_e_need = aa_need[n_rec]; }
  }
  const IssueRec *d_irec = nullptr;
  const ExecRec  *d_xrec = nullptr;
  q.last_dest = -1;
  const int kind = fat ? I->prefetch_dist : 0;
  int       hit  = -1, new_slot = -1, host_sum_n = 0;
  if (n_ops > 0)
  { // an operation list identical to one still sitting in a device slot (repeated Lk(NULL) on one topology) is
    // neither rebuilt nor re-uploaded
    for (int sl = 0; sl < I->ops_slots && hit < 0; ++sl)
      if (I->slot_kind[sl] == kind && I->slot_ops[sl].size() == (size_t)n_ops &&
          memcmp(I->slot_ops[sl].data(), I->pending.data(), sizeof(DevOp) * n_ops) == 0 && I->slot_inl[sl] == I->pending_inl)
        hit = sl;
  }
  if (n_ops > 0 && hit >= 0)
  {
    char *dst = I->d_ops + (size_t)hit * I->ops_slot_bytes;
    if (!fat) ro.ops = reinterpret_cast<const DevOp *>(dst);
    else
    {
      d_irec = reinterpret_cast<const IssueRec *>(dst);
      d_xrec = reinterpret_cast<const ExecRec *>(dst + sizeof(IssueRec) * (n_ops + (n_ops & 1)));
      q.last_dest = I->pending[n_ops - 1].dest;
    }
    q.n_ops = fat ? n_ops + (n_ops & 1) : n_ops;
  }
  else if (n_ops > 0)
  {
    char *dst = I->d_ops + (size_t)I->ops_slot * I->ops_slot_bytes;
    // one or two operations of the lane-per-pattern nucleotide kernel travel in the kernel arguments (phyhip_nt2.hpp):
    // no staging, no copy command, and the device slots keep the long lists they cache
    const bool in_args = fat && (I->soa || I->perm) && I->args_recs && n_ops <= 2;
    IssueRec   arg_ir[2];
    ExecRec    arg_xr[2];
    if (!in_args)
    {
      new_slot = I->ops_slot;
      I->slot_kind[new_slot] = -1; // the slot's old content is gone; it holds the new list only once the copy was issued
    }
    void *st = nullptr;
    if (!fat)
    {
      if (sizeof(DevOp) * (size_t)n_ops > I->ops_slot_bytes)
        return fail(PHYHIP_ERROR_OUT_OF_RANGE, "a launch of %d operations does not fit a device slot of %zu bytes", n_ops, I->ops_slot_bytes);
      rc = I->ring.alloc(sizeof(DevOp) * n_ops, I->stream, &st);
      if (rc) return rc;
      memcpy(st, I->pending.data(), sizeof(DevOp) * n_ops);
      HIPCHK(hipMemcpyAsync(dst, st, sizeof(DevOp) * n_ops, hipMemcpyHostToDevice, I->stream));
      ro.ops = reinterpret_cast<const DevOp *>(dst);
    }
    else
    {
      // one record pair per operation, all address arithmetic done here once.  The kernel alternates two
      // register sets, so an odd list is padded with a re-execution of its last operation (idempotent: same
      // inputs, same output, same address) whose forwarding flags are computed for its own position.
      const int    n_rec = n_ops + (n_ops & 1);
      const size_t ib = sizeof(IssueRec) * n_rec, xb = sizeof(ExecRec) * n_rec;
      if (!in_args && ib + xb > I->ops_slot_bytes)
        return fail(PHYHIP_ERROR_OUT_OF_RANGE, "a launch of %d operation records does not fit a device slot of %zu bytes", n_rec, I->ops_slot_bytes);
      if (!in_args)
      {
        rc = I->ring.alloc(ib + xb, I->stream, &st);
        if (rc) return rc;
      }
      IssueRec *ir = in_args ? arg_ir : reinterpret_cast<IssueRec *>(st);
      ExecRec  *xr = in_args ? arg_xr : reinterpret_cast<ExecRec *>((char *)st + ib);
      const size_t   bufbytes = buf_elems(I) * sizeof(double);
      // spare word of the data descriptors: byte offset of the child's matrix (natural table, or the MFMA
      // A-fragment table for the 20-state kernel)
      const unsigned matbytes = I->perm ? (unsigned)(kAaMat * sizeof(double))
                                        : (unsigned)((size_t)I->C * I->S * I->S * sizeof(double));
      auto desc = [](const void *base, size_t bytes, unsigned x) {
        Desc d;
        d.base = (unsigned long long)(uintptr_t)base; d.bytes = (unsigned)bytes; d.x = x;
        return d;
      };
      auto at = [&](int k) -> const DevOp & { return I->pending[std::min(k, n_ops - 1)]; };
      for (int k = 0; k < n_rec; ++k)
      {
        const DevOp &o  = at(k);
        const int    e1 = (k >= 1 && !big_cmd) ? at(k - 1).dest : -1;
        const int    e2 = (k >= 2 && I->prefetch_dist == 2) ? at(k - 2).dest : -1;
        unsigned     fl = 0;
        const InlineDef *inl = (o.pad & (kOpInl1 | kOpInl2)) ? &I->pending_inl[std::min(k, n_ops - 1)] : nullptr;
        auto child = [&](int c, unsigned tipbit, unsigned f1bit, unsigned f2bit, Desc &data, Desc &scale, Desc &tip,
                         unsigned pmoff, bool in_step) {
          if (in_step)
          { // a virtual tip x tip result computed inside this operation's step (kOpCh1 / kOpCh2, phyhip_kernels.hpp): no load of
            // its own -- its two matrices' offsets, its first tip's row in the auxiliary slot, its second tip's row in the tip slot
            fl |= (tipbit == kOpTip1) ? kOpCh1 : kOpCh2;
            const unsigned long long ab = (unsigned long long)((unsigned)inl->pmA * matbytes) | ((unsigned long long)((unsigned)inl->pmB * matbytes) << 32);
            if (I->perm)
            { // (20 states: the operation's two tip slots carry the plan of its ring item and the second tip's mask row -- below)
              data  = desc(nullptr, 0, pmoff);
              scale = desc(I->d_tipmasks + (size_t)inl->a * I->Ppad, (size_t)I->Ppad * 4, 1);
              return;
            }
            data  = desc(reinterpret_cast<const void *>((uintptr_t)ab), 0, pmoff);
            scale = desc(I->d_tipcodes + (size_t)inl->a * I->Ppad, (size_t)I->Ppad, 1);
            tip   = desc(I->d_tipcodes + (size_t)inl->b * I->Ppad, (size_t)I->Ppad, 1);
            return;
          }
          const bool t = c < I->tips;
          const bool f1 = !t && c == e1, f2 = !t && !f1 && c == e2;
          const bool ld = !t && !f1 && !f2 && !I->no_loads;
          if (t) fl |= tipbit;
          if (f1) fl |= f1bit;
          if (f2) fl |= f2bit;
          const size_t b = ld ? (size_t)(c - I->tips) : 0;
          data  = desc(I->d_partials + b * buf_elems(I), ld ? bufbytes : 0, pmoff);
          scale = desc(I->d_scales + b * scale_elems(I), ld ? scale_elems(I) * 4 : 0, 0);
          tip   = desc(I->d_tipcodes + (size_t)(t ? c : 0) * I->Ppad, (t && !I->soa) ? (size_t)I->Ppad : 0, 0); // (lane-per-pattern kernel: only an in-step child uses this slot)
          // lane-per-pattern nucleotide kernel and the 20-state kernel: ONE auxiliary dword load per child -- the scale
          // descriptor of a tip child points at its tip row instead (spare word 1: the kernel then reads the aligned dword
          // holding the byte)
          if (I->soa && t) scale = desc(I->d_tipcodes + (size_t)c * I->Ppad, (size_t)I->Ppad, 1);
          if (I->perm && t) scale = desc(I->d_tipmasks + (size_t)c * I->Ppad, (size_t)I->Ppad * 4, 1); // (the mask itself)
        };
        child(o.c1, kOpTip1, kOpF11, kOpF12, ir[k].c1_data, ir[k].c1_scale, ir[k].c1_tip, (unsigned)o.pm1 * matbytes, (o.pad & kOpInl1) != 0);
        child(o.c2, kOpTip2, kOpF21, kOpF22, ir[k].c2_data, ir[k].c2_scale, ir[k].c2_tip, (unsigned)o.pm2 * matbytes, (o.pad & kOpInl2) != 0);
        const size_t b = (size_t)(o.dest - I->tips);
        const bool   st_on = !(o.pad & kOpNoStore); // (a result that stays virtual: stores through descriptors of size 0 are dropped)
        xr[k].dst_data  = desc(I->d_partials + b * buf_elems(I), st_on ? bufbytes : 0, fl);
        xr[k].dst_scale = desc(I->d_scales + b * scale_elems(I), st_on ? scale_elems(I) * 4 : 0, 0);
        if (I->perm && has_inl)
        { // the ring item of this operation: first table slot (consumers: dst_scale.x; loader: c2_tip.x), what must be released
          // before it is written (c1_tip.bytes), and -- with an in-step child -- its two tables' offsets and its second tip's masks
          const unsigned long long ab = inl ? ((unsigned long long)((unsigned)inl->pmA * matbytes) | ((unsigned long long)((unsigned)inl->pmB * matbytes) << 32)) : 0ull;
          ir[k].c1_tip = desc(reinterpret_cast<const void *>((uintptr_t)ab), (size_t)aa_need[k], inl ? ((o.pad & kOpInl1) ? 1u : 2u) : 0u);
          ir[k].c2_tip = desc(I->d_tipmasks + (size_t)(inl ? inl->b : 0) * I->Ppad, inl ? (size_t)I->Ppad * 4 : 0, (unsigned)aa_slot[k]);
          xr[k].dst_scale.x = (unsigned)aa_slot[k];
        }
      }
      // (reading short lists straight from the pinned staging memory instead was measured: no gain)
      if (in_args)
      {
        q.recs_in_args = 1; q.n_real_ops = n_ops;
        q.arg_ir[0] = ir[0]; q.arg_ir[1] = ir[1];
        q.arg_xr[0] = xr[0]; q.arg_xr[1] = xr[1];
      }
      else
      {
        HIPCHK(hipMemcpyAsync(dst, st, ib + xb, hipMemcpyHostToDevice, I->stream));
        d_irec = reinterpret_cast<const IssueRec *>(dst);
        d_xrec = reinterpret_cast<const ExecRec *>(dst + ib);
      }
      q.last_dest = I->pending[n_ops - 1].dest;
    }
    q.n_ops = fat ? n_ops + (n_ops & 1) : n_ops;
    if (new_slot >= 0)
    {
      I->slot_ops[new_slot]  = I->pending;
      I->slot_inl[new_slot]  = I->pending_inl;
      I->slot_kind[new_slot] = kind;
      I->ops_slot = (new_slot + 1) % I->ops_slots;
    }
  }
  // small nucleotide alignments: the resident short-launch evaluator (resident_nt2_kernel) may take the call
  const bool rt_grid = resident_short_eligible(I);
  // (a list-form launch of an instance with two wave shapes: traverse_nt2_mixed_kernel's grid, one record per workgroup)
  const bool mixed = I->mix_n4 > 0 && n_ops > 0 && !q.recs_in_args && I->prefetch_dist == 2 && !I->ablate;
  const int  soa_grid = mixed ? I->mix_n2 + I->mix_n4 : I->grid_nt2;
  if (ee && ee->eigen)
  { // Update_Eigen_Lr fused behind the queued partial update(s): no sums, the products go to d_dot
    q.edge_eval = 2; q.e_parent = ee->parent; q.e_child = ee->child; q.e_pm = 0; q.dot_out = I->d_dot;
    memcpy(q.m_evec, I->h_evec.data(), 16 * sizeof(double)); memcpy(q.m_ivec, I->h_ivec.data(), 16 * sizeof(double));
    if (n_ops == 0 && I->args_recs) { q.recs_in_args = 1; q.n_real_ops = 0; }
    if (q.recs_in_args)
    {
      auto untouched = [&](int idx) {
        if (idx < I->tips) return false;
        for (const DevOp &o : I->pending)
          if (o.dest == idx) return false;
        return true;
      };
      q.e_prefetch = (untouched(ee->parent) ? 1 : 0) | (untouched(ee->child) ? 2 : 0);
    }
    // completion as an evaluation's: every workgroup fences its stores and posts an (empty) record the caller waits for -- the
    // stream is clean when phyhip_update_eigen_lr returns, and the resident workgroups can take the call (the only
    // instances that come here: phyhip_update_eigen_lr)
    q.host_blocks = I->h_blocks; q.host_tag = ++I->seq; q.warn = I->h_warn;
    host_sum_n    = soa_grid;
  }
  else if (ee)
  {
    I->warn_current = false;
    q.edge_eval = 1; q.e_parent = ee->parent; q.e_child = ee->child; q.e_pm = ee->pm;
    if (n_ops == 0 && fat && (I->soa || aa_take) && I->args_recs) { q.recs_in_args = 1; q.n_real_ops = 0; } // (evaluation-only short launch)
    if (q.recs_in_args)
    { // short launch: the kernel fetches the sides of the evaluation edge that no queued operation writes up front
      auto untouched = [&](int idx) {
        if (idx < I->tips) return false;
        for (const DevOp &o : I->pending)
          if (o.dest == idx) return false;
        return true;
      };
      q.e_prefetch = (untouched(ee->parent) ? 1 : 0) | (untouched(ee->child) ? 2 : 0);
    }
    const int nblk = I->soa ? soa_grid : (I->perm ? I->grid_aa : (fat ? I->grid_nt : I->grid));
    // (fusing on large grids was measured for one-operation launches too: 61.6 vs 41.9 us per SPR candidate at cfg5)
    fused_sum = !I->class_axis && fuse_reduce(I, nblk) && !(I->host_sum && ee->to_host && !ee->dev_out);
    if (fused_sum)
    { // the traversal kernel's last workgroup finishes the sum and reports to the host
      q.tickets = I->d_tickets; q.result = ee->dev_out ? ee->dev_out : I->d_result;
      q.result_host = ee->to_host ? I->h_result : nullptr; q.warn_host = I->h_warn;
      q.seq = ee->to_host ? ++I->seq : 0ull;
      q.warn_out = ee->warn_out;
    }
    if (!fused_sum && ee->to_host && !ee->dev_out && I->host_sum && !I->class_axis)
    { // the workgroups post their sums to the host, which adds them (wait_result) -- at every grid size: on small grids
      // this replaces the ticket draw of the fused sum (block sum written through, atomic, fence, re-read: ~3 us of
      // dependent memory round trips inside a ~10 us kernel), on large ones the second launch
      q.host_blocks = I->h_blocks; q.host_tag = ++I->seq;
      q.warn        = I->h_warn;   // raised straight in host-mapped memory
      *I->h_warn    = 0;
      host_sum_n    = nblk;
      if (I->eig_api_no && I->api_no == I->eig_api_no + 1 && (I->soa || I->perm))
      { // (the kernels that honour it; eig_api_no is only set for small alignments with the resident evaluator enabled)
        q.fence_post   = 1;
        I->fenced_eval = true;
      }
    }
    if (I->want_site_outputs) { q.site_lnl = I->d_site_lnl; q.site_lk = I->d_site_lk; q.site_cat = I->d_site_cat; }
  }
  // ---- small nucleotide alignments: the resident short-launch evaluator (resident_nt2_kernel) ------------------------
  static const bool rtdbg = kDiag && getenv("PHYHIP_RESIDENT_DEBUG") != nullptr; // (diag build: why an evaluation was launched)
  if (rtdbg && ee)
    fprintf(stderr, "rt: grid_ok %d (res %d spin %d hs %d soa %d co %d cls %d g2 %d abl %d grp %d) hsn %d args %d fresh %d site %d prof %d skip %d dirty_prev %d touched %d\n",
            (int)rt_grid, (int)I->resident, (int)I->spin_wait, (int)I->host_sum, (int)I->soa, I->co != nullptr, (int)I->class_axis, I->grid_nt2,
            I->ablate, I->nt_groups, host_sum_n, q.recs_in_args, q.n_fresh, (int)I->want_site_outputs, (int)I->prof, (int)I->rt_skip,
            (int)I->dirty_prev, (int)I->touched_call);
  if ((rt_grid || resident_aa_eligible(I)) && host_sum_n > 0)
  { // every evaluation of such an instance completes its stores before it posts: the stream is clean once the scalar is back
    q.fence_post   = 1;
    I->fenced_eval = true;
  }
  // ---- small 20-state alignments: the resident form of traverse_aa_kernel (phyhip_aa.hpp) -------------------------------------
  if (aa_take)
  {
    if (!(host_sum_n > 0 && q.recs_in_args && q.n_fresh <= 4 && q.n_up == 0))
      return fail(PHYHIP_ERROR_GENERAL, "20-state resident evaluator: an evaluation it cannot take (%d records, %d matrices)", host_sum_n, q.n_fresh);
    Resident  &R = I->rt;
    // what the workgroups are launched with: everything of the launch form's arguments that does not change per call
    TreeParams sq = base_params(I);
    sq.host_blocks = I->h_blocks; sq.warn = I->h_warn; sq.fence_post = 1; sq.recs_in_args = 1; sq.edge_eval = 1;
    sq.br_len_mult = I->br_len_mult; sq.l_min = I->l_min; sq.l_max = I->l_max; sq.pmats_rw = I->d_pmats;
    if (I->want_site_outputs) { sq.site_lnl = I->d_site_lnl; sq.site_lk = I->d_site_lk; sq.site_cat = I->d_site_cat; }
    // geometry: as few workgroups as hold every wave-tile with at most kAaMaxCons2 consumers each, the tiles spread evenly
    const int ntl = I->grid_aa /* (aa_nw == 1: one tile per workgroup of the launched form) */, wgs = (ntl + kAaMaxCons2 - 1) / kAaMaxCons2,
              nwr = (ntl + wgs - 1) / wgs;
    if (!R.launched || R.grid != wgs || memcmp(&I->rt_static, &sq, sizeof sq) != 0 || resident_gone(R))
    {
      if (R.launched && (R.grid != wgs || memcmp(&I->rt_static, &sq, sizeof sq) != 0)) resident_stop(R);
      AaResident  rs;
      hipStream_t st;
      if ((rc = resident_prepare(I, R, wgs, kResidentAaWords, R.seq, rs.ctl, &st))) return rc;
      rs.evec = I->d_evec; rs.ivec = I->d_ivec; rs.eval = I->d_eval; rs.rates = I->d_catr; rs.pmats_rw = I->d_pmats; rs.afrag_rw = I->d_afrag;
      rs.stamps = nullptr; rs.stamp_wg = 0;
      if (getenv("PHYHIP_RESIDENT_STATS"))
      { // (where workgroup 0's time goes per command: printed when the instance goes)
        if (!I->d_big_stamps)
        {
          HIPCHK(hipMalloc((void **)&I->d_big_stamps, sizeof(unsigned long long) * 32));
          HIPCHK(hipMemsetAsync(I->d_big_stamps, 0, sizeof(unsigned long long) * 16, I->stream));
          HIPCHK(hipStreamSynchronize(I->stream));
        }
        rs.stamps = I->d_big_stamps;
        rs.stamp_wg = std::min(wgs - 1, std::max(0, atoi(getenv("PHYHIP_RESIDENT_STATS")) - 1));
      }
      if (launch_resident_aa(I->C, wgs, nwr, st, sq, (const double *)I->d_afrag, I->nmat_all, (const uint32_t *)I->d_tipmasks, rs))
        return fail(PHYHIP_ERROR_GENERAL, "20-state resident evaluator: no kernel for %d categories", I->C);
      HIPCHK(hipGetLastError());
      memcpy(&I->rt_static, &sq, sizeof sq);
      resident_launched(R, wgs);
    }
    unsigned long long words[kResidentAaWords];
    memset(words, 0, sizeof words);
    const bool changed = I->clean_epoch != I->rt_epoch; // the stream ran something since the last command
    words[0] = q.host_tag;
    words[1] = (unsigned long long)q.n_real_ops | (changed ? 4ull : 0ull) | ((unsigned long long)q.n_fresh << 4) | ((unsigned long long)q.e_prefetch << 8);
    words[2] = (unsigned long long)(unsigned)q.e_parent | ((unsigned long long)(unsigned)q.e_child << 32);
    words[3] = (unsigned long long)(unsigned)q.e_pm | ((unsigned long long)(unsigned)q.last_dest << 32);
    for (int k = 0; k < q.n_fresh; ++k)
    {
      words[4 + k / 2] |= (unsigned long long)(unsigned)q.fresh_idx[k] << (32 * (k & 1));
      memcpy(&words[6 + k], &q.fresh_len[k], 8);
    }
    auto put = [&](int k, const Desc &d) { words[k] = d.base; words[k + 1] = (unsigned long long)d.bytes | ((unsigned long long)d.x << 32); };
    for (int o = 0; o < q.n_real_ops; ++o)
    {
      put(10 + o * 12, q.arg_ir[o].c1_data); put(12 + o * 12, q.arg_ir[o].c2_data);
      put(14 + o * 12, q.arg_ir[o].c1_scale); put(16 + o * 12, q.arg_ir[o].c2_scale);
      put(18 + o * 12, q.arg_xr[o].dst_data); put(20 + o * 12, q.arg_xr[o].dst_scale);
    }
    words[34] = I->model_epoch;
    // kept until the answer is in: an evaluation nobody answers is launched the ordinary way (flush_and_wait)
    I->rt_ops = I->pending; I->rt_pm_idx = I->pm_idx; I->rt_pm_len = I->pm_len;
    resident_send(I, R, words, kResidentAaWords);
    I->rt_epoch = I->clean_epoch;
    I->host_sum_n = host_sum_n; I->host_sum_ns = 1;
    if (fold_pm)
    {
      for (int m : I->pm_idx) I->pm_slot[m] = -1;
      I->pm_idx.clear();
      I->pm_len.clear();
      I->pm_shadow.clear();
    }
    I->pending.clear();
    std::fill(I->mat_in_queue.begin(), I->mat_in_queue.end(), 0);
    return 0;
  }
  if (rt_grid && host_sum_n > 0 && q.recs_in_args && q.n_fresh <= 4 && (q.n_up == 0 || up_resident(I->rt)) && !I->prof && !I->rt_skip)
  {
    bool clean = !I->dirty_prev && !I->touched_call;
    if (clean && I->clean_after)
    { // the report of the last Update_Eigen_Lr (bounded wait, else the ordinary launch)
      volatile unsigned long long *stamp = reinterpret_cast<volatile unsigned long long *>(I->h_result + 3);
      struct timespec t0;
      clock_gettime(CLOCK_MONOTONIC, &t0);
      for (long it = 1; *stamp < I->clean_after && clean; ++it)
      {
        __builtin_ia32_pause();
        if ((it & 255) == 0 && ns_since(t0) > 200000.0) clean = false;
      }
      if (clean) { __atomic_thread_fence(__ATOMIC_ACQUIRE); I->clean_after = 0; ++I->clean_epoch; }
    }
    if (!clean) ++I->rt.n_busy;
    else
    {
      Resident  &R = I->rt;
      // what the workgroups are launched with: everything of the launch form's arguments that does not change per call
      TreeParams sq = base_params(I);
      sq.host_blocks = I->h_blocks; sq.warn = I->h_warn; sq.fence_post = 1; sq.recs_in_args = 1; sq.edge_eval = 1;
      sq.br_len_mult = I->br_len_mult; sq.l_min = I->l_min; sq.l_max = I->l_max; sq.pmats_rw = I->d_pmats;
      sq.dot_out = I->d_dot;
      if (I->want_site_outputs) { sq.site_lnl = I->d_site_lnl; sq.site_lk = I->d_site_lk; sq.site_cat = I->d_site_cat; }
      if (!R.launched || R.grid != I->grid_nt2 || memcmp(&I->rt_static, &sq, sizeof sq) != 0 || resident_gone(R))
      {
        if (R.launched && (R.grid != I->grid_nt2 || memcmp(&I->rt_static, &sq, sizeof sq) != 0)) resident_stop(R);
        ResidentCtl r;
        hipStream_t st;
        if ((rc = resident_prepare(I, R, I->grid_nt2, kResidentNtWords, R.seq, r, &st))) return rc;
#define NT2RES(c_, g_)                                                                                                      \
  hipLaunchKernelGGL((resident_nt2_kernel<c_, g_>), dim3(I->grid_nt2), dim3(64), 0, st, sq, r, (const double *)I->d_pmats,   \
                     (const uint8_t *)I->d_tipcodes, (const double *)I->d_evec, (const double *)I->d_ivec,                    \
                     (const double *)I->d_eval, (const double *)I->d_catr);                                                   \
  break;
        switch (I->C * 8 + I->nt_groups)
        {
          case 1 * 8 + 1: NT2RES(1, 1)
          case 2 * 8 + 1: NT2RES(2, 1)
          case 2 * 8 + 2: NT2RES(2, 2)
          case 3 * 8 + 1: NT2RES(3, 1)
          case 4 * 8 + 1: NT2RES(4, 1)
          case 4 * 8 + 2: NT2RES(4, 2)
          default: return fail(PHYHIP_ERROR_GENERAL, "resident evaluator: no kernel for %d categories in %d groups", I->C, I->nt_groups);
        }
#undef NT2RES
        HIPCHK(hipGetLastError());
        memcpy(&I->rt_static, &sq, sizeof sq);
        resident_launched(R, I->grid_nt2);
      }
      // (host-computed matrices and a record that turned out to live in host memory: the launch below takes them in its arguments)
      if (q.n_up == 0 || resident_push_uploads(R, q.n_up, q.up_val, 16 * I->C))
      {
      unsigned long long words[kResidentNtWords];
      memset(words, 0, sizeof words);
      const bool changed = I->clean_epoch != I->rt_epoch; // the stream ran something since the last command
      words[0] = q.host_tag;
      words[1] = (unsigned long long)q.n_real_ops | (changed ? 4ull : 0ull) | ((unsigned long long)q.n_fresh << 4) |
                 ((unsigned long long)q.e_prefetch << 8) | (q.edge_eval == 2 ? 1ull << 10 : 0ull) | ((unsigned long long)q.n_up << 17);
      words[2] = (unsigned long long)(unsigned)q.e_parent | ((unsigned long long)(unsigned)q.e_child << 32);
      words[3] = (unsigned long long)(unsigned)q.e_pm | ((unsigned long long)(unsigned)q.last_dest << 32);
      for (int k = 0; k < q.n_fresh; ++k)
      {
        words[4 + k / 2] |= (unsigned long long)(unsigned)q.fresh_idx[k] << (32 * (k & 1));
        memcpy(&words[6 + k], &q.fresh_len[k], 8);
      }
      for (int k = 0; k < q.n_up; ++k) words[4 + k / 2] |= (unsigned long long)(unsigned)q.up_idx[k] << (32 * (k & 1)); // (never both kinds: up_ride)
      auto put = [&](int k, const Desc &d) { words[k] = d.base; words[k + 1] = (unsigned long long)d.bytes | ((unsigned long long)d.x << 32); };
      for (int o = 0; o < q.n_real_ops; ++o)
      {
        put(10 + o * 12, q.arg_ir[o].c1_data); put(12 + o * 12, q.arg_ir[o].c2_data);
        put(14 + o * 12, q.arg_ir[o].c1_scale); put(16 + o * 12, q.arg_ir[o].c2_scale);
        put(18 + o * 12, q.arg_xr[o].dst_data); put(20 + o * 12, q.arg_xr[o].dst_scale);
      }
      // kept until the answer is in: an evaluation nobody answers is launched the ordinary way (phyhip_calculate_edge_log_likelihoods)
      I->rt_ops = I->pending; I->rt_pm_idx = I->pm_idx; I->rt_pm_len = I->pm_len;
      keep_uploads(I, q);
      resident_send(I, R, words, kResidentNtWords); // (every sector the workgroups wait for carries the command's number)
      I->rt_epoch = I->clean_epoch;
      I->host_sum_n = host_sum_n; I->host_sum_ns = 1;
      if (fold_pm)
      {
        for (int m : I->pm_idx) I->pm_slot[m] = -1;
        I->pm_idx.clear();
        I->pm_len.clear();
        I->pm_shadow.clear();
      }
      I->pending.clear();
      std::fill(I->mat_in_queue.begin(), I->mat_in_queue.end(), 0);
      return 0;
      }
    }
  }
  // ---- large nucleotide alignments: the large-grid resident evaluator (resident_big_kernel) -------------------------------
  // (launches of such an instance do not fence their stores before they post -- with megabytes of results in the L2s a
  // write-back per wave costs more than the launch; whether the stream is idle again is found by querying it, big_clean)
  if (big_cmd)
  {
    if (!(host_sum_n > 0 && q.recs_in_args && q.n_fresh <= 4 && (q.n_up == 0 || (big_take && q.n_fresh == 0))))
      return fail(PHYHIP_ERROR_GENERAL, "large-grid resident evaluator: an evaluation it cannot take (%d records, %d matrices)", host_sum_n, q.n_fresh);
    if (q.n_up > 0 && !resident_push_uploads(I->rb, q.n_up, q.up_val, 16 * I->C))
      return fail(PHYHIP_ERROR_GENERAL, "large-grid resident evaluator: no upload area for %d host-computed matrices", q.n_up);
    {
      Resident &R = I->rb;
      unsigned long long words[kBigWords];
      memset(words, 0, sizeof words);
      const bool changed = big_take && I->clean_epoch != I->rt_epoch; // the stream ran something since the last command
      const bool dsum = host_sum_n > I->big_device_sum;
      words[0] = q.host_tag;
      words[1] = (unsigned long long)q.n_real_ops | (changed ? kBigChanged : 0ull) | ((unsigned long long)q.n_fresh << 4) |
                 ((unsigned long long)q.e_prefetch << 8) | (q.edge_eval == 2 ? kBigEigen : 0ull) | (dsum ? kBigDeviceSum : 0ull) |
                 (dsum && big_sum_by_group(I, host_sum_n) ? kBigGroupSum : 0ull) | ((unsigned long long)q.n_up << 17);
      words[2] = (unsigned long long)(unsigned)q.e_parent | ((unsigned long long)(unsigned)q.e_child << 32);
      words[3] = (unsigned long long)(unsigned)q.e_pm | ((unsigned long long)(unsigned)q.last_dest << 32);
      for (int k = 0; k < q.n_fresh; ++k)
      {
        words[4 + k / 2] |= (unsigned long long)(unsigned)q.fresh_idx[k] << (32 * (k & 1));
        memcpy(&words[6 + k], &q.fresh_len[k], 8);
      }
      for (int k = 0; k < q.n_up; ++k) words[4 + k / 2] |= (unsigned long long)(unsigned)q.up_idx[k] << (32 * (k & 1));
      auto put = [&](int k, const Desc &d) { words[k] = d.base; words[k + 1] = (unsigned long long)d.bytes | ((unsigned long long)d.x << 32); };
      for (int o = 0; o < q.n_real_ops; ++o)
      {
        put(10 + o * 12, q.arg_ir[o].c1_data); put(12 + o * 12, q.arg_ir[o].c2_data);
        put(14 + o * 12, q.arg_ir[o].c1_scale); put(16 + o * 12, q.arg_ir[o].c2_scale);
        put(18 + o * 12, q.arg_xr[o].dst_data); put(20 + o * 12, q.arg_xr[o].dst_scale);
      }
      if (one_shot)
      { // launched, on the instance's stream: behind the resident workgroups' exit, if there are any (not when they were about
        // to be asked: there are none then, and the streak that launches them at its second call goes on)
        if (!big_try) big_release(I);
        I->touched_call = true;
        hipEvent_t p0 = nullptr, p1 = nullptr;
        const bool timed = I->prof && !ee->eigen; // (Update_Eigen_Lr: the caller's own events are around this call)
        if (timed)
        {
          for (hipEvent_t *e : {&p0, &p1})
          {
            if (!I->prof_spare.empty()) { *e = I->prof_spare.back(); I->prof_spare.pop_back(); }
            else HIPCHK(hipEventCreate(e));
          }
          HIPCHK(hipEventRecord(p0, I->stream));
        }
        if ((rc = big_one_shot(I, words, kBigWords))) return rc;
        if (timed) snprintf(I->prof_kernel, sizeof I->prof_kernel, "resident_big_kernel<%d, %d> (launched for one evaluation)", I->C, I->nt_groups);
        if (timed)
        {
          HIPCHK(hipEventRecord(p1, I->stream));
          I->prof_pairs.emplace_back(p0, p1);
          I->prof_updates += (double)n_queued * (double)I->P;
        }
        I->host_sum_n = dsum ? 1 : host_sum_n; I->host_sum_ns = 1;
      }
      else
      { // kept until the answer is in: an evaluation nobody answers is launched the ordinary way (flush_and_wait)
        I->rt_ops = I->pending; I->rt_pm_idx = I->pm_idx; I->rt_pm_len = I->pm_len;
        keep_uploads(I, q);
        resident_send(I, R, words, kBigWords);
        I->rt_epoch = I->clean_epoch;
        I->host_sum_n = dsum ? 1 : host_sum_n; I->host_sum_ns = 1;
        I->fenced_eval = true; // (nothing went onto the stream: it is as idle as it was found)
      }
      if (fold_pm)
      {
        for (int m : I->pm_idx) I->pm_slot[m] = -1;
        I->pm_idx.clear();
        I->pm_len.clear();
        I->pm_shadow.clear();
      }
      I->pending.clear();
      std::fill(I->mat_in_queue.begin(), I->mat_in_queue.end(), 0);
      // (launched: say when the stream is idle again, so that the resident workgroups can take the next one)
      if (one_shot && big_fit && (rc = stamp_stream(I))) return rc;
      return 0;
    }
  }
  if (!big_try) big_release(I); // (what follows needs the wave slots the large-grid resident workgroups hold, if there are any)
  I->touched_call = true; // (everything below goes onto the stream)
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (I->prof)
  {
    for (hipEvent_t *e : {&e0, &e1})
    { // (phyhip_profile(1) left a supply: creating an event costs about a microsecond of the step that is being timed)
      if (!I->prof_spare.empty()) { *e = I->prof_spare.back(); I->prof_spare.pop_back(); }
      else HIPCHK(hipEventCreate(e));
    }
    HIPCHK(hipEventRecord(e0, I->stream));
  }
  const unsigned long long hp1 = hp_now();
  auto named = [&](const char *fmt, auto... a) { if (I->prof) snprintf(I->prof_kernel, sizeof I->prof_kernel, fmt, a...); };
  rc = dispatch_shape(I, [&](auto s, auto cp) {
    constexpr int S_ = decltype(s)::value, CP_ = decltype(cp)::value;
    if constexpr (S_ == 4 && CP_ <= 4)
    {
      if (I->soa)
      { // lane-per-pattern kernel, instantiated on the exact category count
#ifdef PHYHIP_DIAG
        if ((I->ablate & 8) && I->C == 4 && I->nt_groups <= 2)
        { // PHYHIP_ABLATE=8: cycle stamps of one wave, printed to stderr (diagnostics; costs a sync)
          unsigned long long *&d_dbg = I->d_dbg;
          if (!d_dbg) HIPCHK(hipMalloc((void **)&d_dbg, 64 * 8 * 8));
          if (I->nt_groups == 2)
            hipLaunchKernelGGL((traverse_nt2_kernel<4, 2, true>), dim3(I->grid_nt2), dim3(64), 0, I->stream, q, d_irec, d_xrec, ro.pmats, ro.tip_codes, d_dbg);
          else
            hipLaunchKernelGGL((traverse_nt2_kernel<4, 1, true>), dim3(I->grid_nt2), dim3(64), 0, I->stream, q, d_irec, d_xrec, ro.pmats, ro.tip_codes, d_dbg);
          static int printed = 0;
          if (printed++ == 5)
          {
            unsigned long long h[64 * 8];
            HIPCHK(hipMemcpyAsync(h, d_dbg, sizeof h, hipMemcpyDeviceToHost, I->stream));
            HIPCHK(hipStreamSynchronize(I->stream));
            for (int k = 0; k < 64 && k < q.n_ops; ++k)
            {
              fprintf(stderr, "step %2d:", k);
              for (int i = 1; i < 7; ++i) fprintf(stderr, " %6lld", (long long)(h[k * 8 + i] - h[k * 8 + i - 1]));
              if (k + 1 < 64) fprintf(stderr, "  | next %6lld", (long long)(h[(k + 1) * 8] - h[k * 8 + 6]));
              fprintf(stderr, "  | load issue %6lld of segment 4", (long long)(h[k * 8 + 7] - h[k * 8 + 3]));
              fprintf(stderr, "\n");
            }
          }
          return 0;
        }
#endif
#define NT2LAUNCH(c_, g_, a_)                                                                                               \
  hipLaunchKernelGGL((traverse_nt2_kernel<c_, g_, false, a_>), dim3(I->grid_nt2), dim3(64), 0, I->stream, q, d_irec, d_xrec,  \
                     ro.pmats, ro.tip_codes, (unsigned long long *)nullptr);
#define NT2CASE(c_, g_)                                                                                                     \
  if (mixed && c_ == 4 && g_ == 2)                                                                                          \
  { /* two wave shapes in one launch (phyhip_nt2.hpp): full rounds of two-lane waves + four-lane waves for the rest */      \
    named("traverse_nt2_mixed_kernel<4, %s>", has_inl ? "true" : "false");                                                  \
    if (has_inl)                                                                                                            \
      hipLaunchKernelGGL((traverse_nt2_mixed_kernel<4, true>), dim3(soa_grid), dim3(64), 0, I->stream, q, d_irec, d_xrec, ro.pmats,  \
                         ro.tip_codes, I->mix_n2);                                                                          \
    else                                                                                                                    \
      hipLaunchKernelGGL((traverse_nt2_mixed_kernel<4, false>), dim3(soa_grid), dim3(64), 0, I->stream, q, d_irec, d_xrec, ro.pmats, \
                         ro.tip_codes, I->mix_n2);                                                                          \
  }                                                                                                                         \
  else if (!q.recs_in_args && I->prefetch_dist == 1)                                                                        \
  {                                                                                                                         \
    named("traverse_nt2_kernel<%d, %d, false, 0, 1>", c_, g_);                                                              \
    hipLaunchKernelGGL((traverse_nt2_kernel<c_, g_, false, 0, 1>), dim3(I->grid_nt2), dim3(64), 0, I->stream, q, d_irec, d_xrec, \
                       ro.pmats, ro.tip_codes, (unsigned long long *)nullptr);                                              \
  }                                                                                                                         \
  else if (!q.recs_in_args && has_inl && g_ <= 2)                                                                           \
  { /* a list with in-step tip x tip children: the instantiation that stages four matrices per step */                      \
    named("traverse_nt2_kernel<%d, %d, false, 0, 2, true>", c_, (g_ <= 2 ? g_ : 1));                                        \
    hipLaunchKernelGGL((traverse_nt2_kernel<c_, (g_ <= 2 ? g_ : 1), false, 0, 2, true>), dim3(I->grid_nt2), dim3(64), 0, I->stream, q, d_irec, d_xrec, \
                       ro.pmats, ro.tip_codes, (unsigned long long *)nullptr);                                              \
  }                                                                                                                         \
  else if (!q.recs_in_args) { named("traverse_nt2_kernel<%d, %d, false, 0>", c_, g_); NT2LAUNCH(c_, g_, 0) }                \
  else if (q.n_real_ops == 1) { named("traverse_nt2_kernel<%d, %d, false, 1>", c_, g_); NT2LAUNCH(c_, g_, 1) }              \
  else if (q.n_real_ops == 2) { named("traverse_nt2_kernel<%d, %d, false, 2>", c_, g_); NT2LAUNCH(c_, g_, 2) }              \
  else { named("traverse_nt2_kernel<%d, %d, false, 3>", c_, g_); NT2LAUNCH(c_, g_, 3) }                                     \
  return 0;
        switch (I->C * 8 + I->nt_groups)
        {
          case 1 * 8 + 1: NT2CASE(1, 1)
          case 2 * 8 + 1: NT2CASE(2, 1)
          case 2 * 8 + 2: NT2CASE(2, 2)
          case 3 * 8 + 1: NT2CASE(3, 1)
          case 4 * 8 + 1: NT2CASE(4, 1)
          case 4 * 8 + 2: NT2CASE(4, 2)
          case 4 * 8 + 4: NT2CASE(4, 4)
          default: break;
        }
#undef NT2CASE
#undef NT2LAUNCH
      }
    }
    if constexpr (S_ == 4 && CP_ <= 8 && (CP_ == 8 || kDiag))
    { // first-generation lane = (pattern, category) pipeline: the production kernel for 5..8 categories
      if (!I->generic_nt)
      {
        named("traverse_nt_kernel<%d, 0, %d>", CP_, I->prefetch_dist == 1 ? 1 : 2);
        if (I->prefetch_dist == 1)
        {
          hipLaunchKernelGGL((traverse_nt_kernel<CP_, 0, 1>), dim3(I->grid_nt), dim3(I->block_nt), 0, I->stream, q, d_irec, d_xrec, ro.pmats,
                             ro.tip_codes);
          return 0;
        }
#ifdef PHYHIP_DIAG
        if constexpr (CP_ == 4)
        {
          switch (I->ablate)
          {
#define ABLCASE(a) case a: hipLaunchKernelGGL((traverse_nt_kernel<CP_, a>), dim3(I->grid_nt), dim3(I->block_nt), 0, I->stream, q, d_irec, d_xrec, ro.pmats, ro.tip_codes); return 0;
            ABLCASE(1) ABLCASE(2) ABLCASE(3) ABLCASE(6) ABLCASE(7)
#undef ABLCASE
            default: break;
          }
        }
#endif
        hipLaunchKernelGGL((traverse_nt_kernel<CP_>), dim3(I->grid_nt), dim3(I->block_nt), 0, I->stream, q, d_irec, d_xrec, ro.pmats,
                           ro.tip_codes);
        return 0;
      }
    }
    if constexpr (S_ == 20 && CP_ <= 4)
    {
      if (I->perm)
      {
        const dim3 blk(64 * (I->aa_nw + 1));
#define AACASE(c_)                                                                                                          \
  case c_:                                                                                                                  \
    named("traverse_aa_kernel<%d, false, 0, %s, %s, %d, %s>", c_, q.recs_in_args ? "true" : "false", (!q.recs_in_args && has_inl) ? "true" : "false", \
          q.recs_in_args ? 1 : I->aa_nt, (!q.recs_in_args && I->aa_nt == 1 && I->aa_d2) ? "true" : "false");                \
    if constexpr (kDiag) /* (measured variants of the list form, kept for A/B: profiles/r06_aa_kernel.md) */               \
    if (!q.recs_in_args && I->aa_nt == 1 && I->aa_d2)                                                                       \
    { /* few waves per SIMD: loads two operations ahead */                                                                  \
      if (has_inl)                                                                                                          \
        hipLaunchKernelGGL((traverse_aa_kernel<c_, false, 0, false, true, 1, true>), dim3(I->grid_aa), blk, 0, I->stream, q, d_irec, d_xrec, \
                           (const double *)I->d_afrag, I->nmat_all, (const uint32_t *)I->d_tipmasks, (unsigned long long *)nullptr); \
      else                                                                                                                  \
        hipLaunchKernelGGL((traverse_aa_kernel<c_, false, 0, false, false, 1, true>), dim3(I->grid_aa), blk, 0, I->stream, q, d_irec, d_xrec, \
                           (const double *)I->d_afrag, I->nmat_all, (const uint32_t *)I->d_tipmasks, (unsigned long long *)nullptr); \
      return 0;                                                                                                             \
    }                                                                                                                       \
    if constexpr (kDiag)                                                                                                    \
    if (!q.recs_in_args && I->aa_nt == 2)                                                                                   \
    { /* two wave-tiles per consumer wave */                                                                                \
      const dim3 blk2(64 * ((I->aa_nw + 1) / 2 + 1));                                                                        \
      q.aa_tpw = I->aa_nw;                                                                                                  \
      if (has_inl)                                                                                                          \
        hipLaunchKernelGGL((traverse_aa_kernel<c_, false, 0, false, true, 2>), dim3(I->grid_aa), blk2, 0, I->stream, q, d_irec, d_xrec, \
                           (const double *)I->d_afrag, I->nmat_all, (const uint32_t *)I->d_tipmasks, (unsigned long long *)nullptr); \
      else                                                                                                                  \
        hipLaunchKernelGGL((traverse_aa_kernel<c_, false, 0, false, false, 2>), dim3(I->grid_aa), blk2, 0, I->stream, q, d_irec, d_xrec, \
                           (const double *)I->d_afrag, I->nmat_all, (const uint32_t *)I->d_tipmasks, (unsigned long long *)nullptr); \
      return 0;                                                                                                             \
    }                                                                                                                       \
    if (q.recs_in_args)                                                                                                     \
      hipLaunchKernelGGL((traverse_aa_kernel<c_, false, 0, true>), dim3(I->grid_aa), blk, 0, I->stream, q, d_irec, d_xrec,    \
                         (const double *)I->d_afrag, I->nmat_all, (const uint32_t *)I->d_tipmasks, (unsigned long long *)nullptr); \
    else if (has_inl)                                                                                                       \
      hipLaunchKernelGGL((traverse_aa_kernel<c_, false, 0, false, true>), dim3(I->grid_aa), blk, 0, I->stream, q, d_irec, d_xrec, \
                         (const double *)I->d_afrag, I->nmat_all, (const uint32_t *)I->d_tipmasks, (unsigned long long *)nullptr); \
    else                                                                                                                    \
    hipLaunchKernelGGL((traverse_aa_kernel<c_>), dim3(I->grid_aa), blk, 0, I->stream, q, d_irec, d_xrec,                      \
                       (const double *)I->d_afrag, I->nmat_all, (const uint32_t *)I->d_tipmasks, (unsigned long long *)nullptr); \
    return 0;
#ifdef PHYHIP_DIAG
        if (I->C == 4 && I->ablate >= 256)
        { // PHYHIP_ABLATE = 256 + bits: timing-only ablations of the 20-state kernel (results invalid)
#define AAABL(a_) case a_: hipLaunchKernelGGL((traverse_aa_kernel<4, false, a_>), dim3(I->grid_aa), blk, 0, I->stream, q, d_irec, d_xrec, (const double *)I->d_afrag, I->nmat_all, (const uint32_t *)I->d_tipmasks, (unsigned long long *)nullptr); return 0;
          switch (I->ablate - 256)
          {
            AAABL(1) AAABL(2) AAABL(4) AAABL(8) AAABL(9) AAABL(16) AAABL(18) AAABL(5) AAABL(13) AAABL(31) AAABL(27)
            default: break;
          }
#undef AAABL
        }
        if ((I->ablate & 8) && I->ablate < 256 && I->C == 4)
        { // PHYHIP_ABLATE=8: cycle stamps of one consumer wave, printed to stderr (diagnostics; costs a sync per launch)
          unsigned long long *&d_dbg = I->d_dbg;
          if (!d_dbg) HIPCHK(hipMalloc((void **)&d_dbg, 64 * 8 * 8));
          if (I->ablate & 128) // (stamps of the bare skeleton: every ablation on)
            hipLaunchKernelGGL((traverse_aa_kernel<4, true, 31>), dim3(I->grid_aa), blk, 0, I->stream, q, d_irec, d_xrec,
                               (const double *)I->d_afrag, I->nmat_all, (const uint32_t *)I->d_tipmasks, d_dbg);
          else
          hipLaunchKernelGGL((traverse_aa_kernel<4, true>), dim3(I->grid_aa), blk, 0, I->stream, q, d_irec, d_xrec,
                             (const double *)I->d_afrag, I->nmat_all, (const uint32_t *)I->d_tipmasks, d_dbg);
          static int printed = 0;
          if (printed++ == 5)
          {
            unsigned long long h[64 * 8];
            HIPCHK(hipMemcpyAsync(h, d_dbg, sizeof h, hipMemcpyDeviceToHost, I->stream));
            HIPCHK(hipStreamSynchronize(I->stream));
            for (int k = 0; k < 64 && k < q.n_ops; ++k)
            {
              fprintf(stderr, "step %2d:", k);
              for (int i = 1; i < 7; ++i) fprintf(stderr, " %6lld", (long long)(h[k * 8 + i] - h[k * 8 + i - 1]));
              if (k + 1 < 64) fprintf(stderr, "  | next %6lld", (long long)(h[(k + 1) * 8] - h[k * 8 + 6]));
              fprintf(stderr, "\n");
            }
          }
          return 0;
        }
#endif
        switch (I->C)
        {
          AACASE(1) AACASE(2) AACASE(3) AACASE(4)
          default: break;
        }
#undef AACASE
      }
    }
    named("traverse_kernel<%d, %d>", S_, CP_);
    hipLaunchKernelGGL((traverse_kernel<S_, CP_>), dim3(I->grid), dim3(256), 0, I->stream, q, ro.ops, ro.pmats, ro.tip_codes,
                       ro.code_masks);
    return 0;
  });
  if (rc) return rc;
  if (I->prof)
  {
    HIPCHK(hipEventRecord(e1, I->stream));
    I->prof_pairs.emplace_back(e0, e1);
    I->prof_updates += (double)n_queued * (double)I->P;
    // Minimum traffic of this launch if nothing but the kernel's own register forwarding saved a byte: every result is
    // written once; a child is read unless it is a tip (1 byte per pattern) or the result of one of the previous two
    // operations (forwarded in registers -- exactly the flags computed for the operation records above).
    {
      const double rec = (double)I->C * I->S * 8.0 + 4.0;
      double       rd = 0.0, wr = 0.0;
      for (int k = 0; k < n_ops; ++k)
      {
        const DevOp &o  = I->pending[k];
        if (!(o.pad & kOpNoStore)) wr += rec;
        const int    e1 = k >= 1 ? I->pending[k - 1].dest : -1;
        const int    e2 = (k >= 2 && fat && I->prefetch_dist == 2) ? I->pending[k - 2].dest : -1;
        for (int w = 0; w < 2; ++w)
        {
          const int c = w ? o.c2 : o.c1;
          if (o.pad & (w ? kOpInl2 : kOpInl1)) rd += 2.0; // (an in-step child: its two tip bytes)
          else rd += c < I->tips ? 1.0 : ((fat && (c == e1 || c == e2)) ? 0.0 : rec);
        }
      }
      if (ee)
      { // root edge: both sides unless just produced, pattern weight in; per-pattern outputs out
        for (int c : {ee->parent, ee->child})
          rd += c < I->tips ? 1.0 : ((fat && n_ops > 0 && c == I->pending[n_ops - 1].dest) ? 0.0 : rec);
        rd += 8.0;
        wr += 4.0 + (I->want_site_outputs ? 16.0 + 8.0 * I->C : 0.0);
      }
      I->prof_rd_bytes += rd * (double)I->P;
      I->prof_wr_bytes += wr * (double)I->P;
    }
  }
  if (kDiag) { const unsigned long long hp2 = hp_now(); g_hp.prep += hp1 - hp0; g_hp.launch += hp2 - hp1; ++g_hp.n_launch; }
  HIPCHK(hipGetLastError());
  I->host_sum_n = host_sum_n; I->host_sum_ns = 1;
  if (ee && !ee->eigen && !fused_sum && !host_sum_n && !I->class_axis) // (class axis: the combination kernel follows, no sum here)
  {
    double *out = ee->dev_out ? ee->dev_out : I->d_result;
    const int nsum = I->soa ? soa_grid : (I->perm ? I->grid_aa : (fat ? I->grid_nt : I->grid));
    hipLaunchKernelGGL(final_reduce_kernel, dim3(1), dim3(256), 0, I->stream, (const double *)I->d_block, nsum, 1,
                       nsum, out, ee->to_host ? I->h_result : (double *)nullptr, I->d_warn, I->h_warn,
                       ee->to_host ? ++I->seq : 0ull, ee->warn_out);
    HIPCHK(hipGetLastError());
  }
  if (fold_pm)
  {
    for (int m : I->pm_idx) I->pm_slot[m] = -1;
    I->pm_idx.clear();
    I->pm_len.clear();
    I->pm_shadow.clear();
  }
  I->pending.clear();
  std::fill(I->mat_in_queue.begin(), I->mat_in_queue.end(), 0);
  // an evaluation the large-grid resident workgroups would have taken, had the stream been known to be idle: say when it is
  if (big_fit && host_sum_n > 0 && (rc = stamp_stream(I))) return rc;
  return 0;
}

int flush(Instance *I, const EdgeEval *ee)
{
  const int rc = flush_impl(I, ee);
  if (rc)
  { // a failed launch leaves no half-queued state behind: the operations are dropped (the caller gets the error and
    // PhyML's glue exits on it), queued matrix rebuilds stay queued, no device slot claims a list it never received
    I->pending.clear();
    std::fill(I->mat_in_queue.begin(), I->mat_in_queue.end(), 0);
  }
  return rc;
}

int flush_sync(Instance *I)
{
  int rc = flush(I, nullptr);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(I->stream));
  return 0;
}

int check_partial_index(const Instance *I, int idx, bool allow_tip)
{
  if (idx < 0 || idx >= I->nbuf || (!allow_tip && idx < I->tips))
    return fail(PHYHIP_ERROR_OUT_OF_RANGE, "partials buffer index %d out of range [%d,%d)", idx, allow_tip ? 0 : I->tips, I->nbuf);
  return 0;
}

// Wait until the final reduction has published evaluation `seq` in host-mapped memory.  Spinning on the
// sequence word avoids the stream-synchronise wake-up latency; after 2 ms of spinning (elapsed time, checked every
// 256 polls) fall back to it: evaluations of very large alignments take milliseconds and must not burn a core.
// The host's side of the final sum on large grids: poll the {sum, tag} records the workgroups posted (they arrive roughly in
// launch order), then add them exactly as final_reduce_kernel does -- 256 strided accumulators, then a binary tree -- so
// that the value does not depend on which path produced it.
int wait_host_sum(Instance *I)
{
  const int                n   = I->host_sum_n * I->host_sum_ns, per = I->host_sum_n;
  const unsigned long long tag = I->seq;
  volatile HostBlock      *hb  = I->h_blocks;
  struct timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  bool   synced = false;
  // final_reduce_kernel's order -- 256 strided accumulators per sum, then a binary tree -- with the records taken as they
  // arrive, front to back: accumulator t receives records t, t + 256, ... in that order either way, and one sequential pass
  // over the records costs a fraction of 256 strided ones (thousands of records per evaluation on large grids)
  double acc[2][256];
  for (int k = 0; k < I->host_sum_ns; ++k)
    for (int t = 0; t < 256; ++t) acc[k][t] = 0.0;
  for (int i = 0, k = 0, j = 0; i < n; ++i)
  {
    long it = 0;
    while (hb[i].tag != tag)
    {
      __builtin_ia32_pause();
      if ((++it & 255) == 0 && !synced)
      {
        struct timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const long waited = (t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec);
        if (I->r_inflight)
        { // no answer from the resident workgroups (they may have left just before the command arrived): the caller
          // retires them and launches the evaluation the ordinary way
          if (waited > 30000L && resident_gone(*I->r_inflight)) return kResidentSilent; // they left as the command arrived
          // (a healthy command of the large-grid evaluator takes ~8 ns per tile and operation -- 25 us at 3 126 tiles -- so the
          // give-up time grows with the grid: 100 us + 80 ns per tile)
          const long patience = 100000L + (I->r_inflight == &I->rb ? 80L * (long)I->grid_nt2 : 0L);
          if (waited > patience && ns_since(I->r_inflight->t_launch) > 20e6) return kResidentSilent; // (a first launch loads code: ms)
          continue;
        }
        if (waited > 2000000L || !I->spin_wait)
        {
          HIPCHK(hipStreamSynchronize(I->stream));
          synced = true;
          it = 0;
        }
      }
      else if (synced && it > 100000000L)
        return fail(PHYHIP_ERROR_GENERAL, "evaluation %llu finished without posting block sum %d", tag, i);
    }
    acc[k][j & 255] += hb[i].sum; // (the record is ONE 16-byte store of the device: the sum is there when the tag is)
    if (++j == per) { j = 0; ++k; }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  for (int k = 0; k < I->host_sum_ns; ++k)
  {
    for (int off = 128; off > 0; off >>= 1)
      for (int t = 0; t < off; ++t) acc[k][t] += acc[k][t + off];
    I->h_result[k] = acc[k][0];
  }
  I->host_sum_n    = 0;
  if (I->r_inflight) I->r_inflight->ns_wait += ns_since(I->r_inflight->t_cmd);
  I->r_inflight    = nullptr;
  I->warn_current  = true;
  *reinterpret_cast<volatile unsigned long long *>(I->h_result + 2) = tag;
  return 0;
}

int wait_result_impl(Instance *I);
int wait_result(Instance *I)
{
  const unsigned long long t0 = hp_now();
  const int rc = wait_result_impl(I);
  if (kDiag) { const unsigned long long t1 = hp_now(); g_hp.wait += t1 - t0; ++g_hp.n_wait; if (!g_hp.t_first) g_hp.t_first = t0; g_hp.t_last = t1; }
  return rc;
}
int wait_result_impl(Instance *I)
{
  if (I->host_sum_n > 0) return wait_host_sum(I);
  if (I->spin_wait)
  {
    volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(I->h_result + 2);
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (long it = 0;; ++it)
    {
      if (*flag == I->seq)
      {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        I->warn_current = true;
        return 0;
      }
      __builtin_ia32_pause();
      if ((it & 255) == 255)
      {
        struct timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 2000000L) break;
      }
    }
  }
  HIPCHK(hipStreamSynchronize(I->stream));
  if (*reinterpret_cast<volatile unsigned long long *>(I->h_result + 2) != I->seq)
  { // the stream drained without the hand-over (a faulted launch): do not return a stale scalar, re-arm the ticket counter
    (void)hipMemsetAsync(I->d_tickets, 0, sizeof(unsigned) * (1 + kTicketGroups), I->stream);
    return fail(PHYHIP_ERROR_GENERAL, "evaluation %llu finished without handing its result over", I->seq);
  }
  I->warn_current = true;
  return 0;
}

int collect_profile(Instance *I)
{
  for (auto &pr : I->prof_pairs)
  {
    HIPCHK(hipEventSynchronize(pr.second));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, pr.first, pr.second));
    I->prof_ms += ms;
    I->prof_n += 1;
    for (hipEvent_t e : {pr.first, pr.second})
      if (I->prof_spare.size() < 1024) I->prof_spare.push_back(e);
      else (void)hipEventDestroy(e);
  }
  I->prof_pairs.clear();
  for (auto &pr : I->prof_aux)
  {
    HIPCHK(hipEventSynchronize(pr.b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, pr.a, pr.b));
    I->prof_aux_ms[pr.kind] += ms;
    I->prof_aux_n[pr.kind] += 1;
    (void)hipEventDestroy(pr.a);
    (void)hipEventDestroy(pr.b);
  }
  I->prof_aux.clear();
  return 0;
}

// An evaluation the host waits for (edge sum, or the empty records of a small alignment's Update_Eigen_Lr): queue, launch or
// hand to the resident short-launch evaluator, wait.  An evaluation the resident workgroups do not answer is launched.
int flush_and_wait(Instance *I, EdgeEval &ee, bool flushed)
{
  int rc = flushed ? 0 : flush(I, &ee);
  if (rc) return rc;
  Resident  *const by = I->r_inflight;
  const bool by_resident = by != nullptr;
  rc = wait_result(I);
  if (rc == kResidentSilent)
  { // the resident workgroups had left: retire them for good (no late record can arrive after this), put the evaluation
    // back in the queue and launch it
    ++by->n_silent;
    resident_stop(*by);
    if (by == &I->rb) big_release(I);
    I->r_inflight = nullptr; I->host_sum_n = 0;
    I->pending = I->rt_ops;
    for (const DevOp &o : I->pending) { I->mat_in_queue[o.pm1] = 1; I->mat_in_queue[o.pm2] = 1; }
    for (size_t k = 0; k < I->rt_pm_idx.size(); ++k)
      if (I->pm_slot[I->rt_pm_idx[k]] < 0)
      {
        I->pm_slot[I->rt_pm_idx[k]] = (int)I->pm_idx.size();
        I->pm_idx.push_back(I->rt_pm_idx[k]);
        I->pm_len.push_back(I->rt_pm_len[k]);
        I->pm_shadow.push_back(-1);
      }
    for (int k = 0; k < I->rt_up_n; ++k)
    { // ... and its host-computed matrices: queued again as phyhip_set_transition_matrix queued them
      const int    m = I->rt_up_idx[k];
      const size_t bytes = (size_t)I->C * I->S * I->S * sizeof(double);
      void        *st = nullptr;
      if ((rc = I->ring.alloc(bytes, I->stream, &st))) return rc;
      memcpy(st, I->rt_up_val[k], bytes);
      if (I->up_slot[m] >= 0) continue; // (cannot be: the command took every queued upload)
      I->up_slot[m] = (int)I->up_idx.size();
      I->up_idx.push_back(m);
      I->up_src.push_back((const double *)st);
      I->up_shadow.push_back(-1);
    }
    I->rt_up_n = 0;
    I->rt_skip = true;
    rc = flush(I, &ee);
    I->rt_skip = false;
    if (rc) return rc;
    rc = wait_result(I);
  }
  if (rc) return rc;
  if (I->fenced_eval)
  { // every store of this evaluation -- and so everything queued before it -- is in memory
    I->fenced_eval = false; I->stream_dirty = false; I->clean_after = 0;
    // (a kernel ran, or the small evaluators' dot_prod was rewritten by another set of workgroups: they re-read.  Not the
    // large-grid evaluator's: the wave that evaluates a tile's dLk is the one that wrote its products, phyhip_big.hpp)
    if (!by_resident || (ee.eigen && by != &I->rb)) ++I->clean_epoch;
  }
  return 0;
}

} // namespace phyhip_host

