// phyhip_eigen.hip -- Update_Eigen_Lr and the eigen-basis evaluations (dLk / Lk): entry points and their launches
// (libphyhip.so, gfx950 only; the units and what they share: phyhip_host.hpp)
#include "phyhip_host.hpp"

namespace phyhip_host
{

// dLk in the traversal's tiles (dlk_tile): the launched form of what the large-grid resident workgroups serve
template <int CP> static void launch_dlk64(Instance *I, const DlkParams &q, int dgrid)
{
  if (I->nt_groups == 2) hipLaunchKernelGGL((dlk64_kernel<4, CP, (CP >= 2 ? CP / 2 : 1)>), dim3(dgrid), dim3(64), 0, I->stream, q);
  else hipLaunchKernelGGL((dlk64_kernel<4, CP, CP>), dim3(dgrid), dim3(64), 0, I->stream, q);
}

} // namespace phyhip_host

using namespace phyhip_host;

extern "C" {

// ---- eigen basis -------------------------------------------------------------------------------------

int phyhip_update_eigen_lr(int instance, int left, int rght)
{
  if (Group *G = get_group_nodrain(instance))
  { // (the recorded queue-only calls -- the partial update in front of the products -- are replayed by the same job)
    const int rc = group_parallel(G, [&](int g) -> int {
      const int r = group_replay(G, g);
      return r ? r : phyhip_update_eigen_lr(G->sub_id[g], left, rght);
    });
    G->deferred.clear(); G->deferred_ops = 0;
    return rc;
  }
  GET_INST_RES(I, instance);
  int rc = check_partial_index(I, left, true);
  if (rc) return rc;
  if ((rc = check_partial_index(I, rght, true))) return rc;
  // virtual buffers the products (or the queued updates in front of them) read are stored first: the decisions below see the
  // queue as it will be launched
  devirtualise(I, left); devirtualise(I, rght);
  rewrite_pending(I, nullptr, false);
  // Small nucleotide alignments (the resident short-launch evaluator's range, 2 048 patterns): the queued partial update(s) and
  // the products are ONE launch of the lane-per-pattern kernel (TreeParams::edge_eval 2) -- or, mostly, one command of the
  // resident workgroups.  Measured by chain (1 Update_Eigen_Lr + 5 dLk, docs/history/tools/gpu_fuse_eigen_cross.sh): 40.3 vs 43.8 us at 382
  // patterns, 53.7 vs 56.2 at 2 048; WITHOUT the resident evaluator the fused launch loses to eigen_lr_kernel at every size
  // (+2 us: it stores dot_prod 16 bytes per lane at a 64-byte stride), beyond 4 096 patterns by 4-7 us -- so nowhere else.
  // Large alignments: the same, when the large-grid resident workgroups (phyhip_big.hpp) can take it -- launched, the fused form
  // loses to eigen_lr_kernel there, as a resident command it is the partial update and the products in one trip.
  const bool big_eig = I->NE == 1 && I->C <= 4 && I->fuse_eigen && big_eligible(I) && I->pending.size() <= 2 && I->pm_idx.size() <= 4 &&
                       I->up_idx.empty() && I->args_recs && I->fold_pmats && (I->rb.launched || I->big_streak >= 1) && big_ready(I);
  if (I->NE == 1 && I->C <= 4 && I->fuse_eigen && (resident_short_eligible(I) || big_eig))
  {
    AuxProf  ap(I, 0);
    EdgeEval ee{left, rght, 0, nullptr, false, nullptr};
    ee.eigen = true;
    const bool   wc = I->warn_current;
    const double keep = I->h_result[0];
    // the workgroups (launched or resident) post empty records once their products are in memory
    if ((rc = flush_and_wait(I, ee))) return rc;
    I->h_result[0] = keep; I->warn_current = wc;
    I->eig_api_no = 0;
    return PHYHIP_SUCCESS;
  }
  if ((rc = flush(I, nullptr))) return rc;
  big_release(I, false);
  I->touched_call = true;
  EigenParams e;
  e.t = base_params(I); e.ro = base_ro(I, nullptr); e.left = left; e.rght = rght; e.r_e_vect = I->d_evec; e.l_e_vect = I->d_ivec; e.dot_prod = I->d_dot;
  const bool report = I->resident && (I->S == 4 || dlk_from_len(I)) && I->host_sum && I->spin_wait && I->grid <= kResidentMaxGrid && !I->co; // (see eigen_eval)
  e.tickets = report ? I->d_tickets : nullptr;
  e.stamp_host = reinterpret_cast<unsigned long long *>(I->h_result + 3); e.stamp = report ? ++I->stamp_seq : 0ull;
  AuxProf ap(I, 0);
  rc = dispatch_shape(I, [&](auto s, auto cp) {
    constexpr int S_ = decltype(s)::value, CP_ = decltype(cp)::value;
    const size_t lds = sizeof(double) * 2 * (size_t)(I->class_axis ? I->C : 1) * S_ * S_; // the eigenvectors, staged per workgroup
    const int egrid = (int)(((long long)I->P * I->CP * kEigenSplit<S_> + 255) / 256);
    if (I->class_axis) hipLaunchKernelGGL((eigen_lr_kernel<S_, CP_, true>), dim3(egrid), dim3(256), lds, I->stream, e);
    else hipLaunchKernelGGL((eigen_lr_kernel<S_, CP_, false>), dim3(egrid), dim3(256), lds, I->stream, e);
    return 0;
  });
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  if (report) { I->stream_dirty = false; I->clean_after = e.stamp; }
  else if (big_eligible(I) && !I->prof && (rc = stamp_stream(I))) return rc; // (large instance: the dLk calls that follow can be served resident)
  I->eig_api_no = report ? I->api_no : 0;
  return PHYHIP_SUCCESS;
}

// dev_out != nullptr (sharded evaluation): the two sums stay in device memory (dev_out[0..1]), the warning flag goes to
// *warn_out as a double, nothing is handed to the host and the call does not wait.
static int eigen_eval(Instance *I, double l, bool deriv, double *lnl, double *dlnl, double *dev_out = nullptr,
                      double *warn_out = nullptr)
{
  // the expl table: in the kernel arguments up to 8 categories, staged into device memory beyond (C <= kMaxCategories)
  const bool          expl_in_args = (size_t)I->C * 2 * I->S <= (size_t)kMaxExpl;
  std::vector<double> expl_big;
  if (!expl_in_args) expl_big.assign((size_t)I->C * 2 * I->S, 0.0);
  int rc = flush(I, nullptr);
  if (rc) return rc;
  DlkParams q;
  memset(&q, 0, sizeof q);
  q.dot_prod = I->d_dot; q.wght = I->d_wght; q.fact = I->d_fact; q.cat_w = I->d_catw; q.pi = I->d_pi; q.invar = I->d_invar;
  q.P = I->P; q.C = I->C; q.invar_model = I->invar_model; q.apply_scaling = I->apply_scaling; q.with_derivative = deriv ? 1 : 0;
  // 20 states: the workgroups build the expl table from the edge length themselves (DlkParams::from_len)
  const bool from_len = dlk_from_len(I) && expl_in_args;
  q.eval_dev = I->d_eval; q.rates_dev = I->d_catr; q.br_len_mult = I->br_len_mult; q.l_min = I->l_min; q.l_max = I->l_max;
  q.from_len = from_len ? 1 : 0; q.len = from_len ? l : 0.0;
  // grid-stride kernel.  Measured (us per dLk incl. launch and host hand-over): 512 workgroups + fused final sum win up to
  // ~100 MB of dot_prod (20 states x 100 k patterns: 33 vs 48); beyond, filling every wave slot (2048 workgroups, separate
  // final sum) streams better (4 states x 1 M patterns: 45 vs 64; the one-workgroup-per-256-lanes form took 75)
  const size_t dot_bytes = (size_t)I->P * I->C * I->S * sizeof(double);
  int          dgrid = std::min(I->grid, dot_bytes > (size_t)100 << 20 ? 2048 : 512);
  if (const char *e = diag_env("PHYHIP_DLK_GRID")) dgrid = std::min(I->grid, std::max(1, atoi(e)));
  const bool hsum  = !dev_out && I->host_sum;
  // Large nucleotide alignments with the scalar wanted on the host: the evaluation is cut into one-wave virtual blocks
  // (dlk_tile) -- what the large-grid resident workgroups serve (phyhip_big.hpp) and, launched, dlk64_kernel: the same doubles
  const bool big = hsum && big_shape(I);
  if (big) dgrid = I->n_vdlk;
  q.pinvar = I->pinvar; q.fin.block_sums = I->d_block; q.fin.stride = dgrid; q.fin.warn = I->d_warn;
  const bool fused = !hsum && fuse_reduce(I, dgrid);
  if (hsum)
  { // both sums posted to the host per workgroup (see flush_impl)
    q.fin.host_blocks = I->h_blocks; q.fin.host_tag = ++I->seq; q.fin.warn = I->h_warn;
    *I->h_warn = 0;
  }
  if (fused)
  {
    q.fin.tickets = I->d_tickets; q.fin.result = dev_out ? dev_out : I->d_result;
    q.fin.result_host = dev_out ? nullptr : I->h_result; q.fin.warn_host = I->h_warn;
    q.fin.seq = dev_out ? 0ull : ++I->seq; q.fin.warn_out = warn_out;
  }
  double *const xp = expl_in_args ? q.expl : expl_big.data();
  for (int c = 0; c < (from_len ? 0 : I->C); ++c)
  {
    if (deriv)
    { // src/lk.c:688-726
      const double rr  = I->h_rates[c] * I->br_len_mult;
      double       len = l * rr;
      if (len < I->l_min) len = I->l_min;
      else if (len > I->l_max) len = I->l_max;
      for (int s = 0; s < I->S; ++s)
      {
        const double ev = I->h_eval[s], ex = exp(ev * len);
        xp[c * 2 * I->S + 2 * s]     = ex;
        xp[c * 2 * I->S + 2 * s + 1] = ex * ev * rr;
      }
    }
    else
    { // src/lk.c:594-602
      double len = (l > 0.0 ? l : 0.0) * I->h_rates[c];
      len *= I->br_len_mult;
      if (len < I->l_min) len = I->l_min;
      else if (len > I->l_max) len = I->l_max;
      for (int s = 0; s < I->S; ++s) xp[c * I->S + s] = exp(I->h_eval[s] * len);
    }
  }
  if (!expl_in_args)
  { // (the mixture evaluations' table area, kMaxMixClasses x 2 x 20 doubles: stream order keeps the two apart)
    void        *st = nullptr;
    const size_t eb = expl_big.size() * sizeof(double);
    if ((rc = I->ring.alloc(eb, I->stream, &st))) return rc;
    memcpy(st, expl_big.data(), eb);
    big_release(I, false);
    I->touched_call = true;
    HIPCHK(hipMemcpyAsync(I->d_mixexpl, st, eb, hipMemcpyHostToDevice, I->stream));
    q.expl_dev = I->d_mixexpl;
  }
  // Small alignment, scalar wanted on the host: hand the evaluation to the resident workgroups (resident_dlk_kernel) when
  // nothing of this instance is still running on its stream -- they are not ordered with it.  Right after Update_Eigen_Lr
  // the products are a few microseconds away: poll the stream that long, else launch as usual.
  // (4 states only: a 20-state command takes four 512-byte reads per poll instead of one and the round trip loses to the launch, 14.1-14.9 against
  // 12.4-12.5 us at 2 000 patterns -- measured, docs/history/tools/gpu_resident_ab2.sh)
  if (kDiag && getenv("PHYHIP_RESIDENT_DEBUG") && big)
    fprintf(stderr, "big dLk: eligible %d | dirty %d dirty_prev %d touched %d clean_after %llu stamp %llu streak %d launched %d\n", (int)big_eligible(I),
            (int)I->stream_dirty, (int)I->dirty_prev, (int)I->touched_call, I->clean_after, *reinterpret_cast<volatile unsigned long long *>(I->h_result + 3),
            I->big_streak, (int)I->rb.launched);
  if (big && big_eligible(I) && big_ready(I))
  {
    I->stream_dirty = false; // (found idle; a dLk leaves nothing on the stream)
    const int brc = big_ensure(I);
    if (brc < 0) return brc;
    if (brc == 0)
    {
      Resident          &R = I->rb;
      unsigned long long words[kBigWords];
      memset(words, 0, sizeof words);
      const int  n_expl = I->C * (deriv ? 2 : 1) * I->S;
      const bool changed = I->clean_epoch != I->rt_epoch; // kernels ran on the stream since the last command
      const bool dsum = dgrid > I->big_device_sum;
      words[0] = q.fin.host_tag;
      words[1] = kBigDlk | (q.with_derivative ? kBigDeriv : 0ull) | (q.invar_model ? kBigInvar : 0ull) | (q.apply_scaling ? kBigScaling : 0ull) |
                 (changed ? kBigChanged : 0ull) | (dsum ? kBigDeviceSum : 0ull) | (dsum && big_sum_by_group(I, dgrid) ? kBigGroupSum : 0ull);
      memcpy(&words[2], &q.pinvar, 8);
      memcpy(&words[3], q.expl, sizeof(double) * (size_t)n_expl);
      resident_send(I, R, words, kBigWords);
      I->rb_dlk_api = I->api_no; I->rt_epoch = I->clean_epoch;
      I->host_sum_n = dsum ? 1 : dgrid; I->host_sum_ns = 2;
      rc = wait_result(I);
      if (rc == 0)
      {
        *lnl = I->h_result[0];
        if (dlnl) *dlnl = I->h_result[1];
        return PHYHIP_SUCCESS;
      }
      if (rc != kResidentSilent) return rc;
      // nobody there: make sure of it, then the ordinary launch below repeats the evaluation -- under a NEW tag: workgroups
      // that had started on the command may have posted the (single) record of a device-side final sum under the old one,
      // where the launched form's first tile record goes
      ++R.n_silent;
      resident_stop(R);
      big_release(I);
      I->r_inflight = nullptr; I->host_sum_n = 0;
      q.fin.host_tag = ++I->seq;
    }
  }
  else if (big && big_eligible(I)) { ++I->rb.n_busy; I->big_streak = 0; }
  if (big) big_release(I, false); // (launched on the stream: behind the resident workgroups' exit, if there are any)
  // (20 states, re-measured in round 4 with compact commands -- exp() values only, two 512-byte reads instead of four, ONE
  // polling workgroup, the others on the device-memory mailbox: 10.7 us from command to answer on the recorded proteic search,
  // 18.2 against 17.5 us per scalar-returning call, 13.2 against 13.4 us per dLk at 2 000 patterns -- still no gain: removed)
  if (!big && hsum && I->resident && (I->S == 4 || from_len) && dgrid <= kResidentMaxGrid && I->spin_wait && expl_in_args)
  {
    bool idle = !I->stream_dirty;
    if (idle && I->clean_after)
    { // the report of the last Update_Eigen_Lr: a few microseconds away at most (bounded wait, then the ordinary launch)
      volatile unsigned long long *stamp = reinterpret_cast<volatile unsigned long long *>(I->h_result + 3);
      struct timespec t0;
      clock_gettime(CLOCK_MONOTONIC, &t0);
      for (long it = 1; *stamp < I->clean_after && idle; ++it)
      {
        __builtin_ia32_pause();
        if ((it & 255) == 0 && ns_since(t0) > 200000.0) idle = false;
      }
      if (idle) { __atomic_thread_fence(__ATOMIC_ACQUIRE); I->clean_after = 0; ++I->clean_epoch; }
    }
    if (!idle) ++I->rd.n_busy;
    if (idle)
    {
      DlkParams qs = q; // what stays the same from call to call
      qs.with_derivative = 0; qs.invar_model = 0; qs.apply_scaling = 0; qs.pinvar = 0.0; qs.fin.host_tag = 0; qs.len = 0.0;
      memset(qs.expl, 0, sizeof qs.expl);
      const DlkParams &o = I->r_static;
      Resident  &R = I->rd;
      const bool same = R.launched && R.grid == dgrid && o.dot_prod == qs.dot_prod && o.wght == qs.wght && o.fact == qs.fact &&
                        o.cat_w == qs.cat_w && o.pi == qs.pi && o.invar == qs.invar && o.P == qs.P && o.C == qs.C &&
                        o.fin.host_blocks == qs.fin.host_blocks && o.fin.stride == qs.fin.stride && o.fin.warn == qs.fin.warn &&
                        o.from_len == qs.from_len && o.eval_dev == qs.eval_dev && o.rates_dev == qs.rates_dev && o.br_len_mult == qs.br_len_mult &&
                        o.l_min == qs.l_min && o.l_max == qs.l_max;
      if (!same)
      {
        resident_stop(R);
        if ((rc = resident_launch_dlk(I, qs, dgrid, R.seq))) return rc;
      }
      else if (resident_gone(R))
      { // the workgroups have left (idle)
        if ((rc = resident_launch_dlk(I, qs, dgrid, R.seq))) return rc;
      }
      unsigned long long words[kResidentWords];
      const int          n_words = from_len ? 4 : 3 + I->C * 2 * I->S;
      const bool         changed = I->api_no != R.api_no + 1; // something else was called since the last command
      words[0] = q.fin.host_tag;
      words[1] = (q.with_derivative ? 1u : 0u) | (q.invar_model ? 2u : 0u) | (q.apply_scaling ? 4u : 0u) | (changed ? 8u : 0u) | (from_len ? 16u : 0u);
      memcpy(&words[2], &q.pinvar, 8);
      if (from_len) memcpy(&words[3], &l, 8);
      else memcpy(&words[3], q.expl, sizeof(double) * (size_t)(n_words - 3));
      resident_send(I, R, words, n_words);
      I->host_sum_n = dgrid; I->host_sum_ns = 2;
      rc = wait_result(I);
      if (rc == 0)
      {
        *lnl = I->h_result[0];
        if (dlnl) *dlnl = I->h_result[1];
        return PHYHIP_SUCCESS;
      }
      if (rc != kResidentSilent) return rc;
      // nobody there: make sure of it (after this no resident workgroup can still write a record), then the ordinary
      // launch below repeats the evaluation under the same tag
      ++R.n_silent;
      resident_stop(R);
      I->r_inflight = nullptr; I->host_sum_n = 0;
    }
  }
  // the large-grid kernel launched for this one evaluation (BigArgs::n_one_shot): 256 workgroups that add per workgroup and
  // post one record per sum, instead of one workgroup and two records per tile
  const bool one_shot = big && hsum && !dev_out && I->big_oneshot && I->spin_wait && dgrid > I->big_device_sum && big_sum_by_group(I, dgrid);
  if (one_shot)
  {
    unsigned long long words[kBigWords];
    memset(words, 0, sizeof words);
    const int n_expl = I->C * (deriv ? 2 : 1) * I->S;
    words[0] = q.fin.host_tag;
    words[1] = kBigDlk | (q.with_derivative ? kBigDeriv : 0ull) | (q.invar_model ? kBigInvar : 0ull) | (q.apply_scaling ? kBigScaling : 0ull) |
               kBigDeviceSum | kBigGroupSum;
    memcpy(&words[2], &q.pinvar, 8);
    memcpy(&words[3], q.expl, sizeof(double) * (size_t)n_expl);
    {
      AuxProf ap(I, 1);
      if ((rc = big_one_shot(I, words, kBigWords))) return rc;
    }
    I->host_sum_n = 1; I->host_sum_ns = 2;
    if (big_eligible(I) && !I->prof && (rc = stamp_stream(I))) return rc; // (the next one can go to the resident workgroups)
    if ((rc = wait_result(I))) return rc;
    *lnl = I->h_result[0];
    if (dlnl) *dlnl = I->h_result[1];
    return PHYHIP_SUCCESS;
  }
  {
  AuxProf ap(I, 1);
  rc = dispatch_shape(I, [&](auto s, auto cp) {
    constexpr int S_ = decltype(s)::value, CP_ = decltype(cp)::value;
    const unsigned long long h1 = hp_now();
    // (the last-workgroup sum instead of 2 x 3 126 records was measured for this launch: 37 against 28-33 us per call -- a ticket
    // per one-wave workgroup is thousands of atomics at the memory side)
    if (big) { if constexpr (S_ == 4 && CP_ <= 4) launch_dlk64<CP_>(I, q, dgrid); }
    else hipLaunchKernelGGL((dlk_kernel<S_, CP_>), dim3(dgrid), dim3(256), 0, I->stream, q);
    if (kDiag) { g_hp.launch += hp_now() - h1; ++g_hp.n_launch; }
    return 0;
  });
  }
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  if (hsum) { I->host_sum_n = dgrid; I->host_sum_ns = 2; }
  if (big && big_eligible(I) && !I->prof && (rc = stamp_stream(I))) return rc; // (the next one can go to the resident workgroups)
  if (!fused && !hsum)
  {
    hipLaunchKernelGGL(final_reduce_kernel, dim3(1), dim3(256), 0, I->stream, (const double *)I->d_block, dgrid, 2, dgrid,
                       dev_out ? dev_out : I->d_result, dev_out ? (double *)nullptr : I->h_result, I->d_warn, I->h_warn,
                       dev_out ? 0ull : ++I->seq, warn_out);
    HIPCHK(hipGetLastError());
  }
  if (dev_out)
  {
    I->warn_current = false;
    return PHYHIP_SUCCESS;
  }
  if ((rc = wait_result(I))) return rc;
  *lnl = I->h_result[0];
  if (dlnl) *dlnl = I->h_result[1];
  return PHYHIP_SUCCESS;
}

// dLk / eigen-basis Lk on the shards + the collective (count 3: warning, lnL, dlnL)
static int group_eigen_eval(Group *G, double l, bool deriv, double *lnl, double *dlnl)
{
  if (group_combines_on_host(G, false))
  { // shard by shard through the plain entry points (resident dLk evaluators included), added here in shard order
    std::vector<double> pl(G->sub.size(), 0.0), pd(G->sub.size(), 0.0);
    std::vector<int>    warn(G->sub.size(), 0);
    int rc = group_parallel(G, [&](int g) -> int {
      double lg = l; // (clamped by the entry point: the same value on every shard)
      int    r = group_replay(G, g);
      if (r) return r;
      r = deriv ? phyhip_calculate_eigen_lnl_dlnl(G->sub_id[g], &lg, &pl[g], &pd[g]) : phyhip_calculate_eigen_lnl(G->sub_id[g], l, &pl[g]);
      return r ? r : phyhip_get_numerical_warning(G->sub_id[g], &warn[g]);
    });
    G->deferred.clear(); G->deferred_ops = 0;
    if (rc) return rc;
    double a = 0.0, b = 0.0;
    for (size_t g = 0; g < pl.size(); ++g) { a += pl[g]; b += pd[g]; }
    *lnl = a;
    if (dlnl) *dlnl = b;
    return group_collect_warning(G, warn);
  }
  if (group_drain(G)) return group_take_drain_error(G);
  int rc = group_parallel(G, [&](int g) -> int {
    double *slot = shard_slot(G->co->ctx[G->ctx_of[g]], G->k_of[g]);
    return eigen_eval(G->sub[g], l, deriv, nullptr, nullptr, slot + 1, slot);
  });
  if (rc) return rc;
  rc = reduce_and_publish(*G->co, 3, G->sub[0]);
  if (rc) return rc;
  *lnl = G->sub[0]->h_result[0];
  if (dlnl) *dlnl = G->sub[0]->h_result[1];
  G->last_warn  = *G->sub[0]->h_warn;
  G->warn_valid = true;
  return PHYHIP_SUCCESS;
}

static int rank_eigen_eval(Instance *I, double l, bool deriv, double *lnl, double *dlnl)
{
  int rc = eigen_eval(I, l, deriv, nullptr, nullptr, I->d_red + 1, I->d_red);
  if (rc) return rc;
  if ((rc = reduce_and_publish(*I->co, 3, I))) return rc;
  *lnl = I->h_result[0];
  if (dlnl) *dlnl = I->h_result[1];
  return PHYHIP_SUCCESS;
}

int phyhip_calculate_eigen_lnl_dlnl(int instance, double *l, double *outLnL, double *outDLnL)
{
  if (Group *G = get_group_nodrain(instance))
  { // (no call is entered on a shard here: the shards' own entry points or the group's launches below do that)
    if (std::isnan(*l)) return fail(PHYHIP_ERROR_FLOATING_POINT, "branch length is NaN");
    const Instance *I0 = G->sub[0];
    if (*l < I0->l_min) *l = I0->l_min;
    else if (*l > I0->l_max) *l = I0->l_max;
    return group_eigen_eval(G, *l, true, outLnL, outDLnL);
  }
  GET_INST_RES(I, instance);
  I_call.leave_untouched(); // (queues nothing by itself; flush() says so if it does)
  if (std::isnan(*l)) return fail(PHYHIP_ERROR_FLOATING_POINT, "branch length is NaN"); // src/lk.c:671
  if (*l < I->l_min) *l = I->l_min;                                                     // src/lk.c:673-674
  else if (*l > I->l_max) *l = I->l_max;
  if (I->co) return rank_eigen_eval(I, *l, true, outLnL, outDLnL);
  return eigen_eval(I, *l, true, outLnL, outDLnL);
}

int phyhip_calculate_eigen_lnl(int instance, double l, double *outLnL)
{
  if (Group *G = get_group_nodrain(instance)) return group_eigen_eval(G, l, false, outLnL, nullptr);
  GET_INST_RES(I, instance);
  I_call.leave_untouched();
  if (I->co) return rank_eigen_eval(I, l, false, outLnL, nullptr);
  return eigen_eval(I, l, false, outLnL, nullptr);
}

int phyhip_get_dot_prod(int instance, double *out)
{
  if (Group *G = get_group(instance))
    return group_each(G, [&](int id, long long lo, long long) { return phyhip_get_dot_prod(id, out + lo * G->C * G->S); });
  GET_INST(I, instance);
  int rc = flush_sync(I);
  if (rc) return rc;
  HIPCHK(hipMemcpy(out, I->d_dot, (size_t)I->P * I->C * I->S * sizeof(double), hipMemcpyDeviceToHost));
  return PHYHIP_SUCCESS;
}

} // extern "C"
