// phyhip_mixture.hip -- mixtures: class instances combined on the device, and classes on the category axis of one instance
// (libphyhip.so, gfx950 only; the units and what they share: phyhip_host.hpp)
#include "phyhip_host.hpp"

using namespace phyhip_host;

extern "C" {

static int mixture_lnl_impl(const int *instances, int count, const int *parent, const int *child, const int *pm,
                            const double *classProba, const double *rMatWeight, const double *eFrqWeight, double rMatWeightSum,
                            double eFrqWeightSum, double sumProbas, double *outLnL, const MixOut *mo)
{
  if (count < 1 || count > kMaxMixClasses) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "1..%d mixture classes", kMaxMixClasses);
  Instance *I0 = nullptr;
  MixParams q;
  memset(&q, 0, sizeof q);
  // An entry of the list is ONE class (an instance with one category) or a GROUP of classes on the category axis of one
  // instance (PHYHIP_FLAG_CLASS_AXIS: its C classes, in category order): a mixture of K classes is ceil(K / 4) traversal
  // launches instead of K.  The per-class tables (classProba, ...) run over the classes in list order.
  int nc = 0;
  for (int k = 0; k < count; ++k)
  {
    GET_INST(I, instances[k]);
    if (k == 0) I0 = I;
    const int Ck = I->class_axis ? I->C : 1;
    if ((!I->class_axis && I->C != 1) || I->P != I0->P || I->dev != I0->dev)
      return fail(PHYHIP_ERROR_OUT_OF_RANGE, "mixture entry %d: needs one category (or the class axis), the same pattern count and the same device", k);
    if (nc + Ck > kMaxMixClasses) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "more than %d mixture classes", kMaxMixClasses);
    int rc = check_partial_index(I, parent[k], true);
    if (rc) return rc;
    if ((rc = check_partial_index(I, child[k], true))) return rc;
    if (pm[k] < 0 || pm[k] >= I->nmat) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "matrix index %d", pm[k]);
    // the entry's own edge evaluation: leaves unscaled_site_lk_cat and fact_sum_scale (per class) in its device arrays, no host sync
    EdgeEval ee{parent[k], child[k], pm[k], I->class_axis ? nullptr : I->d_result, false, nullptr};
    if ((rc = flush(I, &ee))) return rc;
    if (I != I0)
    { // the combination runs on the first instance's stream, after every class stream
      HIPCHK(hipEventRecord(I->ev_sync, I->stream));
      HIPCHK(hipStreamWaitEvent(I0->stream, I->ev_sync, 0));
    }
    for (int c = 0; c < Ck; ++c, ++nc)
    {
      q.site_cat[nc] = I->d_site_cat + c; q.cat_stride[nc] = (unsigned short)Ck; q.fact[nc] = I->d_fact + (size_t)c * I->P;
      q.proba[nc] = classProba[nc]; q.r_w[nc] = rMatWeight[nc]; q.e_w[nc] = eFrqWeight[nc];
    }
  }
  q.count = nc; q.P = I0->P; q.r_sum = rMatWeightSum; q.e_sum = eFrqWeightSum; q.sum_probas = sumProbas;
  q.wght = I0->d_wght; q.site_lnl = I0->d_site_lnl;
  fill_mixture_invariant(I0, q);
  const int grid = (int)((I0->P + 255) / 256);
  mix_finish_setup(I0, q.fin, grid, 1, mo);
  hipLaunchKernelGGL(mixture_combine_kernel, dim3(grid), dim3(256), 0, I0->stream, q);
  HIPCHK(hipGetLastError());
  if (mo) return PHYHIP_SUCCESS;
  int rc = wait_result(I0);
  if (rc) return rc;
  *outLnL = I0->h_result[0];
  return PHYHIP_SUCCESS;
}

int phyhip_calculate_mixture_log_likelihood(const int *instances, int count, const int *parent, const int *child, const int *pm,
                                            const double *classProba, const double *rMatWeight, const double *eFrqWeight,
                                            double rMatWeightSum, double eFrqWeightSum, double sumProbas, double *outLnL)
{
  if (count >= 1 && get_group(instances[0]))
  { // class instances that are sharded instances: every shard combines its own patterns, ONE all-reduce of {warning, lnL}
    std::vector<Group *> Gs;
    int rc = mixture_groups(instances, count, Gs);
    if (rc) return rc;
    Group *G0 = Gs[0];
    rc = group_parallel(G0, [&](int g) -> int {
      int ids[kMaxMixClasses];
      for (int k = 0; k < count; ++k) ids[k] = Gs[k]->sub_id[g];
      double      *slot = shard_slot(G0->co->ctx[G0->ctx_of[g]], G0->k_of[g]);
      const MixOut mo{slot + 1, slot};
      return mixture_lnl_impl(ids, count, parent, child, pm, classProba, rMatWeight, eFrqWeight, rMatWeightSum, eFrqWeightSum, sumProbas,
                              nullptr, &mo);
    });
    if (rc) return rc;
    if ((rc = reduce_and_publish(*G0->co, 2, G0->sub[0]))) return rc;
    *outLnL        = G0->sub[0]->h_result[0];
    G0->last_warn  = *G0->sub[0]->h_warn;
    G0->warn_valid = true;
    return PHYHIP_SUCCESS;
  }
  return mixture_lnl_impl(instances, count, parent, child, pm, classProba, rMatWeight, eFrqWeight, rMatWeightSum, eFrqWeightSum, sumProbas,
                          outLnL, nullptr);
}

static int mixture_dlnl_impl(const int *instances, int count, const int *left, const int *right, double *l, const double *classProba,
                             const double *rMatWeight, const double *eFrqWeight, double rMatWeightSum, double eFrqWeightSum,
                             double sumProbas, double *outLnL, double *outDLnL, const MixOut *mo)
{
  if (count < 1 || count > kMaxMixClasses) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "1..%d mixture classes", kMaxMixClasses);
  if (std::isnan(*l)) return fail(PHYHIP_ERROR_FLOATING_POINT, "branch length is NaN");
  Instance *I0 = nullptr;
  std::vector<double> expl;
  std::vector<Instance *> cls;
  int nc = 0; // classes so far (an entry of the list is one class, or the C classes of a class-axis instance: mixture_lnl_impl)
  for (int k = 0; k < count; ++k)
  {
    GET_INST(I, instances[k]);
    if (k == 0)
    {
      I0 = I;
      if (*l < I->l_min) *l = I->l_min; // src/lk.c:672-673 (dLk clamps before diverting to MIXT_dLk)
      else if (*l > I->l_max) *l = I->l_max;
      expl.assign((size_t)kMaxMixClasses * 2 * I->S, 0.0);
    }
    const int Ck = I->class_axis ? I->C : 1;
    if ((!I->class_axis && I->C != 1) || I->P != I0->P || I->dev != I0->dev || I->S != I0->S)
      return fail(PHYHIP_ERROR_OUT_OF_RANGE, "mixture entry %d: needs one category (or the class axis), the same shape and the same device", k);
    if (nc + Ck > kMaxMixClasses) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "more than %d mixture classes", kMaxMixClasses);
    int rc = check_partial_index(I, left[k], true);
    if (rc) return rc;
    if ((rc = check_partial_index(I, right[k], true))) return rc;
    devirtualise(I, left[k]); devirtualise(I, right[k]); // (the combination kernel reads their scale vectors)
    if ((rc = flush(I, nullptr))) return rc; // queued partial updates write the scale vectors read below
    for (int c = 0; c < Ck; ++c, ++nc)
    { // src/mixt.c:3056-3114
      const double rr  = 1.0 * I->br_len_mult * I->h_rates[c];
      double       len = (*l) * rr;
      if (len < I->l_min) len = I->l_min;
      else if (len > I->l_max) len = I->l_max;
      for (int s = 0; s < I->S; ++s)
      {
        const double ev = I->h_eval[(size_t)(I->class_axis ? c : 0) * I->S + s], ex = exp(ev * len);
        expl[(size_t)nc * 2 * I->S + 2 * s]     = ex;
        expl[(size_t)nc * 2 * I->S + 2 * s + 1] = ex * ev * rr;
      }
    }
    cls.push_back(I);
  }
  expl.resize((size_t)nc * 2 * I0->S);
  // expl pairs of all classes: staged copy into the first instance's matrix scratch area
  void        *st = nullptr;
  const size_t eb = expl.size() * sizeof(double);
  int rc = I0->ring.alloc(eb, I0->stream, &st);
  if (rc) return rc;
  memcpy(st, expl.data(), eb);
  HIPCHK(hipMemcpyAsync(I0->d_mixexpl, st, eb, hipMemcpyHostToDevice, I0->stream));
  auto launch = [&](auto s_) -> int {
    constexpr int S_ = decltype(s_)::value;
    MixDlkParams<S_> q;
    memset(&q, 0, sizeof q);
    int n = 0;
    for (int k = 0; k < count; ++k)
    {
      Instance *I = cls[k];
      if (I != I0)
      {
        HIPCHK(hipEventRecord(I->ev_sync, I->stream));
        HIPCHK(hipStreamWaitEvent(I0->stream, I->ev_sync, 0));
      }
      const int Ck = I->class_axis ? I->C : 1;
      for (int c = 0; c < Ck; ++c, ++n)
      {
        q.dot[n] = I->d_dot + (size_t)c * S_; q.dot_stride[n] = (unsigned short)(Ck * S_);
        q.scale_l[n] = left[k] < I->tips ? nullptr : I->d_scales + (size_t)(left[k] - I->tips) * scale_elems(I) + (size_t)c * I->Ppad;
        q.scale_r[n] = right[k] < I->tips ? nullptr : I->d_scales + (size_t)(right[k] - I->tips) * scale_elems(I) + (size_t)c * I->Ppad;
        q.proba[n] = classProba[n]; q.r_w[n] = rMatWeight[n]; q.e_w[n] = eFrqWeight[n];
      }
    }
    q.count = n; q.P = I0->P; q.r_sum = rMatWeightSum; q.e_sum = eFrqWeightSum; q.sum_probas = sumProbas;
    q.expl = I0->d_mixexpl; q.wght = I0->d_wght;
    fill_mixture_invariant(I0, q);
    const int grid = (int)((I0->P + 255) / 256);
    mix_finish_setup(I0, q.fin, grid, 2, mo);
    hipLaunchKernelGGL((mixture_dlk_kernel<S_>), dim3(grid), dim3(256), 0, I0->stream, q);
    HIPCHK(hipGetLastError());
    return 0;
  };
  if (I0->S == 4) rc = launch(std::integral_constant<int, 4>());
  else if (I0->S == 20) rc = launch(std::integral_constant<int, 20>());
  else return fail(PHYHIP_ERROR_NO_IMPLEMENTATION, "mixtures: 4 or 20 states");
  if (rc) return rc;
  if (mo) return PHYHIP_SUCCESS;
  if ((rc = wait_result(I0))) return rc;
  *outLnL = I0->h_result[0];
  if (outDLnL) *outDLnL = I0->h_result[1];
  return PHYHIP_SUCCESS;
}

int phyhip_calculate_mixture_eigen_lnl_dlnl(const int *instances, int count, const int *left, const int *right, double *l,
                                            const double *classProba, const double *rMatWeight, const double *eFrqWeight,
                                            double rMatWeightSum, double eFrqWeightSum, double sumProbas, double *outLnL,
                                            double *outDLnL)
{
  if (count >= 1 && get_group(instances[0]))
  { // sharded class instances: ONE all-reduce of {warning, lnL, dlnL}
    std::vector<Group *> Gs;
    int rc = mixture_groups(instances, count, Gs);
    if (rc) return rc;
    Group *G0 = Gs[0];
    std::vector<double> ls(G0->sub.size(), *l); // every shard clamps its own copy the same way
    rc = group_parallel(G0, [&](int g) -> int {
      int ids[kMaxMixClasses];
      for (int k = 0; k < count; ++k) ids[k] = Gs[k]->sub_id[g];
      double      *slot = shard_slot(G0->co->ctx[G0->ctx_of[g]], G0->k_of[g]);
      const MixOut mo{slot + 1, slot};
      return mixture_dlnl_impl(ids, count, left, right, &ls[g], classProba, rMatWeight, eFrqWeight, rMatWeightSum, eFrqWeightSum,
                               sumProbas, nullptr, nullptr, &mo);
    });
    if (rc) return rc;
    *l = ls[0];
    if ((rc = reduce_and_publish(*G0->co, 3, G0->sub[0]))) return rc;
    *outLnL = G0->sub[0]->h_result[0];
    if (outDLnL) *outDLnL = G0->sub[0]->h_result[1];
    G0->last_warn  = *G0->sub[0]->h_warn;
    G0->warn_valid = true;
    return PHYHIP_SUCCESS;
  }
  return mixture_dlnl_impl(instances, count, left, right, l, classProba, rMatWeight, eFrqWeight, rMatWeightSum, eFrqWeightSum, sumProbas,
                           outLnL, outDLnL, nullptr);
}

int phyhip_set_mixture_invariant_sites(int instance, int invar_model, double pinvar, const short *invar, const double *piInvariantClass)
{
  if (Group *G = get_group(instance))
    return group_each(G, [&](int id, long long lo, long long) {
      return phyhip_set_mixture_invariant_sites(id, invar_model, pinvar, invar ? invar + lo : nullptr, piInvariantClass);
    });
  GET_INST(I, instance);
  if (invar_model && (!invar || !piInvariantClass)) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "invar_model set but invar / frequencies missing");
  const bool same_sites = !invar_model || (I->h_invar_set && !memcmp(I->h_invar.data(), invar, I->P * sizeof(short)));
  if (!same_sites)
  { // the per-pattern table shares the slot of phyhip_set_invariant_sites: a mixture's class instances run without +I
    int rc = flush_sync(I);
    if (rc) return rc;
    HIPCHK(hipMemcpy(I->d_invar, invar, I->P * sizeof(short), hipMemcpyHostToDevice));
    I->h_invar.assign(invar, invar + I->P);
    I->h_invar_set = true;
  }
  I->mix_invar_model = invar_model ? 1 : 0;
  I->mix_pinvar      = pinvar;
  if (invar_model) for (int s = 0; s < I->S; ++s) I->mix_pi_inv[s] = piInvariantClass[s];
  return PHYHIP_SUCCESS;
}

// ---- mixtures on the class axis of ONE instance --------------------------------------------------------------------

static int class_mixture_lnl_impl(int instance, int parent, int child, int pm, const double *classProba, const double *rMatWeight,
                                  const double *eFrqWeight, double rMatWeightSum, double eFrqWeightSum, double sumProbas,
                                  double *outLnL, const MixOut *mo)
{
  GET_INST(I, instance);
  if (!I->class_axis) return fail(PHYHIP_ERROR_GENERAL, "instance %d was not created with PHYHIP_FLAG_CLASS_AXIS", instance);
  int rc = check_partial_index(I, parent, true);
  if (rc) return rc;
  if ((rc = check_partial_index(I, child, true))) return rc;
  if (pm < 0 || pm >= I->nmat) return fail(PHYHIP_ERROR_OUT_OF_RANGE, "matrix index %d", pm);
  // ONE traversal launch for all classes (queued updates + the per-class edge likelihoods and scale exponents) ...
  EdgeEval ee{parent, child, pm, nullptr, false, nullptr};
  if ((rc = flush(I, &ee))) return rc;
  // ... and the site loop of MIXT_Lk (src/mixt.c:1027-1135) over them
  MixParams q;
  memset(&q, 0, sizeof q);
  for (int k = 0; k < I->C; ++k)
  {
    q.site_cat[k] = I->d_site_cat + k; q.fact[k] = I->d_fact + (size_t)k * I->P; q.cat_stride[k] = (unsigned short)I->C;
    q.proba[k] = classProba[k]; q.r_w[k] = rMatWeight[k]; q.e_w[k] = eFrqWeight[k];
  }
  q.count = I->C; q.P = I->P; q.r_sum = rMatWeightSum; q.e_sum = eFrqWeightSum; q.sum_probas = sumProbas;
  q.wght = I->d_wght; q.site_lnl = I->d_site_lnl;
  fill_mixture_invariant(I, q);
  const int grid = (int)((I->P + 255) / 256);
  mix_finish_setup(I, q.fin, grid, 1, mo);
  hipLaunchKernelGGL(mixture_combine_kernel, dim3(grid), dim3(256), 0, I->stream, q);
  HIPCHK(hipGetLastError());
  if (mo) return PHYHIP_SUCCESS;
  if ((rc = wait_result(I))) return rc;
  *outLnL = I->h_result[0];
  return PHYHIP_SUCCESS;
}

int phyhip_calculate_class_mixture_log_likelihood(int instance, int parent, int child, int pm, const double *classProba,
                                                  const double *rMatWeight, const double *eFrqWeight, double rMatWeightSum,
                                                  double eFrqWeightSum, double sumProbas, double *outLnL)
{
  if (Group *G = get_group(instance))
  {
    int rc = group_parallel(G, [&](int g) -> int {
      double      *slot = shard_slot(G->co->ctx[G->ctx_of[g]], G->k_of[g]);
      const MixOut mo{slot + 1, slot};
      return class_mixture_lnl_impl(G->sub_id[g], parent, child, pm, classProba, rMatWeight, eFrqWeight, rMatWeightSum, eFrqWeightSum,
                                    sumProbas, nullptr, &mo);
    });
    if (rc) return rc;
    if ((rc = reduce_and_publish(*G->co, 2, G->sub[0]))) return rc;
    *outLnL       = G->sub[0]->h_result[0];
    G->last_warn  = *G->sub[0]->h_warn;
    G->warn_valid = true;
    return PHYHIP_SUCCESS;
  }
  return class_mixture_lnl_impl(instance, parent, child, pm, classProba, rMatWeight, eFrqWeight, rMatWeightSum, eFrqWeightSum, sumProbas,
                                outLnL, nullptr);
}

static int class_mixture_dlnl_impl(int instance, int left, int right, double *l, const double *classProba, const double *rMatWeight,
                                   const double *eFrqWeight, double rMatWeightSum, double eFrqWeightSum, double sumProbas,
                                   double *outLnL, double *outDLnL, const MixOut *mo)
{
  GET_INST(I, instance);
  if (!I->class_axis) return fail(PHYHIP_ERROR_GENERAL, "instance %d was not created with PHYHIP_FLAG_CLASS_AXIS", instance);
  if (std::isnan(*l)) return fail(PHYHIP_ERROR_FLOATING_POINT, "branch length is NaN");
  if (*l < I->l_min) *l = I->l_min; // src/lk.c:672-673 (dLk clamps before diverting to MIXT_dLk)
  else if (*l > I->l_max) *l = I->l_max;
  int rc = check_partial_index(I, left, true);
  if (rc) return rc;
  if ((rc = check_partial_index(I, right, true))) return rc;
  if ((rc = flush(I, nullptr))) return rc; // queued partial updates write the scale vectors read below
  std::vector<double> expl((size_t)I->C * 2 * I->S);
  for (int k = 0; k < I->C; ++k)
  { // src/mixt.c:3056-3114
    const double rr  = 1.0 * I->br_len_mult * I->h_rates[k];
    double       len = (*l) * rr;
    if (len < I->l_min) len = I->l_min;
    else if (len > I->l_max) len = I->l_max;
    for (int s = 0; s < I->S; ++s)
    {
      const double ev = I->h_eval[(size_t)k * I->S + s], ex = exp(ev * len);
      expl[(size_t)k * 2 * I->S + 2 * s]     = ex;
      expl[(size_t)k * 2 * I->S + 2 * s + 1] = ex * ev * rr;
    }
  }
  void        *st = nullptr;
  const size_t eb = expl.size() * sizeof(double);
  if ((rc = I->ring.alloc(eb, I->stream, &st))) return rc;
  memcpy(st, expl.data(), eb);
  HIPCHK(hipMemcpyAsync(I->d_mixexpl, st, eb, hipMemcpyHostToDevice, I->stream));
  auto launch = [&](auto s_) {
    constexpr int S_ = decltype(s_)::value;
    MixDlkParams<S_> q;
    memset(&q, 0, sizeof q);
    for (int k = 0; k < I->C; ++k)
    {
      q.dot[k]     = I->d_dot + (size_t)k * I->S; q.dot_stride[k] = (unsigned short)(I->C * I->S);
      q.scale_l[k] = left < I->tips ? nullptr : I->d_scales + (size_t)(left - I->tips) * scale_elems(I) + (size_t)k * I->Ppad;
      q.scale_r[k] = right < I->tips ? nullptr : I->d_scales + (size_t)(right - I->tips) * scale_elems(I) + (size_t)k * I->Ppad;
      q.proba[k] = classProba[k]; q.r_w[k] = rMatWeight[k]; q.e_w[k] = eFrqWeight[k];
    }
    q.count = I->C; q.P = I->P; q.r_sum = rMatWeightSum; q.e_sum = eFrqWeightSum; q.sum_probas = sumProbas;
    q.expl = I->d_mixexpl; q.wght = I->d_wght;
    fill_mixture_invariant(I, q);
    const int grid = (int)((I->P + 255) / 256);
    mix_finish_setup(I, q.fin, grid, 2, mo);
    hipLaunchKernelGGL((mixture_dlk_kernel<S_>), dim3(grid), dim3(256), 0, I->stream, q);
  };
  if (I->S == 4) launch(std::integral_constant<int, 4>());
  else launch(std::integral_constant<int, 20>());
  HIPCHK(hipGetLastError());
  if (mo) return PHYHIP_SUCCESS;
  if ((rc = wait_result(I))) return rc;
  *outLnL = I->h_result[0];
  if (outDLnL) *outDLnL = I->h_result[1];
  return PHYHIP_SUCCESS;
}

int phyhip_calculate_class_mixture_eigen_lnl_dlnl(int instance, int left, int right, double *l, const double *classProba,
                                                  const double *rMatWeight, const double *eFrqWeight, double rMatWeightSum,
                                                  double eFrqWeightSum, double sumProbas, double *outLnL, double *outDLnL)
{
  if (Group *G = get_group(instance))
  {
    std::vector<double> ls(G->sub.size(), *l);
    int rc = group_parallel(G, [&](int g) -> int {
      double      *slot = shard_slot(G->co->ctx[G->ctx_of[g]], G->k_of[g]);
      const MixOut mo{slot + 1, slot};
      return class_mixture_dlnl_impl(G->sub_id[g], left, right, &ls[g], classProba, rMatWeight, eFrqWeight, rMatWeightSum, eFrqWeightSum,
                                     sumProbas, nullptr, nullptr, &mo);
    });
    if (rc) return rc;
    *l = ls[0];
    if ((rc = reduce_and_publish(*G->co, 3, G->sub[0]))) return rc;
    *outLnL = G->sub[0]->h_result[0];
    if (outDLnL) *outDLnL = G->sub[0]->h_result[1];
    G->last_warn  = *G->sub[0]->h_warn;
    G->warn_valid = true;
    return PHYHIP_SUCCESS;
  }
  return class_mixture_dlnl_impl(instance, left, right, l, classProba, rMatWeight, eFrqWeight, rMatWeightSum, eFrqWeightSum, sumProbas,
                                 outLnL, outDLnL, nullptr);
}

} // extern "C"
