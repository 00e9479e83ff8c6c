/*
 * glue_driver.c -- TEST INFRASTRUCTURE / integration demonstrator.  Runs the REAL, unmodified PhyML tree search
 * (spr.c, optimiz.c, ... compiled from /root/reference/src into oracle/_ref/libphyml_ref.so) with its likelihood
 * surface served by the phyhip engine (libphyhip.so, include/phyhip.h) on the GPU.
 *
 * The reference delegates this path only behind `#ifdef BEAGLE` (src/lk.c:1300-1302, 585-587, 2360), which needs
 * BEAGLE's headers to compile; instead of editing sources this driver uses the same mechanism as
 * oracle/trace_driver.c: the reference objects are position-independent, so every call to Lk / dLk /
 * Update_Partial_Lk / Update_PMat_At_Given_Edge / Update_Eigen_Lr -- from spr.c and optimiz.c as well as from inside
 * lk.c -- goes through the PLT and binds to the definitions below.  They are the glue of INTEGRATION.md section 2
 * in executable form:
 *
 *   Update_PMat_At_Given_Edge  original (host PMat into b->Pij_rr, src/lk.c:2238)  + upload (beagleSetTransitionMatrix
 *                              call site, src/lk.c:2360); GLUE_DEVICE_PMAT=1: device matrices from the eigen system
 *   Update_Partial_Lk          gates of src/lk.c:1285-1297, buffers resolved with the reference's Set_All_Partial_Lk,
 *                              one queued phyhip_update_partials (update_beagle_partials, src/beagle_utils.c:214-262)
 *   Lk                         control flow of src/lk.c:443-606 (model refresh, matrix refresh, traversals through the
 *                              reference's own Post/Pre_Order_Lk, root edge) + the device edge evaluation in place of the
 *                              site loop (calc_edgelks_beagle, src/lk.c:585-587)
 *   Update_Eigen_Lr, dLk       device eigen-basis products and derivative (src/lk.c:1038-1114, 655-753)
 *
 * modes (env GLUE_MODE):
 *   device (default)  the search is driven ONLY by device results; host partial buffers are never written
 *   check             the original runs as well (host buffers stay valid) and every scalar the device returns is
 *                     compared with the reference's own, call by call, over the whole search; the search follows the
 *                     reference's values.  Prints the worst relative differences.
 *   host              every call is forwarded to the reference untouched (no device): the CPU-only run of the same
 *                     command, same output format -- how the expected values of tests/golden/search_expected.json are made
 *
 * usage: phyml_glue_driver [--gtr-rr a,..,f] -- <phyml command line>
 * prints: GLUE_DRIVER {json}   (final lnL, tree, call counts, worst per-call differences in check mode)
 */
#define _GNU_SOURCE
#include <dlfcn.h>

#define main ref_driver_main_unused
#include "ref_driver.c"
#undef main

#include "../include/phyhip.h"

static int g_inst = -1, g_check = 0, g_device_pmat = 0, g_host = 0;
static t_tree *g_tree = NULL;
static long    g_n_lk = 0, g_n_lk_full = 0, g_n_upd = 0, g_n_dlk = 0, g_n_pmat = 0, g_n_eig = 0;
static double  g_worst_lnl = 0.0, g_worst_dlnl = 0.0;

static void die(const char *what)
{
  fprintf(stderr, "\n== glue_driver: %s: %s\n", what, phyhip_get_last_error());
  exit(5); /* the reference's convention: message, then Exit (src/beagle_utils.c:246-249) */
}
#define OK(call) do { if ((call) < 0) die(#call); } while (0)

/* pointer -> device index tables */
#define MAXID 8192
static const void *g_bufptr[MAXID]; static int g_nbuf = 0, g_bufcap = 0;
static const void *g_matptr[MAXID]; static int g_nmat = 0, g_matcap = 0;

static int buf_id(const t_tree *tree, const void *p_lk)
{
  for (int i = 0; i < g_nbuf; ++i) if (g_bufptr[i] == p_lk) return tree->n_otu + i;
  if (g_nbuf == g_bufcap) { fprintf(stderr, "glue_driver: more partial buffers than the instance holds\n"); exit(5); }
  g_bufptr[g_nbuf] = p_lk;
  return tree->n_otu + g_nbuf++;
}
static int mat_id(const void *pij)
{
  for (int i = 0; i < g_nmat; ++i) if (g_matptr[i] == pij) return i;
  if (g_nmat == g_matcap) { fprintf(stderr, "glue_driver: more matrices than the instance holds\n"); exit(5); }
  g_matptr[g_nmat] = pij;
  return g_nmat++;
}
static void edge_sides(const t_tree *tree, const t_edge *b, int *left, int *right)
{ /* src/lk.c:605-606 */
  *left  = buf_id(tree, b->p_lk_left);
  *right = b->rght->tax ? b->rght->num : buf_id(tree, b->p_lk_rght);
}

static void push_model(t_tree *tree)
{ /* update_beagle_ras / _efrqs / _eigen, src/beagle_utils.c:273-395 */
  t_mod *m = tree->mod;
  OK(phyhip_set_category_rates(g_inst, m->ras->gamma_rr->v));
  OK(phyhip_set_category_weights(g_inst, 0, m->ras->gamma_r_proba->v));
  OK(phyhip_set_state_frequencies(g_inst, 0, m->e_frq->pi->v));
  OK(phyhip_set_eigen_decomposition(g_inst, 0, m->eigen->r_e_vect, m->eigen->l_e_vect, m->eigen->e_val));
  OK(phyhip_set_phyml_options(g_inst, m->l_min, m->l_max, m->br_len_mult->v, tree->apply_lk_scaling));
  OK(phyhip_set_invariant_sites(g_inst, m->ras->invar, m->ras->pinvar->v, tree->data->invar));
}

static void ensure_instance(t_tree *tree)
{ /* create_beagle_instance, src/beagle_utils.c:97-190: after Make_Tree_For_Lk (src/main.c:235) */
  if (g_inst >= 0)
  {
    if (tree != g_tree) { fprintf(stderr, "glue_driver: a second tree object reached the likelihood surface\n"); exit(5); }
    return;
  }
  if (tree->is_mixt_tree || tree->n_root || tree->mod->gamma_mgf_bl == YES || tree->mod->log_l == YES || tree->mod->use_m4mod)
  { fprintf(stderr, "glue_driver: unsupported tree kind (mixture / rooted / mgf / log_l / m4)\n"); exit(5); }
  const int n = tree->n_otu, P = tree->data->n_pattern, S = tree->mod->ns, C = tree->mod->ras->n_catg;
  g_bufcap = 3 * n - 2;  /* internal edge sides + both sides of the two spare SPR edges (src/make.c:96-104) */
  g_matcap = 2 * n - 1;
  g_inst = phyhip_create_instance(n, n + g_bufcap, 0, S, P, 1, g_matcap, C, 0, NULL, 0, 0, 0, NULL);
  if (g_inst < 0) die("phyhip_create_instance");
  g_tree = tree;
  OK(phyhip_set_pattern_weights(g_inst, tree->data->wght));
  for (int t = 0; t < n; ++t) OK(phyhip_set_tip_partials(g_inst, tree->a_nodes[t]->num, tree->a_nodes[t]->b[0]->p_lk_tip_r));
  push_model(tree);
}

static void track(double *worst, double dev, double ref, double floor_)
{
  const double d = fabs(dev - ref) / fmax(floor_, fabs(ref));
  if (d > *worst) *worst = d;
}

/* ---- the interposed surface ---------------------------------------------------------------------------------- */

void Update_PMat_At_Given_Edge(t_edge *b_fcus, t_tree *tree)
{
  static void (*real)(t_edge *, t_tree *) = NULL;
  if (!real) real = (void (*)(t_edge *, t_tree *))dlsym(RTLD_NEXT, "Update_PMat_At_Given_Edge");
  ++g_n_pmat;
  if (g_host) { real(b_fcus, tree); return; }
  ensure_instance(tree);
  if (b_fcus->has_zero_br_len == YES) { fprintf(stderr, "glue_driver: zero-length edge flag not supported\n"); exit(5); }
  const int m = mat_id(b_fcus->Pij_rr);
  if (g_device_pmat && !g_check)
  { /* src/lk.c:2344: matrices built on the device from (U, lambda, U^-1, length) */
    const double len = b_fcus->l->v;
    OK(phyhip_update_transition_matrices(g_inst, 0, &m, NULL, NULL, &len, 1));
    return;
  }
  real(b_fcus, tree); /* host PMat (src/models.c:257-373) into b->Pij_rr ... */
  OK(phyhip_set_transition_matrix(g_inst, m, b_fcus->Pij_rr, -1.0)); /* ... and the upload of src/lk.c:2360 */
}

void Update_Partial_Lk(t_tree *tree, t_edge *b, t_node *d)
{
  static void (*real)(t_tree *, t_edge *, t_node *) = NULL;
  if (!real) real = (void (*)(t_tree *, t_edge *, t_node *))dlsym(RTLD_NEXT, "Update_Partial_Lk");
  /* src/lk.c:1285-1297 */
  if (b->left == d && b->update_partial_lk_left == NO) return;
  if (b->rght == d && b->update_partial_lk_rght == NO) return;
  if (d->tax) return;
  ++g_n_upd;
  if (g_host) { real(tree, b, d); return; }
  ensure_instance(tree);
  t_node *n_v1 = NULL, *n_v2 = NULL;
  phydbl *p_lk = NULL, *p_lk_v1 = NULL, *p_lk_v2 = NULL, *Pij1 = NULL, *Pij2 = NULL, *tPij1 = NULL, *tPij2 = NULL;
  int    *sum_scale = NULL, *sum_scale_v1 = NULL, *sum_scale_v2 = NULL, *p_lk_loc = NULL;
  Set_All_Partial_Lk(&n_v1, &n_v2, &p_lk, &sum_scale, &p_lk_loc, &Pij1, &tPij1, &p_lk_v1, &sum_scale_v1, &Pij2, &tPij2,
                     &p_lk_v2, &sum_scale_v2, d, b, tree);
  phyhip_operation op;
  op.destinationPartials = buf_id(tree, p_lk);
  op.destinationScaleWrite = op.destinationScaleRead = PHYHIP_OP_NONE;
  op.child1Partials = n_v1->tax ? n_v1->num : buf_id(tree, p_lk_v1);
  op.child1TransitionMatrix = mat_id(Pij1);
  op.child2Partials = n_v2->tax ? n_v2->num : buf_id(tree, p_lk_v2);
  op.child2TransitionMatrix = mat_id(Pij2);
  OK(phyhip_update_partials(g_inst, &op, 1, PHYHIP_OP_NONE));
  if (g_check) real(tree, b, d);
}

void Update_Eigen_Lr(t_edge *b, t_tree *tree)
{
  static void (*real)(t_edge *, t_tree *) = NULL;
  if (!real) real = (void (*)(t_edge *, t_tree *))dlsym(RTLD_NEXT, "Update_Eigen_Lr");
  ++g_n_eig;
  if (g_host) { real(b, tree); return; }
  ensure_instance(tree);
  int l, r;
  edge_sides(tree, b, &l, &r);
  OK(phyhip_update_eigen_lr(g_inst, l, r));
  if (g_check) real(b, tree);
}

static double device_edge_value(t_tree *tree, const t_edge *b)
{
  double lnl = 0.0;
  if (tree->use_eigen_lr == YES) OK(phyhip_calculate_eigen_lnl(g_inst, b->l->v, &lnl)); /* src/lk.c:592-603 */
  else
  {
    int l, r, pm = mat_id(b->Pij_rr), zero = 0;
    edge_sides(tree, b, &l, &r);
    OK(phyhip_calculate_edge_log_likelihoods(g_inst, &l, &r, &pm, NULL, NULL, &zero, &zero, NULL, 1, &lnl, NULL, NULL));
  }
  return lnl;
}

phydbl Lk(t_edge *b, t_tree *tree)
{
  static phydbl (*real)(t_edge *, t_tree *) = NULL;
  if (!real) real = (phydbl (*)(t_edge *, t_tree *))dlsym(RTLD_NEXT, "Lk");
  ++g_n_lk;
  if (!b) ++g_n_lk_full;
  if (g_host) return real(b, tree);
  if (g_check)
  {
    const phydbl ref = real(b, tree); /* drives the device through the nested surface calls as well */
    ensure_instance(tree);
    if (!b) push_model(tree);
    const t_edge *e = b ? b : tree->a_nodes[tree->tip_root]->b[0];
    track(&g_worst_lnl, device_edge_value(tree, e), ref, 1.0);
    return ref;
  }
  /* ---- control flow of src/lk.c:443-606 (unrooted, no mixture), device evaluation instead of the site loop ---- */
  tree->numerical_warning = NO;
  if (!b)
  { /* src/lk.c:489-495 */
    Update_Boundaries(tree->mod);
    Update_RAS(tree->mod);
    Update_Efrq(tree->mod);
    Update_Eigen(tree->mod);
  }
  ensure_instance(tree);
  if (!b) push_model(tree);
  if (tree->mod->s_opt->skip_tree_traversal == NO)
  {
    if (!b)
    { /* src/lk.c:500-512, 560-565 */
      for (int br = 0; br < 2 * tree->n_otu - 3; ++br) Update_PMat_At_Given_Edge(tree->a_edges[br], tree);
      Post_Order_Lk(tree->a_nodes[tree->tip_root], tree->a_nodes[tree->tip_root]->v[0], tree);
      if (tree->both_sides == YES) Pre_Order_Lk(tree->a_nodes[tree->tip_root], tree->a_nodes[tree->tip_root]->v[0], tree);
    }
    else if (tree->use_eigen_lr == NO) Update_PMat_At_Given_Edge(b, tree); /* src/lk.c:513-528 */
  }
  if (!b) b = tree->a_nodes[tree->tip_root]->b[0]; /* src/lk.c:569-580 */
  tree->c_lnL = .0;
  tree->sum_min_sum_scale = .0;
  if (tree->update_eigen_lr == YES) Update_Eigen_Lr(b, tree); /* src/lk.c:590 */
  tree->c_lnL = device_edge_value(tree, b);
  {
    int w = 0;
    OK(phyhip_get_numerical_warning(g_inst, &w));
    if (w) tree->numerical_warning = YES;
  }
  return tree->c_lnL;
}

phydbl dLk(phydbl *l, t_edge *b, t_tree *tree)
{
  static phydbl (*real)(phydbl *, t_edge *, t_tree *) = NULL;
  if (!real) real = (phydbl (*)(phydbl *, t_edge *, t_tree *))dlsym(RTLD_NEXT, "dLk");
  ++g_n_dlk;
  if (g_host) return real(l, b, tree);
  ensure_instance(tree);
  if (g_check)
  {
    double x = *l, lnl = 0.0, dlnl = 0.0;
    const phydbl ref = real(l, b, tree); /* its Update_Eigen_Lr (if any) reaches the device through the wrapper */
    OK(phyhip_calculate_eigen_lnl_dlnl(g_inst, &x, &lnl, &dlnl));
    track(&g_worst_lnl, lnl, ref, 1.0);
    track(&g_worst_dlnl, dlnl, tree->c_dlnL, 1.0);
    return ref;
  }
  /* src/lk.c:655-753 */
  tree->numerical_warning = NO;
  if (tree->update_eigen_lr == YES) Update_Eigen_Lr(b, tree);
  double lnl = 0.0, dlnl = 0.0;
  OK(phyhip_calculate_eigen_lnl_dlnl(g_inst, l, &lnl, &dlnl)); /* clamps *l like src/lk.c:672-673 */
  tree->c_dlnL = dlnl;
  tree->c_lnL  = lnl;
  return tree->c_lnL;
}

/* ---------------------------------------------------------------------------------------------------------------- */

int main(int argc, char **argv)
{
  drv_opt o; memset(&o, 0, sizeof o);
  o.both_sides = 0;
  int i = 1;
  for (; i < argc; ++i)
  {
    if (!strcmp(argv[i], "--")) { ++i; break; }
    else if (!strcmp(argv[i], "--gtr-rr") && i + 1 < argc)
    {
      o.have_rr = 1;
      if (sscanf(argv[++i], "%lf,%lf,%lf,%lf,%lf,%lf", o.rr, o.rr + 1, o.rr + 2, o.rr + 3, o.rr + 4, o.rr + 5) != 6)
      { fprintf(stderr, "bad --gtr-rr\n"); return 2; }
    }
    else { fprintf(stderr, "usage: %s [--gtr-rr a,..,f] -- <phyml args>\n", argv[0]); return 2; }
  }
  int    pargc = argc - i + 1;
  char **pargv = malloc(sizeof(char *) * (pargc + 1));
  pargv[0] = argv[0];
  for (int k = 1; k < pargc; ++k) pargv[k] = argv[i + k - 1];
  pargv[pargc] = NULL;
  const char *mode = getenv("GLUE_MODE");
  g_check = mode && !strcmp(mode, "check");
  g_host  = mode && !strcmp(mode, "host");
  g_device_pmat = getenv("GLUE_DEVICE_PMAT") && atoi(getenv("GLUE_DEVICE_PMAT"));

  const double t0 = now_s();
  t_tree *tree = setup_tree(pargc, pargv, &o); /* src/main.c:73-258, first Lk(NULL) included */
  const double lnl_init = tree->c_lnL;
  /* src/main.c:262-275 */
  if (tree->mod->s_opt->opt_topo) Global_Spr_Search(tree);
  else if (tree->mod->s_opt->opt_subst_param || tree->mod->s_opt->opt_bl_one_by_one) Round_Optimize(tree, ROUND_MAX);
  /* src/main.c:281-282 */
  Set_Both_Sides(YES, tree);
  Lk(NULL, tree);
  const double dt = now_s() - t0;
  char *nwk = Write_Tree(tree);
  printf("\nGLUE_DRIVER {\"mode\": \"%s\", \"device_pmat\": %d, \"lnL_init\": %.17g, \"lnL_final\": %.17g, \"seconds\": %.3f, "
         "\"calls\": {\"Lk\": %ld, \"Lk_full\": %ld, \"Update_Partial_Lk\": %ld, \"dLk\": %ld, \"Update_PMat\": %ld, "
         "\"Update_Eigen_Lr\": %ld}, \"worst_rel_lnL\": %.3g, \"worst_rel_dlnL\": %.3g, \"buffers\": %d, \"matrices\": %d, "
         "\"tree\": \"%s\"}\n",
         g_host ? "host" : (g_check ? "check" : "device"), g_device_pmat, lnl_init, tree->c_lnL, dt, g_n_lk, g_n_lk_full, g_n_upd, g_n_dlk, g_n_pmat,
         g_n_eig, g_worst_lnl, g_worst_dlnl, g_nbuf, g_nmat, nwk ? nwk : "");
  fflush(stdout);
  if (g_inst >= 0) OK(phyhip_finalize_instance(g_inst));
  _exit(0);
}
