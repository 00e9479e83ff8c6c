/*
 * mixt_driver.c -- TEST INFRASTRUCTURE.  Golden vectors for the mixture path: runs the REAL
 * reference in XML mode (src/io.c:5033 PhyML_XML, e.g. the LG4X mixture of examples/lg4x) with MIXT_Lk interposed (same
 * mechanism as trace_driver.c: the reference objects are linked as a shared object, calls to global functions go through
 * the PLT, no source is modified), and at the chosen call dumps what MIXT_Lk (src/mixt.c:730-1160) combined:
 *
 *   per class tree k (each has n_catg = 1, its own rate matrix / frequencies / eigen system, src/mixt.c:2603-2640):
 *     model block, the rate multiplier of the class (mixt_tree->mod->ras->gamma_rr[parent_class_number], applied to
 *     branch lengths in Update_PMat_At_Given_Edge, src/lk.c:2298), the factors of its mixture coefficient
 *     (gamma_r_proba, r_mat_weight, e_frq_weight and their normalisers, src/mixt.c:1048-1053), the per-site class
 *     likelihood unscaled_site_lk_cat and the scale exponent sum at the evaluation edge
 *   mixture level: pattern weights, tips, edges, per-site log-likelihoods c_lnL_sorted, lnL
 *
 * In "dlk" mode the dump is taken at a MIXT_dLk call (src/mixt.c:2962-3340) instead: the branch length it was given, the
 * returned lnL and dlnL, and per class the eigen-basis products dot_prod (src/lk.c:1038-1114) it combined.
 *
 * usage: phyml_mixt_driver <out.phyg> <call index> [lk|dlk] -- --xml=<file>
 */
#define _GNU_SOURCE
#include <dlfcn.h>

#define main ref_driver_main_unused
#include "ref_driver.c"
#undef main
#include "mixt.h"

static int         g_want = 0, g_calls = 0, g_dlk_mode = 0, g_dlk_calls = 0;
static double      g_dlk_l = 0.0, g_dlk_dlnl = 0.0;
static const char *g_path = NULL;

static void dump_mixture(t_edge *mixt_b_in, t_tree *mixt_tree, phydbl lnl)
{
  if (mixt_tree->n_root || mixt_tree->next_mixt) { fprintf(stderr, "mixt_driver: rooted or multi-partition input not supported\n"); exit(4); }
  if (mixt_tree->mod->ras->invar == YES) { fprintf(stderr, "mixt_driver: +I mixtures not supported\n"); exit(4); }
  t_edge *mixt_b = mixt_b_in ? mixt_b_in : mixt_tree->a_nodes[0]->b[0]; /* src/mixt.c:889 */
  const int n = mixt_tree->n_otu, P = mixt_tree->data->n_pattern, NE = 2 * n - 3;
  int K = 0;
  for (t_tree *t = mixt_tree->next; t && t->is_mixt_tree == NO; t = t->next) ++K;
  const int S = mixt_tree->next->mod->ns;
  g_out = fopen(g_path, "wb");
  if (!g_out) { perror(g_path); exit(2); }
  fwrite("PHYG", 1, 4, g_out);
  rec_scalar("n_otu", n); rec_scalar("n_pattern", P); rec_scalar("ns", S); rec_scalar("n_classes", K);
  rec_scalar("lnL", lnl); rec_scalar("eval_edge", mixt_b->num);
  if (g_dlk_mode) { rec_scalar("dlk_l", g_dlk_l); rec_scalar("dlnL", g_dlk_dlnl); }
  rec_scalar("datatype", mixt_tree->io->datatype);
  rec_f64("wght", mixt_tree->data->wght, P);
  rec_f64("c_lnL_sorted", mixt_tree->c_lnL_sorted, P);
  rec_f64("mixt_gamma_rr", mixt_tree->mod->ras->gamma_rr->v, mixt_tree->mod->ras->n_catg);
  rec_f64("mixt_gamma_r_proba", mixt_tree->mod->ras->gamma_r_proba->v, mixt_tree->mod->ras->n_catg);
  const double r_sum = MIXT_Get_Sum_Chained_Scalar_Dbl(mixt_tree->next->mod->r_mat_weight);
  const double e_sum = MIXT_Get_Sum_Chained_Scalar_Dbl(mixt_tree->next->mod->e_frq_weight);
  const double sum_p = MIXT_Get_Sum_Of_Probas_Across_Mixtures(r_sum, e_sum, mixt_tree);
  rec_scalar("r_mat_weight_sum", r_sum); rec_scalar("e_frq_weight_sum", e_sum); rec_scalar("sum_probas", sum_p);
  {
    int *el = malloc(sizeof(int) * NE), *er = malloc(sizeof(int) * NE);
    double *len = malloc(sizeof(double) * NE);
    for (int e = 0; e < NE; ++e)
    { el[e] = mixt_tree->a_edges[e]->left->num; er[e] = mixt_tree->a_edges[e]->rght->num; len[e] = mixt_tree->a_edges[e]->l->v; }
    rec_i32("edge_left", el, NE); rec_i32("edge_rght", er, NE); rec_f64("edge_len", len, NE);
    free(el); free(er); free(len);
  }
  {
    t_tree *t0 = mixt_tree->next;
    int    *mask = malloc(sizeof(int) * (size_t)n * P);
    for (int t = 0; t < n; ++t)
    {
      const double *tipv = t0->a_nodes[t]->b[0]->p_lk_tip_r;
      for (int p = 0; p < P; ++p)
      {
        int m = 0;
        for (int s = 0; s < S; ++s)
        {
          const double x = tipv[(size_t)p * S + s];
          if (x != 0.0 && x != 1.0) { fprintf(stderr, "mixt_driver: tip vector entry not 0/1\n"); exit(3); }
          if (x == 1.0) m |= (1 << s);
        }
        mask[(size_t)t * P + p] = m;
      }
    }
    uint64_t d[2] = {(uint64_t)n, (uint64_t)P};
    rec("tip_mask", 1, 2, d, mask);
    free(mask);
  }
  int     k = 0;
  t_edge *b = mixt_b->next;
  char    nm[64];
  for (t_tree *t = mixt_tree->next; t && t->is_mixt_tree == NO; t = t->next, b = b->next, ++k)
  {
    if (t->mod->ras->n_catg != 1 || t->mod->ns != S || t->mod->ras->invar == YES)
    { fprintf(stderr, "mixt_driver: class tree %d is not a plain single-category class\n", k); exit(4); }
    const int pcn = t->mod->ras->parent_class_number;
#define NM(x) (snprintf(nm, sizeof nm, "class%d_%s", k, x), nm)
    rec_scalar(NM("parent_class_number"), pcn);
    rec_scalar(NM("rate"), mixt_tree->mod->ras->gamma_rr->v[pcn]);      /* src/lk.c:2298 */
    rec_scalar(NM("own_gamma_rr"), t->mod->ras->gamma_rr->v[0]);
    rec_scalar(NM("proba"), mixt_tree->mod->ras->gamma_r_proba->v[pcn]);  /* src/mixt.c:1049-1050 */
    rec_scalar(NM("r_mat_weight"), t->mod->r_mat_weight->v);
    rec_scalar(NM("e_frq_weight"), t->mod->e_frq_weight->v);
    rec_scalar(NM("l_min"), t->mod->l_min); rec_scalar(NM("l_max"), t->mod->l_max);
    rec_scalar(NM("br_len_mult"), t->mod->br_len_mult->v);
    rec_f64(NM("pi"), t->mod->e_frq->pi->v, S);
    rec_f64(NM("e_val"), t->mod->eigen->e_val, S);
    rec_f64_2(NM("r_e_vect"), t->mod->eigen->r_e_vect, S, S);
    rec_f64_2(NM("l_e_vect"), t->mod->eigen->l_e_vect, S, S);
    rec_f64(NM("unscaled_site_lk_cat"), t->unscaled_site_lk_cat, P);
    {
      int *f = malloc(sizeof(int) * P);
      for (int p = 0; p < P; ++p)
        f[p] = (b->sum_scale_left ? b->sum_scale_left[p] : 0) + (b->sum_scale_rght ? b->sum_scale_rght[p] : 0); /* src/mixt.c:1018-1025 */
      rec_i32(NM("fact"), f, P);
      free(f);
    }
    { uint64_t d[2] = {(uint64_t)S, (uint64_t)S}; rec(NM("Pij_eval_edge"), 0, 2, d, b->Pij_rr); }
    if (g_dlk_mode) rec_f64_2(NM("dot_prod"), t->dot_prod, P, S);
#undef NM
  }
  fclose(g_out);
  printf("\nMIXT_DRIVER lnL=%.17g classes=%d n_otu=%d n_pattern=%d ns=%d call=%d\n", lnl, K, n, P, S, g_calls);
  fflush(stdout);
  _exit(0);
}

phydbl MIXT_Lk(t_edge *mixt_b, t_tree *mixt_tree)
{
  static phydbl (*real)(t_edge *, t_tree *) = NULL;
  if (!real) real = (phydbl (*)(t_edge *, t_tree *))dlsym(RTLD_NEXT, "MIXT_Lk");
  const phydbl v = real(mixt_b, mixt_tree);
  if (!g_dlk_mode && g_calls++ == g_want) dump_mixture(mixt_b, mixt_tree, v);
  return v;
}

phydbl MIXT_dLk(phydbl *l, t_edge *mixt_b, t_tree *mixt_tree)
{
  static phydbl (*real)(phydbl *, t_edge *, t_tree *) = NULL;
  if (!real) real = (phydbl (*)(phydbl *, t_edge *, t_tree *))dlsym(RTLD_NEXT, "MIXT_dLk");
  const double x = *l;
  const phydbl v = real(l, mixt_b, mixt_tree);
  if (g_dlk_mode && g_dlk_calls++ == g_want)
  {
    g_dlk_l = x; g_dlk_dlnl = mixt_tree->c_dlnL; g_calls = g_dlk_calls;
    dump_mixture(mixt_b, mixt_tree, v);
  }
  return v;
}

int main(int argc, char **argv)
{
  int sep = 3;
  if (argc > 4 && (!strcmp(argv[3], "dlk") || !strcmp(argv[3], "lk"))) { g_dlk_mode = !strcmp(argv[3], "dlk"); sep = 4; }
  if (argc < sep + 2 || strcmp(argv[sep], "--")) { fprintf(stderr, "usage: %s <out.phyg> <call index> [lk|dlk] -- --xml=<file>\n", argv[0]); return 2; }
  g_path = argv[1];
  g_want = atoi(argv[2]);
  int    pargc = argc - sep;
  char **pargv = malloc(sizeof(char *) * (pargc + 1));
  pargv[0] = argv[0];
  for (int k = 1; k < pargc; ++k) pargv[k] = argv[sep + k];
  pargv[pargc] = NULL;
  Get_Input(pargc, pargv); /* XML mode runs the whole analysis inside (src/cl.c:335) */
  fprintf(stderr, "mixt_driver: the run ended before MIXT_Lk call %d\n", g_want);
  return 3;
}
