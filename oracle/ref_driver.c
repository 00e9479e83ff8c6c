/*
 * ref_driver.c -- TEST INFRASTRUCTURE.  Drives the *real* PhyML reference (compiled from
 * /root/reference/src by oracle/Makefile into oracle/_ref/libphyml_ref.a) to
 *   (1) dump golden vectors for the likelihood hot path (mode "dump"), and
 *   (2) time the reference's own AVX `Lk(NULL,tree)` on the host cores (mode "bench").
 *
 * This file is this repo's own code: it only *calls* the reference's public functions, in the
 * order the reference's program entry establishes (src/main.c:73-260: Get_Input, Get_Seq,
 * Make_Model_Complete, Compact_Data, Init_Model, Set_Model_Parameters, tree construction,
 * Connect_CSeqs_To_Nodes, Make_Tree_For_Pars/Lk, Make_Spr, Lk).  It is never linked into the
 * product library and nothing under phyml_amd/ may call it.
 *
 * usage:  phyml_ref_driver dump  <out.phyg> [driver opts] -- <phyml command line>
 *         phyml_ref_driver bench <reps>     [driver opts] -- <phyml command line>
 * driver opts:
 *   --gtr-rr a,b,c,d,e,f   set GTR exchangeabilities (the documented -m a,b,.. form crashes)
 *   --zero-weights k       zero the weight of every k-th pattern (bootstrap-like zero-weight columns)
 *   --one-side             both_sides = NO  (default for dump: YES, so every edge side is filled)
 *   --full-edges n         number of edges whose partial vectors are dumped in full (default 6)
 *   --eigen-edges n        number of edges on which dot_prod / dLk triples are dumped (default 3)
 *   --pmat-edges n         dump the P-matrices of the first n edges only (default: all)
 *   --model-only           dump scalars + model block (pi, rates, eigen system) + lnL only
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include <time.h>

#include "utilities.h"
#include "lk.h"
#include "models.h"
#include "io.h"
#include "make.h"
#include "init.h"
#include "free.h"
#include "pars.h"
#include "spr.h"
#include "optimiz.h"

/* ------------------------------------------------------------------------------------------- */
/* tiny named-array container ("PHYG"): u32 magic, then records                                */
/*   u32 name_len | name | u8 dtype (0=f64 1=i32 2=i16 3=u8) | u32 ndim | u64 dims[] | payload */
/* ------------------------------------------------------------------------------------------- */
static FILE *g_out = NULL;

static void rec(const char *name, int dtype, int ndim, const uint64_t *dims, const void *data)
{
  static const size_t esz[4] = {8, 4, 2, 1};
  uint32_t nl = (uint32_t)strlen(name), nd = (uint32_t)ndim;
  uint8_t  dt = (uint8_t)dtype;
  uint64_t n  = 1;
  for (int i = 0; i < ndim; ++i) n *= dims[i];
  fwrite(&nl, 4, 1, g_out);
  fwrite(name, 1, nl, g_out);
  fwrite(&dt, 1, 1, g_out);
  fwrite(&nd, 4, 1, g_out);
  fwrite(dims, 8, ndim, g_out);
  fwrite(data, esz[dtype], n, g_out);
}
static void rec_f64(const char *name, const double *v, uint64_t n) { rec(name, 0, 1, &n, v); }
static void rec_f64_2(const char *name, const double *v, uint64_t a, uint64_t b)
{ uint64_t d[2] = {a, b}; rec(name, 0, 2, d, v); }
static void rec_i32(const char *name, const int *v, uint64_t n) { rec(name, 1, 1, &n, v); }
static void rec_i16(const char *name, const short *v, uint64_t n) { rec(name, 2, 1, &n, v); }
static void rec_u8_2(const char *name, const unsigned char *v, uint64_t a, uint64_t b)
{ uint64_t d[2] = {a, b}; rec(name, 3, 2, d, v); }
static void rec_scalar(const char *name, double x) { rec_f64(name, &x, 1); }

/* ------------------------------------------------------------------------------------------- */

typedef struct
{
  int    have_rr;
  double rr[6];
  int    zero_every;
  int    both_sides;
  int    n_full_edges;
  int    n_eigen_edges;
  int    n_pmat_edges;
  int    model_only;
} drv_opt;

static t_tree *setup_tree(int argc, char **argv, const drv_opt *o)
{
  option *io = (option *)Get_Input(argc, argv);
  if (!io) { fprintf(stderr, "ref_driver: Get_Input returned NULL\n"); exit(2); }
  srand(io->r_seed < 0 ? 1 : io->r_seed);
  io->n_trees = 1;

  Get_Seq(io);
  Make_Model_Complete(io->mod);
  Set_Model_Name(io->mod);
  t_mod *mod = io->mod;

  if (o->have_rr)
    for (int i = 0; i < 6; ++i) mod->r_mat->rr_val->v[i] = log(o->rr[i]);

  calign *cdata = Compact_Data(io->data, io);
  Free_Seq(io->data, cdata->n_otu);

  Init_Model(cdata, mod, io);
  if (o->have_rr)
    for (int i = 0; i < 6; ++i) mod->r_mat->rr_val->v[i] = log(o->rr[i]);
  Set_Model_Parameters(mod);

  t_tree *tree = (io->in_tree == 2) ? Read_User_Tree(cdata, mod, io) : Dist_And_BioNJ(cdata, mod, io);
  if (!tree) { fprintf(stderr, "ref_driver: no tree\n"); exit(2); }

  tree->mod = mod;  tree->io = io;  tree->data = cdata;
  tree->n_root = NULL;  tree->e_root = NULL;  tree->n_tot_bl_opt = 0;

  Set_Both_Sides(o->both_sides ? YES : NO, tree);
  Connect_CSeqs_To_Nodes(tree->data, tree->io, tree);
  Make_Tree_For_Pars(tree);
  Make_Tree_For_Lk(tree);
  Make_Spr(tree);
  Br_Len_Not_Involving_Invar(tree);
  Unscale_Br_Len_Multiplier_Tree(tree);

  if (o->zero_every > 0)
    for (int s = 0; s < tree->data->n_pattern; ++s)
      if (s % o->zero_every == o->zero_every - 1) tree->data->wght[s] = 0.0;

  Set_Update_Eigen(YES, tree->mod);
  Lk(NULL, tree);
  Set_Update_Eigen(NO, tree->mod);
  return tree;
}

/* sum / sum of squares / scale-sum digest of one edge side; zero-weight patterns are skipped
   because the reference leaves them untouched (src/avx.c:515-520) */
static void side_digest(const t_tree *tree, const double *plk, const int *scale, double out[3])
{
  const int P = tree->data->n_pattern, CS = tree->mod->ras->n_catg * tree->mod->ns;
  double s1 = 0., s2 = 0., sc = 0.;
  for (int p = 0; p < P; ++p)
  {
    if (!(tree->data->wght[p] > SMALL)) continue;
    for (int k = 0; k < CS; ++k) { double x = plk[(size_t)p * CS + k]; s1 += x; s2 += x * x; }
    sc += scale ? scale[p] : 0;
  }
  out[0] = s1; out[1] = s2; out[2] = sc;
}

static void dump(t_tree *tree, const drv_opt *o)
{
  const int n = tree->n_otu, P = tree->data->n_pattern, S = tree->mod->ns,
            C = tree->mod->ras->n_catg, NE = 2 * n - 3, NN = 2 * n - 2;
  char   nm[128];
  double lnl0 = tree->c_lnL;

  rec_scalar("n_otu", n); rec_scalar("n_pattern", P); rec_scalar("ns", S); rec_scalar("ncatg", C);
  rec_scalar("datatype", tree->io->datatype);
  rec_scalar("both_sides", tree->both_sides);
  rec_scalar("lnL", lnl0);
  rec_scalar("l_min", tree->mod->l_min); rec_scalar("l_max", tree->mod->l_max);
  rec_scalar("br_len_mult", tree->mod->br_len_mult->v);
  rec_scalar("invar_model", tree->mod->ras->invar);
  rec_scalar("pinvar", tree->mod->ras->pinvar->v);
  rec_scalar("apply_lk_scaling", tree->apply_lk_scaling);
  rec_scalar("tip_root", tree->tip_root);
  rec_scalar("alpha", tree->mod->ras->alpha->v);

  rec_f64("wght", tree->data->wght, P);
  rec_i16("invar", tree->data->invar, P);
  rec_f64("pi", tree->mod->e_frq->pi->v, S);
  rec_f64("gamma_rr", tree->mod->ras->gamma_rr->v, C);
  rec_f64("gamma_r_proba", tree->mod->ras->gamma_r_proba->v, C);
  rec_f64("e_val", tree->mod->eigen->e_val, S);
  rec_f64_2("r_e_vect", tree->mod->eigen->r_e_vect, S, S);
  rec_f64_2("l_e_vect", tree->mod->eigen->l_e_vect, S, S);

  if (o->model_only) return;

  /* tips: characters, digit states, ambiguity flags, and the 0/1 tip vector as a bit mask */
  {
    unsigned char *chars = malloc((size_t)n * P);
    short *dst = malloc(sizeof(short) * (size_t)n * P), *amb = malloc(sizeof(short) * (size_t)n * P);
    int   *mask = malloc(sizeof(int) * (size_t)n * P);
    for (int t = 0; t < n; ++t)
    {
      const t_node *nd = tree->a_nodes[t];
      const double *tipv = nd->b[0]->p_lk_tip_r;
      for (int p = 0; p < P; ++p)
      {
        chars[(size_t)t * P + p] = (unsigned char)nd->c_seq->state[p];
        dst[(size_t)t * P + p]   = nd->c_seq->d_state[p];
        amb[(size_t)t * P + p]   = nd->c_seq->is_ambigu[p];
        int m = 0;
        for (int s = 0; s < S; ++s)
        {
          double x = tipv[(size_t)p * S + s];
          if (x != 0.0 && x != 1.0) { fprintf(stderr, "ref_driver: tip vector entry not 0/1\n"); exit(3); }
          if (x == 1.0) m |= (1 << s);
        }
        mask[(size_t)t * P + p] = m;
      }
    }
    rec_u8_2("tip_chars", chars, n, P);
    { uint64_t d[2] = {(uint64_t)n, (uint64_t)P}; rec("tip_d_state", 2, 2, d, dst); rec("tip_is_ambigu", 2, 2, d, amb);
      rec("tip_mask", 1, 2, d, mask); }
    free(chars); free(dst); free(amb); free(mask);
  }

  /* topology + lengths + P-matrices */
  {
    int *el = malloc(sizeof(int) * NE), *er = malloc(sizeof(int) * NE);
    double *len = malloc(sizeof(double) * NE);
    int *nv = malloc(sizeof(int) * NN * 3), *nb = malloc(sizeof(int) * NN * 3);
    const int NPM = (o->n_pmat_edges >= 0 && o->n_pmat_edges < NE) ? o->n_pmat_edges : NE;
    double *pm = malloc(sizeof(double) * (size_t)(NPM + 1) * C * S * S);
    for (int e = 0; e < NE; ++e)
    {
      const t_edge *b = tree->a_edges[e];
      if (b->num != e) { fprintf(stderr, "ref_driver: edge numbering assumption broken\n"); exit(3); }
      el[e] = b->left->num; er[e] = b->rght->num; len[e] = b->l->v;
      if (e < NPM) memcpy(pm + (size_t)e * C * S * S, b->Pij_rr, sizeof(double) * C * S * S);
      for (int c = 0; c < C; ++c) for (int i = 0; i < S; ++i) for (int j = 0; j < S; ++j)
        if (b->tPij_rr[c * S * S + j * S + i] != b->Pij_rr[c * S * S + i * S + j])
        { fprintf(stderr, "ref_driver: tPij is not the transpose of Pij\n"); exit(3); }
    }
    for (int k = 0; k < NN; ++k)
    {
      const t_node *nd = tree->a_nodes[k];
      for (int i = 0; i < 3; ++i)
      {
        nv[k * 3 + i] = (nd->v[i]) ? nd->v[i]->num : -1;
        nb[k * 3 + i] = (nd->b[i]) ? nd->b[i]->num : -1;
      }
    }
    rec_i32("edge_left", el, NE); rec_i32("edge_rght", er, NE); rec_f64("edge_len", len, NE);
    { uint64_t d[2] = {(uint64_t)NN, 3}; rec("node_v", 1, 2, d, nv); rec("node_b", 1, 2, d, nb); }
    { uint64_t d[4] = {(uint64_t)NPM, (uint64_t)C, (uint64_t)S, (uint64_t)S}; rec("Pij_rr", 0, 4, d, pm); }
    free(el); free(er); free(len); free(nv); free(nb); free(pm);
  }

  /* per-site outputs of the root-edge evaluation that Lk(NULL) just performed */
  rec_f64("c_lnL_sorted", tree->c_lnL_sorted, P);
  rec_f64("cur_site_lk", tree->cur_site_lk, P);
  rec_i32("fact_sum_scale", tree->fact_sum_scale, P);
  rec_f64_2("unscaled_site_lk_cat", tree->unscaled_site_lk_cat, P, C);

  /* digests of every edge side + a few edges in full */
  {
    double *dg = calloc((size_t)NE * 2 * 3, sizeof(double));
    int *valid = calloc((size_t)NE * 2, sizeof(int));
    for (int e = 0; e < NE; ++e)
    {
      const t_edge *b = tree->a_edges[e];
      /* which sides hold a computed internal partial vector?  left: computed by post-order iff left is
         internal and (both_sides or the side points away from tip_root); we only trust both_sides=YES */
      if (!b->left->tax && b->p_lk_left && tree->both_sides) { side_digest(tree, b->p_lk_left, b->sum_scale_left, dg + (e * 2 + 0) * 3); valid[e * 2 + 0] = 1; }
      if (!b->rght->tax && b->p_lk_rght && tree->both_sides) { side_digest(tree, b->p_lk_rght, b->sum_scale_rght, dg + (e * 2 + 1) * 3); valid[e * 2 + 1] = 1; }
    }
    { uint64_t d[3] = {(uint64_t)NE, 2, 3}; rec("side_digest", 0, 3, d, dg); }
    { uint64_t d[2] = {(uint64_t)NE, 2}; rec("side_valid", 1, 2, d, valid); }

    int nfull = 0;
    int *full_ids = malloc(sizeof(int) * (o->n_full_edges + 1));
    for (int k = 0; k < NE && nfull < o->n_full_edges; ++k)
    {
      int e = (int)(((long)k * 7919) % NE); /* spread over the tree */
      int dup = 0; for (int q = 0; q < nfull; ++q) if (full_ids[q] == e) dup = 1;
      if (dup || !(valid[e * 2] || valid[e * 2 + 1])) continue;
      const t_edge *b = tree->a_edges[e];
      if (valid[e * 2])
      {
        snprintf(nm, sizeof nm, "p_lk_left_%d", e);      rec_f64_2(nm, b->p_lk_left, P, (uint64_t)C * S);
        snprintf(nm, sizeof nm, "sum_scale_left_%d", e); rec_i32(nm, b->sum_scale_left, P);
      }
      if (valid[e * 2 + 1])
      {
        snprintf(nm, sizeof nm, "p_lk_rght_%d", e);      rec_f64_2(nm, b->p_lk_rght, P, (uint64_t)C * S);
        snprintf(nm, sizeof nm, "sum_scale_rght_%d", e); rec_i32(nm, b->sum_scale_rght, P);
      }
      full_ids[nfull++] = e;
    }
    rec_i32("full_edges", full_ids, nfull);
    free(dg); free(valid); free(full_ids);
  }

  /* lnL evaluated at every edge (pulley principle; needs both sides) */
  if (tree->both_sides)
  {
    double *el = malloc(sizeof(double) * NE);
    for (int e = 0; e < NE; ++e) el[e] = Lk(tree->a_edges[e], tree);
    rec_f64("edge_lnL", el, NE);
    free(el);

    /* eigen-basis products and derivative triples on a few edges (Br_Len_Opt's call pattern,
       src/optimiz.c:622-630) */
    int ndone = 0;
    int *eig_ids = malloc(sizeof(int) * (o->n_eigen_edges + 1));
    double *trip = malloc(sizeof(double) * (o->n_eigen_edges + 1) * 3 * 3);
    for (int k = 0; k < NE && ndone < o->n_eigen_edges; ++k)
    {
      int e = (int)(((long)k * 104729 + 3) % NE);
      int dup = 0; for (int q = 0; q < ndone; ++q) if (eig_ids[q] == e) dup = 1;
      if (dup) continue;
      t_edge *b = tree->a_edges[e];
      Set_Update_Eigen_Lr(YES, tree); Set_Use_Eigen_Lr(NO, tree);
      Lk(b, tree);
      snprintf(nm, sizeof nm, "dot_prod_%d", e); rec_f64_2(nm, tree->dot_prod, P, (uint64_t)C * S);
      snprintf(nm, sizeof nm, "eig_fact_sum_scale_%d", e); rec_i32(nm, tree->fact_sum_scale, P);
      Set_Update_Eigen_Lr(NO, tree); Set_Use_Eigen_Lr(YES, tree);
      const double mult[3] = {0.5, 1.0, 3.0};
      for (int t = 0; t < 3; ++t)
      {
        double l = b->l->v * mult[t];
        dLk(&l, b, tree);
        trip[(ndone * 3 + t) * 3 + 0] = l;
        trip[(ndone * 3 + t) * 3 + 1] = tree->c_lnL;
        trip[(ndone * 3 + t) * 3 + 2] = tree->c_dlnL;
      }
      /* Lk(b) in the eigen basis at the current length (src/lk.c:592-603,625-629) */
      { double v = Lk(b, tree); snprintf(nm, sizeof nm, "eig_lnL_%d", e); rec_scalar(nm, v); }
      Set_Update_Eigen_Lr(NO, tree); Set_Use_Eigen_Lr(NO, tree);
      eig_ids[ndone++] = e;
    }
    rec_i32("eigen_edges", eig_ids, ndone);
    { uint64_t d[3] = {(uint64_t)ndone, 3, 3}; rec("dlk_triples", 0, 3, d, trip); }
    free(eig_ids); free(trip);
  }
}

static double now_s(void)
{
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv)
{
  if (argc < 4) { fprintf(stderr, "usage: %s dump <out>|bench <reps> [opts] -- <phyml args>\n", argv[0]); return 2; }
  const char *mode = argv[1], *arg = argv[2];
  drv_opt o; memset(&o, 0, sizeof o);
  o.both_sides = 1; o.n_full_edges = 6; o.n_eigen_edges = 3; o.n_pmat_edges = -1;

  int i = 3;
  for (; i < argc; ++i)
  {
    if (!strcmp(argv[i], "--")) { ++i; break; }
    else if (!strcmp(argv[i], "--gtr-rr") && i + 1 < argc)
    {
      o.have_rr = 1;
      if (sscanf(argv[++i], "%lf,%lf,%lf,%lf,%lf,%lf", o.rr, o.rr + 1, o.rr + 2, o.rr + 3, o.rr + 4, o.rr + 5) != 6)
      { fprintf(stderr, "bad --gtr-rr\n"); return 2; }
    }
    else if (!strcmp(argv[i], "--zero-weights") && i + 1 < argc) o.zero_every = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--one-side")) o.both_sides = 0;
    else if (!strcmp(argv[i], "--full-edges") && i + 1 < argc) o.n_full_edges = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--eigen-edges") && i + 1 < argc) o.n_eigen_edges = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--pmat-edges") && i + 1 < argc) o.n_pmat_edges = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--model-only")) o.model_only = 1;
    else { fprintf(stderr, "unknown driver option %s\n", argv[i]); return 2; }
  }
  /* rebuild an argv for the reference's own command-line parser */
  int    pargc = argc - i + 1;
  char **pargv = malloc(sizeof(char *) * (pargc + 1));
  pargv[0] = argv[0];
  for (int k = 1; k < pargc; ++k) pargv[k] = argv[i + k - 1];
  pargv[pargc] = NULL;

  if (!strcmp(mode, "dump"))
  {
    t_tree *tree = setup_tree(pargc, pargv, &o);
    const double lnl_root = tree->c_lnL;
    g_out = fopen(arg, "wb");
    if (!g_out) { perror(arg); return 2; }
    fwrite("PHYG", 1, 4, g_out);
    dump(tree, &o);
    fclose(g_out);
    printf("\nREF_DRIVER lnL=%.17g n_otu=%d n_pattern=%d ns=%d ncatg=%d\n", lnl_root, tree->n_otu,
           tree->data->n_pattern, tree->mod->ns, tree->mod->ras->n_catg);
    return 0;
  }
  else if (!strcmp(mode, "bench"))
  {
    int reps = atoi(arg);
    o.both_sides = 0;
    t_tree *tree = setup_tree(pargc, pargv, &o);
    Set_Both_Sides(NO, tree);
    double lnl = Lk(NULL, tree); /* warm */
    double t0 = now_s();
    for (int r = 0; r < reps; ++r) lnl = Lk(NULL, tree);
    double dt = (now_s() - t0) / (reps > 0 ? reps : 1);
    double upd = (double)tree->data->n_pattern * (tree->n_otu - 2);
    printf("\nREF_BENCH {\"lnL\": %.17g, \"n_otu\": %d, \"n_pattern\": %d, \"ns\": %d, \"ncatg\": %d, "
           "\"reps\": %d, \"s_per_lk\": %.9g, \"site_updates_per_s\": %.9g}\n",
           lnl, tree->n_otu, tree->data->n_pattern, tree->mod->ns, tree->mod->ras->n_catg, reps, dt, upd / dt);
    return 0;
  }
  fprintf(stderr, "unknown mode %s\n", mode);
  return 2;
}
