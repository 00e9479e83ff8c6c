/*
 * trace_driver.c -- TEST INFRASTRUCTURE.  Records the stream of likelihood-surface calls that a REAL PhyML
 * tree search makes (spr.c / optimiz.c driving Lk, Update_Partial_Lk, Update_PMat_At_Given_Edge, Update_Eigen_Lr,
 * dLk), at buffer level, together with the scalar every call returned (SURVEY 7.1 step 10b, 8b "caller
 * counterpart").
 *
 * How: the reference objects (compiled, unmodified, from /root/reference/src by oracle/Makefile) are linked as a
 * shared object, oracle/_ref/libphyml_ref.so.  They are position-independent code, so every call to a global
 * function -- including the calls lk.c makes to its own Update_Partial_Lk / Update_PMat_At_Given_Edge -- goes
 * through the PLT.  This executable defines functions with those names and the reference's signatures; the
 * dynamic linker binds the library's calls to them, they log and forward to the originals (dlsym RTLD_NEXT).
 * No reference source is modified, copied or stubbed.
 *
 * Buffers are identified by POINTER (first sight = next id): SPR's prune / graft swap buffer pointers between
 * edges (src/utilities.c:6235-6398, 6618-6686) without touching contents, so a pointer-level stream captures the
 * rearrangements with no SWAP record.  Which buffers a call reads and writes is resolved with the reference's own
 * Set_All_Partial_Lk (src/lk.c:2922) at the moment of the call.
 *
 * Record kinds are those of include/phyhip_lk.h (PHL_REC_*):
 *   0 SET_PMAT  a = matrix, x = b->l->v
 *   1 UPDATE    a = dest, b = child1 (tip: taxon number), c = matrix1, d = child2, e = matrix2
 *   2 EDGE_LNL  a = left buffer, b = right buffer or tip, c = matrix           -> out = returned lnL
 *   3 EIGEN_LR  a = left, b = right or tip
 *   4 DLK       x = *l on entry                                               -> out = lnL, out2 = dlnL
 *   5 EIGEN_LNL x = b->l->v (Lk(b) in the eigen basis, src/lk.c:592-603)       -> out = returned lnL
 * Partials buffers are numbered n_otu, n_otu+1, ... in order of first appearance; tips are 0..n_otu-1.
 *
 * usage: phyml_trace_driver <out.phyg> <max_records> [--gtr-rr a,..,f] -- <phyml command line>
 */
#define _GNU_SOURCE
#include <dlfcn.h>

#define main ref_driver_main_unused
#include "ref_driver.c"
#undef main

/* ------------------------------------------------------------------------------------------- */
static int     g_on = 0, g_max = 0, g_n = 0, g_cap = 0;
static int    *g_kind, *g_a, *g_b, *g_c, *g_d, *g_e;
static double *g_x, *g_o1, *g_o2;
static t_tree *g_tree = NULL;
static const char *g_path = NULL;
static void    finish_and_exit(void);

static void push(int kind, int a, int b, int c, int d, int e, double x, double o1, double o2)
{
  if (g_n == g_cap)
  {
    g_cap = g_cap ? 2 * g_cap : 4096;
    g_kind = realloc(g_kind, sizeof(int) * g_cap); g_a = realloc(g_a, sizeof(int) * g_cap);
    g_b = realloc(g_b, sizeof(int) * g_cap); g_c = realloc(g_c, sizeof(int) * g_cap);
    g_d = realloc(g_d, sizeof(int) * g_cap); g_e = realloc(g_e, sizeof(int) * g_cap);
    g_x = realloc(g_x, sizeof(double) * g_cap); g_o1 = realloc(g_o1, sizeof(double) * g_cap);
    g_o2 = realloc(g_o2, sizeof(double) * g_cap);
  }
  g_kind[g_n] = kind; g_a[g_n] = a; g_b[g_n] = b; g_c[g_n] = c; g_d[g_n] = d; g_e[g_n] = e;
  g_x[g_n] = x; g_o1[g_n] = o1; g_o2[g_n] = o2;
  ++g_n;
}

/* pointer -> id tables (a few hundred entries at most) */
#define MAXID 4096
static const void *g_bufptr[MAXID]; static const void *g_bufscale[MAXID]; static int g_nbuf = 0;
static const void *g_matptr[MAXID]; static int g_nmat = 0;

static int buf_id(const t_tree *tree, const void *p_lk, const void *sum_scale)
{
  for (int i = 0; i < g_nbuf; ++i)
    if (g_bufptr[i] == p_lk)
    {
      /* the scale vector must travel with its partials buffer (one device buffer index covers both) */
      if (sum_scale && g_bufscale[i] && g_bufscale[i] != sum_scale)
      { fprintf(stderr, "trace_driver: partials buffer %d changed its scale vector\n", i); exit(4); }
      if (sum_scale) g_bufscale[i] = sum_scale;
      return tree->n_otu + i;
    }
  if (g_nbuf == MAXID) { fprintf(stderr, "trace_driver: too many buffers\n"); exit(4); }
  g_bufptr[g_nbuf] = p_lk; g_bufscale[g_nbuf] = sum_scale;
  return tree->n_otu + g_nbuf++;
}
static int mat_id(const void *pij)
{
  for (int i = 0; i < g_nmat; ++i) if (g_matptr[i] == pij) return i;
  if (g_nmat == MAXID) { fprintf(stderr, "trace_driver: too many matrices\n"); exit(4); }
  g_matptr[g_nmat] = pij;
  return g_nmat++;
}
static void check_tree(const t_tree *tree, const char *who)
{
  if (tree->is_mixt_tree || tree->n_root || tree->mod->gamma_mgf_bl == YES || tree->mod->log_l == YES)
  { fprintf(stderr, "trace_driver: %s on an unsupported tree kind (mixture / rooted / mgf / log_l)\n", who); exit(4); }
  if (g_tree && tree != g_tree) { fprintf(stderr, "trace_driver: %s on a second tree object\n", who); exit(4); }
}
/* the two sides of an edge as the evaluation reads them (src/lk.c:605-606) */
static void edge_sides(const t_tree *tree, const t_edge *b, int *left, int *right)
{
  *left  = buf_id(tree, b->p_lk_left, b->sum_scale_left);
  *right = b->rght->tax ? b->rght->num : buf_id(tree, b->p_lk_rght, b->sum_scale_rght);
}
static void maybe_stop(void) { if (g_on && g_n >= g_max) finish_and_exit(); }

/* ---- interposed surface -------------------------------------------------------------------- */

void Update_Partial_Lk(t_tree *tree, t_edge *b, t_node *d)
{
  static void (*real)(t_tree *, t_edge *, t_node *) = NULL;
  if (!real) real = (void (*)(t_tree *, t_edge *, t_node *))dlsym(RTLD_NEXT, "Update_Partial_Lk");
  /* the gates of src/lk.c:1285-1297: a call that returns early changes no buffer and is not recorded */
  const int runs = !(b->left == d && b->update_partial_lk_left == NO) && !(b->rght == d && b->update_partial_lk_rght == NO) &&
                   !d->tax;
  if (g_on && runs)
  {
    check_tree(tree, "Update_Partial_Lk");
    t_node *n_v1 = NULL, *n_v2 = NULL;
    phydbl *p_lk = NULL, *p_lk_v1 = NULL, *p_lk_v2 = NULL, *Pij1 = NULL, *Pij2 = NULL, *tPij1 = NULL, *tPij2 = NULL;
    int    *sum_scale = NULL, *sum_scale_v1 = NULL, *sum_scale_v2 = NULL, *p_lk_loc = NULL;
    Set_All_Partial_Lk(&n_v1, &n_v2, &p_lk, &sum_scale, &p_lk_loc, &Pij1, &tPij1, &p_lk_v1, &sum_scale_v1, &Pij2, &tPij2,
                       &p_lk_v2, &sum_scale_v2, d, b, tree);
    if (!n_v1 || !n_v2) { fprintf(stderr, "trace_driver: internal node with a missing neighbour\n"); exit(4); }
    const int dest = buf_id(tree, p_lk, sum_scale);
    const int c1   = n_v1->tax ? n_v1->num : buf_id(tree, p_lk_v1, sum_scale_v1);
    const int c2   = n_v2->tax ? n_v2->num : buf_id(tree, p_lk_v2, sum_scale_v2);
    push(1, dest, c1, mat_id(Pij1), c2, mat_id(Pij2), 0.0, 0.0, 0.0);
  }
  real(tree, b, d);
  maybe_stop();
}

void Update_PMat_At_Given_Edge(t_edge *b_fcus, t_tree *tree)
{
  static void (*real)(t_edge *, t_tree *) = NULL;
  if (!real) real = (void (*)(t_edge *, t_tree *))dlsym(RTLD_NEXT, "Update_PMat_At_Given_Edge");
  if (g_on)
  {
    check_tree(tree, "Update_PMat_At_Given_Edge");
    if (b_fcus->has_zero_br_len == YES) { fprintf(stderr, "trace_driver: zero-length edge flag not supported\n"); exit(4); }
    push(0, mat_id(b_fcus->Pij_rr), 0, 0, 0, 0, b_fcus->l->v, 0.0, 0.0);
  }
  real(b_fcus, tree);
  maybe_stop();
}

void Update_Eigen_Lr(t_edge *b, t_tree *tree)
{
  static void (*real)(t_edge *, t_tree *) = NULL;
  if (!real) real = (void (*)(t_edge *, t_tree *))dlsym(RTLD_NEXT, "Update_Eigen_Lr");
  if (g_on)
  {
    check_tree(tree, "Update_Eigen_Lr");
    int l, r;
    edge_sides(tree, b, &l, &r);
    push(3, l, r, 0, 0, 0, 0.0, 0.0, 0.0);
  }
  real(b, tree);
  maybe_stop();
}

phydbl Lk(t_edge *b, t_tree *tree)
{
  static phydbl (*real)(t_edge *, t_tree *) = NULL;
  if (!real) real = (phydbl (*)(t_edge *, t_tree *))dlsym(RTLD_NEXT, "Lk");
  const phydbl v = real(b, tree); /* the matrix refreshes and partial updates it performs are recorded on the way */
  if (g_on)
  {
    check_tree(tree, "Lk");
    const t_edge *e = b ? b : tree->a_nodes[tree->tip_root]->b[0]; /* src/lk.c:569-580, unrooted */
    if (tree->use_eigen_lr == YES) push(5, 0, 0, 0, 0, 0, e->l->v, v, 0.0);
    else
    {
      int l, r;
      edge_sides(tree, e, &l, &r);
      push(2, l, r, mat_id(e->Pij_rr), 0, 0, 0.0, v, 0.0);
    }
  }
  maybe_stop();
  return v;
}

phydbl dLk(phydbl *l, t_edge *b, t_tree *tree)
{
  static phydbl (*real)(phydbl *, t_edge *, t_tree *) = NULL;
  if (!real) real = (phydbl (*)(phydbl *, t_edge *, t_tree *))dlsym(RTLD_NEXT, "dLk");
  const double x = *l;
  const phydbl v = real(l, b, tree);
  if (g_on)
  {
    check_tree(tree, "dLk");
    push(4, 0, 0, 0, 0, 0, x, v, tree->c_dlnL);
  }
  maybe_stop();
  return v;
}

/* ---- output ---------------------------------------------------------------------------------- */

static void write_header(t_tree *tree)
{
  const int n = tree->n_otu, P = tree->data->n_pattern, S = tree->mod->ns, C = tree->mod->ras->n_catg, NE = 2 * n - 3;
  drv_opt o; memset(&o, 0, sizeof o); o.model_only = 1;
  dump(tree, &o); /* scalars, weights, +I data, model block */
  int *mask = malloc(sizeof(int) * (size_t)n * P);
  for (int t = 0; t < n; ++t)
  {
    const double *tipv = tree->a_nodes[t]->b[0]->p_lk_tip_r;
    for (int p = 0; p < P; ++p)
    {
      int m = 0;
      for (int s = 0; s < S; ++s)
      {
        const double x = tipv[(size_t)p * S + s];
        if (x != 0.0 && x != 1.0) { fprintf(stderr, "trace_driver: tip vector entry not 0/1\n"); exit(3); }
        if (x == 1.0) m |= (1 << s);
      }
      mask[(size_t)t * P + p] = m;
    }
  }
  { uint64_t d[2] = {(uint64_t)n, (uint64_t)P}; rec("tip_mask", 1, 2, d, mask); }
  free(mask);
  int *el = malloc(sizeof(int) * NE), *er = malloc(sizeof(int) * NE);
  double *len = malloc(sizeof(double) * NE);
  for (int e = 0; e < NE; ++e)
  { el[e] = tree->a_edges[e]->left->num; er[e] = tree->a_edges[e]->rght->num; len[e] = tree->a_edges[e]->l->v; }
  rec_i32("edge_left", el, NE); rec_i32("edge_rght", er, NE); rec_f64("edge_len", len, NE);
  free(el); free(er); free(len);
  (void)C;
}

static void finish_and_exit(void)
{
  g_on = 0;
  rec_scalar("trace_n_buffers", g_nbuf);
  rec_scalar("trace_n_matrices", g_nmat);
  rec_i32("trace_kind", g_kind, g_n); rec_i32("trace_a", g_a, g_n); rec_i32("trace_b", g_b, g_n);
  rec_i32("trace_c", g_c, g_n); rec_i32("trace_d", g_d, g_n); rec_i32("trace_e", g_e, g_n);
  rec_f64("trace_x", g_x, g_n); rec_f64("trace_out", g_o1, g_n); rec_f64("trace_out2", g_o2, g_n);
  fclose(g_out);
  int cnt[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < g_n; ++i) cnt[g_kind[i]]++;
  printf("\nTRACE_DRIVER records=%d set_pmat=%d update=%d edge_lnl=%d eigen_lr=%d dlk=%d eigen_lnl=%d buffers=%d matrices=%d\n",
         g_n, cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5], g_nbuf, g_nmat);
  fflush(stdout);
  _exit(0);
}

int main(int argc, char **argv)
{
  if (argc < 5) { fprintf(stderr, "usage: %s <out.phyg> <max_records> [--gtr-rr a,..,f] -- <phyml args>\n", argv[0]); return 2; }
  g_path = argv[1];
  g_max  = atoi(argv[2]);
  drv_opt o; memset(&o, 0, sizeof o);
  o.both_sides = 0;
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
    else { fprintf(stderr, "unknown driver option %s\n", argv[i]); return 2; }
  }
  int    pargc = argc - i + 1;
  char **pargv = malloc(sizeof(char *) * (pargc + 1));
  pargv[0] = argv[0];
  for (int k = 1; k < pargc; ++k) pargv[k] = argv[i + k - 1];
  pargv[pargc] = NULL;

  g_out = fopen(g_path, "wb");
  if (!g_out) { perror(g_path); return 2; }
  fwrite("PHYG", 1, 4, g_out);

  /* everything from the first Lk(NULL) of the program entry on is recorded (src/main.c:256-258) */
  g_on = 1;
  t_tree *tree = setup_tree(pargc, pargv, &o);
  g_tree = tree;
  g_on = 0;
  write_header(tree); /* model block, tips, weights: state before the search (the model stays fixed: no -o r / alpha) */
  g_on = 1;
  /* src/main.c:262-275 */
  if (tree->mod->s_opt->opt_topo) Global_Spr_Search(tree);
  else if (tree->mod->s_opt->opt_subst_param || tree->mod->s_opt->opt_bl_one_by_one) Round_Optimize(tree, ROUND_MAX);
  /* src/main.c:281-282 */
  Set_Both_Sides(YES, tree);
  Lk(NULL, tree);
  printf("\nTRACE_DRIVER final lnL=%.17g\n", tree->c_lnL);
  finish_and_exit();
  return 0;
}
