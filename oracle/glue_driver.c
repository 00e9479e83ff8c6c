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
 * Mixtures (XML mode, e.g. examples/lg4x; one partition element, no +I): every class tree of the mixture (src/mixt.c:2603-2640) gets
 * its own instance; MIXT_Update_Partial_Lk / MIXT_Update_PMat_At_Given_Edge loop over the class trees and come back
 * through the wrappers; MIXT_Lk is interposed too and, after the original, repeats the evaluation with
 * phyhip_calculate_mixture_log_likelihood over the class instances (GLUE_MAX_MIXT=n stops after n comparisons).
 *
 * usage: phyml_glue_driver [--gtr-rr a,..,f] -- <phyml command line>      (or: -- --xml=<file>)
 * prints: GLUE_DRIVER {json}   (final lnL, tree, call counts, worst per-call differences in check mode)
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <link.h>

#define main ref_driver_main_unused
#include "ref_driver.c"
#undef main

#include "mixt.h"
#include "../include/phyhip.h"
#define kMaxClasses 64

static int g_check = 0, g_device_pmat = 0, g_host = 0;
static long    g_n_lk = 0, g_n_lk_full = 0, g_n_upd = 0, g_n_dlk = 0, g_n_pmat = 0, g_n_eig = 0, g_n_mixt = 0, g_n_mixt_skipped = 0, g_n_alias = 0;
static double  g_worst_lnl = 0.0, g_worst_dlnl = 0.0, g_worst_mixt = 0.0, g_worst_mixt_dlnl = 0.0;
static long    g_n_mixt_dlk = 0;
static double  g_last_mixt_lnl = 0.0, g_best_full_lnl = -1e300;
/* branch-support phase (aLRT_From_String, src/utilities.c:9282): alrt.c reads the per-pattern log-likelihoods of the last
   Lk() (src/alrt.c:453,555,682), so every device evaluation is followed by the download hook of SURVEY 8(f) rank 3 */
static int     g_site_outputs = 0;
static long    g_n_dropped = 0, g_n_created = 0;
static long    g_n_site_dl = 0;
static double  g_worst_site_lnl = 0.0;

/* ---- the reference's random stream ------------------------------------------------------------------------------
   PhyML draws from libc's global rand() (SPR's Rgamma jitter src/spr.c:813, Permutate, the RELL resampling of the SH-like
   supports src/alrt.c:1148).  libamdhip64 and libhsa-runtime64 import rand/srand too, so in a process that talks to the
   GPU the global stream is no longer the one `--r_seed` set up, and host / check / device runs of one command would not be
   comparable where PhyML is stochastic.  rand/srand are therefore interposed like the likelihood surface: callers inside
   libphyml_ref.so or this executable get a private stream -- glibc's own generator (random_r on a private TYPE_3 state
   gives exactly the srand/rand sequence) -- everybody else gets libc's global one. */
static uintptr_t g_ref_lo[4], g_ref_hi[4];
static int       g_ref_n = -1;
static int       note_object(struct dl_phdr_info *info, size_t size, void *data)
{
  (void)size; (void)data;
  const char *nm = info->dlpi_name ? info->dlpi_name : "";
  if (nm[0] && !strstr(nm, "libphyml_ref")) return 0; /* "" is the executable */
  for (int k = 0; k < info->dlpi_phnum && g_ref_n < 4; ++k)
    if (info->dlpi_phdr[k].p_type == PT_LOAD && (info->dlpi_phdr[k].p_flags & PF_X))
    {
      g_ref_lo[g_ref_n] = info->dlpi_addr + info->dlpi_phdr[k].p_vaddr;
      g_ref_hi[g_ref_n] = g_ref_lo[g_ref_n] + info->dlpi_phdr[k].p_memsz;
      ++g_ref_n;
    }
  return 0;
}
static int from_reference(const void *ret)
{
  if (g_ref_n < 0) { g_ref_n = 0; dl_iterate_phdr(note_object, NULL); }
  for (int k = 0; k < g_ref_n; ++k) if ((uintptr_t)ret >= g_ref_lo[k] && (uintptr_t)ret < g_ref_hi[k]) return 1;
  return 0;
}
static struct random_data g_rng;
static char               g_rng_state[128];
static int                g_rng_ready = 0;
static void               private_seed(unsigned seed)
{
  if (!g_rng_ready) { memset(&g_rng, 0, sizeof g_rng); initstate_r(seed, g_rng_state, sizeof g_rng_state, &g_rng); g_rng_ready = 1; }
  else srandom_r(seed, &g_rng);
}
void srand(unsigned seed)
{
  static void (*real)(unsigned) = NULL;
  if (from_reference(__builtin_return_address(0))) { private_seed(seed); return; }
  if (!real) real = (void (*)(unsigned))dlsym(RTLD_NEXT, "srand");
  real(seed);
}
int rand(void)
{
  static int (*real)(void) = NULL;
  if (from_reference(__builtin_return_address(0)))
  {
    int32_t r;
    if (!g_rng_ready) private_seed(1); /* glibc: rand() before srand() behaves as srand(1) */
    random_r(&g_rng, &r);
    return (int)r;
  }
  if (!real) real = (int (*)(void))dlsym(RTLD_NEXT, "rand");
  return real();
}

static void die(const char *what)
{
  fprintf(stderr, "\n== glue_driver: %s: %s\n", what, phyhip_get_last_error());
  exit(5); /* the reference's convention: message, then Exit (src/beagle_utils.c:246-249) */
}
#define OK(call) do { if ((call) < 0) die(#call); } while (0)

/* GLUE_HOSTPROF=1: where the HOST's time goes in a device-driven run -- cycles (rdtsc) inside each call of the C ABI and inside
   the interposed surface functions around them (pointer look-ups, Set_All_Partial_Lk, the reference's own PMat), reported per
   call and as a share of the run in the GLUE_DRIVER line (tools/host_share.py, profiles/r05_host_share.md).  The evaluation
   slots include the wait for the device. */
enum { HP_ABI_UPDATE_PARTIALS, HP_WRAP_UPDATE_PARTIAL_LK, HP_ABI_MATRIX, HP_HOST_PMAT, HP_WRAP_UPDATE_PMAT, HP_ABI_EDGE_LNL, HP_ABI_EIGEN_EVAL,
       HP_ABI_UPDATE_EIGEN_LR, HP_ABI_WARNING, HP_PUSH_MODEL, HP_SLOTS };
static const char *const hp_name[HP_SLOTS] = {"phyhip_update_partials", "Update_Partial_Lk wrapper (incl. the call)", "phyhip_set_transition_matrix / update_transition_matrices",
                                              "reference PMat on the host", "Update_PMat_At_Given_Edge wrapper (incl. both)", "phyhip_calculate_edge_log_likelihoods (incl. wait)",
                                              "phyhip_calculate_eigen_lnl[_dlnl] (incl. wait)", "phyhip_update_eigen_lr", "phyhip_get_numerical_warning", "push_model"};
static int                g_hp_on = 0;
static unsigned long long g_hp_cyc[HP_SLOTS], g_hp_n[HP_SLOTS], g_hp_c0 = 0;
static inline unsigned long long hp_tick(void) { return g_hp_on ? __builtin_ia32_rdtsc() : 0ull; }
static inline void hp_add(int slot, unsigned long long t0) { if (g_hp_on) { g_hp_cyc[slot] += __builtin_ia32_rdtsc() - t0; ++g_hp_n[slot]; } }
#define TOK(slot, call) do { const unsigned long long t_ = hp_tick(); OK(call); hp_add(slot, t_); } while (0)

/* one device instance per tree object that reaches the surface: the tree of an ordinary run, or every class tree of a
   mixture (src/mixt.c:2603-2640); pointer -> device index tables per instance */
#define MAXID 2048
#define MAXCTX (kMaxClasses + 32) /* one context per class tree: a 64-class profile mixture and then some */
typedef struct
{
  t_tree     *tree;
  int         inst;
  const void *bufptr[MAXID]; int nbuf, bufcap;
  int        *scaleptr[MAXID]; /* host sum_scale vector of each partials buffer (known once it was a destination) */
  const void *matptr[MAXID]; int nmat, matcap;
  /* GLUE_CLASS_AXIS=1: the class trees of a mixture share ONE instance whose categories are the classes
     (PHYHIP_FLAG_CLASS_AXIS).  Every class tree keeps its own pointer tables, but ids are handed out by class 0 only:
     the reference loops over the class trees for every surface call (MIXT_Update_Partial_Lk, src/mixt.c:1191-1250), so
     class k > 0 sees exactly the call class 0 just made and registers its pointers under the same ids. */
  int         cls, K;          /* class index / class count (K = 0: ordinary context) */
  void       *batch;           /* clsbatch_t of the element (class-0 context only) */
  int         is_class_tree;   /* tree->mixt_tree != NULL when the context was made (the tree object may be gone later) */
} ctx_t;
static int g_class_axis = 0;
/* The partial updates class 0 queued since the last mixture evaluation.  The reference walks the class trees one after
   the other for a whole traversal (MIXT_Post_Order_Lk, src/mixt.c:656-688: a full post-order per class tree) or one
   operation at a time (MIXT_Update_Partial_Lk), so class k's j-th call of the batch is class 0's j-th operation. */
#define MAXCLSOPS 8192
typedef struct
{ /* one per partition element (= per class-axis instance), hung on the class-0 context */
  phyhip_operation ops[MAXCLSOPS];
  int              nops, pos[kMaxClasses];
} clsbatch_t;
static void cls_batch_done(clsbatch_t *q, int K)
{
  for (int k = 1; k < K; ++k)
    if (q->pos[k] != q->nops)
    { fprintf(stderr, "glue_driver: class tree %d made %d of class 0's %d partial updates before the evaluation\n", k, q->pos[k], q->nops); exit(6); }
  q->nops = 0;
  for (int k = 0; k < kMaxClasses; ++k) q->pos[k] = 0;
}
static int g_cls_mat = -1;          /* the matrix id class 0 refreshed last */
static ctx_t g_ctx[MAXCTX];
static int   g_nctx = 0;
#define g_inst (g_ctx[0].inst)
#define g_nbuf (g_ctx[0].nbuf)
#define g_nmat (g_ctx[0].nmat)

static int buf_id(ctx_t *c, const void *p_lk)
{
  for (int i = 0; i < c->nbuf; ++i) if (c->bufptr[i] == p_lk) return c->tree->n_otu + i;
  if (c->nbuf == c->bufcap) { fprintf(stderr, "glue_driver: more partial buffers than the instance holds\n"); exit(5); }
  c->bufptr[c->nbuf] = p_lk;
  return c->tree->n_otu + c->nbuf++;
}
static int mat_id(ctx_t *c, const void *pij)
{
  for (int i = 0; i < c->nmat; ++i) if (c->matptr[i] == pij) return i;
  if (c->nmat == c->matcap) { fprintf(stderr, "glue_driver: more matrices than the instance holds\n"); exit(5); }
  c->matptr[c->nmat] = pij;
  return c->nmat++;
}
/* class k > 0: the pointer must stand for the id class 0 just used */
static void mirror_buf(ctx_t *c, const void *p_lk, int id)
{
  const int i = id - c->tree->n_otu;
  if (i < 0) return; /* tip */
  for (int j = 0; j < c->nbuf; ++j)
    if (c->bufptr[j] == p_lk)
    {
      if (j == i) return;
      fprintf(stderr, "glue_driver: class tree %d diverged from class 0 (its buffer has id %d, class 0 used %d)\n", c->cls,
              j + c->tree->n_otu, id);
      exit(6);
    }
  if (c->bufptr[i] != NULL)
  { fprintf(stderr, "glue_driver: class tree %d diverged from class 0 (partials buffer id %d already taken)\n", c->cls, id); exit(6); }
  c->bufptr[i] = p_lk;
  if (i >= c->nbuf) c->nbuf = i + 1;
}
static void mirror_mat(ctx_t *c, const void *pij, int id)
{
  if (c->matptr[id] == NULL) { c->matptr[id] = pij; if (id >= c->nmat) c->nmat = id + 1; return; }
  if (c->matptr[id] != pij)
  {
    int at = -1;
    for (int j = 0; j < c->nmat; ++j) if (c->matptr[j] == pij) at = j;
    fprintf(stderr, "glue_driver: class tree %d diverged from class 0 (matrix id %d holds %p, call has %p which is id %d; nmat %d)\n",
            c->cls, id, c->matptr[id], pij, at, c->nmat);
    exit(6);
  }
}
static void edge_sides(ctx_t *c, const t_edge *b, int *left, int *right)
{ /* src/lk.c:605-606 */
  *left  = buf_id(c, b->p_lk_left);
  *right = b->rght->tax ? b->rght->num : buf_id(c, b->p_lk_rght);
}

static void push_model(ctx_t *c)
{ /* update_beagle_ras / _efrqs / _eigen, src/beagle_utils.c:273-395 */
  t_tree *tree = c->tree;
  t_mod  *m = tree->mod;
  if (c->K > 0)
  { /* class axis: this class tree's rate, frequencies and eigen system go to class slot c->cls of the shared instance */
    static double rates[kMaxClasses];
    rates[c->cls] = m->ras->gamma_rr->v[0] * tree->mixt_tree->mod->ras->gamma_rr->v[m->ras->parent_class_number];
    if (c->cls == c->K - 1) OK(phyhip_set_category_rates(c->inst, rates)); /* (callers push the classes in order) */
    OK(phyhip_set_state_frequencies(c->inst, c->cls, m->e_frq->pi->v));
    OK(phyhip_set_eigen_decomposition(c->inst, c->cls, m->eigen->r_e_vect, m->eigen->l_e_vect, m->eigen->e_val));
    if (c->cls == 0)
    {
      OK(phyhip_set_phyml_options(c->inst, m->l_min, m->l_max, m->br_len_mult->v, tree->apply_lk_scaling));
      OK(phyhip_set_invariant_sites(c->inst, m->ras->invar, m->ras->pinvar->v, tree->data->invar));
    }
    return;
  }
  if (tree->mixt_tree)
  { /* class tree: one category whose rate is the class rate of the mixture (src/lk.c:2298) */
    const double rate = m->ras->gamma_rr->v[0] * tree->mixt_tree->mod->ras->gamma_rr->v[m->ras->parent_class_number], one = 1.0;
    OK(phyhip_set_category_rates(c->inst, &rate));
    OK(phyhip_set_category_weights(c->inst, 0, &one));
  }
  else
  {
    OK(phyhip_set_category_rates(c->inst, m->ras->gamma_rr->v));
    OK(phyhip_set_category_weights(c->inst, 0, m->ras->gamma_r_proba->v));
  }
  OK(phyhip_set_state_frequencies(c->inst, 0, m->e_frq->pi->v));
  OK(phyhip_set_eigen_decomposition(c->inst, 0, m->eigen->r_e_vect, m->eigen->l_e_vect, m->eigen->e_val));
  OK(phyhip_set_phyml_options(c->inst, m->l_min, m->l_max, m->br_len_mult->v, tree->apply_lk_scaling));
  OK(phyhip_set_invariant_sites(c->inst, m->ras->invar, m->ras->pinvar->v, tree->data->invar));
}

static ctx_t *ensure_instance(t_tree *tree)
{ /* create_beagle_instance, src/beagle_utils.c:97-190: after Make_Tree_For_Lk (src/main.c:235) */
  for (int i = 0; i < g_nctx; ++i) if (g_ctx[i].tree == tree) return &g_ctx[i];
  if (tree->is_mixt_tree || tree->n_root || tree->mod->gamma_mgf_bl == YES || tree->mod->log_l == YES ||
      (tree->mod->use_m4mod && tree->mixt_tree))
  { fprintf(stderr, "glue_driver: unsupported tree kind (rooted / mgf / log_l / m4 class tree, or the mixture tree itself)\n"); exit(5); }
  if (!tree->mixt_tree && g_nctx > 0)
  { /* a fresh tree object for the same data (aLRT_From_String rebuilds the tree from its Newick string and calls
       Make_Tree_For_Lk again; under BEAGLE it creates a fresh instance there too, src/utilities.c:9314): the old instance goes */
    /* (the replaced tree object may already be freed -- bootstrap replicates: never look into it) */
    if (g_nctx != 1 || g_ctx[0].is_class_tree) { fprintf(stderr, "glue_driver: an ordinary tree after class trees\n"); exit(5); }
    OK(phyhip_finalize_instance(g_ctx[0].inst));
    g_nctx = 0;
  }
  if (g_nctx == MAXCTX) { fprintf(stderr, "glue_driver: too many class trees\n"); exit(5); }
  if (g_class_axis && tree->mixt_tree && (tree->mod->ns == 20 || tree->mod->ns == 4))
  { /* all class trees of the mixture at once: contexts in class order, one instance */
    int K = 0; /* class trees that compute (the invariant class of a +I mixture never reaches the surface) */
    for (t_tree *t = tree->mixt_tree->next; t && t->is_mixt_tree == NO; t = t->next) if (t->mod->ras->invar == NO) ++K;
    /* (what the class axis is built for: 20 states x up to 4 classes, 4 states x 1, 2 or 4 classes) */
    if (K >= 1 && K <= 4 && !(tree->mod->ns == 4 && K == 3) && g_nctx + K <= MAXCTX && tree->mod->ras->invar == NO)
    {
      ctx_t *first = NULL, *mine = NULL;
      int    k = 0;
      for (t_tree *t = tree->mixt_tree->next; t && t->is_mixt_tree == NO; t = t->next)
      {
        if (t->mod->ras->invar == YES) continue;
        ctx_t *c = &g_ctx[g_nctx++];
        memset(c, 0, sizeof *c);
        c->tree = t; c->cls = k; c->K = K; c->is_class_tree = 1;
        c->bufcap = 3 * t->n_otu - 2; c->matcap = 2 * t->n_otu - 1;
        if (k == 0)
        {
          first = c;
          c->batch = calloc(1, sizeof(clsbatch_t));
          ++g_n_created;
          c->inst = phyhip_create_instance(t->n_otu, t->n_otu + c->bufcap, 0, t->mod->ns, t->data->n_pattern, K, c->matcap, K, 0, NULL, 0,
                                           0, PHYHIP_FLAG_CLASS_AXIS, NULL);
          if (c->inst < 0) die("phyhip_create_instance (class axis)");
          OK(phyhip_set_pattern_weights(c->inst, t->data->wght));
          for (int i = 0; i < t->n_otu; ++i) OK(phyhip_set_tip_partials(c->inst, t->a_nodes[i]->num, t->a_nodes[i]->b[0]->p_lk_tip_r));
        }
        else c->inst = first->inst;
        if (t == tree) mine = c;
        ++k;
      }
      for (ctx_t *c = first; c < first + K; ++c) push_model(c);
      return mine;
    }
  }
  ctx_t *c = &g_ctx[g_nctx++];
  memset(c, 0, sizeof *c);
  ++g_n_created;
  const int n = tree->n_otu, P = tree->data->n_pattern, S = tree->mod->ns, C = tree->mod->ras->n_catg;
  c->tree = tree;
  c->is_class_tree = tree->mixt_tree != NULL;
  c->bufcap = 3 * n - 2;  /* internal edge sides + both sides of the two spare SPR edges (src/make.c:96-104) */
  c->matcap = 2 * n - 1;
  if (c->bufcap > MAXID || c->matcap > MAXID) { fprintf(stderr, "glue_driver: tree too large for the index tables\n"); exit(5); }
  {
    /* GLUE_DEVICES="0,1,2,3": the sharded (multi-GPU) instance of include/phyhip.h -- the resource list PhyML would pass
       where the BEAGLE glue passes its own (src/beagle_utils.c:119-133).  Plain tree models only. */
    int devs[64], nd = 0;
    const char *e = getenv("GLUE_DEVICES");
    if (e && !tree->is_mixt_tree && !tree->mixt_tree)
      for (const char *p = e; *p && nd < 64;)
      {
        devs[nd++] = atoi(p);
        while (*p && *p != ',') ++p;
        if (*p == ',') ++p;
      }
    c->inst = phyhip_create_instance(n, n + c->bufcap, 0, S, P, 1, c->matcap, C, 0, nd ? devs : NULL, nd,
                                     0, (nd == 1 ? PHYHIP_FLAG_SHARDED : 0) |
                                            /* `--cov` (mod->use_m4mod): the reference's generic loop, src/lk.c:1303-1324 */
                                            (tree->mod->use_m4mod ? PHYHIP_FLAG_GENERIC_LOOP : 0), NULL);
  }
  if (c->inst < 0) die("phyhip_create_instance");
  OK(phyhip_set_pattern_weights(c->inst, tree->data->wght));
  for (int t = 0; t < n; ++t) OK(phyhip_set_tip_partials(c->inst, tree->a_nodes[t]->num, tree->a_nodes[t]->b[0]->p_lk_tip_r));
  push_model(c);
  return c;
}

static void first_bad(double d, double dev, double ref); /* GLUE_FIRST_BAD=1 (below) */
static void track(double *worst, double dev, double ref, double floor_)
{
  const double d = fabs(dev - ref) / fmax(floor_, fabs(ref));
  if (d > *worst) *worst = d;
  if (d > 1e-8) first_bad(d, dev, ref);
}

/* GLUE_FIRST_BAD=1 (check mode, developer aid): the first compared scalar that is off by more than 1e-8, with the call counts so far
   and the instance's virtual-buffer counters */
#define TRN 64
static struct { char k; int a, b, c, d, e; double x; } g_tr[TRN];
static unsigned long g_trn = 0;
static void tr(char k, int a, int b, int c, int d, int e, double x)
{
  g_tr[g_trn % TRN].k = k; g_tr[g_trn % TRN].a = a; g_tr[g_trn % TRN].b = b; g_tr[g_trn % TRN].c = c; g_tr[g_trn % TRN].d = d;
  g_tr[g_trn % TRN].e = e; g_tr[g_trn % TRN].x = x; ++g_trn;
}
static void first_bad(double d, double dev, double ref)
{
  static int done = 0;
  if (done || !getenv("GLUE_FIRST_BAD")) return;
  done = 1;
  for (unsigned long i = g_trn > TRN ? g_trn - TRN : 0; i < g_trn; ++i)
    fprintf(stderr, "GLUE_TRACE %lu %c %d %d %d %d %d %.6g\n", i, g_tr[i % TRN].k, g_tr[i % TRN].a, g_tr[i % TRN].b, g_tr[i % TRN].c, g_tr[i % TRN].d,
            g_tr[i % TRN].e, g_tr[i % TRN].x);
  long long vs[4] = {0, 0, 0, 0};
  if (g_nctx > 0) phyhip_get_virtual_stats(g_ctx[0].inst, vs);
  fprintf(stderr, "GLUE_FIRST_BAD rel %.3g device %.17g reference %.17g | calls so far: Lk %ld (full %ld) Update_Partial_Lk %ld dLk %ld Update_PMat %ld "
                  "Update_Eigen_Lr %ld | virtual now %lld skipped %lld reissued %lld stored %lld\n",
          d, dev, ref, g_n_lk, g_n_lk_full, g_n_upd, g_n_dlk, g_n_pmat, g_n_eig, vs[0], vs[1], vs[2], vs[3]);
}

/* ---- the interposed surface ---------------------------------------------------------------------------------- */

/* Bootstrap (src/utilities.c:3884-4110) builds a new tree object per replicate that ALIASES the original's buffers
   (Share_Lk_Struct, :4046), rewrites the tip vectors in place (Init_Partial_Lk_Tips_Double, :4050) and carries resampled
   pattern weights; the object of the previous replicate has been freed and its address may be reused.  So when the tips of
   a tree are (re)initialised, whatever instance stood for that address is dropped; the next surface call creates a fresh
   one with the new tips and weights (the re-uploadable weights / tips of SURVEY Appendix A). */
void Init_Partial_Lk_Tips_Double(t_tree *tree)
{
  static void (*real)(t_tree *) = NULL;
  if (!real) real = (void (*)(t_tree *))dlsym(RTLD_NEXT, "Init_Partial_Lk_Tips_Double");
  real(tree);
  if (g_host) return;
  for (int i = 0; i < g_nctx; ++i)
    if (g_ctx[i].tree == tree)
    {
      OK(phyhip_finalize_instance(g_ctx[i].inst));
      g_ctx[i] = g_ctx[--g_nctx];
      ++g_n_dropped;
      break;
    }
}

/* Leave-one-out cross-validation (MIXT_Maxfold_Cv, src/mixt.c:4198-4290): one character of one tip is hidden, the pendant
   edge optimised, the character restored -- Init_Partial_Lk_Tips_Double_One_Character (src/lk.c:2092) rewrites ONE pattern of
   the tip vector in place.  The mixture tree's call fans out to the class trees (src/mixt.c:2674-2686) and comes back here. */
static long g_n_tipchar = 0, g_n_cv_vec = 0, g_n_cv_bad = 0;
void Init_Partial_Lk_Tips_Double_One_Character(int node_id, int site, t_tree *tree)
{
  static void (*real)(int, int, t_tree *) = NULL;
  if (!real) real = (void (*)(int, int, t_tree *))dlsym(RTLD_NEXT, "Init_Partial_Lk_Tips_Double_One_Character");
  real(node_id, site, tree);
  if (g_host || tree->is_mixt_tree == YES) return;
  for (int i = 0; i < g_nctx; ++i)
    if (g_ctx[i].tree == tree)
    {
      if (g_ctx[i].K > 0 && g_ctx[i].cls > 0) return; /* class axis: the classes share ONE instance, class 0 pushed it */
      OK(phyhip_set_tip_partials_at_pattern(g_ctx[i].inst, node_id, site, tree->a_nodes[node_id]->b[0]->p_lk_tip_r + (size_t)site * tree->mod->ns));
      ++g_n_tipchar;
      return;
    }
}

/* ... and what the loop reads afterwards (CV_State_Probs_Core, src/cv.c:294-420): p_lk_left of the pendant edge of every
   class tree (or of the tree) at the hidden site.  Check mode: the device's vector is downloaded (phyhip_get_partials, the hook
   of SURVEY 8f rank 3) and compared BIT FOR BIT with the host's at that site. */
void CV_State_Probs_Core(phydbl **state_probs, short int **truth, phydbl **site_loglk, phydbl **weights, int *n_prob_vectors,
                         int tax_id, int site, int true_d_state, phydbl patt_weight, t_tree *tree)
{
  static void (*real)(phydbl **, short int **, phydbl **, phydbl **, int *, int, int, int, phydbl, t_tree *) = NULL;
  if (!real) real = (void (*)(phydbl **, short int **, phydbl **, phydbl **, int *, int, int, int, phydbl, t_tree *))dlsym(RTLD_NEXT, "CV_State_Probs_Core");
  if (!g_host && g_check)
    for (t_tree *t = tree->is_mixt_tree == YES ? tree->next : tree; t && t->is_mixt_tree == NO; t = (tree->is_mixt_tree == YES) ? t->next : NULL)
    {
      if (t->mod->ras->invar == YES) continue;
      ctx_t *c = NULL;
      for (int i = 0; i < g_nctx; ++i) if (g_ctx[i].tree == t) c = &g_ctx[i];
      if (!c) continue;
      const t_edge *b = t->a_nodes[tax_id]->b[0];
      int id = -1;
      for (int i = 0; i < c->nbuf; ++i) if (c->bufptr[i] == (const void *)b->p_lk_left) id = t->n_otu + i;
      if (id < 0) continue; /* (never a destination on the device: nothing to compare) */
      const int ns = t->mod->ns, P = t->data->n_pattern, K = c->K > 0 ? c->K : 1, C = K > 1 ? K : t->mod->ras->n_catg;
      double   *v = (double *)malloc(sizeof(double) * (size_t)P * C * ns);
      OK(phyhip_get_partials(c->inst, id, PHYHIP_OP_NONE, v));
      const int     ncat = t->mod->ras->n_catg; /* host layout [site][catg][state]; class axis: the class is the category */
      const double *dev = v + ((size_t)site * C + (K > 1 ? c->cls : 0)) * ns, *host = b->p_lk_left + (size_t)site * ncat * ns;
      ++g_n_cv_vec;
      if (memcmp(dev, host, sizeof(double) * (size_t)(K > 1 ? 1 : ncat) * ns) != 0) ++g_n_cv_bad;
      free(v);
    }
  real(state_probs, truth, site_loglk, weights, n_prob_vectors, tax_id, site, true_d_state, patt_weight, tree);
}

void Update_PMat_At_Given_Edge(t_edge *b_fcus, t_tree *tree)
{
  static void (*real)(t_edge *, t_tree *) = NULL;
  if (!real) real = (void (*)(t_edge *, t_tree *))dlsym(RTLD_NEXT, "Update_PMat_At_Given_Edge");
  ++g_n_pmat;
  if (g_host || tree->is_mixt_tree) { real(b_fcus, tree); return; } /* mixture tree: the original loops over the class trees */
  if (tree->mixt_tree && tree->mod->ras->invar == YES) { real(b_fcus, tree); return; } /* invariant class: not on the device */
  const unsigned long long hp_w = hp_tick();
  ctx_t *c = ensure_instance(tree);
  if (b_fcus->has_zero_br_len == YES) { fprintf(stderr, "glue_driver: zero-length edge flag not supported\n"); exit(5); }
  if (c->K > 0)
  { /* class axis: class 0 names the matrix; the K class blocks are uploaded together when the last class has built its own
       (or built on the device, per class, from one call: the class trees share the edge length) */
    int id = -1;
    if (c->cls == 0) id = g_cls_mat = mat_id(c, b_fcus->Pij_rr);
    else
    { /* known pointer: its id (some flows walk one class tree edge by edge); new pointer: the id class 0 just named */
      for (int j = 0; j < c->nmat; ++j) if (c->matptr[j] == b_fcus->Pij_rr) id = j;
      if (id < 0) { mirror_mat(c, b_fcus->Pij_rr, g_cls_mat); id = g_cls_mat; }
    }
    if (g_device_pmat && !g_check)
    {
      const double len = b_fcus->l->v; /* the class trees share the edge length: any class's call rebuilds all classes */
      OK(phyhip_update_transition_matrices(c->inst, 0, &id, NULL, NULL, &len, 1));
      return;
    }
    real(b_fcus, tree);
    {
      const int ss = tree->mod->ns * tree->mod->ns;
      static double blk[4 * 400];
      ctx_t *c0 = c - c->cls;
      for (int k = 0; k < c->K; ++k) if (!(c0 + k)->matptr[id]) return; /* not every class has named this matrix yet */
      for (int k = 0; k < c->K; ++k) memcpy(blk + (size_t)k * ss, (c0 + k)->matptr[id], sizeof(double) * ss);
      OK(phyhip_set_transition_matrix(c->inst, id, blk, -1.0)); /* (queued; the last upload of a matrix wins) */
    }
    return;
  }
  const int m = mat_id(c, b_fcus->Pij_rr);
  if (g_device_pmat && !g_check)
  { /* src/lk.c:2344: matrices built on the device from (U, lambda, U^-1, length) */
    const double len = b_fcus->l->v;
    TOK(HP_ABI_MATRIX, phyhip_update_transition_matrices(c->inst, 0, &m, NULL, NULL, &len, 1));
    hp_add(HP_WRAP_UPDATE_PMAT, hp_w);
    return;
  }
  {
    const unsigned long long t_ = hp_tick();
    real(b_fcus, tree); /* host PMat (src/models.c:257-373) into b->Pij_rr ... */
    hp_add(HP_HOST_PMAT, t_);
  }
  tr('P', m, 0, 0, 0, 0, b_fcus->l->v);
  TOK(HP_ABI_MATRIX, phyhip_set_transition_matrix(c->inst, m, b_fcus->Pij_rr, -1.0)); /* ... and the upload of src/lk.c:2360 */
  hp_add(HP_WRAP_UPDATE_PMAT, hp_w);
}

void Update_Partial_Lk(t_tree *tree, t_edge *b, t_node *d)
{
  static void (*real)(t_tree *, t_edge *, t_node *) = NULL;
  if (!real) real = (void (*)(t_tree *, t_edge *, t_node *))dlsym(RTLD_NEXT, "Update_Partial_Lk");
  /* src/lk.c:1285-1297 */
  if (b->left == d && b->update_partial_lk_left == NO) return;
  if (b->rght == d && b->update_partial_lk_rght == NO) return;
  if (tree->is_mixt_tree) { real(tree, b, d); return; } /* MIXT_Update_Partial_Lk: per class tree, back through this wrapper */
  /* `--alias_subpatt` (src/lk.c:1294-1296): the application's own bookkeeping, at the place the reference calls it.  Nothing
     of the likelihood reads what it writes (include/phyhip_lk.h), so the device path needs nothing from it.  (When real()
     runs below -- host mode, check mode on an internal node -- it makes this call itself: not twice.) */
  const int alias_here = tree->io->do_alias_subpatt == YES && tree->update_alias_subpatt == YES &&
                         !(g_host || (g_check && !d->tax) || (tree->mixt_tree && tree->mod->ras->invar == YES));
  if (alias_here) { Alias_One_Subpatt((d == b->left) ? b->rght : b->left, d, tree); ++g_n_alias; }
  if (d->tax) return;
  ++g_n_upd;
  if (g_host || (tree->mixt_tree && tree->mod->ras->invar == YES)) { real(tree, b, d); return; }
  const unsigned long long hp_w = hp_tick();
  ctx_t *c = ensure_instance(tree);
  t_node *n_v1 = NULL, *n_v2 = NULL;
  phydbl *p_lk = NULL, *p_lk_v1 = NULL, *p_lk_v2 = NULL, *Pij1 = NULL, *Pij2 = NULL, *tPij1 = NULL, *tPij2 = NULL;
  int    *sum_scale = NULL, *sum_scale_v1 = NULL, *sum_scale_v2 = NULL, *p_lk_loc = NULL;
  Set_All_Partial_Lk(&n_v1, &n_v2, &p_lk, &sum_scale, &p_lk_loc, &Pij1, &tPij1, &p_lk_v1, &sum_scale_v1, &Pij2, &tPij2,
                     &p_lk_v2, &sum_scale_v2, d, b, tree);
  if (c->K > 0 && c->cls > 0)
  { /* class axis, class k > 0: the operation class 0 queued covers this class; only register / verify the pointers */
    clsbatch_t *q = (clsbatch_t *)(c - c->cls)->batch;
    if (q->pos[c->cls] >= q->nops) { fprintf(stderr, "glue_driver: class tree %d is ahead of class 0\n", c->cls); exit(6); }
    const phyhip_operation *o = &q->ops[q->pos[c->cls]++];
    mirror_buf(c, p_lk, o->destinationPartials);
    c->scaleptr[o->destinationPartials - tree->n_otu] = sum_scale;
    if (!n_v1->tax) mirror_buf(c, p_lk_v1, o->child1Partials);
    if (!n_v2->tax) mirror_buf(c, p_lk_v2, o->child2Partials);
    if ((n_v1->tax ? n_v1->num : -1) != (o->child1Partials < tree->n_otu ? o->child1Partials : -1) ||
        (n_v2->tax ? n_v2->num : -1) != (o->child2Partials < tree->n_otu ? o->child2Partials : -1))
    { fprintf(stderr, "glue_driver: class tree %d diverged from class 0 (tip child)\n", c->cls); exit(6); }
    if (g_check) real(tree, b, d);
    return;
  }
  phyhip_operation op;
  op.destinationPartials = buf_id(c, p_lk);
  c->scaleptr[op.destinationPartials - tree->n_otu] = sum_scale;
  op.destinationScaleWrite = op.destinationScaleRead = PHYHIP_OP_NONE;
  op.child1Partials = n_v1->tax ? n_v1->num : buf_id(c, p_lk_v1);
  op.child1TransitionMatrix = mat_id(c, Pij1);
  op.child2Partials = n_v2->tax ? n_v2->num : buf_id(c, p_lk_v2);
  op.child2TransitionMatrix = mat_id(c, Pij2);
  tr('U', op.destinationPartials, op.child1Partials, op.child2Partials, op.child1TransitionMatrix, op.child2TransitionMatrix, 0.0);
  TOK(HP_ABI_UPDATE_PARTIALS, phyhip_update_partials(c->inst, &op, 1, PHYHIP_OP_NONE));
  hp_add(HP_WRAP_UPDATE_PARTIAL_LK, hp_w);
  if (c->K > 0)
  {
    clsbatch_t *q = (clsbatch_t *)c->batch;
    if (q->nops == MAXCLSOPS) { fprintf(stderr, "glue_driver: too many partial updates between two mixture evaluations\n"); exit(5); }
    q->ops[q->nops++] = op;
  }
  if (g_check) real(tree, b, d);
}

void Update_Eigen_Lr(t_edge *b, t_tree *tree)
{
  static void (*real)(t_edge *, t_tree *) = NULL;
  if (!real) real = (void (*)(t_edge *, t_tree *))dlsym(RTLD_NEXT, "Update_Eigen_Lr");
  ++g_n_eig;
  if (g_host || tree->is_mixt_tree || (tree->mixt_tree && tree->mod->ras->invar == YES)) { real(b, tree); return; }
  ctx_t *c = ensure_instance(tree);
  if (c->K > 0 && c->cls > 0) { if (g_check) real(b, tree); return; } /* class axis: class 0's call covered every class */
  int l, r;
  edge_sides(c, b, &l, &r);
  tr('E', l, r, 0, 0, 0, 0.0);
  TOK(HP_ABI_UPDATE_EIGEN_LR, phyhip_update_eigen_lr(c->inst, l, r));
  if (g_check) real(b, tree);
}

static void site_outputs(t_tree *tree, ctx_t *c)
{ /* what Lk_Core leaves per pattern for host readers (src/lk.c:855-857): device mode writes them where alrt.c / io.c look,
     check mode compares them with what the original just wrote */
  const int P = tree->data->n_pattern, C = tree->mod->ras->n_catg;
  ++g_n_site_dl;
  if (!g_check)
  {
    OK(phyhip_get_site_outputs(c->inst, tree->c_lnL_sorted, tree->cur_site_lk, tree->unscaled_site_lk_cat, tree->fact_sum_scale));
    return;
  }
  double *lnl = malloc(sizeof(double) * P), *cat = malloc(sizeof(double) * P * C);
  int    *f = malloc(sizeof(int) * P);
  OK(phyhip_get_site_outputs(c->inst, lnl, NULL, cat, f));
  for (int p = 0; p < P; ++p)
  {
    if (tree->data->wght[p] <= SMALL) continue; /* skipped by both (src/lk.c:627) */
    track(&g_worst_site_lnl, lnl[p], tree->c_lnL_sorted[p], 1.0);
    if (f[p] != tree->fact_sum_scale[p]) { fprintf(stderr, "glue_driver: scale exponent of pattern %d differs\n", p); exit(6); }
    for (int k = 0; k < C; ++k) track(&g_worst_site_lnl, cat[p * C + k], tree->unscaled_site_lk_cat[p * C + k], 1e-300);
  }
  free(lnl); free(cat); free(f);
}

static double device_edge_value(t_tree *tree, const t_edge *b)
{
  ctx_t *c = ensure_instance(tree);
  double lnl = 0.0;
  if (tree->use_eigen_lr == YES) { TOK(HP_ABI_EIGEN_EVAL, phyhip_calculate_eigen_lnl(c->inst, b->l->v, &lnl)); tr('G', 0, 0, 0, 0, 0, lnl); } /* src/lk.c:592-603 */
  else
  {
    int l, r, pm = mat_id(c, b->Pij_rr), zero = 0;
    edge_sides(c, b, &l, &r);
    TOK(HP_ABI_EDGE_LNL, phyhip_calculate_edge_log_likelihoods(c->inst, &l, &r, &pm, NULL, NULL, &zero, &zero, NULL, 1, &lnl, NULL, NULL));
    tr('L', l, r, pm, 0, 0, lnl);
    if (g_site_outputs) site_outputs(tree, c);
  }
  return lnl;
}

/* the class trees of a mixture and what the combination needs from each (src/mixt.c:1048-1053) */
typedef struct
{
  int    K, ids[kMaxClasses], lft[kMaxClasses], rgt[kMaxClasses], pms[kMaxClasses];
  double proba[kMaxClasses], rw[kMaxClasses], ew[kMaxClasses], r_sum, e_sum, sum_p;
} mix_t;
static int gather_classes(t_edge *e, t_tree *mixt_tree, mix_t *m)
{
  m->K = 0;
  t_edge *b = e->next;
  t_tree *inv_tree = NULL;
  for (t_tree *t = mixt_tree->next; t && t->is_mixt_tree == NO; t = t->next, b = b->next)
  {
    if (t->mod->ras->invar == YES) { inv_tree = t; continue; } /* the invariant class: no class tree on the device (src/mixt.c:899) */
    if (m->K == kMaxClasses) return 0;
    ctx_t *c = ensure_instance(t);
    push_model(c);
    m->ids[m->K] = c->inst;
    if (c->K > 0 && c->cls > 0) m->lft[m->K] = m->rgt[m->K] = m->pms[m->K] = 0; /* class axis: class 0's ids stand for all */
    else
    {
      edge_sides(c, b, &m->lft[m->K], &m->rgt[m->K]);
      m->pms[m->K] = mat_id(c, b->Pij_rr);
    }
    m->proba[m->K] = mixt_tree->mod->ras->gamma_r_proba->v[t->mod->ras->parent_class_number];
    m->rw[m->K] = t->mod->r_mat_weight->v; m->ew[m->K] = t->mod->e_frq_weight->v;
    ++m->K;
  }
  m->r_sum = MIXT_Get_Sum_Chained_Scalar_Dbl(mixt_tree->next->mod->r_mat_weight);
  m->e_sum = MIXT_Get_Sum_Chained_Scalar_Dbl(mixt_tree->next->mod->e_frq_weight);
  m->sum_p = MIXT_Get_Sum_Of_Probas_Across_Mixtures(m->r_sum, m->e_sum, mixt_tree);
  if (m->K == 0) return 0;
  /* +I mixture (src/mixt.c:1079-1112): pinvar of the mixture, the constant-state table of the element, the invariant
     class tree's frequencies */
  if (mixt_tree->mod->ras->invar == YES && !inv_tree) return 0;
  OK(phyhip_set_mixture_invariant_sites(m->ids[0], mixt_tree->mod->ras->invar == YES, mixt_tree->mod->ras->pinvar->v,
                                        mixt_tree->data->invar, inv_tree ? inv_tree->mod->e_frq->pi->v : NULL));
  return 1;
}
static t_tree *first_class_tree(t_tree *mixt_tree)
{ /* the first class tree that computes: the invariant class of a +I mixture comes first in the chain (src/xml.c) */
  t_tree *t = mixt_tree->next;
  while (t && t->is_mixt_tree == NO && t->mod->ras->invar == YES) t = t->next;
  return t;
}
/* the two mixture evaluations, over class instances or on the class axis of one instance */
static void mixture_lnl(const mix_t *m, t_tree *mixt_tree, double *lnl)
{
  ctx_t *c0 = ensure_instance(first_class_tree(mixt_tree));
  if (c0->K > 0)
  {
    cls_batch_done((clsbatch_t *)c0->batch, c0->K);
    OK(phyhip_calculate_class_mixture_log_likelihood(c0->inst, m->lft[0], m->rgt[0], m->pms[0], m->proba, m->rw, m->ew, m->r_sum,
                                                     m->e_sum, m->sum_p, lnl));
  }
  else
    OK(phyhip_calculate_mixture_log_likelihood(m->ids, m->K, m->lft, m->rgt, m->pms, m->proba, m->rw, m->ew, m->r_sum, m->e_sum,
                                               m->sum_p, lnl));
}
static void mixture_dlnl(const mix_t *m, t_tree *mixt_tree, double *l, double *lnl, double *dlnl)
{
  ctx_t *c0 = ensure_instance(first_class_tree(mixt_tree));
  if (c0->K > 0)
  {
    cls_batch_done((clsbatch_t *)c0->batch, c0->K);
    OK(phyhip_calculate_class_mixture_eigen_lnl_dlnl(c0->inst, m->lft[0], m->rgt[0], l, m->proba, m->rw, m->ew, m->r_sum, m->e_sum,
                                                     m->sum_p, lnl, dlnl));
  }
  else
    OK(phyhip_calculate_mixture_eigen_lnl_dlnl(m->ids, m->K, m->lft, m->rgt, l, m->proba, m->rw, m->ew, m->r_sum, m->e_sum, m->sum_p,
                                               lnl, dlnl));
}
static int mixture_supported(const t_tree *mixt_tree)
{
  for (const t_tree *mt = mixt_tree; mt; mt = mt->next_mixt) /* every element of the data partition (src/mixt.c:862) */
    if (mt->n_root || mt->mod->gamma_mgf_bl == YES) return 0;
  return 1;
}

static double g_t0 = 0.0;
static long   g_max_mixt = 0; /* GLUE_MAX_MIXT: stop after this many compared MIXT_Lk calls (bounded test runs) */
static void report_xml_and_exit(void)
{
  printf("\nGLUE_DRIVER {\"mode\": \"%s\", \"xml\": 1, \"seconds\": %.3f, \"calls\": {\"Lk\": %ld, \"MIXT_Lk\": %ld, \"MIXT_dLk\": %ld, \"MIXT_skipped\": %ld, "
         "\"Update_Partial_Lk\": %ld, \"Update_PMat\": %ld, \"Update_Eigen_Lr\": %ld, \"dLk\": %ld}, \"class_instances\": %d, "
         "\"worst_rel_mixture_lnL\": %.3g, \"worst_rel_mixture_dlnL\": %.3g, \"last_mixture_lnL\": %.17g, \"best_full_lnL\": %.17g, "
         "\"tip_characters_rewritten\": %ld, \"cv_vectors_compared\": %ld, \"cv_vector_mismatches\": %ld}\n",
         g_host ? "host" : (g_check ? "check" : "device"), now_s() - g_t0, g_n_lk, g_n_mixt, g_n_mixt_dlk, g_n_mixt_skipped, g_n_upd, g_n_pmat, g_n_eig, g_n_dlk,
         g_nctx, g_worst_mixt, g_worst_mixt_dlnl, g_last_mixt_lnl, g_best_full_lnl, g_n_tipchar, g_n_cv_vec, g_n_cv_bad);
  fflush(stdout);
  _exit(0);
}

/* MIXT_Lk (src/mixt.c:730-1160).
   check mode: the original runs (its per-class matrix refreshes and partial updates reach the class instances through the
   wrappers above), then the device repeats the evaluation of the same edge over the class instances; results compared.
   device mode: the control flow of src/mixt.c:754-860 with the reference's own per-class loops (MIXT_Update_PMat_At_Given_Edge,
   MIXT_Post/Pre_Order_Lk, which come back through the wrappers), the device combination in place of the site loop. */
phydbl MIXT_Lk(t_edge *mixt_b, t_tree *mixt_tree)
{
  static phydbl (*real)(t_edge *, t_tree *) = NULL;
  if (!real) real = (phydbl (*)(t_edge *, t_tree *))dlsym(RTLD_NEXT, "MIXT_Lk");
  if (g_host) { g_last_mixt_lnl = real(mixt_b, mixt_tree); return g_last_mixt_lnl; }
  ++g_n_mixt;
  mix_t m;
  if (g_check)
  {
    const phydbl ref = real(mixt_b, mixt_tree);
    g_last_mixt_lnl = ref;
    if (!mixture_supported(mixt_tree) || mixt_tree->next->use_eigen_lr == YES)
    { ++g_n_mixt_skipped; return ref; } /* eigen-basis Lk and +I mixtures: not on the device */
    double lnl = 0.0;
    {
      t_edge *mb = mixt_b;
      for (t_tree *mt = mixt_tree; mt; mt = mt->next_mixt, mb = mb ? mb->next_mixt : NULL)
      { /* the elements of the data partition: c_lnL is their sum (src/mixt.c:862,1161-1176) */
        t_edge *e = mb ? mb : mt->a_nodes[0]->b[0]; /* src/mixt.c:889 */
        double  x = 0.0;
        if (!gather_classes(e, mt, &m)) { ++g_n_mixt_skipped; return ref; }
        mixture_lnl(&m, mt, &x);
        lnl += x;
      }
    }
    track(&g_worst_mixt, lnl, ref, 1.0);
    if (g_max_mixt && g_n_mixt + g_n_mixt_dlk - g_n_mixt_skipped >= g_max_mixt) report_xml_and_exit();
    return ref;
  }
  /* ---- device mode ---- */
  if (!mixture_supported(mixt_tree) || mixt_tree->next->use_eigen_lr == YES)
  { fprintf(stderr, "glue_driver: this mixture (rooted / partitioned / +I / eigen-basis Lk) runs in check mode only\n"); exit(5); }
  if (!mixt_b)
  { /* src/mixt.c:754-782: every element of the partition, every class tree of the chain */
    for (t_tree *mt = mixt_tree; mt; mt = mt->next_mixt) Update_RAS(mt->mod);
    for (t_tree *t = mixt_tree->next; t; t = t->next)
      if (t->is_mixt_tree == NO && (!Update_Boundaries(t->mod) || !Update_Efrq(t->mod) || !Update_Eigen(t->mod)))
      { fprintf(stderr, "glue_driver: model update failed\n"); exit(5); }
    if (g_device_pmat) /* device-built matrices need the refreshed class rates / eigen systems on the device first */
      for (t_tree *t = mixt_tree->next; t; t = t->next) if (t->is_mixt_tree == NO && t->mod->ras->invar == NO) push_model(ensure_instance(t));
    for (int br = 0; br < 2 * mixt_tree->n_otu - 3; ++br) MIXT_Update_PMat_At_Given_Edge(mixt_tree->a_edges[br], mixt_tree); /* :784-787 */
    MIXT_Post_Order_Lk(mixt_tree->a_nodes[0], mixt_tree->a_nodes[0]->v[0], mixt_tree);                                    /* :850-857 */
    if (mixt_tree->both_sides == YES) MIXT_Pre_Order_Lk(mixt_tree->a_nodes[0], mixt_tree->a_nodes[0]->v[0], mixt_tree);
  }
  else
  {
    if (g_device_pmat)
      for (t_tree *t = mixt_tree->next; t; t = t->next) if (t->is_mixt_tree == NO && t->mod->ras->invar == NO) push_model(ensure_instance(t));
    MIXT_Update_PMat_At_Given_Edge(mixt_b, mixt_tree); /* :806 */
  }
  double lnl = 0.0;
  {
    t_edge *mb = mixt_b;
    for (t_tree *mt = mixt_tree; mt; mt = mt->next_mixt, mb = mb ? mb->next_mixt : NULL)
    { /* the elements of the data partition (src/mixt.c:862-1159), their sum (:1161-1176) */
      t_edge *e = mb ? mb : mt->a_nodes[0]->b[0];
      t_edge *b = e->next;
      for (t_tree *t = mt->next; t && t->is_mixt_tree == NO; t = t->next, b = b->next)
      {
        t->c_lnL = 0.0; t->numerical_warning = NO;
        if (t->mod->ras->invar == YES) continue;              /* :899 */
        if (t->update_eigen_lr == YES) Update_Eigen_Lr(b, t); /* :929 */
      }
      if (!gather_classes(e, mt, &m)) { fprintf(stderr, "glue_driver: unsupported class tree in the mixture\n"); exit(5); }
      double x = 0.0;
      mixture_lnl(&m, mt, &x);
      lnl += x;
      mt->numerical_warning = NO;
      int w = 0;
      OK(phyhip_get_numerical_warning(m.ids[0], &w));
      if (w) mt->numerical_warning = YES;
    }
  }
  for (t_tree *mt = mixt_tree; mt; mt = mt->next_mixt) mt->c_lnL = lnl;
  g_last_mixt_lnl = lnl;
  if (!mixt_b && lnl > g_best_full_lnl) g_best_full_lnl = lnl;
  if (g_max_mixt && g_n_mixt + g_n_mixt_dlk >= g_max_mixt) report_xml_and_exit();
  return lnl;
}

static void report_xml_and_exit(void);
/* MIXT_dLk (src/mixt.c:2962-3340): check mode compares, device mode serves the call (MIXT_Update_Eigen_Lr loops over the
   class trees and comes back through the Update_Eigen_Lr wrapper). */
phydbl MIXT_dLk(phydbl *l, t_edge *mixt_b, t_tree *mixt_tree)
{
  static phydbl (*real)(phydbl *, t_edge *, t_tree *) = NULL;
  if (!real) real = (phydbl (*)(phydbl *, t_edge *, t_tree *))dlsym(RTLD_NEXT, "MIXT_dLk");
  if (g_host) return real(l, mixt_b, mixt_tree);
  ++g_n_mixt_dlk;
  mix_t m;
  if (g_check)
  {
    double       x = *l;
    const phydbl ref = real(l, mixt_b, mixt_tree);
    if (!mixture_supported(mixt_tree)) { ++g_n_mixt_skipped; return ref; }
    double lnl = 0.0, dlnl = 0.0;
    {
      t_edge *mb = mixt_b;
      for (t_tree *mt = mixt_tree; mt; mt = mt->next_mixt, mb = mb->next_mixt)
      { /* every element with the same length (src/mixt.c:3009-3030), sums at :3325-3360 */
        double a = 0.0, d = 0.0, xl = x;
        if (!gather_classes(mb, mt, &m)) { ++g_n_mixt_skipped; return ref; }
        mixture_dlnl(&m, mt, &xl, &a, &d);
        lnl += a; dlnl += d;
      }
    }
    track(&g_worst_mixt, lnl, ref, 1.0);
    track(&g_worst_mixt_dlnl, dlnl, mixt_tree->c_dlnL, 1.0);
    if (g_max_mixt && g_n_mixt + g_n_mixt_dlk - g_n_mixt_skipped >= g_max_mixt) report_xml_and_exit();
    return ref;
  }
  if (!mixture_supported(mixt_tree)) { fprintf(stderr, "glue_driver: this mixture runs in check mode only\n"); exit(5); }
  if (mixt_tree->update_eigen_lr == YES) MIXT_Update_Eigen_Lr(mixt_b, mixt_tree); /* src/mixt.c:2990-2991 */
  {
    t_edge *b = mixt_b; /* src/mixt.c:2993-3000: l must be the length of this edge */
    while (b && &(b->l->v) != l) b = b->next;
    if (!b) { fprintf(stderr, "glue_driver: MIXT_dLk with a length that is not the edge's own\n"); exit(5); }
  }
  double lnl = 0.0, dlnl = 0.0;
  {
    t_edge *mb = mixt_b;
    const double l_in = *l;
    for (t_tree *mt = mixt_tree; mt; mt = mt->next_mixt, mb = mb->next_mixt)
    {
      double a = 0.0, d = 0.0, xl = l_in;
      if (!gather_classes(mb, mt, &m)) { fprintf(stderr, "glue_driver: unsupported class tree in the mixture\n"); exit(5); }
      mixture_dlnl(&m, mt, mt == mixt_tree ? l : &xl, &a, &d);
      lnl += a; dlnl += d;
    }
  }
  for (t_tree *t = mixt_tree; t; t = t->next) { t->c_lnL = .0; t->c_dlnL = .0; } /* :3026-3032 */
  for (t_tree *mt = mixt_tree; mt; mt = mt->next_mixt) { mt->c_lnL = lnl; mt->c_dlnL = dlnl; } /* :3325-3360 */
  return mixt_tree->c_lnL;
}

phydbl Lk(t_edge *b, t_tree *tree)
{
  static phydbl (*real)(t_edge *, t_tree *) = NULL;
  if (!real) real = (phydbl (*)(t_edge *, t_tree *))dlsym(RTLD_NEXT, "Lk");
  ++g_n_lk;
  if (!b) { ++g_n_lk_full; tr('F', 0, 0, 0, 0, 0, 0.0); }
  if (g_host || tree->is_mixt_tree) return real(b, tree); /* mixture: src/lk.c:465-472 diverts to MIXT_Lk (above) */
  if (tree->mixt_tree && tree->mod->ras->invar == YES) return real(b, tree); /* the invariant class of a +I mixture: host only */
  if (tree->mixt_tree && !g_check) { fprintf(stderr, "glue_driver: Lk() on a class tree outside MIXT_Lk is served in check mode only\n"); exit(5); }
  if (g_check)
  {
    const phydbl ref = real(b, tree); /* drives the device through the nested surface calls as well */
    if (!b) push_model(ensure_instance(tree));
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
  if (!b) { const unsigned long long t_ = hp_tick(); push_model(ensure_instance(tree)); hp_add(HP_PUSH_MODEL, t_); }
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
    TOK(HP_ABI_WARNING, phyhip_get_numerical_warning(ensure_instance(tree)->inst, &w));
    if (w) tree->numerical_warning = YES;
  }
  return tree->c_lnL;
}

phydbl dLk(phydbl *l, t_edge *b, t_tree *tree)
{
  static phydbl (*real)(phydbl *, t_edge *, t_tree *) = NULL;
  if (!real) real = (phydbl (*)(phydbl *, t_edge *, t_tree *))dlsym(RTLD_NEXT, "dLk");
  ++g_n_dlk;
  if (g_host || tree->is_mixt_tree || tree->mixt_tree) return real(l, b, tree); /* mixture: src/lk.c:679-685 diverts to MIXT_dLk */
  ctx_t *c = ensure_instance(tree);
  if (g_check)
  {
    double x = *l, lnl = 0.0, dlnl = 0.0;
    const phydbl ref = real(l, b, tree); /* its Update_Eigen_Lr (if any) reaches the device through the wrapper */
    OK(phyhip_calculate_eigen_lnl_dlnl(c->inst, &x, &lnl, &dlnl));
    tr('D', 0, 0, 0, 0, 0, lnl);
    track(&g_worst_lnl, lnl, ref, 1.0);
    track(&g_worst_dlnl, dlnl, tree->c_dlnL, 1.0);
    return ref;
  }
  /* src/lk.c:655-753 */
  tree->numerical_warning = NO;
  if (tree->update_eigen_lr == YES) Update_Eigen_Lr(b, tree);
  double lnl = 0.0, dlnl = 0.0;
  TOK(HP_ABI_EIGEN_EVAL, phyhip_calculate_eigen_lnl_dlnl(c->inst, l, &lnl, &dlnl)); /* clamps *l like src/lk.c:672-673 */
  tree->c_dlnL = dlnl;
  tree->c_lnL  = lnl;
  return tree->c_lnL;
}

/* Host readers of device-resident state: ancestral.c walks p_lk_left/rght and sum_scale_left/rght of
   every edge (src/ancestral.c:677-869).  Device mode: every device buffer is downloaded to the host vector it stands
   for (phyhip_get_partials / phyhip_get_scale_factors, the beagleGetPartials hook of src/beagle_utils.c:252).  Check mode:
   the downloads are compared bit for bit with the host vectors the reference computed itself. */
static long g_mirror_buffers = 0, g_mirror_mismatch = 0;
static void mirror_partials(t_tree *tree)
{
  ctx_t       *c = ensure_instance(tree);
  const int    P = tree->data->n_pattern, C = tree->mod->ras->n_catg, S = tree->mod->ns;
  const size_t n = (size_t)P * C * S;
  double      *tmp = malloc(sizeof(double) * n);
  int         *sc = malloc(sizeof(int) * P);
  for (int i = 0; i < c->nbuf; ++i)
  {
    double *host = (double *)c->bufptr[i];
    if (!c->scaleptr[i]) continue; /* never a destination: a spare SPR buffer */
    ++g_mirror_buffers;
    if (!g_check)
    {
      OK(phyhip_get_partials(c->inst, tree->n_otu + i, PHYHIP_OP_NONE, host));
      OK(phyhip_get_scale_factors(c->inst, tree->n_otu + i, c->scaleptr[i]));
      continue;
    }
    OK(phyhip_get_partials(c->inst, tree->n_otu + i, PHYHIP_OP_NONE, tmp));
    OK(phyhip_get_scale_factors(c->inst, tree->n_otu + i, sc));
    for (int p = 0; p < P; ++p)
    {
      if (tree->data->wght[p] <= SMALL) continue;
      if (memcmp(tmp + (size_t)p * C * S, host + (size_t)p * C * S, sizeof(double) * C * S) || sc[p] != c->scaleptr[i][p]) ++g_mirror_mismatch;
    }
  }
  free(tmp); free(sc);
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
  g_class_axis  = getenv("GLUE_CLASS_AXIS") && atoi(getenv("GLUE_CLASS_AXIS"));

  g_hp_on = getenv("GLUE_HOSTPROF") && atoi(getenv("GLUE_HOSTPROF"));
  const double t0 = now_s();
  g_t0 = t0;
  g_hp_c0 = g_hp_on ? __builtin_ia32_rdtsc() : 0ull;
  if (getenv("GLUE_MAX_MIXT")) g_max_mixt = atol(getenv("GLUE_MAX_MIXT"));
  for (int k = 1; k < pargc; ++k)
    if (!strncmp(pargv[k], "--xml", 5))
    { /* XML mode (mixtures): the whole analysis runs inside Get_Input (src/cl.c:335, src/io.c:5033) */
      Get_Input(pargc, pargv);
      report_xml_and_exit();
    }
  t_tree *tree = setup_tree(pargc, pargv, &o); /* src/main.c:73-258, first Lk(NULL) included */
  const double lnl_init = tree->c_lnL;
  /* src/main.c:262-275 */
  if (tree->mod->s_opt->opt_topo) Global_Spr_Search(tree);
  else if (tree->mod->s_opt->opt_subst_param || tree->mod->s_opt->opt_bl_one_by_one) Round_Optimize(tree, ROUND_MAX);
  /* src/main.c:281-282 */
  Set_Both_Sides(YES, tree);
  Lk(NULL, tree);
  const double lnl_final = tree->c_lnL;
  char *nwk = Write_Tree(tree);
  if (tree->io->ancestral == YES)
  { /* src/main.c:288: marginal ancestral reconstruction reads the partial vectors of the evaluation above */
    if (!g_host)
    {
      mirror_partials(tree);
      site_outputs(tree, ensure_instance(tree));
    }
    Ancestral_Sequences(tree, YES);
  }
  if (tree->io->print_site_lnl)
  { /* src/main.c:323, src/io.c:1870-2013: per-site likelihoods, per-category likelihoods and posterior mean rates */
    if (!g_host) site_outputs(tree, ensure_instance(tree));
    Print_Site_Lk(tree, tree->io->fp_out_lk);
    fflush(tree->io->fp_out_lk);
  }
  char *support_nwk = NULL;
  if (tree->io->do_boot || tree->io->do_tbe || tree->io->do_bayesboot)
  { /* src/main.c:351-369: every replicate is a new tree object on resampled weights (see Init_Partial_Lk_Tips_Double above) */
    support_nwk = Bootstrap_From_String(Write_Tree(tree), tree->data, tree->mod, tree->io);
  }
  else if (tree->io->ratio_test != NO)
  { /* src/main.c:371-375: fast branch supports on the most likely tree (a new tree object, a new instance) */
    g_site_outputs = 1;
    support_nwk = aLRT_From_String(Write_Tree(tree), tree->data, tree->mod, tree->io);
  }
  const double dt = now_s() - t0;
  if (g_hp_on)
  { /* one line of its own, in front of the result line */
    const double cyc_per_ns = (double)(__builtin_ia32_rdtsc() - g_hp_c0) / (dt * 1e9);
    printf("\nGLUE_HOSTPROF {\"seconds\": %.3f, \"cycles_per_ns\": %.4f, \"slots\": {", dt, cyc_per_ns);
    for (int k = 0; k < HP_SLOTS; ++k)
      printf("%s\"%s\": {\"calls\": %llu, \"ns_per_call\": %.1f, \"seconds\": %.4f, \"share\": %.4f}", k ? ", " : "", hp_name[k], g_hp_n[k],
             g_hp_n[k] ? (double)g_hp_cyc[k] / cyc_per_ns / (double)g_hp_n[k] : 0.0, (double)g_hp_cyc[k] / cyc_per_ns * 1e-9,
             (double)g_hp_cyc[k] / cyc_per_ns * 1e-9 / dt);
    printf("}}\n");
  }
  long long vs[4] = {0, 0, 0, 0}; /* virtual buffers of the (first) instance: now | stores skipped | recomputed for a reader | stored on demand */
  if (!g_host && g_nctx > 0) phyhip_get_virtual_stats(g_ctx[0].inst, vs);
  printf("\nGLUE_DRIVER {\"mode\": \"%s\", \"virtual_buffers\": [%lld, %lld, %lld, %lld], \"device_pmat\": %d, \"lnL_init\": %.17g, \"lnL_final\": %.17g, \"seconds\": %.3f, "
         "\"calls\": {\"Lk\": %ld, \"Lk_full\": %ld, \"Update_Partial_Lk\": %ld, \"dLk\": %ld, \"Update_PMat\": %ld, "
         "\"Update_Eigen_Lr\": %ld}, \"worst_rel_lnL\": %.3g, \"worst_rel_dlnL\": %.3g, \"buffers\": %d, \"matrices\": %d, "
         "\"site_output_downloads\": %ld, \"worst_rel_site_output\": %.3g, \"mirrored_buffers\": %ld, \"mirror_mismatches\": %ld, \"instances_created\": %ld, "
         "\"alias_one_subpatt_calls_made_here\": %ld, \"support_tree\": \"%s\", \"tree\": \"%s\"}\n",
         g_host ? "host" : (g_check ? "check" : "device"), vs[0], vs[1], vs[2], vs[3], g_device_pmat, lnl_init, lnl_final, dt, g_n_lk, g_n_lk_full, g_n_upd, g_n_dlk, g_n_pmat,
         g_n_eig, g_worst_lnl, g_worst_dlnl, g_nbuf, g_nmat, g_n_site_dl, g_worst_site_lnl, g_mirror_buffers, g_mirror_mismatch, g_n_created, g_n_alias, support_nwk ? support_nwk : "", nwk ? nwk : "");
  fflush(stdout);
  for (int k = 0; k < g_nctx; ++k) OK(phyhip_finalize_instance(g_ctx[k].inst));
  _exit(0);
}
