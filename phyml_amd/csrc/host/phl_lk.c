/*
 * phl_lk.c -- host-side mirror of PhyML's likelihood surface (plain C) on top of the phyhip C ABI.
 *
 * What stays on the host in the reference stays on the host here: tree traversal order
 * (Post_Order_Lk / Pre_Order_Lk, src/lk.c:282-393), the resolution of "which buffers feed this edge
 * side" (Set_All_Partial_Lk / Set_Partial_Lk_One_Side, src/lk.c:2922-3271), the bookkeeping flags, and
 * (optionally) the transition matrices (PMat, src/models.c:257-373).  Everything that touches a
 * pattern runs in libphyhip.so on the GPU.  See include/phyhip_lk.h for the interface contract.
 */
#include "../../../include/phyhip_lk.h"
#include "../../../include/phyhip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SMALL_PIJ 1.E-100 /* src/utilities.h:478 */

static void (*g_exit_handler)(const char *) = NULL;

void Set_Exit_Handler(void (*handler)(const char *msg)) { g_exit_handler = handler; }

/* the reference prints and calls Exit() (src/utilities.c:1105); so do we, unless a handler is installed */
static void Lk_Exit(const char *where, const char *msg)
{
  char buf[640];
  snprintf(buf, sizeof buf, "\n. Err. in %s: %s\n", where, msg);
  fputs(buf, stderr);
  fflush(NULL);
  if (g_exit_handler) g_exit_handler(buf);
  else exit(1);
}

#define CHK(call)                                                                                           \
  do                                                                                                        \
  {                                                                                                         \
    int rc_ = (call);                                                                                       \
    if (rc_ < 0)                                                                                            \
    {                                                                                                       \
      Lk_Exit(#call, phyhip_get_last_error());                                                              \
      return;                                                                                               \
    }                                                                                                       \
  } while (0)
#define CHKV(call, ret)                                                                                     \
  do                                                                                                        \
  {                                                                                                         \
    int rc_ = (call);                                                                                       \
    if (rc_ < 0)                                                                                            \
    {                                                                                                       \
      Lk_Exit(#call, phyhip_get_last_error());                                                              \
      return ret;                                                                                           \
    }                                                                                                       \
  } while (0)

/* ------------------------------------------------------------------------------------------------ */
/* construction                                                                                       */
/* ------------------------------------------------------------------------------------------------ */

t_mod *Make_Model_Basic(int ns, int n_catg)
{
  t_mod *m         = (t_mod *)calloc(1, sizeof(t_mod));
  m->ns            = ns;
  m->n_catg        = n_catg;
  m->pi            = (phydbl *)calloc(ns, sizeof(phydbl));
  m->gamma_rr      = (phydbl *)calloc(n_catg, sizeof(phydbl));
  m->gamma_r_proba = (phydbl *)calloc(n_catg, sizeof(phydbl));
  m->e_val         = (phydbl *)calloc(ns, sizeof(phydbl));
  m->r_e_vect      = (phydbl *)calloc((size_t)ns * ns, sizeof(phydbl));
  m->l_e_vect      = (phydbl *)calloc((size_t)ns * ns, sizeof(phydbl));
  m->l_min         = 1.E-8; /* src/init.c:711-714 */
  m->l_max         = 100.0;
  m->br_len_mult   = 1.0;
  for (int c = 0; c < n_catg; ++c)
  {
    m->gamma_rr[c]      = 1.0;
    m->gamma_r_proba[c] = 1.0 / n_catg;
  }
  return m;
}

void Free_Model(t_mod *m)
{
  if (!m) return;
  free(m->pi); free(m->gamma_rr); free(m->gamma_r_proba); free(m->e_val); free(m->r_e_vect); free(m->l_e_vect);
  free(m);
}

t_tree *Make_Tree_From_Edges(int n_otu, const int *edge_left, const int *edge_rght, const phydbl *edge_len,
                             const int *neighbour_v, const int *neighbour_b)
{
  const int n_nodes = 2 * n_otu - 2, n_edges = 2 * n_otu - 3;
  t_tree   *tree    = (t_tree *)calloc(1, sizeof(t_tree));
  tree->n_otu       = n_otu;
  tree->b_inst      = -1;
  tree->a_nodes     = (t_node **)calloc(n_nodes, sizeof(t_node *));
  tree->a_edges     = (t_edge **)calloc(n_edges, sizeof(t_edge *));
  tree->apply_lk_scaling = YES;
  for (int i = 0; i < n_nodes; ++i)
  {
    tree->a_nodes[i]      = (t_node *)calloc(1, sizeof(t_node));
    tree->a_nodes[i]->num = i;
    tree->a_nodes[i]->tax = i < n_otu;
  }
  for (int e = 0; e < n_edges; ++e)
  {
    t_edge *b = tree->a_edges[e] = (t_edge *)calloc(1, sizeof(t_edge));
    b->num  = e;
    b->left = tree->a_nodes[edge_left[e]];
    b->rght = tree->a_nodes[edge_rght[e]];
    b->l    = edge_len[e];
    b->update_partial_lk_left = b->update_partial_lk_rght = YES;
    if (b->left->tax && !b->rght->tax)
    { /* keep the reference's invariant: a tip is always on the right (src/make.c:418-423) */
      t_node *t = b->left;
      b->left = b->rght;
      b->rght = t;
    }
  }
  if (neighbour_v && neighbour_b)
  {
    for (int i = 0; i < n_nodes; ++i)
      for (int k = 0; k < 3; ++k)
        if (neighbour_v[i * 3 + k] >= 0)
        {
          tree->a_nodes[i]->v[k] = tree->a_nodes[neighbour_v[i * 3 + k]];
          tree->a_nodes[i]->b[k] = tree->a_edges[neighbour_b[i * 3 + k]];
        }
  }
  else
  {
    for (int e = 0; e < n_edges; ++e)
    {
      t_edge *b       = tree->a_edges[e];
      t_node *ends[2] = {b->left, b->rght};
      for (int s = 0; s < 2; ++s)
      {
        t_node *nd = ends[s];
        int     k  = 0;
        while (k < 3 && nd->b[k]) ++k;
        if (k == 3) { Lk_Exit("Make_Tree_From_Edges", "node with more than three edges"); return NULL; }
        nd->b[k] = b;
        nd->v[k] = ends[1 - s];
      }
    }
  }
  return tree;
}

void Free_Tree(t_tree *tree)
{
  if (!tree) return;
  for (int i = 0; i < 2 * tree->n_otu - 2; ++i) free(tree->a_nodes[i]);
  for (int e = 0; e < 2 * tree->n_otu - 3; ++e)
  {
    free(tree->a_edges[e]->Pij_rr);
    free(tree->a_edges[e]);
  }
  free(tree->a_nodes); free(tree->a_edges); free(tree->wght); free(tree->invar);
  free(tree);
}

void Update_Model_On_Device(t_tree *tree)
{
  const t_mod *m = tree->mod;
  CHK(phyhip_set_category_rates(tree->b_inst, m->gamma_rr));                            /* update_beagle_ras   */
  CHK(phyhip_set_category_weights(tree->b_inst, 0, m->gamma_r_proba));
  CHK(phyhip_set_state_frequencies(tree->b_inst, 0, m->pi));                            /* update_beagle_efrqs */
  CHK(phyhip_set_eigen_decomposition(tree->b_inst, 0, m->r_e_vect, m->l_e_vect, m->e_val)); /* update_beagle_eigen */
  CHK(phyhip_set_phyml_options(tree->b_inst, m->l_min, m->l_max, m->br_len_mult, tree->apply_lk_scaling));
  CHK(phyhip_set_invariant_sites(tree->b_inst, m->invar, m->pinvar, m->invar ? tree->invar : NULL));
}

void Make_Tree_For_Lk(t_tree *tree, int n_pattern, const phydbl *wght, const short *invar, int device)
{
  Make_Tree_For_Lk_On_Devices(tree, n_pattern, wght, invar, device >= 0 ? &device : NULL, device >= 0 ? 1 : 0, NO);
}

/* Same with a list of devices: more than one entry (or flags & 1) gives the sharded instance of include/phyhip.h --
   contiguous pattern ranges, one per device, and ONE RCCL all-reduce behind every Lk()/dLk().  Nothing else
   in this file knows about it: tree->b_inst is used exactly like a single-device instance.  flags & 2: the categories are
   the classes of a mixture (PHYHIP_FLAG_CLASS_AXIS); the per-class models beyond class 0 are pushed by the caller. */
void Make_Tree_For_Lk_On_Devices(t_tree *tree, int n_pattern, const phydbl *wght, const short *invar, const int *devices,
                                 int n_devices, int flags)
{
  const int n = tree->n_otu, n_edges = 2 * n - 3;
  const t_mod *m = tree->mod;
  if (!m) { Lk_Exit("Make_Tree_For_Lk", "tree->mod is NULL"); return; }
  tree->n_pattern = n_pattern;
  tree->wght      = (phydbl *)malloc(sizeof(phydbl) * n_pattern);
  memcpy(tree->wght, wght, sizeof(phydbl) * n_pattern);
  tree->invar = (short *)malloc(sizeof(short) * n_pattern);
  if (invar) memcpy(tree->invar, invar, sizeof(short) * n_pattern);
  else for (int i = 0; i < n_pattern; ++i) tree->invar[i] = -1;

  /* buffer indices: tips first (src/lk.c:2229), then one per internal edge side; one matrix per edge */
  int next = n;
  for (int e = 0; e < n_edges; ++e)
  {
    t_edge *b          = tree->a_edges[e];
    b->Pij_rr_idx      = e;
    b->p_lk_tip_idx    = b->rght->tax ? b->rght->num : -1;
    b->p_lk_left_idx   = b->left->tax ? b->left->num : next++;
    b->p_lk_rght_idx   = b->rght->tax ? b->rght->num : next++;
    b->Pij_rr          = (phydbl *)calloc((size_t)m->n_catg * m->ns * m->ns, sizeof(phydbl));
  }
  /* spare partials buffers and matrices, the counterpart of the reference's extra SPR edges
     (Make_Extra_Edge_Lk, src/make.c:750): a regraft candidate is evaluated into these */
  tree->spare_p_lk_idx = next;
  tree->spare_Pij_idx  = n_edges;
  next += PHL_N_SPARE;
  phyhip_instance_details det;
  int inst = phyhip_create_instance(n, next, 0, m->ns, n_pattern, 1, n_edges + PHL_N_SPARE, m->n_catg, 0,
                                    n_devices > 0 ? devices : NULL, n_devices, 0,
                                    ((flags & 1) ? PHYHIP_FLAG_SHARDED : 0) | ((flags & 2) ? PHYHIP_FLAG_CLASS_AXIS : 0) |
                                        (m->use_m4mod ? PHYHIP_FLAG_GENERIC_LOOP : 0), &det);
  if (inst < 0) { Lk_Exit("phyhip_create_instance", phyhip_get_last_error()); return; }
  tree->b_inst = inst;
  CHK(phyhip_set_pattern_weights(inst, tree->wght));
  Update_Model_On_Device(tree);
}

void Free_Tree_Lk(t_tree *tree)
{
  if (tree->b_inst >= 0) phyhip_finalize_instance(tree->b_inst);
  tree->b_inst = -1;
}

void Init_Partial_Lk_Tips_Double_One_Tip(t_tree *tree, int tax_id, const phydbl *p_lk_tip)
{
  CHK(phyhip_set_tip_partials(tree->b_inst, tax_id, p_lk_tip));
}

void Init_Partial_Lk_Tips_States_One_Tip(t_tree *tree, int tax_id, const int *states)
{
  CHK(phyhip_set_tip_states(tree->b_inst, tax_id, states));
}

/* ---- a14: character -> 0/1 tip vector (src/lk.c:26-69 nucleotides, :122-161 amino acids) ---------------------- */
/* Allowed-state sets as bit masks, one table per alphabet, built on first use.  Nucleotides: IUPAC codes, U = T,
   N X ? O - = any state.  Amino acids: the 20 letters in PhyML's order ARNDCQEGHILKMFPSTWYV; the reference resolves
   B to asparagine and Z to glutamine (src/lk.c:149-150) rather than to two-state sets; X ? - = any state. */
static unsigned g_nt_mask[256], g_aa_mask[256];
static int      g_tip_tables_ready = 0;
static void build_tip_tables(void)
{
  static const struct { char c; unsigned m; } nt[] = {
      {'A', 1}, {'C', 2}, {'G', 4}, {'T', 8}, {'U', 8}, {'M', 1 | 2}, {'R', 1 | 4}, {'W', 1 | 8}, {'S', 2 | 4}, {'Y', 2 | 8},
      {'K', 4 | 8}, {'B', 2 | 4 | 8}, {'D', 1 | 4 | 8}, {'H', 1 | 2 | 8}, {'V', 1 | 2 | 4}, {'N', 15}, {'X', 15}, {'?', 15},
      {'O', 15}, {'-', 15}};
  static const char aa_order[] = "ARNDCQEGHILKMFPSTWYV";
  for (int i = 0; i < 256; ++i) g_nt_mask[i] = g_aa_mask[i] = 0u;
  for (unsigned i = 0; i < sizeof nt / sizeof nt[0]; ++i) g_nt_mask[(unsigned char)nt[i].c] = nt[i].m;
  for (int i = 0; i < 20; ++i) g_aa_mask[(unsigned char)aa_order[i]] = 1u << i;
  g_aa_mask[(unsigned char)'B'] = 1u << 2;
  g_aa_mask[(unsigned char)'Z'] = 1u << 5;
  g_aa_mask[(unsigned char)'X'] = g_aa_mask[(unsigned char)'?'] = g_aa_mask[(unsigned char)'-'] = (1u << 20) - 1u;
  g_tip_tables_ready = 1;
}
static void tip_vector(const unsigned *table, int ns, char state, int pos, phydbl *p_lk, const char *who)
{
  if (!g_tip_tables_ready) build_tip_tables();
  const unsigned m = table[(unsigned char)state];
  if (!m)
  {
    char msg[96];
    snprintf(msg, sizeof msg, "unknown character state '%c' at position %d", state, pos);
    Lk_Exit(who, msg);
    return;
  }
  for (int s = 0; s < ns; ++s) p_lk[pos + s] = ((m >> s) & 1u) ? 1.0 : 0.0;
}
void Init_Tips_At_One_Site_Nucleotides_Float(char state, int pos, phydbl *p_lk)
{
  tip_vector(g_nt_mask, 4, state, pos, p_lk, "Init_Tips_At_One_Site_Nucleotides_Float");
}
void Init_Tips_At_One_Site_AA_Float(char aa, int pos, phydbl *p_lk)
{
  tip_vector(g_aa_mask, 20, aa, pos, p_lk, "Init_Tips_At_One_Site_AA_Float");
}
void Init_Partial_Lk_Tips_Chars_One_Tip(t_tree *tree, int tax_id, const char *seq)
{ /* the per-tip loop of Init_Partial_Lk_Tips_Double (src/lk.c:2060-2118) + the upload */
  const int ns = tree->mod->ns, P = tree->n_pattern;
  if (ns != 4 && ns != 20) { Lk_Exit("Init_Partial_Lk_Tips_Chars_One_Tip", "4 or 20 states"); return; }
  phydbl *v = (phydbl *)malloc(sizeof(phydbl) * (size_t)P * ns);
  for (int p = 0; p < P; ++p)
  {
    if (ns == 4) Init_Tips_At_One_Site_Nucleotides_Float(seq[p], p * ns, v);
    else Init_Tips_At_One_Site_AA_Float(seq[p], p * ns, v);
  }
  Init_Partial_Lk_Tips_Double_One_Tip(tree, tax_id, v);
  free(v);
}

/* ------------------------------------------------------------------------------------------------ */
/* a12: transition matrices                                                                           */
/* ------------------------------------------------------------------------------------------------ */

void PMat(phydbl l, const t_mod *mod, int pos, phydbl *Pij)
{
  const int ns = mod->ns;
  Pij += pos;
  if (l < 0.0)
  { /* src/models.c:331-338,356-361 */
    for (int i = 0; i < ns * ns; ++i) Pij[i] = 0.0;
    for (int i = 0; i < ns; ++i) Pij[i * ns + i] = 1.0;
    return;
  }
  phydbl expt[64], uexpt[64 * 64];
  const phydbl *U = mod->r_e_vect, *V = mod->l_e_vect, *R = mod->e_val;
  for (int k = 0; k < ns; ++k) expt[k] = exp(R[k] * l);
  for (int i = 0; i < ns; ++i)
    for (int k = 0; k < ns; ++k) uexpt[i * ns + k] = U[i * ns + k] * expt[k];
  for (int i = 0; i < ns; ++i)
  {
    phydbl *row = Pij + i * ns, sum = 0.0;
    for (int j = 0; j < ns; ++j)
    {
      phydbl acc = 0.0;
      for (int k = 0; k < ns; ++k) acc = fma(uexpt[i * ns + k], V[k * ns + j], acc);
      if (acc < SMALL_PIJ) acc = SMALL_PIJ; /* src/models.c:293 */
      row[j] = acc;
    }
    for (int j = 0; j < ns; ++j) sum += row[j]; /* :296-298 */
    for (int j = 0; j < ns; ++j) row[j] /= sum;
  }
}

void Update_PMat_At_Given_Edge(t_edge *b, t_tree *tree)
{
  const t_mod *m = tree->mod;
  if (tree->host_pmat)
  {
    for (int c = 0; c < m->n_catg; ++c)
    { /* src/lk.c:2296-2300 */
      phydbl len = (b->l > 0.0 ? b->l : 0.0) * m->gamma_rr[c];
      len *= m->br_len_mult;
      if (len < m->l_min) len = m->l_min;
      else if (len > m->l_max) len = m->l_max;
      PMat(len, m, c * m->ns * m->ns, b->Pij_rr);
    }
    CHK(phyhip_set_transition_matrix(tree->b_inst, b->Pij_rr_idx, b->Pij_rr, -1)); /* src/lk.c:2360 */
  }
  else
  {
    int    idx[1] = {b->Pij_rr_idx};
    double len[1] = {b->l};
    CHK(phyhip_update_transition_matrices(tree->b_inst, 0, idx, NULL, NULL, len, 1)); /* src/lk.c:2344 */
  }
}

static void Update_All_PMat(t_tree *tree)
{
  const int n_edges = 2 * tree->n_otu - 3;
  if (tree->host_pmat)
  {
    for (int e = 0; e < n_edges; ++e) Update_PMat_At_Given_Edge(tree->a_edges[e], tree);
    return;
  }
  int    *idx = (int *)malloc(sizeof(int) * n_edges);
  double *len = (double *)malloc(sizeof(double) * n_edges);
  for (int e = 0; e < n_edges; ++e)
  {
    idx[e] = tree->a_edges[e]->Pij_rr_idx;
    len[e] = tree->a_edges[e]->l;
  }
  int rc = phyhip_update_transition_matrices(tree->b_inst, 0, idx, NULL, NULL, len, n_edges);
  free(idx); free(len);
  if (rc < 0) Lk_Exit("phyhip_update_transition_matrices", phyhip_get_last_error());
}

/* ------------------------------------------------------------------------------------------------ */
/* a2 / a5: one edge-side update                                                                      */
/* ------------------------------------------------------------------------------------------------ */

/* buffer index of the subtree seen from d across edge b (Set_Partial_Lk_One_Side, src/lk.c:3228-3259) */
static int Child_Buffer(const t_node *d, const t_edge *b)
{
  if (d == b->left) return b->rght->tax ? b->p_lk_tip_idx : b->p_lk_rght_idx;
  return b->p_lk_left_idx;
}

static int Fill_Operation(const t_edge *b, const t_node *d, phyhip_operation *op)
{
  int k = 0, child[2] = {-1, -1}, mat[2] = {-1, -1};
  for (int i = 0; i < 3; ++i)
    if (d->b[i] != b)
    { /* first other neighbour -> v1, second -> v2 (src/lk.c:2958-2986) */
      if (k == 2) return -1;
      child[k] = Child_Buffer(d, d->b[i]);
      mat[k]   = d->b[i]->Pij_rr_idx;
      ++k;
    }
  if (k != 2) return -1;
  op->destinationPartials    = (d == b->left) ? b->p_lk_left_idx : b->p_lk_rght_idx;
  op->destinationScaleWrite  = PHYHIP_OP_NONE;
  op->destinationScaleRead   = PHYHIP_OP_NONE;
  op->child1Partials         = child[0];
  op->child1TransitionMatrix = mat[0];
  op->child2Partials         = child[1];
  op->child2TransitionMatrix = mat[1];
  return 0;
}

/* The gates of Update_Partial_Lk (src/lk.c:1285-1297); 1: *op is the operation to queue */
static int Partial_Lk_Operation(t_tree *tree, t_edge *b, t_node *d, phyhip_operation *op)
{
  if (b->left == d && b->update_partial_lk_left == NO) return 0; /* src/lk.c:1285-1286 */
  if (b->rght == d && b->update_partial_lk_rght == NO) return 0;
  if (tree->do_alias_subpatt == YES && tree->update_alias_subpatt == YES && tree->alias_one_subpatt) /* :1294-1296 */
    tree->alias_one_subpatt((d == b->left) ? b->rght : b->left, d, tree);
  if (d->tax) return 0;                                          /* :1297 */
  if (Fill_Operation(b, d, op) < 0) { Lk_Exit("Update_Partial_Lk", "node is not an internal node of degree 3"); return 0; }
  return 1;
}

void Update_Partial_Lk(t_tree *tree, t_edge *b, t_node *d)
{
  phyhip_operation op;
  if (!Partial_Lk_Operation(tree, b, d, &op)) return;
  CHK(phyhip_update_partials(tree->b_inst, &op, 1, PHYHIP_OP_NONE)); /* src/lk.c:1301 */
  tree->n_edges_traversed++;
}

/* A traversal hands its operations over in ONE call of the queueing entry point, in the order the reference would make its
   calls (update_beagle_partials, src/beagle_utils.c:214-262, passes them one by one; the entry point takes a list, as BEAGLE's
   does): a call costs ~60 ns of table lookup and checks, a 100-taxon post-order has 98 of them in front of its launch. */
typedef struct { phyhip_operation *ops; int n, cap; } op_batch;
static void Batch_Partial_Lk(op_batch *q, t_tree *tree, t_edge *b, t_node *d)
{
  if (q->n == q->cap)
  {
    q->cap = q->cap ? 2 * q->cap : 256;
    q->ops = (phyhip_operation *)realloc(q->ops, sizeof(phyhip_operation) * (size_t)q->cap);
  }
  if (Partial_Lk_Operation(tree, b, d, &q->ops[q->n])) q->n++;
}
static void Batch_Submit(op_batch *q, t_tree *tree)
{
  if (q->n > 0)
  {
    const int rc = phyhip_update_partials(tree->b_inst, q->ops, q->n, PHYHIP_OP_NONE);
    tree->n_edges_traversed += q->n;
    free(q->ops);
    q->ops = NULL; q->n = q->cap = 0;
    if (rc < 0) Lk_Exit("phyhip_update_partials", phyhip_get_last_error());
    return;
  }
  free(q->ops);
  q->ops = NULL; q->cap = 0;
}

/* ------------------------------------------------------------------------------------------------ */
/* a13: traversals (unrooted)                                                                          */
/* ------------------------------------------------------------------------------------------------ */

void Post_Order_Lk(t_node *a, t_node *d, t_tree *tree)
{
  /* recursion of src/lk.c:308-349, iterative so that 4000-taxon combs cannot exhaust the C stack */
  typedef struct { t_node *a, *d; int i, dir; } frame;
  frame *st = (frame *)malloc(sizeof(frame) * (size_t)(2 * tree->n_otu));
  int    sp = 0;
  op_batch q = {NULL, 0, 0};
  st[sp++]  = (frame){a, d, 0, -1};
  while (sp > 0)
  {
    frame *f = &st[sp - 1];
    if (f->d->tax) { --sp; continue; }
    if (f->i < 3)
    {
      const int i = f->i++;
      if (f->d->v[i] != f->a) st[sp++] = (frame){f->d, f->d->v[i], 0, -1};
      else f->dir = i;
      continue;
    }
    if (f->dir < 0) { free(st); free(q.ops); Lk_Exit("Post_Order_Lk", "direction towards the ancestor not found"); return; }
    Batch_Partial_Lk(&q, tree, f->d->b[f->dir], f->d);
    --sp;
  }
  free(st);
  Batch_Submit(&q, tree);
}

void Pre_Order_Lk(t_node *a, t_node *d, t_tree *tree)
{
  /* src/lk.c:383-391: update the side of d->b[i] facing d, then descend */
  typedef struct { t_node *a, *d; int i; } frame;
  frame *st = (frame *)malloc(sizeof(frame) * (size_t)(2 * tree->n_otu));
  int    sp = 0;
  op_batch q = {NULL, 0, 0};
  st[sp++]  = (frame){a, d, 0};
  while (sp > 0)
  {
    frame *f = &st[sp - 1];
    if (f->d->tax || f->i == 3) { --sp; continue; }
    const int i = f->i++;
    if (f->d->v[i] != f->a)
    {
      Batch_Partial_Lk(&q, tree, f->d->b[i], f->d);
      st[sp++] = (frame){f->d, f->d->v[i], 0};
    }
  }
  free(st);
  Batch_Submit(&q, tree);
}

void Update_All_Partial_Lk(t_tree *tree)
{
  if (tree->e_root)
  { /* a rooted input tree whose root is ignored (tree->n_root != NULL, ignore_root == YES -- the only rooted form the
       `phyml` program evaluates, src/init.c:145): the two subtrees of the root edge (src/lk.c:420-429) */
    t_edge *e = tree->e_root;
    Post_Order_Lk(e->rght, e->left, tree);
    Post_Order_Lk(e->left, e->rght, tree);
    if (tree->both_sides == YES)
    {
      Pre_Order_Lk(e->rght, e->left, tree);
      Pre_Order_Lk(e->left, e->rght, tree);
    }
    return;
  }
  /* src/lk.c:432-437 */
  t_node *r = tree->a_nodes[tree->tip_root];
  Post_Order_Lk(r, r->v[0], tree);
  if (tree->both_sides == YES) Pre_Order_Lk(r, r->v[0], tree);
}

void Update_Partial_Lk_Along_A_Path(t_node **path, int path_length, t_tree *tree)
{ /* src/lk.c:2379-2411 */
  for (int i = 0; i < path_length - 1; ++i)
  {
    int j;
    for (j = 0; j < 3; ++j)
      if (path[i]->v[j] == path[i + 1])
      {
        Update_Partial_Lk(tree, path[i]->b[j], path[i]);
        break;
      }
    if (j == 3) { Lk_Exit("Update_Partial_Lk_Along_A_Path", "consecutive path nodes are not neighbours"); return; }
  }
}

void Set_Both_Sides(int yesno, t_tree *tree) { tree->both_sides = (short)yesno; }
void Set_Use_Eigen_Lr(int yesno, t_tree *tree) { tree->use_eigen_lr = (short)yesno; }
void Set_Update_Eigen_Lr(int yesno, t_tree *tree) { tree->update_eigen_lr = (short)yesno; }

/* ------------------------------------------------------------------------------------------------ */
/* a1: Lk                                                                                              */
/* ------------------------------------------------------------------------------------------------ */

void Update_Eigen_Lr(t_edge *b, t_tree *tree)
{
  const int left = b->left->tax ? b->left->num : b->p_lk_left_idx; /* src/avx.c:68-69 */
  const int rght = b->rght->tax ? b->p_lk_tip_idx : b->p_lk_rght_idx;
  CHK(phyhip_update_eigen_lr(tree->b_inst, left, rght));
}

static t_edge *Traverse_For_Lk(t_tree *tree)
{
  Update_All_PMat(tree);          /* src/lk.c:500-512 */
  Update_All_Partial_Lk(tree);    /* src/lk.c:545-564 */
  if (tree->e_root) return tree->e_root;      /* src/lk.c:573-576 (ignore_root == YES) */
  return tree->a_nodes[tree->tip_root]->b[0]; /* src/lk.c:578-579 */
}

phydbl Lk(t_edge *b, t_tree *tree)
{
  tree->numerical_warning = NO;   /* src/lk.c:454 */
  tree->old_lnL = tree->c_lnL;    /* :474 */
  if (b == NULL) b = Traverse_For_Lk(tree);
  else if (tree->use_eigen_lr == NO) Update_PMat_At_Given_Edge(b, tree); /* :515-527 */

  if (tree->update_eigen_lr == YES) Update_Eigen_Lr(b, tree);            /* :590 */

  double lnl = 0.0;
  if (tree->use_eigen_lr == YES)
    CHKV(phyhip_calculate_eigen_lnl(tree->b_inst, b->l, &lnl), 0.0);      /* :592-603,625-629 */
  else
  {
    int parent[1] = {b->p_lk_left_idx};                                   /* src/beagle_utils.c:338-340 */
    int child[1]  = {b->rght->tax ? b->p_lk_tip_idx : b->p_lk_rght_idx};
    int pmat[1]   = {b->Pij_rr_idx}, zero[1] = {0};
    CHKV(phyhip_calculate_edge_log_likelihoods(tree->b_inst, parent, child, pmat, NULL, NULL, zero, zero, NULL, 1, &lnl, NULL,
                                               NULL), 0.0);
  }
  tree->c_lnL = lnl;
  return tree->c_lnL;
}

phydbl Update_Lk_At_Given_Edge(t_edge *b_fcus, t_tree *tree)
{ /* src/lk.c:2478-2484: both sides of the edge, then Lk(b) -- two queued operations + the evaluation = one launch */
  Update_Partial_Lk(tree, b_fcus, b_fcus->left);
  Update_Partial_Lk(tree, b_fcus, b_fcus->rght);
  tree->c_lnL = Lk(b_fcus, tree);
  return tree->c_lnL;
}

void Lk_Shard_Device(t_tree *tree, double *device_out)
{
  t_edge *b = Traverse_For_Lk(tree);
  CHK(phyhip_calculate_edge_log_likelihoods_device(tree->b_inst, b->p_lk_left_idx,
                                                   b->rght->tax ? b->p_lk_tip_idx : b->p_lk_rght_idx, b->Pij_rr_idx, device_out));
}

phydbl dLk(phydbl *l, t_edge *b, t_tree *tree)
{
  tree->numerical_warning = NO;
  if (isnan(*l)) { Lk_Exit("dLk", "branch length is NaN"); return 0.0; } /* src/lk.c:671 */
  if (tree->update_eigen_lr == YES) Update_Eigen_Lr(b, tree);             /* :686 */
  double lnl = 0.0, dlnl = 0.0;
  CHKV(phyhip_calculate_eigen_lnl_dlnl(tree->b_inst, l, &lnl, &dlnl), 0.0);
  tree->c_dlnL = dlnl;                                                     /* :749-750 */
  tree->c_lnL  = lnl;
  return tree->c_lnL;
}

/* Br_Len_Newton (a harness, not a reference function) keeps the call pattern of the caller Br_Len_Opt (src/optimiz.c:607-663): Lk(b) fills dot_prod, then the
   length is optimised on dLk alone, then b's matrices are refreshed.  The 1-D search here is a safeguarded
   Newton/bisection on dlnL (the reference uses a spline search on the same two quantities). */
phydbl Br_Len_Newton(phydbl *l, t_edge *b, t_tree *tree)
{
  const t_mod *m = tree->mod;
  Set_Update_Eigen_Lr(YES, tree);
  Set_Use_Eigen_Lr(NO, tree);
  const phydbl lk_begin = Lk(b, tree);
  Set_Update_Eigen_Lr(NO, tree);
  Set_Use_Eigen_Lr(YES, tree);

  phydbl lo = m->l_min, hi = m->l_max, x = *l, best_l = *l, best_lk = lk_begin;
  if (x < lo) x = lo;
  if (x > hi) x = hi;
  for (int it = 0; it < 40; ++it)
  {
    phydbl xl = x;
    dLk(&xl, b, tree);
    if (tree->c_lnL > best_lk) { best_lk = tree->c_lnL; best_l = xl; }
    const phydbl g = tree->c_dlnL;
    if (g > 0.0) lo = xl; else hi = xl;
    if (hi - lo < 1.E-10 * (1.0 + hi) || fabs(g) < 1.E-9) break;
    /* secant-free step: geometric bisection is robust across the eight decades of [l_min,l_max] */
    x = (lo <= 0.0) ? 0.5 * (lo + hi) : sqrt(lo * hi);
  }
  *l   = best_l;
  b->l = best_l;
  dLk(l, b, tree);
  Update_PMat_At_Given_Edge(b, tree);
  Set_Update_Eigen_Lr(NO, tree);
  Set_Use_Eigen_Lr(NO, tree);
  if (tree->c_lnL < lk_begin - 1.E-6 * fabs(lk_begin))
  { /* src/optimiz.c:656-661 */
    Lk_Exit("Br_Len_Newton", "likelihood decreased");
    return tree->c_lnL;
  }
  return tree->c_lnL;
}

/* ------------------------------------------------------------------------------------------------ */
/* caller-side counterpart: replay of a recorded surface-call stream            */
/* ------------------------------------------------------------------------------------------------ */

/* Matrix refresh by index (real edge or spare): the body of Update_PMat_At_Given_Edge for a bare length */
static int Replay_Set_PMat(t_tree *tree, int idx, phydbl l, phydbl *scratch)
{
  const t_mod *m = tree->mod;
  if (tree->host_pmat)
  {
    for (int c = 0; c < m->n_catg; ++c)
    {
      phydbl len = (l > 0.0 ? l : 0.0) * m->gamma_rr[c];
      len *= m->br_len_mult;
      if (len < m->l_min) len = m->l_min;
      else if (len > m->l_max) len = m->l_max;
      PMat(len, m, c * m->ns * m->ns, scratch);
    }
    return phyhip_set_transition_matrix(tree->b_inst, idx, scratch, -1);
  }
  int    i1[1] = {idx};
  double l1[1] = {l};
  return phyhip_update_transition_matrices(tree->b_inst, 0, i1, NULL, NULL, l1, 1);
}

void Replay_Surface_Trace(t_tree *tree, int n_rec, const int *kind, const int *a, const int *b, const int *c, const int *d,
                          const int *e, const phydbl *x, phydbl *out, phydbl *out2)
{
  const t_mod *m = tree->mod;
  phydbl *scratch = (phydbl *)malloc(sizeof(phydbl) * (size_t)m->n_catg * m->ns * m->ns);
  int     zero[1] = {0};
  for (int i = 0; i < n_rec; ++i)
  {
    int rc = 0;
    out[i] = out2[i] = 0.0;
    switch (kind[i])
    {
    case PHL_REC_SET_PMAT: rc = Replay_Set_PMat(tree, a[i], x[i], scratch); break;
    case PHL_REC_UPDATE:
    {
      phyhip_operation op = {a[i], PHYHIP_OP_NONE, PHYHIP_OP_NONE, b[i], c[i], d[i], e[i]};
      rc = phyhip_update_partials(tree->b_inst, &op, 1, PHYHIP_OP_NONE);
      tree->n_edges_traversed++;
      break;
    }
    case PHL_REC_EDGE_LNL:
    {
      int parent[1] = {a[i]}, child[1] = {b[i]}, pm[1] = {c[i]};
      double lnl = 0.0;
      rc = phyhip_calculate_edge_log_likelihoods(tree->b_inst, parent, child, pm, NULL, NULL, zero, zero, NULL, 1, &lnl, NULL, NULL);
      out[i] = tree->c_lnL = lnl;
      break;
    }
    case PHL_REC_EIGEN_LR: rc = phyhip_update_eigen_lr(tree->b_inst, a[i], b[i]); break;
    case PHL_REC_DLK:
    {
      double l = x[i], lnl = 0.0, dlnl = 0.0;
      rc = phyhip_calculate_eigen_lnl_dlnl(tree->b_inst, &l, &lnl, &dlnl);
      out[i] = tree->c_lnL = lnl;
      out2[i] = tree->c_dlnL = dlnl;
      break;
    }
    case PHL_REC_EIGEN_LNL:
    {
      double lnl = 0.0;
      rc = phyhip_calculate_eigen_lnl(tree->b_inst, x[i], &lnl);
      out[i] = tree->c_lnL = lnl;
      break;
    }
    default: rc = -1; break;
    }
    if (rc < 0)
    {
      free(scratch);
      Lk_Exit("Replay_Surface_Trace", rc == -1 && kind[i] > PHL_REC_EIGEN_LNL ? "unknown record kind" : phyhip_get_last_error());
      return;
    }
  }
  free(scratch);
}

/* ------------------------------------------------------------------------------------------------ */
/* download hooks                                                                                      */
/* ------------------------------------------------------------------------------------------------ */

void Get_Partial_Lk(t_tree *tree, t_edge *b, t_node *d, phydbl *p_lk, int *sum_scale)
{
  const int idx = (d == b->left) ? b->p_lk_left_idx : b->p_lk_rght_idx;
  if (d->tax) { Lk_Exit("Get_Partial_Lk", "tips hold no partial vector"); return; }
  if (p_lk) CHK(phyhip_get_partials(tree->b_inst, idx, PHYHIP_OP_NONE, p_lk));
  if (sum_scale) CHK(phyhip_get_scale_factors(tree->b_inst, idx, sum_scale));
}

void Get_Site_Lk(t_tree *tree, phydbl *c_lnL_sorted, phydbl *cur_site_lk, phydbl *unscaled_site_lk_cat, int *fact_sum_scale)
{
  CHK(phyhip_get_site_outputs(tree->b_inst, c_lnL_sorted, cur_site_lk, unscaled_site_lk_cat, fact_sum_scale));
}
