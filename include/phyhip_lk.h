/*
 * phyhip_lk.h -- host-side mirror (plain C) of PhyML's likelihood surface, implemented on the phyhip
 * C ABI (include/phyhip.h).  This is what spr.c / optimiz.c call in the reference (src/lk.h:27-159);
 * names, argument order and side effects follow the reference so that a caller written against lk.h
 * (or a parity test) reads the same:
 *
 *   Lk(b,tree)                      src/lk.c:443     full (b == NULL) or single-edge log-likelihood -> tree->c_lnL
 *   dLk(&l,b,tree)                  src/lk.c:655     lnL and dlnL/dl in the eigen basis -> tree->c_lnL, tree->c_dlnL
 *   Update_Partial_Lk(tree,b,d)     src/lk.c:1282    one edge-side partial vector (queued on the device)
 *   Update_PMat_At_Given_Edge(b,t)  src/lk.c:2238    transition matrices of one edge
 *   Post_Order_Lk / Pre_Order_Lk    src/lk.c:282/357 traversals issuing Update_Partial_Lk
 *   Update_All_Partial_Lk           src/lk.c:401
 *   Update_Partial_Lk_Along_A_Path  src/lk.c:2379
 *   Update_Lk_At_Given_Edge(b,tree) src/lk.c:2478    both sides of b refreshed, then Lk(b)
 *   Update_Eigen_Lr(b,tree)         src/lk.c:1038
 *   Set_Both_Sides / Set_Use_Eigen_Lr / Set_Update_Eigen_Lr   src/utilities.c:11614-11640
 *   Make_Tree_For_Lk / Free_Tree_Lk src/make.c:17 / src/free.c:387   (device instance instead of the host slab)
 *   Br_Len_Newton(&l,b,tree)        NOT a reference function: a harness that drives the surface the way the caller
 *                                   Br_Len_Opt (src/optimiz.c:607-663) does -- Lk(b) with update_eigen_lr, then dLk alone,
 *                                   then the matrix refresh -- with a safeguarded Newton search on dlnL in place of the
 *                                   reference's Br_Len_Spline (src/optimiz.c:2244); optimiz.c itself stays a caller
 *
 * The structs are this repo's own minimal versions of t_tree/t_edge/t_node/t_mod: only the fields the
 * hot path reads, with the reference's field names (src/utilities.h:640-1010).  Unrooted trees, and rooted
 * input trees the way the `phyml` program evaluates them (root ignored: tree->e_root).  Errors follow the reference's
 * convention: message on stderr, then Exit() (src/utilities.c:1105) -- replaceable via Set_Exit_Handler.
 */
#ifndef PHYHIP_LK_H
#define PHYHIP_LK_H

#ifdef __cplusplus
extern "C" {
#endif

#define YES 1
#define NO 0
typedef double phydbl; /* src/utilities.h:462 */

struct __Edge;
typedef struct __Node
{
  struct __Node *v[3]; /* neighbours (NULL beyond the first for a tip) */
  struct __Edge *b[3]; /* b[i] connects this node to v[i] */
  int            num;
  int            tax;  /* 1: tip */
} t_node;

typedef struct __Edge
{
  t_node *left, *rght; /* a tip is always on the right (src/make.c:418-423) */
  int     num;
  phydbl  l;           /* b->l->v */
  /* device buffer indices: the fields the BEAGLE seam adds to t_edge (src/utilities.h:746-763) */
  int     Pij_rr_idx;
  int     p_lk_left_idx, p_lk_rght_idx, p_lk_tip_idx;
  short   update_partial_lk_left, update_partial_lk_rght; /* src/utilities.h:826-827 */
  phydbl *Pij_rr;      /* host copy [C][S][S], valid when tree->host_pmat == YES */
} t_edge;

typedef struct __Model
{
  int     ns, n_catg;            /* mod->ns, mod->ras->n_catg */
  phydbl *pi;                    /* mod->e_frq->pi->v */
  phydbl *gamma_rr;              /* mod->ras->gamma_rr->v */
  phydbl *gamma_r_proba;         /* mod->ras->gamma_r_proba->v */
  phydbl *e_val, *r_e_vect, *l_e_vect; /* mod->eigen */
  phydbl  l_min, l_max;          /* src/init.c:711-714 */
  phydbl  br_len_mult;
  int     invar;                 /* mod->ras->invar */
  phydbl  pinvar;
  int     use_m4mod;             /* mod->use_m4mod (`phyml --cov`, src/cl.c:753-757): Update_Partial_Lk sends the data through the
                                    generic loop instead of the SIMD kernels, src/lk.c:1303-1324 -- the device instance is then
                                    created with PHYHIP_FLAG_GENERIC_LOOP (that loop's arithmetic) */
} t_mod;

typedef struct __Tree
{
  t_node **a_nodes; /* [2n-2], tips first */
  t_edge **a_edges; /* [2n-3] */
  t_mod   *mod;
  int      n_otu;
  int      n_pattern;      /* tree->data->n_pattern */
  phydbl  *wght;           /* tree->data->wght */
  short   *invar;          /* tree->data->invar */
  int      b_inst;         /* device instance id (tree->b_inst, src/utilities.h:999-1001) */
  int      tip_root;
  short    both_sides, use_eigen_lr, update_eigen_lr, apply_lk_scaling, numerical_warning;
  short    host_pmat;      /* YES: PMat() on the host + upload (src/lk.c:2315,2360); NO: device PMat (src/lk.c:2344) */
  phydbl   c_lnL, old_lnL, c_dlnL;
  int      n_edges_traversed; /* counter like src/utilities.h:1018 */
  int      spare_p_lk_idx;    /* first of PHL_N_SPARE spare partials buffers (extra SPR edges, src/make.c:750) */
  int      spare_Pij_idx;     /* first of PHL_N_SPARE spare transition-matrix buffers */
  t_edge  *e_root;            /* rooted input tree with the root ignored (tree->n_root != NULL, ignore_root == YES, the only
                                 rooted form the `phyml` program evaluates): the edge the root sits on, else NULL.  The
                                 ignore_root == NO special cases of src/lk.c:2988-3146 are not built: the reference's own AVX
                                 path cannot run them (oracle/probe_rooted.sh) */
  /* `phyml --alias_subpatt` (src/cl.c:502, off by default src/init.c:624): Update_Partial_Lk calls Alias_One_Subpatt on the
     node opposite d before anything else (src/lk.c:1294-1296), tips included.  That function only maintains the host
     application's patt_id / p_lk_loc arrays (src/utilities.c:13547-13666); no likelihood function of the reference reads them
     (they are written at src/lk.c:2501-2513 and in Alias_One_Subpatt, nowhere indexed else), so the option changes no number
     of this path.  The gate is mirrored, the bookkeeping stays the application's: its own function goes here. */
  short    do_alias_subpatt, update_alias_subpatt; /* tree->io->do_alias_subpatt, tree->update_alias_subpatt */
  void   (*alias_one_subpatt)(struct __Node *a, struct __Node *d, struct __Tree *tree);
} t_tree;

#define PHL_N_SPARE 4

/* ---- construction ------------------------------------------------------------------------------ */

/* Topology from edge arrays (left/right node numbers, tips 0..n-1).  neighbour_v/neighbour_b may be NULL
   (neighbour order = edge order) or give the reference's own v[]/b[] order as [2n-2][3] node / edge numbers. */
t_tree *Make_Tree_From_Edges(int n_otu, const int *edge_left, const int *edge_rght, const phydbl *edge_len,
                             const int *neighbour_v, const int *neighbour_b);
t_mod  *Make_Model_Basic(int ns, int n_catg);
void    Free_Model(t_mod *mod);
void    Free_Tree(t_tree *tree);

/* Allocates the device instance (one partials buffer per internal edge side, one matrix per edge), uploads
   weights, +I data and the model.  device < 0: default device. */
void Make_Tree_For_Lk(t_tree *tree, int n_pattern, const phydbl *wght, const short *invar, int device);
/* the multi-GPU form: pattern shards over `devices` (see phyhip_create_instance), one RCCL all-reduce per Lk()/dLk() */
void Make_Tree_For_Lk_On_Devices(t_tree *tree, int n_pattern, const phydbl *wght, const short *invar, const int *devices,
                                 int n_devices, int flags /* 1: sharded even for one device; 2: class axis */);
void Free_Tree_Lk(t_tree *tree);
/* tip data: 0/1 tip vector [pattern][state] (a_nodes[i]->b[0]->p_lk_tip_r) or compact states */
void Init_Partial_Lk_Tips_Double_One_Tip(t_tree *tree, int tax_id, const phydbl *p_lk_tip);
void Init_Partial_Lk_Tips_States_One_Tip(t_tree *tree, int tax_id, const int *states);
/* character encoders (src/lk.c:26-69, 122-161): one alignment character -> ns 0/1 entries at p_lk[pos..]; and a whole
   compressed sequence (n_pattern characters) of one taxon, encoded and uploaded (src/lk.c:2060-2118) */
void Init_Tips_At_One_Site_Nucleotides_Float(char state, int pos, phydbl *p_lk);
void Init_Tips_At_One_Site_AA_Float(char aa, int pos, phydbl *p_lk);
void Init_Partial_Lk_Tips_Chars_One_Tip(t_tree *tree, int tax_id, const char *seq);
/* push model changes: update_beagle_ras / _efrqs / _eigen of the seam (src/beagle_utils.c:273-395) */
void Update_Model_On_Device(t_tree *tree);

/* ---- the surface ----------------------------------------------------------------------------------- */
phydbl Lk(t_edge *b, t_tree *tree);
phydbl dLk(phydbl *l, t_edge *b, t_tree *tree);
void   Update_Partial_Lk(t_tree *tree, t_edge *b, t_node *d);
void   Update_PMat_At_Given_Edge(t_edge *b_fcus, t_tree *tree);
void   Post_Order_Lk(t_node *a, t_node *d, t_tree *tree);
void   Pre_Order_Lk(t_node *a, t_node *d, t_tree *tree);
void   Update_All_Partial_Lk(t_tree *tree);
void   Update_Partial_Lk_Along_A_Path(t_node **path, int path_length, t_tree *tree);
phydbl Update_Lk_At_Given_Edge(t_edge *b_fcus, t_tree *tree); /* src/lk.c:2478-2484 */
void   Update_Eigen_Lr(t_edge *b, t_tree *tree);
void   Set_Both_Sides(int yesno, t_tree *tree);
void   Set_Use_Eigen_Lr(int yesno, t_tree *tree);
void   Set_Update_Eigen_Lr(int yesno, t_tree *tree);
phydbl Br_Len_Newton(phydbl *l, t_edge *b, t_tree *tree);
/* host P-matrix (src/models.c:257-326, 353-373) -- used when tree->host_pmat == YES */
void   PMat(phydbl l, const t_mod *mod, int pos, phydbl *Pij);

/* sharded evaluation: same as Lk(NULL,tree) but the shard's lnL is left in device memory (no sync) */
void   Lk_Shard_Device(t_tree *tree, double *device_out);

/* Caller-side counterpart for tree search: replays a recorded stream of surface calls
   -- the calls spr.c / optimiz.c make through Update_PMat_At_Given_Edge, Update_Partial_Lk, Lk(b), Update_Eigen_Lr
   and dLk (src/spr.c:543,643-646; src/optimiz.c:622-632) -- at buffer-index level, in one C loop, and records the
   scalar every call returned.  Streams recorded from real PhyML searches (oracle/trace_driver.c) use the same
   records with buffer ids in order of first appearance.  kind[i]: */
#define PHL_REC_SET_PMAT 0 /* a = matrix index, x = edge length                               */
#define PHL_REC_UPDATE   1 /* a = dest, b = child1, c = matrix1, d = child2, e = matrix2       */
#define PHL_REC_EDGE_LNL 2 /* a = left buffer, b = right buffer or tip, c = matrix -> out = lnL */
#define PHL_REC_EIGEN_LR 3 /* a = left, b = right                                              */
#define PHL_REC_DLK      4 /* x = length -> out = lnL, out2 = dlnL                             */
#define PHL_REC_EIGEN_LNL 5 /* x = length -> out = lnL: Lk(b) in the eigen basis (src/lk.c:592-603) */
void Replay_Surface_Trace(t_tree *tree, int n_rec, const int *kind, const int *a, const int *b, const int *c, const int *d,
                          const int *e, const phydbl *x, phydbl *out, phydbl *out2);

/* download hooks for host readers (ancestral.c, cv.c, io.c) */
void Get_Partial_Lk(t_tree *tree, t_edge *b, t_node *d, phydbl *p_lk, int *sum_scale);
void Get_Site_Lk(t_tree *tree, phydbl *c_lnL_sorted, phydbl *cur_site_lk, phydbl *unscaled_site_lk_cat, int *fact_sum_scale);

void Set_Exit_Handler(void (*handler)(const char *msg));

#ifdef __cplusplus
}
#endif
#endif
