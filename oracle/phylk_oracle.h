/*
 * phylk_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of PhyML's Felsenstein-pruning likelihood path (the functions SURVEY.md
 * section 8a lists), written from scratch against flat arrays.  It exists so that tests/, the
 * smoke test and bench.py's `cpu_baseline` leg can check the HIP engine; nothing under phyml_amd/
 * (the product) may include, link or call it.
 *
 * Pinning: every function below is checked in tests/test_oracle_golden.py against golden vectors
 * dumped from the REAL reference (oracle/ref_driver.c linked with the reference's own objects):
 * partial vectors and scale vectors bit-for-bit, per-site log-likelihoods, lnL, dot products, dlnL.
 *
 * Layouts (same as the reference):
 *   partial vector  [pattern][catg][state]      tip vector [pattern][state] (0/1 doubles)
 *   Pij_rr          [catg][from i][to j]        scale vector [pattern] int
 */
#ifndef PHYLK_ORACLE_H
#define PHYLK_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NT 0
#define ORC_AA 1

/* one side of an edge as the kernels see it */
typedef struct
{
  const double *p_lk;       /* internal: [P][C][S]; tip: [P][S] */
  const int    *sum_scale;  /* [P] or NULL (tips) */
  int           is_tip;
  const short  *is_ambigu;  /* tips only, [P] */
  const short  *d_state;    /* tips only, [P] */
} orc_side;

/* K6: character -> 0/1 state vector, digit state and ambiguity flag.
   ref: src/lk.c:26-69 (nt), :122-161 (aa); Is_Ambigu / Assign_State src/utilities.c:2979-3039,3399-3408.
   Returns 0, or -1 on an unknown character. */
int orc_init_tip(int datatype, const unsigned char *chars, int n_pattern, double *p_lk_tip,
                 short *d_state, short *is_ambigu);

/* K5: P = U diag(exp(lambda*len)) V, floor 1e-100, row-renormalise; len < 0 -> identity.
   ref: src/models.c:257-326 (PMat_Empirical), :331-338, :353-373 (PMat). */
void orc_pmat(double len, int ns, const double *r_e_vect, const double *l_e_vect, const double *e_val,
              double *Pij);

/* a12: all rate classes of one edge.  len_c = MAX(0,l)*rate_c*br_len_mult clamped to [l_min,l_max].
   ref: src/lk.c:2280-2316 (Update_PMat_At_Given_Edge). */
void orc_update_pmat_edge(double l, int ns, int ncatg, const double *gamma_rr, double br_len_mult,
                          double l_min, double l_max, const double *r_e_vect, const double *l_e_vect,
                          const double *e_val, double *Pij_rr);

/* K1: one edge-side partial vector from its two children, then the 2^256 rescale rule.
   arith = 1 follows the AVX kernel's operation order (src/avx.c:301-634: column-wise FMA chain),
   arith = 0 the scalar kernel's (src/lk.c:1659-1768,3328-3406: row-wise multiply-add). */
void orc_update_partial(int n_pattern, int ncatg, int ns, const double *wght,
                        const orc_side *v1, const double *Pij1,
                        const orc_side *v2, const double *Pij2,
                        double *plk0, int *sum_scale0, int apply_scaling, int arith);

/* K2: root-edge site likelihoods and the weighted log sum.
   ref: src/lk.c:608-645 (site loop of Lk), :767-861 (Lk_Core), src/avx.c:110-215 /
   src/lk.c:1185-1218 (one class), :2696-2803 (Pull_Scaling_Factors, SCALE_FAST), :1226-1273 (Invariant_Lk).
   Per-site outputs may be NULL.  Returns lnL; *numerical_warning is set like src/lk.c:847-851. */
double orc_edge_lnl(int n_pattern, int ncatg, int ns, const double *wght,
                    const orc_side *left, const orc_side *rght, const double *Pij_rr,
                    const double *pi, const double *gamma_r_proba,
                    int invar_model, double pinvar, const short *invar,
                    int apply_scaling, int arith,
                    double *c_lnL_sorted, double *cur_site_lk, double *unscaled_site_lk_cat,
                    int *fact_sum_scale, int *numerical_warning);

/* K3: dot_prod[site][c][i] = (sum_j R[j][i] pi_j left_j) * (sum_j L[i][j] rght_j).
   ref: src/lk.c:1038-1114, src/avx.c:21-105. */
void orc_update_eigen_lr(int n_pattern, int ncatg, int ns, const double *wght,
                         const orc_side *left, const orc_side *rght,
                         const double *r_e_vect, const double *l_e_vect, const double *pi,
                         double *dot_prod, int arith);

/* K4: lnL and dlnL/dl from dot_prod in the eigen basis.  *l is clamped in place.
   ref: src/lk.c:655-753 (dLk), :955-1032 (Lk_dLk_Core_Eigen_Lr), src/avx.c:250-276.
   fact_sum_scale is the vector the preceding edge evaluation left behind. */
void orc_dlk(double *l, int n_pattern, int ncatg, int ns, const double *wght, const double *dot_prod,
             const double *e_val, const double *gamma_rr, const double *gamma_r_proba, double br_len_mult,
             double l_min, double l_max, int invar_model, double pinvar, const short *invar,
             const double *pi, const int *fact_sum_scale, int apply_scaling,
             double *lnL, double *dlnL);

/* Lk(b) with use_eigen_lr = YES: lnL from dot_prod * exp(lambda*len_c).
   ref: src/lk.c:592-603,625-629 and :866-950 (Lk_Core_Eigen_Lr). */
double orc_lk_eigen(double l, int n_pattern, int ncatg, int ns, const double *wght, const double *dot_prod,
                    const double *e_val, const double *gamma_rr, const double *gamma_r_proba,
                    double br_len_mult, double l_min, double l_max, int invar_model, double pinvar,
                    const short *invar, const double *pi, const int *fact_sum_scale, int apply_scaling);

#ifdef __cplusplus
}
#endif
#endif
