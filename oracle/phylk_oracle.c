/*
 * phylk_oracle.c -- TEST INFRASTRUCTURE ONLY (see phylk_oracle.h).
 *
 * CPU restatement of the reference's likelihood hot path on flat arrays.  Written from scratch;
 * each function cites the reference lines whose behaviour it follows.  The `arith` switch selects
 * the floating-point operation order of the reference's AVX kernels (src/avx.c: per output state a
 * fused-multiply-add chain over the input states) or of its scalar kernels (src/lk.c: multiply, add).
 */
#include "phylk_oracle.h"

#include <float.h>
#include <math.h>
#include <stddef.h>
#include <string.h>

#define ORC_SMALL      DBL_MIN            /* src/utilities.h:476 */
#define ORC_SMALL_PIJ  1.E-100            /* src/utilities.h:478 */
#define ORC_LARGE      256                /* src/utilities.h:507 */
#define ORC_LOG2       0.69314718055994528623 /* src/utilities.h:267 */
#define ORC_MAX_NS     64

/* ------------------------------------------------------------------------------------------- */
/* K6  tips                                                                                    */
/* ------------------------------------------------------------------------------------------- */

/* allowed-state mask of an IUPAC nucleotide code in A,C,G,T order (src/lk.c:26-69) */
static int nt_mask(unsigned char ch)
{
  switch (ch)
  {
  case 'A': return 1;
  case 'C': return 2;
  case 'G': return 4;
  case 'T': case 'U': return 8;
  case 'M': return 1 | 2;
  case 'R': return 1 | 4;
  case 'W': return 1 | 8;
  case 'S': return 2 | 4;
  case 'Y': return 2 | 8;
  case 'K': return 4 | 8;
  case 'B': return 2 | 4 | 8;
  case 'D': return 1 | 4 | 8;
  case 'H': return 1 | 2 | 8;
  case 'V': return 1 | 2 | 4;
  case 'N': case 'X': case '?': case 'O': case '-': return 15;
  default: return -1;
  }
}

/* amino-acid index in the reference's order ARNDCQEGHILKMFPSTWYV; B->N, Z->Q (src/lk.c:129-151) */
static int aa_index(unsigned char ch)
{
  static const char order[] = "ARNDCQEGHILKMFPSTWYV";
  const char *p;
  if (ch == 'B') return 2;
  if (ch == 'Z') return 5;
  p = (ch != 0) ? strchr(order, ch) : NULL;
  return p ? (int)(p - order) : -1;
}

int orc_init_tip(int datatype, const unsigned char *chars, int n_pattern, double *p_lk_tip,
                 short *d_state, short *is_ambigu)
{
  const int ns = (datatype == ORC_NT) ? 4 : 20;
  for (int p = 0; p < n_pattern; ++p)
  {
    const unsigned char ch = chars[p];
    double *v = p_lk_tip + (size_t)p * ns;
    if (datatype == ORC_NT)
    {
      const int m = nt_mask(ch);
      if (m < 0) return -1;
      for (int s = 0; s < 4; ++s) v[s] = (m >> s) & 1 ? 1.0 : 0.0;
      /* only A,C,G,T,U are unambiguous (src/utilities.c:2985-3003) */
      const int unamb = (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T' || ch == 'U');
      if (is_ambigu) is_ambigu[p] = unamb ? 0 : 1;
      if (d_state) d_state[p] = unamb ? (short)(ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3) : (short)-1;
    }
    else
    {
      /* X ? - are fully ambiguous (src/lk.c:153); '.' is flagged ambiguous too (src/utilities.c:3009) */
      const int gap = (ch == 'X' || ch == '?' || ch == '-');
      if (gap)
        for (int s = 0; s < 20; ++s) v[s] = 1.0;
      else
      {
        const int k = aa_index(ch);
        if (k < 0) return -1;
        for (int s = 0; s < 20; ++s) v[s] = 0.0;
        v[k] = 1.0;
        if (d_state) d_state[p] = (short)k;
      }
      if (is_ambigu) is_ambigu[p] = gap ? 1 : 0;
      if (gap && d_state) d_state[p] = -1;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* K5  transition matrices                                                                     */
/* ------------------------------------------------------------------------------------------- */

void orc_pmat(double len, int ns, const double *U, const double *V, const double *R, double *Pij)
{
  if (len < 0.0)
  { /* src/models.c:331-338,356-361 */
    for (int i = 0; i < ns * ns; ++i) Pij[i] = 0.0;
    for (int i = 0; i < ns; ++i) Pij[i * ns + i] = 1.0;
    return;
  }
  double expt[ORC_MAX_NS], uexpt[ORC_MAX_NS * ORC_MAX_NS];
  for (int k = 0; k < ns; ++k) expt[k] = exp(R[k] * len);                 /* src/models.c:275 */
  for (int i = 0; i < ns; ++i)
    for (int k = 0; k < ns; ++k) uexpt[i * ns + k] = U[i * ns + k] * expt[k]; /* :278-279 */
  for (int i = 0; i < ns; ++i)
  {
    double *row = Pij + i * ns;
    for (int j = 0; j < ns; ++j)
    {
      /* the reference accumulates into memory with separate multiply and add; gcc -O3 with -mfma
         contracts `x += a*b` into an fma, and so does this file (same flags), which keeps the
         golden P-matrices reproducible to the last bit */
      double acc = 0.0;
      for (int k = 0; k < ns; ++k) acc += uexpt[i * ns + k] * V[k * ns + j]; /* :288-292 */
      if (acc < ORC_SMALL_PIJ) acc = ORC_SMALL_PIJ;                          /* :293 */
      row[j] = acc;
    }
    double sum = 0.0;
    for (int j = 0; j < ns; ++j) sum += row[j];                              /* :296-298 */
    for (int j = 0; j < ns; ++j) row[j] /= sum;
  }
}

void orc_update_pmat_edge(double l, int ns, int ncatg, const double *gamma_rr, double br_len_mult,
                          double l_min, double l_max, const double *U, const double *V,
                          const double *R, double *Pij_rr)
{
  for (int c = 0; c < ncatg; ++c)
  {
    double len = (l > 0.0 ? l : 0.0) * gamma_rr[c];   /* src/lk.c:2296 */
    len *= br_len_mult;                               /* :2297 */
    if (len < l_min) len = l_min;                     /* :2299-2300 */
    else if (len > l_max) len = l_max;
    orc_pmat(len, ns, U, V, R, Pij_rr + (size_t)c * ns * ns);
  }
}

/* ------------------------------------------------------------------------------------------- */
/* K1  partial-likelihood update                                                               */
/* ------------------------------------------------------------------------------------------- */

/* u[i] = sum_j P[i][j] v[j] */
static void matvec(const double *P, const double *v, int ns, double *u, int arith)
{
  if (arith)
  { /* src/avx.c:593-616: u = col_0 * v0, then u = fma(col_j, v_j, u).  arith == 2 (the generic loop as the reference's build
       compiles it, src/lk.c:1463-1476: acc = 0, acc += P[i][j] * v[j] contracted to a fused multiply-add chain) gives the same
       doubles: fma(a, b, +0.0) is the rounded product */
    for (int i = 0; i < ns; ++i) u[i] = P[i * ns] * v[0];
    for (int j = 1; j < ns; ++j)
      for (int i = 0; i < ns; ++i) u[i] = fma(P[i * ns + j], v[j], u[i]);
  }
  else
  { /* src/lk.c:3342-3356 */
    for (int i = 0; i < ns; ++i)
    {
      volatile double acc = 0.0; /* volatile: forbid contraction, this is the non-FMA scalar order */
      for (int j = 0; j < ns; ++j) { volatile double t = P[i * ns + j] * v[j]; acc = acc + t; }
      u[i] = acc;
    }
  }
}

void orc_update_partial(int P, int C, int S, const double *wght,
                        const orc_side *v1, const double *Pij1,
                        const orc_side *v2, const double *Pij2,
                        double *plk0, int *sum_scale0, int apply_scaling, int arith)
{
  const int CS = C * S, SS = S * S;
  const double two_to_large = ldexp(1.0, ORC_LARGE), inv_two_to_large = ldexp(1.0, -ORC_LARGE);
  double u1[ORC_MAX_NS], u2[ORC_MAX_NS];

  for (int site = 0; site < P; ++site)
  {
    if (!(wght[site] > ORC_SMALL))
    { /* src/avx.c:399,515-520: zero-weight patterns untouched; the generic loop zeroes them, src/lk.c:1581-1584 */
      if (arith == 2)
        for (int i = 0; i < CS; ++i) plk0[(size_t)site * CS + i] = 0.0;
      continue;
    }

    int amb1 = 1, amb2 = 1, st1 = -1, st2 = -1; /* src/avx.c:401-414 */
    if (v1->is_tip) { amb1 = v1->is_ambigu[site]; if (!amb1) st1 = v1->d_state[site]; }
    if (v2->is_tip) { amb2 = v2->is_ambigu[site]; if (!amb2) st2 = v2->d_state[site]; }
    /* arith == 2: Update_Partial_Lk_Generic under mod->use_m4mod (the `--cov` door, src/cl.c:753-757, src/lk.c:1303-1324):
       every tip counts as ambiguous (src/lk.c:1431-1435: full sums, the same doubles as the look-up) and there is NO all-ones
       shortcut -- a fully ambiguous subtree yields the rounded row sums of the matrices, not exactly 1.0 */
    if (arith == 2) amb1 = amb2 = 1;

    double *out = plk0 + (size_t)site * CS;
    for (int c = 0; c < C; ++c)
    {
      const double *P1 = Pij1 + (size_t)c * SS, *P2 = Pij2 + (size_t)c * SS;
      const double *x1 = v1->is_tip ? v1->p_lk + (size_t)site * S : v1->p_lk + (size_t)site * CS + c * S;
      const double *x2 = v2->is_tip ? v2->p_lk + (size_t)site * S : v2->p_lk + (size_t)site * CS + c * S;
      double *o = out + c * S;

      if (!amb1 && !amb2)
      { /* Exex: src/avx.c:527-536, src/lk.c:3377-3382 */
        for (int i = 0; i < S; ++i) o[i] = P1[i * S + st1] * P2[i * S + st2];
      }
      else if (amb1 && !amb2)
      { /* Exin: src/avx.c:554-564 */
        matvec(P1, x1, S, u1, arith);
        for (int i = 0; i < S; ++i) o[i] = P2[i * S + st2] * u1[i];
      }
      else if (!amb1 && amb2)
      {
        matvec(P2, x2, S, u2, arith);
        for (int i = 0; i < S; ++i) o[i] = P1[i * S + st1] * u2[i];
      }
      else
      { /* Inin with the all-ones shortcut: src/avx.c:575-587, src/lk.c:3338-3361 */
        int k;
        for (k = 0; k < S; ++k)
          if (x1[k] > 1.0 || x1[k] < 1.0 || x2[k] > 1.0 || x2[k] < 1.0) break;
        if (k != S || arith == 2)
        {
          matvec(P1, x1, S, u1, arith);
          matvec(P2, x2, S, u2, arith);
          for (int i = 0; i < S; ++i) o[i] = u1[i] * u2[i];
        }
        else
          for (int i = 0; i < S; ++i) o[i] = 1.0;
      }
    }

    /* SCALE_FAST: src/avx.c:460-513 */
    const int s1 = v1->sum_scale ? v1->sum_scale[site] : 0;
    const int s2 = v2->sum_scale ? v2->sum_scale[site] : 0;
    sum_scale0[site] = s1 + s2;
    double largest = -DBL_MAX;
    for (int i = 0; i < CS; ++i) if (out[i] > largest) largest = out[i];
    if (largest < inv_two_to_large && apply_scaling)
    {
      for (int i = 0; i < CS; ++i) out[i] *= two_to_large;
      sum_scale0[site] += ORC_LARGE;
    }
  }
}

/* ------------------------------------------------------------------------------------------- */
/* K2  edge log-likelihood                                                                     */
/* ------------------------------------------------------------------------------------------- */

/* one rate class at one site; Pij rows are indexed by the RIGHT-side state */
static double one_class(const double *left, const double *rght, const double *Pij, const double *pi,
                        int ns, int ambiguity_check, int state, int arith)
{
  if (arith)
  { /* src/avx.c:110-215 */
    if (ambiguity_check == 0)
    {
      const double *row = Pij + state * ns;
      if (ns == 4)
      { /* :121-123: elementwise product, then AVX_Vect_Norm = (x0+x2)+(x1+x3) (src/avx.c:281-289) */
        /* (volatile: the reference multiplies as a vector and hands the products to AVX_Vect_Norm, a call -- its binary has no fused
           multiply-add here, and this restatement must not grow one under -mfma) */
        volatile double q0 = row[0] * left[0], q1 = row[1] * left[1], q2 = row[2] * left[2], q3 = row[3] * left[3];
        return pi[state] * ((q0 + q2) + (q1 + q3));
      }
      /* :163-176: the blockwise products (stored, in the reference's binary, before they are added: no fused multiply-add), their
         per-lane sums in block order, then the horizontal norm */
      double lane[4] = {0., 0., 0., 0.};
      for (int b = 0; b < ns / 4; ++b)
        for (int k = 0; k < 4; ++k)
        {
          volatile double prod = row[b * 4 + k] * left[b * 4 + k];
          lane[k] = lane[k] + prod;
        }
      return pi[state] * ((lane[0] + lane[2]) + (lane[1] + lane[3]));
    }
    else
    {
      /* :130-148 / :184-210: pijplk[k] = fma-chain over left states i of P[k][i]*left[i] starting
         from 0 (fma(P,left,0)), then * (rght[k]*pi[k]), then the blockwise horizontal norm */
      double acc[ORC_MAX_NS];
      for (int k = 0; k < ns; ++k) acc[k] = 0.0;
      for (int i = 0; i < ns; ++i)
        for (int k = 0; k < ns; ++k) acc[k] = fma(Pij[k * ns + i], left[i], acc[k]);
      double lk = 0.0;
      for (int b = 0; b < ns / 4; ++b)
      {
        volatile double q0 = acc[b * 4 + 0] * (rght[b * 4 + 0] * pi[b * 4 + 0]);
        volatile double q1 = acc[b * 4 + 1] * (rght[b * 4 + 1] * pi[b * 4 + 1]);
        volatile double q2 = acc[b * 4 + 2] * (rght[b * 4 + 2] * pi[b * 4 + 2]);
        volatile double q3 = acc[b * 4 + 3] * (rght[b * 4 + 3] * pi[b * 4 + 3]);
        const double nrm = (q0 + q2) + (q1 + q3);
        lk = (ns == 4) ? nrm : lk + nrm;
      }
      return lk;
    }
  }
  else
  { /* src/lk.c:1185-1218 */
    double lk = 0.0;
    if (ambiguity_check == 0)
    {
      volatile double sum = 0.0;
      const double *row = Pij + state * ns;
      for (int l = 0; l < ns; ++l) { volatile double t = row[l] * left[l]; sum = sum + t; }
      lk += sum * pi[state];
    }
    else
    {
      for (int k = 0; k < ns; ++k)
      {
        if (rght[k] > 0.0)
        {
          volatile double sum = 0.0;
          for (int l = 0; l < ns; ++l) { volatile double t = Pij[k * ns + l] * left[l]; sum = sum + t; }
          volatile double t2 = sum * pi[k];
          volatile double t3 = t2 * rght[k];
          lk = lk + t3;
        }
      }
    }
    return lk;
  }
}

/* src/lk.c:1226-1273; *issue is set when the scaled invariant likelihood overflows */
static double invariant_lk(int fact_sum_scale, int site, const short *invar, const double *pi,
                           int apply_scaling, int *issue)
{
  double inv = 0.0;
  *issue = 0;
  if (invar[site] > -0.5)
  {
    inv = pi[invar[site]];
    if (apply_scaling)
    {
      int exponent = fact_sum_scale;
      do
      {
        const int piece = exponent < 63 ? exponent : 63;
        inv *= (double)((unsigned long long)1 << piece);
        exponent -= piece;
      } while (exponent != 0);
    }
    if (isinf(inv)) *issue = 1;
  }
  return inv;
}

/* the tail of Lk_Core shared by the plain and the eigen-basis form: src/lk.c:816-857 / :906-947 */
static double site_tail(double site_lk, int site, int *fact, int invar_model, double pinvar,
                        const short *invar, const double *pi, int apply_scaling, int *warn)
{
  if (invar_model)
  {
    int issue = 0;
    double inv = invariant_lk(fact[site], site, invar, pi, apply_scaling, &issue);
    if (issue)
    {
      fact[site] = 0;
      inv = invariant_lk(0, site, invar, pi, apply_scaling, &issue);
      site_lk = inv * pinvar;
    }
    else
      site_lk = site_lk * (1. - pinvar) + inv * pinvar;
  }
  if (site_lk < ORC_SMALL) { site_lk = ORC_SMALL; if (warn) *warn = 1; }
  return log(site_lk) - (double)ORC_LOG2 * fact[site];
}

double orc_edge_lnl(int P, int C, int S, const double *wght,
                    const orc_side *left, const orc_side *rght, const double *Pij_rr,
                    const double *pi, const double *cat_w,
                    int invar_model, double pinvar, const short *invar,
                    int apply_scaling, int arith,
                    double *c_lnL_sorted, double *cur_site_lk, double *unscaled_site_lk_cat,
                    int *fact_sum_scale, int *numerical_warning)
{
  const int CS = C * S, SS = S * S;
  double lnL = 0.0;
  if (numerical_warning) *numerical_warning = 0;
  for (int site = 0; site < P; ++site)
  {
    if (!(wght[site] > ORC_SMALL)) continue; /* src/lk.c:632 */
    int amb = -1, state = -1;                /* src/lk.c:610-621: only a right-hand tip can be "observed" */
    if (rght->is_tip) { amb = rght->is_ambigu[site]; if (amb == 0) state = rght->d_state[site]; }

    double cat_lk[64];
    for (int c = 0; c < C; ++c)
    {
      const double *l = left->is_tip ? left->p_lk + (size_t)site * S : left->p_lk + (size_t)site * CS + c * S;
      const double *r = rght->is_tip ? rght->p_lk + (size_t)site * S : rght->p_lk + (size_t)site * CS + c * S;
      cat_lk[c] = one_class(l, r, Pij_rr + (size_t)c * SS, pi, S, amb, state, arith);
    }
    /* Pull_Scaling_Factors, SCALE_FAST: src/lk.c:2701-2705,2777-2801 */
    int fact = 0;
    if (apply_scaling)
      fact = (left->sum_scale ? left->sum_scale[site] : 0) + (rght->sum_scale ? rght->sum_scale[site] : 0);
    int fact_local[1];
    double site_lk = 0.0;
    for (int c = 0; c < C; ++c)
    {
      if (unscaled_site_lk_cat) unscaled_site_lk_cat[(size_t)site * C + c] = cat_lk[c];
      volatile double t = cat_lk[c] * cat_w[c]; /* src/lk.c:818 */
      site_lk = site_lk + t;
    }
    /* site_tail indexes fact[site]; hand it a one-element view */
    fact_local[0] = fact;
    const short inv_one = invar ? invar[site] : (short)-1;
    const double lsl = site_tail(site_lk, 0, fact_local, invar_model, pinvar, &inv_one, pi, apply_scaling,
                                 numerical_warning);
    if (fact_sum_scale) fact_sum_scale[site] = fact_local[0];
    if (c_lnL_sorted) c_lnL_sorted[site] = lsl;
    if (cur_site_lk) cur_site_lk[site] = exp(lsl);
    {
      volatile double t = wght[site] * lsl; /* src/lk.c:856 */
      lnL = lnL + t;
    }
  }
  return lnL;
}

/* ------------------------------------------------------------------------------------------- */
/* K3 / K4  eigen basis                                                                        */
/* ------------------------------------------------------------------------------------------- */

void orc_update_eigen_lr(int P, int C, int S, const double *wght,
                         const orc_side *left, const orc_side *rght,
                         const double *R, const double *L, const double *pi,
                         double *dot_prod, int arith)
{
  const int CS = C * S;
  double lp[ORC_MAX_NS], a[ORC_MAX_NS], b[ORC_MAX_NS];
  for (int site = 0; site < P; ++site)
  {
    if (!(wght[site] > ORC_SMALL)) continue; /* src/avx.c:75,94-103 */
    for (int c = 0; c < C; ++c)
    {
      const double *l = left->is_tip ? left->p_lk + (size_t)site * S : left->p_lk + (size_t)site * CS + c * S;
      const double *r = rght->is_tip ? rght->p_lk + (size_t)site * S : rght->p_lk + (size_t)site * CS + c * S;
      double *dp = dot_prod + (size_t)site * CS + c * S;
      if (arith)
      { /* src/avx.c:79-84: (R^T (l.pi)) * (L r) with column-wise FMA chains.  _r_ev holds rows of R
           (r_ev + i*ns), so AVX_Matrix_Vect_Prod yields a[k] = sum_i R[i][k] * lp[i];  _l_ev holds rows
           of the *transposed* L, yielding b[k] = sum_i L[k][i] * r[i]. */
        for (int i = 0; i < S; ++i) lp[i] = l[i] * pi[i];
        for (int k = 0; k < S; ++k) { a[k] = R[k] * lp[0]; b[k] = L[k * S] * r[0]; }
        for (int i = 1; i < S; ++i)
          for (int k = 0; k < S; ++k)
          {
            a[k] = fma(R[i * S + k], lp[i], a[k]);
            b[k] = fma(L[k * S + i], r[i], b[k]);
          }
        for (int k = 0; k < S; ++k) dp[k] = a[k] * b[k];
      }
      else
      { /* src/lk.c:1086-1095 */
        for (int i = 0; i < S; ++i)
        {
          volatile double lft = 0.0, rgt = 0.0;
          for (int j = 0; j < S; ++j)
          {
            volatile double t1 = R[j * S + i] * l[j];
            volatile double t2 = t1 * pi[j];
            lft = lft + t2;
            volatile double t3 = L[i * S + j] * r[j];
            rgt = rgt + t3;
          }
          dp[i] = lft * rgt;
        }
      }
    }
  }
}

static double clamp(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

void orc_dlk(double *l, int P, int C, int S, const double *wght, const double *dot_prod,
             const double *e_val, const double *gamma_rr, const double *cat_w, double br_len_mult,
             double l_min, double l_max, int invar_model, double pinvar, const short *invar,
             const double *pi, const int *fact_sum_scale, int apply_scaling,
             double *lnL, double *dlnL)
{
  const int CS = C * S;
  double expl[2 * CS]; /* [category][state][value, derivative] */
  *l = clamp(*l, l_min, l_max);                                  /* src/lk.c:673-674 */
  for (int c = 0; c < C; ++c)
  {
    const double rr = gamma_rr[c] * br_len_mult;                 /* :690-692 */
    const double len = clamp((*l) * rr, l_min, l_max);           /* :694,704-705 */
    for (int s = 0; s < S; ++s)
    {
      const double ev = e_val[s], ex = exp(ev * len);            /* :712-713 */
      expl[c * 2 * S + 2 * s]     = ex;                          /* :723 */
      expl[c * 2 * S + 2 * s + 1] = ex * ev * rr;                /* :724 */
    }
  }
  double dlnlk = 0.0, lnlk = 0.0;
  for (int site = 0; site < P; ++site)
  {
    if (!(wght[site] > ORC_SMALL)) continue;
    double lk = 0.0, dlk = 0.0;
    for (int c = 0; c < C; ++c)
    { /* src/avx.c:250-276: lanes (lk,dlk,lk,dlk) accumulate pairs of states, then lane0+lane2 / lane1+lane3 */
      const double *dp = dot_prod + (size_t)site * CS + c * S, *ex = expl + c * 2 * S;
      double z0 = 0., z1 = 0., z2 = 0., z3 = 0.;
      for (int i = 0; i < S / 2; ++i)
      {
        z0 = fma(dp[2 * i], ex[4 * i], z0);
        z1 = fma(dp[2 * i], ex[4 * i + 1], z1);
        z2 = fma(dp[2 * i + 1], ex[4 * i + 2], z2);
        z3 = fma(dp[2 * i + 1], ex[4 * i + 3], z3);
      }
      volatile double t1 = (z0 + z2) * cat_w[c], t2 = (z1 + z3) * cat_w[c]; /* src/lk.c:996-997 */
      lk = lk + t1;
      dlk = dlk + t2;
    }
    if (invar_model)
    { /* src/lk.c:1005-1025 */
      int issue = 0;
      const double inv = invariant_lk(fact_sum_scale[site], site, invar, pi, apply_scaling, &issue);
      if (issue) { lk = inv * pinvar; dlk = 0.0; }
      else { lk = lk * (1. - pinvar) + inv * pinvar; dlk = dlk * (1. - pinvar); }
    }
    if (lk < ORC_SMALL) lk = ORC_SMALL;                          /* :1027-1031 */
    dlk /= lk;                                                   /* :742 */
    {
      volatile double t1 = wght[site] * dlk;
      volatile double t2 = wght[site] * (log(lk) - (double)ORC_LOG2 * fact_sum_scale[site]); /* :744-745 */
      dlnlk = dlnlk + t1;
      lnlk = lnlk + t2;
    }
  }
  *lnL = lnlk;
  *dlnL = dlnlk;
}

double orc_lk_eigen(double l, int P, int C, int S, const double *wght, const double *dot_prod,
                    const double *e_val, const double *gamma_rr, const double *cat_w,
                    double br_len_mult, double l_min, double l_max, int invar_model, double pinvar,
                    const short *invar, const double *pi, const int *fact_sum_scale, int apply_scaling)
{
  const int CS = C * S;
  double expl[CS];
  for (int c = 0; c < C; ++c)
  { /* src/lk.c:594-602 */
    double len = (l > 0.0 ? l : 0.0) * gamma_rr[c];
    len *= br_len_mult;
    len = clamp(len, l_min, l_max);
    for (int s = 0; s < S; ++s) expl[c * S + s] = exp(e_val[s] * len);
  }
  double lnL = 0.0;
  for (int site = 0; site < P; ++site)
  {
    if (!(wght[site] > ORC_SMALL)) continue;
    double site_lk = 0.0;
    for (int c = 0; c < C; ++c)
    { /* src/avx.c:220-245: elementwise products, blockwise add, horizontal norm */
      const double *dp = dot_prod + (size_t)site * CS + c * S, *ex = expl + c * S;
      double lane[4] = {0., 0., 0., 0.};
      for (int b = 0; b < S / 4; ++b)
        for (int k = 0; k < 4; ++k) { volatile double q = dp[b * 4 + k] * ex[b * 4 + k]; lane[k] = lane[k] + q; }
      volatile double t = ((lane[0] + lane[2]) + (lane[1] + lane[3])) * cat_w[c];
      site_lk = site_lk + t;
    }
    int fact_local[1] = {fact_sum_scale[site]};
    const short inv_one = invar ? invar[site] : (short)-1;
    const double lsl = site_tail(site_lk, 0, fact_local, invar_model, pinvar, &inv_one, pi, apply_scaling, NULL);
    volatile double t = wght[site] * lsl;
    lnL = lnL + t;
  }
  return lnL;
}
