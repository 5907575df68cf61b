/*
 * gar_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 * See gar_oracle.h for the contract.  Every function cites the reference
 * (Simple-Robotics/aligator) file:line whose arithmetic it restates.
 * Nothing here is copied from the reference: the reference is Eigen
 * expression templates, this is plain loops over raw arrays.
 */
#include "gar_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* small dense helpers (column-major unless stated)                          */
/* ------------------------------------------------------------------------ */
static double *dalloc(long n) {
  if (n <= 0)
    n = 1;
  return (double *)calloc((size_t)n, sizeof(double));
}
static void dcopy(long n, const double *src, double *dst) {
  if (n > 0)
    memcpy(dst, src, (size_t)n * sizeof(double));
}
#define CM(a, ld, i, j) ((a)[(long)(j) * (ld) + (i)]) /* column-major */
#define RM(a, ld, i, j) ((a)[(long)(i) * (ld) + (j)]) /* row-major    */

/* C(MxN, generic strides) += alpha * A(MxK) * B(KxN), generic strides.
 * The loop nest is chosen so that the innermost loop runs over whichever
 * index is unit-stride in C when possible (axpy form). */
static void mm_acc(int M, int N, int K, double alpha, const double *A, long ars,
                   long acs, const double *B, long brs, long bcs, double *C,
                   long crs, long ccs) {
  if (M <= 0 || N <= 0 || K <= 0)
    return;
  if (crs == 1 && ars == 1) { /* column-major C and A: axpy over i */
    for (int j = 0; j < N; ++j) {
      double *cj = C + (long)j * ccs;
      for (int k = 0; k < K; ++k) {
        const double b = alpha * B[(long)k * brs + (long)j * bcs];
        const double *ak = A + (long)k * acs;
        for (int i = 0; i < M; ++i)
          cj[i] += ak[i] * b;
      }
    }
  } else if (ccs == 1 && bcs == 1) { /* row-major C and B: axpy over j */
    for (int i = 0; i < M; ++i) {
      double *ci = C + (long)i * crs;
      for (int k = 0; k < K; ++k) {
        const double a = alpha * A[(long)i * ars + (long)k * acs];
        const double *bk = B + (long)k * brs;
        for (int j = 0; j < N; ++j)
          ci[j] += a * bk[j];
      }
    }
  } else if (acs == 1 && brs == 1) { /* dot-product form, contiguous k */
    /* register-blocked: 2 rows of A against 4 columns of B, eight SIMD partial-sum accumulators
     * (what an optimised BLAS / Eigen's gebp kernel does for these sizes; the summation order inside a
     * dot product is the vector unit's, as there) */
    int i = 0;
    for (; i + 2 <= M; i += 2) {
      const double *a0 = A + (long)i * ars, *a1 = a0 + ars;
      int j = 0;
      for (; j + 4 <= N; j += 4) {
        const double *b0 = B + (long)j * bcs, *b1 = b0 + bcs, *b2 = b1 + bcs, *b3 = b2 + bcs;
        double s00 = 0, s01 = 0, s02 = 0, s03 = 0, s10 = 0, s11 = 0, s12 = 0, s13 = 0;
#pragma omp simd reduction(+ : s00, s01, s02, s03, s10, s11, s12, s13)
        for (int k = 0; k < K; ++k) {
          const double x0 = a0[k], x1 = a1[k];
          s00 += x0 * b0[k]; s01 += x0 * b1[k]; s02 += x0 * b2[k]; s03 += x0 * b3[k];
          s10 += x1 * b0[k]; s11 += x1 * b1[k]; s12 += x1 * b2[k]; s13 += x1 * b3[k];
        }
        double *c0 = C + (long)i * crs + (long)j * ccs, *c1 = c0 + crs;
        c0[0] += alpha * s00; c0[ccs] += alpha * s01; c0[2 * ccs] += alpha * s02; c0[3 * ccs] += alpha * s03;
        c1[0] += alpha * s10; c1[ccs] += alpha * s11; c1[2 * ccs] += alpha * s12; c1[3 * ccs] += alpha * s13;
      }
      for (; j < N; ++j) {
        const double *bj = B + (long)j * bcs;
        double s0 = 0.0, s1 = 0.0;
#pragma omp simd reduction(+ : s0, s1)
        for (int k = 0; k < K; ++k) {
          s0 += a0[k] * bj[k];
          s1 += a1[k] * bj[k];
        }
        C[(long)i * crs + (long)j * ccs] += alpha * s0;
        C[(long)(i + 1) * crs + (long)j * ccs] += alpha * s1;
      }
    }
    for (; i < M; ++i)
      for (int j = 0; j < N; ++j) {
        const double *ai = A + (long)i * ars;
        const double *bj = B + (long)j * bcs;
        double s = 0.0;
#pragma omp simd reduction(+ : s)
        for (int k = 0; k < K; ++k)
          s += ai[k] * bj[k];
        C[(long)i * crs + (long)j * ccs] += alpha * s;
      }
  } else {
    for (int j = 0; j < N; ++j)
      for (int i = 0; i < M; ++i) {
        double s = 0.0;
        for (int k = 0; k < K; ++k)
          s += A[(long)i * ars + (long)k * acs] * B[(long)k * brs + (long)j * bcs];
        C[(long)i * crs + (long)j * ccs] += alpha * s;
      }
  }
}

static double inf_norm(int n, const double *x) {
  double m = 0.0;
  for (int i = 0; i < n; ++i) {
    double a = fabs(x[i]);
    if (a > m || a != a)
      m = a;
  }
  return m;
}

/* ------------------------------------------------------------------------ */
/* LqrKnotTpl / LqrProblemTpl (lqr-problem.hxx:28-72, 232-241, 267-283)      */
/* ------------------------------------------------------------------------ */
static void knot_alloc(ora_knot *k, int nx, int nu, int nc, int nx2, int nth) {
  k->nx = nx;
  k->nu = nu;
  k->nc = nc;
  k->nx2 = nx2;
  k->nth = nth;
  k->Q = dalloc((long)nx * nx);
  k->S = dalloc((long)nx * nu);
  k->R = dalloc((long)nu * nu);
  k->q = dalloc(nx);
  k->r = dalloc(nu);
  k->A = dalloc((long)nx2 * nx);
  k->B = dalloc((long)nx2 * nu);
  k->f = dalloc(nx2);
  k->C = dalloc((long)nc * nx);
  k->D = dalloc((long)nc * nu);
  k->d = dalloc(nc);
  k->Gth = dalloc((long)nth * nth);
  k->Gx = dalloc((long)nx * nth);
  k->Gu = dalloc((long)nu * nth);
  k->Gv = dalloc((long)nc * nth);
  k->gamma = dalloc(nth);
}
static void knot_free(ora_knot *k) {
  free(k->Q); free(k->S); free(k->R); free(k->q); free(k->r);
  free(k->A); free(k->B); free(k->f);
  free(k->C); free(k->D); free(k->d);
  free(k->Gth); free(k->Gx); free(k->Gu); free(k->Gv); free(k->gamma);
}
static void knot_copy(const ora_knot *s, ora_knot *d) {
  knot_alloc(d, s->nx, s->nu, s->nc, s->nx2, s->nth);
  dcopy((long)s->nx * s->nx, s->Q, d->Q);
  dcopy((long)s->nx * s->nu, s->S, d->S);
  dcopy((long)s->nu * s->nu, s->R, d->R);
  dcopy(s->nx, s->q, d->q);
  dcopy(s->nu, s->r, d->r);
  dcopy((long)s->nx2 * s->nx, s->A, d->A);
  dcopy((long)s->nx2 * s->nu, s->B, d->B);
  dcopy(s->nx2, s->f, d->f);
  dcopy((long)s->nc * s->nx, s->C, d->C);
  dcopy((long)s->nc * s->nu, s->D, d->D);
  dcopy(s->nc, s->d, d->d);
  dcopy((long)s->nth * s->nth, s->Gth, d->Gth);
  dcopy((long)s->nx * s->nth, s->Gx, d->Gx);
  dcopy((long)s->nu * s->nth, s->Gu, d->Gu);
  dcopy((long)s->nc * s->nth, s->Gv, d->Gv);
  dcopy(s->nth, s->gamma, d->gamma);
}

ora_problem *ora_problem_new(int N, const int *dims5, int nc0) {
  ora_problem *p = (ora_problem *)calloc(1, sizeof(ora_problem));
  p->N = N;
  p->nc0 = nc0;
  p->stages = (ora_knot *)calloc((size_t)(N + 1), sizeof(ora_knot));
  for (int t = 0; t <= N; ++t) {
    const int *dm = dims5 + 5 * t;
    knot_alloc(&p->stages[t], dm[0], dm[1], dm[2], dm[3], dm[4]);
  }
  int nx0 = p->stages[0].nx;
  p->G0 = dalloc((long)nc0 * nx0); /* lqr-problem.hxx:267-273 */
  p->g0 = dalloc(nc0);
  return p;
}
ora_problem *ora_problem_copy(const ora_problem *s) {
  ora_problem *p = (ora_problem *)calloc(1, sizeof(ora_problem));
  p->N = s->N;
  p->nc0 = s->nc0;
  p->stages = (ora_knot *)calloc((size_t)(s->N + 1), sizeof(ora_knot));
  for (int t = 0; t <= s->N; ++t)
    knot_copy(&s->stages[t], &p->stages[t]);
  int nx0 = p->stages[0].nx;
  p->G0 = dalloc((long)s->nc0 * nx0);
  p->g0 = dalloc(s->nc0);
  dcopy((long)s->nc0 * nx0, s->G0, p->G0);
  dcopy(s->nc0, s->g0, p->g0);
  return p;
}
void ora_problem_free(ora_problem *p) {
  if (!p)
    return;
  for (int t = 0; t <= p->N; ++t)
    knot_free(&p->stages[t]);
  free(p->stages);
  free(p->G0);
  free(p->g0);
  free(p);
}
/* lqr-problem.hxx:232-241 */
void ora_knot_add_parameterization(ora_knot *k, int nth) {
  free(k->Gth); free(k->Gx); free(k->Gu); free(k->Gv); free(k->gamma);
  k->nth = nth;
  k->Gth = dalloc((long)nth * nth);
  k->Gx = dalloc((long)k->nx * nth);
  k->Gu = dalloc((long)k->nu * nth);
  k->Gv = dalloc((long)k->nc * nth);
  k->gamma = dalloc(nth);
}
/* lqr-problem.hpp:156-162 */
void ora_problem_add_parameterization(ora_problem *p, int nth) {
  for (int t = 0; t <= p->N; ++t)
    ora_knot_add_parameterization(&p->stages[t], nth);
}
double *ora_knot_block(ora_problem *p, int t, const char *name) {
  ora_knot *k = &p->stages[t];
#define BLK(n) if (!strcmp(name, #n)) return k->n
  BLK(Q); BLK(S); BLK(R); BLK(q); BLK(r); BLK(A); BLK(B); BLK(f);
  BLK(C); BLK(D); BLK(d); BLK(Gth); BLK(Gx); BLK(Gu); BLK(Gv); BLK(gamma);
#undef BLK
  return NULL;
}
const int *ora_knot_dims(const ora_problem *p, int t) { return &p->stages[t].nx; }

/* ------------------------------------------------------------------------ */
/* BunchKaufman (core/bunchkaufman.hpp)                                      */
/* ------------------------------------------------------------------------ */
#define BK_BLOCKSIZE 32 /* bunchkaufman.hpp:531 */
#define BK_OK 0
#define BK_ISSUE 1

ora_bk *ora_bk_new(int n) {
  ora_bk *bk = (ora_bk *)calloc(1, sizeof(ora_bk));
  bk->n = n;
  bk->L = dalloc((long)n * n);
  bk->subdiag = dalloc(n);
  bk->piv = (int *)calloc((size_t)(n > 0 ? n : 1), sizeof(int));
  bk->blocksize = n <= BK_BLOCKSIZE ? 0 : BK_BLOCKSIZE; /* :362-369, :547 */
  bk->W = dalloc((long)n * BK_BLOCKSIZE);
  bk->info = BK_ISSUE;
  return bk;
}
void ora_bk_free(ora_bk *bk) {
  if (!bk)
    return;
  free(bk->L);
  free(bk->subdiag);
  free(bk->piv);
  free(bk->W);
  free(bk);
}

static const double BK_ALPHA_NUM = 17.0;
static double bk_alpha(void) { return (1.0 + sqrt(BK_ALPHA_NUM)) / 8.0; } /* :29 */

static void swapd(double *a, double *b) {
  double t = *a;
  *a = *b;
  *b = t;
}

/* bunch_kaufman_in_place_unblocked (bunchkaufman.hpp:23-169); a is n x n,
 * col-major with leading dimension lda, only the lower triangle is used. */
static int bk_unblocked(double *a, int lda, int n, int *piv, int *pivot_count) {
#define A_(i, j) CM(a, lda, i, j)
  const double alpha = bk_alpha();
  *pivot_count = 0;
  if (n == 0)
    return BK_OK;
  if (n == 1) { /* :36-43 */
    if (fabs(A_(0, 0)) == 0.0)
      return BK_ISSUE;
    A_(0, 0) = 1.0 / A_(0, 0);
    return BK_OK;
  }
  int k = 0;
  while (k < n) {
    int k_step = 1;
    double abs_akk = fabs(A_(k, k));
    int imax = 0;
    double colmax = 0.0;
    if (k + 1 < n) { /* :52-54, first maximiser */
      colmax = fabs(A_(k + 1, k));
      for (int i = k + 2; i < n; ++i)
        if (fabs(A_(i, k)) > colmax) {
          colmax = fabs(A_(i, k));
          imax = i - (k + 1);
        }
    }
    imax += k + 1;
    int kp;
    if (fmax(abs_akk, colmax) == 0.0)
      return BK_ISSUE; /* :58-59 */
    if (abs_akk >= colmax * alpha) {
      kp = k;
    } else {
      double rowmax = 0.0; /* :63-73 */
      for (int j = k; j < imax; ++j)
        rowmax = fmax(rowmax, fabs(A_(imax, j)));
      for (int i = imax + 1; i < n; ++i)
        rowmax = fmax(rowmax, fabs(A_(i, imax)));
      if (abs_akk >= (alpha * colmax) * (colmax / rowmax)) {
        kp = k;
      } else if (fabs(A_(imax, imax)) >= alpha * rowmax) {
        kp = imax;
      } else {
        kp = imax;
        k_step = 2;
      }
    }
    int kk = k + k_step - 1;
    if (kp != kk) { /* symmetric interchange, :86-102 */
      *pivot_count += 1;
      for (int i = kp + 1; i < n; ++i)
        swapd(&A_(i, kk), &A_(i, kp));
      for (int j = kk + 1; j < kp; ++j) {
        double tmp = A_(j, kk);
        A_(j, kk) = A_(kp, j);
        A_(kp, j) = tmp;
      }
      swapd(&A_(kk, kk), &A_(kp, kp));
      if (k_step == 2)
        swapd(&A_(k + 1, k), &A_(kp, k));
    }
    if (k_step == 1) { /* :104-121 */
      double d11 = 1.0 / A_(k, k);
      A_(k, k) = d11;
      int m = n - k - 1;
      for (int j = 0; j < m; ++j) {
        double d11xj = A_(k + 1 + j, k) * d11;
        for (int i = j; i < m; ++i)
          A_(k + 1 + i, k + 1 + j) -= d11xj * A_(k + 1 + i, k);
      }
      for (int i = 0; i < m; ++i)
        A_(k + 1 + i, k) *= d11;
    } else { /* 2x2 pivot, :122-149 */
      double d21_abs = fabs(A_(k + 1, k));
      double d21_inv = 1.0 / d21_abs;
      double d11 = d21_inv * A_(k + 1, k + 1);
      double d22 = d21_inv * A_(k, k);
      double t = 1.0 / ((d11 * d22) - 1.0);
      double d = t * d21_inv;
      double d21 = A_(k + 1, k) * d21_inv;
      A_(k, k) = d11 * d;
      A_(k + 1, k) = -d21 * d;
      A_(k + 1, k + 1) = d22 * d;
      for (int j = k + 2; j < n; ++j) {
        double wk = ((A_(j, k) * d11) - (A_(j, k + 1) * d21)) * d;
        double wkp1 = ((A_(j, k + 1) * d22) - (A_(j, k) * d21)) * d;
        for (int i = j; i < n; ++i)
          A_(i, j) -= A_(i, k) * wk + A_(i, k + 1) * wkp1;
        A_(j, k) = wk;
        A_(j, k + 1) = wkp1;
      }
    }
    if (k_step == 1) { /* :156-161 */
      piv[k] = kp;
    } else {
      piv[k] = -1 - kp;
      piv[k + 1] = -1 - kp;
    }
    k += k_step;
  }
  return BK_OK;
#undef A_
}

/* bunch_kaufman_in_place_one_block (bunchkaufman.hpp:172-344). */
static int bk_one_block(double *a, int lda, int n, double *w, int ldw, int nb,
                        int *piv, int *pivot_count, int *processed_cols) {
#define A_(i, j) CM(a, lda, i, j)
#define W_(i, j) CM(w, ldw, i, j)
  const double alpha = bk_alpha();
  *pivot_count = 0;
  *processed_cols = 0;
  if (n == 0)
    return BK_OK;
  int k = 0;
  while (k < n && k + 1 < nb) {
    /* w(:,k) = a(:,k) - a(k:, 0:k) w(k, 0:k)^T   (:190-196) */
    for (int i = k; i < n; ++i)
      W_(i, k) = A_(i, k);
    for (int c = 0; c < k; ++c) {
      double wkc = W_(k, c);
      for (int i = k; i < n; ++i)
        W_(i, k) -= A_(i, c) * wkc;
    }
    int k_step = 1;
    double abs_akk = fabs(W_(k, k));
    int imax = 0;
    double colmax = 0.0;
    if (k + 1 < n) {
      colmax = fabs(W_(k + 1, k));
      for (int i = k + 2; i < n; ++i)
        if (fabs(W_(i, k)) > colmax) {
          colmax = fabs(W_(i, k));
          imax = i - (k + 1);
        }
    }
    imax += k + 1;
    int kp;
    if (fmax(abs_akk, colmax) == 0.0)
      return BK_ISSUE;
    if (abs_akk >= colmax * alpha) {
      kp = k;
    } else { /* :215-250 */
      for (int j = k; j < imax; ++j)
        W_(j, k + 1) = A_(imax, j);
      for (int i = imax; i < n; ++i)
        W_(i, k + 1) = A_(i, imax);
      for (int c = 0; c < k; ++c) {
        double wic = W_(imax, c);
        for (int i = k; i < n; ++i)
          W_(i, k + 1) -= A_(i, c) * wic;
      }
      double rowmax = 0.0;
      for (int i = k; i < imax; ++i)
        rowmax = fmax(rowmax, fabs(W_(i, k + 1)));
      for (int i = imax + 1; i < n; ++i)
        rowmax = fmax(rowmax, fabs(W_(i, k + 1)));
      if (abs_akk >= (alpha * colmax) * (colmax / rowmax)) {
        kp = k;
      } else if (fabs(W_(imax, k + 1)) >= alpha * rowmax) {
        kp = imax;
        for (int i = k; i < n; ++i)
          W_(i, k) = W_(i, k + 1);
      } else {
        kp = imax;
        k_step = 2;
      }
    }
    int kk = k + k_step - 1;
    if (kp != kk) { /* :253-265 */
      *pivot_count += 1;
      A_(kp, kp) = A_(kk, kk);
      for (int j = kk + 1; j < kp; ++j)
        A_(kp, j) = A_(j, kk);
      for (int i = kp + 1; i < n; ++i)
        A_(i, kp) = A_(i, kk);
      for (int c = 0; c < k; ++c)
        swapd(&A_(kk, c), &A_(kp, c));
      for (int c = 0; c <= kk; ++c)
        swapd(&W_(kk, c), &W_(kp, c));
    }
    if (k_step == 1) { /* :267-276 */
      for (int i = k; i < n; ++i)
        A_(i, k) = W_(i, k);
      double d11 = 1.0 / W_(k, k);
      A_(k, k) = d11;
      for (int i = k + 1; i < n; ++i)
        A_(i, k) *= d11;
    } else { /* :277-303 */
      double d21_abs = fabs(W_(k + 1, k));
      double d21_inv = 1.0 / d21_abs;
      double d11 = d21_inv * W_(k + 1, k + 1);
      double d22 = d21_inv * W_(k, k);
      double t = 1.0 / ((d11 * d22) - 1.0);
      double d21 = W_(k + 1, k) * d21_inv;
      double d = t * d21_inv;
      A_(k, k) = d11 * d;
      A_(k + 1, k) = -d21 * d;
      A_(k + 1, k + 1) = d22 * d;
      for (int j = k + 2; j < n; ++j) {
        double wk = ((W_(j, k) * d11) - (W_(j, k + 1) * d21)) * d;
        double wkp1 = ((W_(j, k + 1) * d22) - (W_(j, k) * d21)) * d;
        A_(j, k) = wk;
        A_(j, k + 1) = wkp1;
      }
    }
    if (k_step == 1) {
      piv[k] = kp;
    } else {
      piv[k] = -1 - kp;
      piv[k + 1] = -1 - kp;
    }
    k += k_step;
  }
  /* trailing update, lower triangle only (:320-324):
   * a_right.tril -= a_left * w(k:, 0:k)^T */
  for (int j = k; j < n; ++j)
    for (int c = 0; c < k; ++c) {
      double wjc = W_(j, c);
      for (int i = j; i < n; ++i)
        A_(i, j) -= A_(i, c) * wjc;
    }
  int j = k - 1;
  *processed_cols = k;
  for (;;) { /* :328-343 */
    int jj = j;
    int jp = piv[j];
    if (jp < 0) {
      jp = -1 - jp;
      j -= 1;
    }
    if (j == 0)
      return BK_OK;
    j -= 1;
    if (jp != jj)
      for (int c = 0; c <= j; ++c)
        swapd(&A_(jp, c), &A_(jj, c));
    if (j == 0)
      return BK_OK;
  }
#undef A_
#undef W_
}

/* bunch_kaufman_in_place (bunchkaufman.hpp:348-420). */
static int bk_in_place(ora_bk *bk) {
  const int n = bk->n;
  double *a = bk->L;
  const int blocksize = bk->blocksize;
  int k = 0;
  bk->pivot_count = 0;
  while (k < n) {
    int kb = 0, kpc = 0, info;
    double *ablk = a + (long)k * n + k;
    if (blocksize != 0 && blocksize < n - k) {
      info = bk_one_block(ablk, n, n - k, bk->W, n, blocksize, bk->piv + k, &kpc, &kb);
    } else {
      info = bk_unblocked(ablk, n, n - k, bk->piv + k, &kpc);
      kb = n - k;
    }
    if (info != BK_OK)
      return info;
    for (int j = k; j < k + kb; ++j) { /* :375-386 */
      if (bk->piv[j] >= 0)
        bk->piv[j] += k;
      else
        bk->piv[j] -= k;
    }
    bk->pivot_count += kpc;
    k += kb;
  }
  k = 0; /* :393-404 */
  while (k < n) {
    if (bk->piv[k] < 0) {
      bk->subdiag[k] = CM(a, n, k + 1, k);
      bk->subdiag[k + 1] = 0.0;
      CM(a, n, k + 1, k) = 0.0;
      k += 2;
    } else {
      bk->subdiag[k] = 0.0;
      k += 1;
    }
  }
  k = 0; /* :406-417 */
  while (k < n) {
    int p = bk->piv[k];
    if (p < 0) {
      p = -1 - p;
      for (int c = 0; c < k; ++c)
        swapd(&CM(a, n, k + 1, c), &CM(a, n, p, c));
      k += 2;
    } else {
      for (int c = 0; c < k; ++c)
        swapd(&CM(a, n, k, c), &CM(a, n, p, c));
      k += 1;
    }
  }
  return BK_OK;
}

/* BunchKaufman::compute (bunchkaufman.hpp:653-676). */
int ora_bk_compute(ora_bk *bk, const double *src, int lda) {
  const int n = bk->n;
  memset(bk->L, 0, sizeof(double) * (size_t)(n > 0 ? (long)n * n : 1));
  memset(bk->subdiag, 0, sizeof(double) * (size_t)(n > 0 ? n : 1));
  memset(bk->piv, 0, sizeof(int) * (size_t)(n > 0 ? n : 1));
  bk->blocksize = n <= BK_BLOCKSIZE ? 0 : BK_BLOCKSIZE;
  memset(bk->W, 0, sizeof(double) * (size_t)(n > 0 ? (long)n * BK_BLOCKSIZE : 1));
  for (int j = 0; j < n; ++j) /* lower triangle only (:670-671) */
    for (int i = j; i < n; ++i)
      CM(bk->L, n, i, j) = CM(src, lda, i, j);
  bk->info = bk_in_place(bk);
  return bk->info;
}

static void swap_rows(double *x, int rs, int cs, int ncols, int r1, int r2) {
  if (r1 == r2)
    return;
  for (int c = 0; c < ncols; ++c)
    swapd(&x[(long)r1 * rs + (long)c * cs], &x[(long)r2 * rs + (long)c * cs]);
}

/* bunch_kaufman_solve_in_place (bunchkaufman.hpp:451-518). */
void ora_bk_solve_in_place(const ora_bk *bk, double *x, int rs, int cs, int ncols) {
  const int n = bk->n;
  const double *L = bk->L;
#define X_(i, c) x[(long)(i) * rs + (long)(c) * cs]
  int k = 0;
  while (k < n) { /* :458-468 */
    int p = bk->piv[k];
    if (p < 0) {
      p = -1 - p;
      swap_rows(x, rs, cs, ncols, k + 1, p);
      k += 2;
    } else {
      swap_rows(x, rs, cs, ncols, k, p);
      k += 1;
    }
  }
  /* unit-lower solve (:472) */
  for (int c = 0; c < ncols; ++c)
    for (int j = 0; j < n; ++j) {
      double xj = X_(j, c);
      if (xj != 0.0)
        for (int i = j + 1; i < n; ++i)
          X_(i, c) -= CM(L, n, i, j) * xj;
    }
  k = 0; /* inverse-D multiply (:474-502) */
  while (k < n) {
    int p = bk->piv[k];
    if (p < 0) {
      double akp1k = bk->subdiag[k];
      double ak = CM(L, n, k, k);
      double akp1 = CM(L, n, k + 1, k + 1);
      for (int c = 0; c < ncols; ++c) {
        double xk = X_(k, c), xkp1 = X_(k + 1, c);
        X_(k, c) = xk * ak + xkp1 * akp1k;
        X_(k + 1, c) = xkp1 * akp1 + xk * akp1k;
      }
      k += 2;
    } else {
      double dk = CM(L, n, k, k);
      for (int c = 0; c < ncols; ++c)
        X_(k, c) *= dk;
      k += 1;
    }
  }
  /* unit-upper (L^T) solve (:504) */
  for (int c = 0; c < ncols; ++c)
    for (int j = n - 1; j >= 0; --j) {
      double s = X_(j, c);
      for (int i = j + 1; i < n; ++i)
        s -= CM(L, n, i, j) * X_(i, c);
      X_(j, c) = s;
    }
  k = n; /* reverse interchanges (:506-517) */
  while (k > 0) {
    k -= 1;
    int p = bk->piv[k];
    if (p < 0) {
      p = -1 - p;
      swap_rows(x, rs, cs, ncols, k, p);
      k -= 1;
    } else {
      swap_rows(x, rs, cs, ncols, k, p);
    }
  }
#undef X_
}

/* ------------------------------------------------------------------------ */
/* StageFactor (riccati-kernel.hxx:11-50)                                    */
/* ------------------------------------------------------------------------ */
static void value_alloc(ora_value *v, int nx, int nth) {
  v->Vxx = dalloc((long)nx * nx);
  v->vx = dalloc(nx);
  v->Vxt = dalloc((long)nx * nth);
  v->Vtt = dalloc((long)nth * nth);
  v->vt = dalloc(nth);
}
static void value_free(ora_value *v) {
  free(v->Vxx); free(v->vx); free(v->Vxt); free(v->Vtt); free(v->vt);
}
static void factor_alloc(ora_stage_factor *d, int nx, int nu, int nc, int nx2, int nth) {
  d->nx = nx; d->nu = nu; d->nc = nc; d->nx2 = nx2; d->nth = nth;
  d->Qhat = dalloc((long)nx * nx);
  d->Rhat = dalloc((long)nu * nu);
  d->Shat = dalloc((long)nx * nu);
  d->qhat = dalloc(nx);
  d->rhat = dalloc(nu);
  d->AtV = dalloc((long)nx * nx2);
  d->BtV = dalloc((long)nu * nx2);
  d->Gxhat = dalloc((long)nx * nth);
  d->Guhat = dalloc((long)nu * nth);
  int nr = nu + nc + nx2;
  d->ff = dalloc(nr);
  d->fb = dalloc((long)nr * nx);
  d->fth = dalloc((long)nr * nth);
  d->kktMat = dalloc((long)(nu + nc) * (nu + nc));
  d->kktChol = ora_bk_new(nu + nc);
  value_alloc(&d->vm, nx, nth);
}
static void factor_free(ora_stage_factor *d) {
  free(d->Qhat); free(d->Rhat); free(d->Shat); free(d->qhat); free(d->rhat);
  free(d->AtV); free(d->BtV); free(d->Gxhat); free(d->Guhat);
  free(d->ff); free(d->fb); free(d->fth); free(d->kktMat);
  ora_bk_free(d->kktChol);
  value_free(&d->vm);
}

/* ------------------------------------------------------------------------ */
/* ProximalRiccatiKernel::terminalSolve (riccati-kernel.hxx:130-193)         */
/* ------------------------------------------------------------------------ */
void ora_terminal_solve(const ora_knot *m, double mueq, ora_stage_factor *d) {
  const int nx = m->nx, nu = m->nu, nc = m->nc, nth = m->nth;
  const int n = nu + nc;
  double *kff = d->ff, *zff = d->ff + nu;
  double *K = d->fb, *Z = d->fb + (long)nu * nx;     /* row-major, ld nx  */
  double *Kth = d->fth, *Zth = d->fth + (long)nu * nth; /* row-major, ld nth */
  ora_value *vc = &d->vm;

  if (nu == 0) { /* :146-149 */
    for (int i = 0; i < nc; ++i) {
      for (int j = 0; j < nx; ++j)
        RM(Z, nx, i, j) = CM(m->C, nc, i, j) / mueq;
      zff[i] = m->d[i] / mueq;
      for (int j = 0; j < nth; ++j)
        RM(Zth, nth, i, j) = 0.0;
    }
  } else { /* :150-172 */
    double *M = d->kktMat;
    for (int j = 0; j < nu; ++j)
      for (int i = 0; i < nu; ++i)
        CM(M, n, i, j) = CM(m->R, nu, i, j);
    for (int i = 0; i < nc; ++i)
      for (int j = 0; j < nu; ++j) {
        CM(M, n, j, nu + i) = CM(m->D, nc, i, j);
        CM(M, n, nu + i, j) = CM(m->D, nc, i, j);
      }
    for (int i = 0; i < nc; ++i)
      CM(M, n, nu + i, nu + i) = -mueq; /* off-diagonals stay 0 from the ctor */
    ora_bk_compute(d->kktChol, M, n);
    for (int i = 0; i < nu; ++i)
      kff[i] = -m->r[i];
    for (int i = 0; i < nc; ++i)
      zff[i] = -m->d[i];
    for (int i = 0; i < nu; ++i)
      for (int j = 0; j < nx; ++j)
        RM(K, nx, i, j) = -CM(m->S, nx, j, i);
    for (int i = 0; i < nc; ++i)
      for (int j = 0; j < nx; ++j)
        RM(Z, nx, i, j) = -CM(m->C, nc, i, j);
    ora_bk_solve_in_place(d->kktChol, d->ff, 1, 1, 1);
    ora_bk_solve_in_place(d->kktChol, d->fb, nx, 1, nx);
    if (nth > 0) { /* :166-171 */
      for (int i = 0; i < nu; ++i)
        for (int j = 0; j < nth; ++j)
          RM(Kth, nth, i, j) = -CM(m->Gu, nu, i, j);
      for (int i = 0; i < nc; ++i)
        for (int j = 0; j < nth; ++j)
          RM(Zth, nth, i, j) = 0.0;
      ora_bk_solve_in_place(d->kktChol, d->fth, nth, 1, nth);
    }
  }
  /* :175-178  Vxx = Q + C^T Z ; vx = q + C^T zff */
  dcopy((long)nx * nx, m->Q, vc->Vxx);
  mm_acc(nx, nx, nc, 1.0, m->C, nc, 1, Z, nx, 1, vc->Vxx, 1, nx);
  dcopy(nx, m->q, vc->vx);
  mm_acc(nx, 1, nc, 1.0, m->C, nc, 1, zff, 1, 1, vc->vx, 1, nx);
  if (nu > 0) { /* :180-183 */
    mm_acc(nx, nx, nu, 1.0, m->S, 1, nx, K, nx, 1, vc->Vxx, 1, nx);
    mm_acc(nx, 1, nu, 1.0, m->S, 1, nx, kff, 1, 1, vc->vx, 1, nx);
  }
  if (nth > 0) { /* :185-192 */
    dcopy((long)nx * nth, m->Gx, vc->Vxt);
    mm_acc(nx, nth, nu, 1.0, K, 1, nx, m->Gu, 1, nu, vc->Vxt, 1, nx);
    dcopy((long)nth * nth, m->Gth, vc->Vtt);
    mm_acc(nth, nth, nu, 1.0, m->Gu, nu, 1, Kth, nth, 1, vc->Vtt, 1, nth);
    dcopy(nth, m->gamma, vc->vt);
    mm_acc(nth, 1, nu, 1.0, m->Gu, nu, 1, kff, 1, 1, vc->vt, 1, nth);
  }
}

/* ------------------------------------------------------------------------ */
/* ProximalRiccatiKernel::stageKernelSolve (riccati-kernel.hxx:209-312)      */
/* ------------------------------------------------------------------------ */
int ora_stage_kernel_solve(const ora_knot *m, ora_stage_factor *d, ora_value *vn,
                           double mueq) {
  const int nx = m->nx, nu = m->nu, nc = m->nc, nx2 = m->nx2, nth = m->nth;
  const int n = nu + nc;
  /* :216  vn.Vxx <- selfadjointView<Lower>, IN PLACE on stage t+1's storage */
  for (int j = 0; j < nx2; ++j)
    for (int i = j + 1; i < nx2; ++i)
      CM(vn->Vxx, nx2, j, i) = CM(vn->Vxx, nx2, i, j);
  /* :217-218  vplus = vx' + Vxx' f */
  double vplus[nx2 > 0 ? nx2 : 1]; /* (the reference allocates this temporary from its pmr arena, :215) */
  dcopy(nx2, vn->vx, vplus);
  mm_acc(nx2, 1, nx2, 1.0, vn->Vxx, 1, nx2, m->f, 1, 1, vplus, 1, nx2);
  /* :220-221  AtV = A^T Vxx' (row-major nx x nx2), BtV = B^T Vxx' */
  memset(d->AtV, 0, sizeof(double) * (size_t)((long)nx * nx2 > 0 ? (long)nx * nx2 : 1));
  memset(d->BtV, 0, sizeof(double) * (size_t)((long)nu * nx2 > 0 ? (long)nu * nx2 : 1));
  mm_acc(nx, nx2, nx2, 1.0, m->A, nx2, 1, vn->Vxx, 1, nx2, d->AtV, nx2, 1);
  mm_acc(nu, nx2, nx2, 1.0, m->B, nx2, 1, vn->Vxx, 1, nx2, d->BtV, nx2, 1);
  /* :224-228 */
  dcopy((long)nx * nx, m->Q, d->Qhat);
  mm_acc(nx, nx, nx2, 1.0, d->AtV, nx2, 1, m->A, 1, nx2, d->Qhat, 1, nx);
  dcopy((long)nu * nu, m->R, d->Rhat);
  mm_acc(nu, nu, nx2, 1.0, d->BtV, nx2, 1, m->B, 1, nx2, d->Rhat, 1, nu);
  dcopy((long)nx * nu, m->S, d->Shat);
  mm_acc(nx, nu, nx2, 1.0, d->AtV, nx2, 1, m->B, 1, nx2, d->Shat, 1, nx);
  dcopy(nx, m->q, d->qhat);
  mm_acc(nx, 1, nx2, 1.0, m->A, nx2, 1, vplus, 1, 1, d->qhat, 1, nx);
  dcopy(nu, m->r, d->rhat);
  mm_acc(nu, 1, nx2, 1.0, m->B, nx2, 1, vplus, 1, 1, d->rhat, 1, nu);

  /* :232-241  kktMat = sym_from_lower([Rhat D^T; D -mu I]); factorise */
  double *M = d->kktMat;
  for (int j = 0; j < nu; ++j)
    for (int i = 0; i < nu; ++i)
      CM(M, n, i, j) = CM(d->Rhat, nu, i, j);
  for (int i = 0; i < nc; ++i)
    for (int j = 0; j < nu; ++j) {
      CM(M, n, j, nu + i) = CM(m->D, nc, i, j);
      CM(M, n, nu + i, j) = CM(m->D, nc, i, j);
    }
  for (int i = 0; i < nc; ++i)
    CM(M, n, nu + i, nu + i) = -mueq;
  for (int j = 0; j < n; ++j)
    for (int i = j + 1; i < n; ++i)
      CM(M, n, j, i) = CM(M, n, i, j);
  if (ora_bk_compute(d->kktChol, M, n) != BK_OK)
    return 0; /* reference throws "Failed stage LDL factorization" */

  double *kff = d->ff, *zff = d->ff + nu, *yff = d->ff + nu + nc;
  double *K = d->fb, *Z = d->fb + (long)nu * nx, *Aff = d->fb + (long)(nu + nc) * nx;
  /* :248-256 */
  for (int i = 0; i < nu; ++i)
    kff[i] = -d->rhat[i];
  for (int i = 0; i < nc; ++i)
    zff[i] = -m->d[i];
  for (int i = 0; i < nu; ++i)
    for (int j = 0; j < nx; ++j)
      RM(K, nx, i, j) = -CM(d->Shat, nx, j, i);
  for (int i = 0; i < nc; ++i)
    for (int j = 0; j < nx; ++j)
      RM(Z, nx, i, j) = -CM(m->C, nc, i, j);
  /* :261-262 */
  ora_bk_solve_in_place(d->kktChol, d->ff, 1, 1, 1);
  ora_bk_solve_in_place(d->kktChol, d->fb, nx, 1, nx);
  /* :266-267  yff = f + B kff ; Aff = A + B K */
  dcopy(nx2, m->f, yff);
  mm_acc(nx2, 1, nu, 1.0, m->B, 1, nx2, kff, 1, 1, yff, 1, nx2);
  for (int i = 0; i < nx2; ++i)
    for (int j = 0; j < nx; ++j)
      RM(Aff, nx, i, j) = CM(m->A, nx2, i, j);
  mm_acc(nx2, nx, nu, 1.0, m->B, 1, nx2, K, nx, 1, Aff, nx, 1);
  /* :272-277 */
  ora_value *vc = &d->vm;
  dcopy((long)nx * nx, d->Qhat, vc->Vxx);
  mm_acc(nx, nx, nu, 1.0, d->Shat, 1, nx, K, nx, 1, vc->Vxx, 1, nx);
  mm_acc(nx, nx, nc, 1.0, m->C, nc, 1, Z, nx, 1, vc->Vxx, 1, nx);
  dcopy(nx, d->qhat, vc->vx);
  mm_acc(nx, 1, nu, 1.0, d->Shat, 1, nx, kff, 1, 1, vc->vx, 1, nx);
  mm_acc(nx, 1, nc, 1.0, m->C, nc, 1, zff, 1, 1, vc->vx, 1, nx);

  if (nth > 0) { /* :278-311 */
    double *Kth = d->fth, *Zth = d->fth + (long)nu * nth;
    double *Yth = d->fth + (long)(nu + nc) * nth;
    dcopy((long)nx * nth, m->Gx, d->Gxhat);
    mm_acc(nx, nth, nx2, 1.0, m->A, nx2, 1, vn->Vxt, 1, nx2, d->Gxhat, 1, nx);
    dcopy((long)nu * nth, m->Gu, d->Guhat);
    mm_acc(nu, nth, nx2, 1.0, m->B, nx2, 1, vn->Vxt, 1, nx2, d->Guhat, 1, nu);
    for (int i = 0; i < nu; ++i)
      for (int j = 0; j < nth; ++j)
        RM(Kth, nth, i, j) = -CM(d->Guhat, nu, i, j);
    for (int i = 0; i < nc; ++i)
      for (int j = 0; j < nth; ++j)
        RM(Zth, nth, i, j) = -CM(m->Gv, nc, i, j);
    ora_bk_solve_in_place(d->kktChol, d->fth, nth, 1, nth);
    /* Yth = B Kth (:295) */
    memset(Yth, 0, sizeof(double) * (size_t)((long)nx2 * nth));
    mm_acc(nx2, nth, nu, 1.0, m->B, 1, nx2, Kth, nth, 1, Yth, nth, 1);
    /* vt = gamma + vt' + Gu^T kff + Vxt'^T yff  (:298-301) */
    for (int i = 0; i < nth; ++i)
      vc->vt[i] = m->gamma[i] + vn->vt[i];
    mm_acc(nth, 1, nu, 1.0, m->Gu, nu, 1, kff, 1, 1, vc->vt, 1, nth);
    mm_acc(nth, 1, nx2, 1.0, vn->Vxt, nx2, 1, yff, 1, 1, vc->vt, 1, nth);
    /* Vxt = Gx + K^T Gu + Aff^T Vxt'  (:304-306) */
    dcopy((long)nx * nth, m->Gx, vc->Vxt);
    mm_acc(nx, nth, nu, 1.0, K, 1, nx, m->Gu, 1, nu, vc->Vxt, 1, nx);
    mm_acc(nx, nth, nx2, 1.0, Aff, 1, nx, vn->Vxt, 1, nx2, vc->Vxt, 1, nx);
    /* Vtt = Gth + Vtt' + Gu^T Kth + Vxt'^T Yth  (:308-310) */
    for (long i = 0; i < (long)nth * nth; ++i)
      vc->Vtt[i] = m->Gth[i] + vn->Vtt[i];
    mm_acc(nth, nth, nu, 1.0, m->Gu, nu, 1, Kth, nth, 1, vc->Vtt, 1, nth);
    mm_acc(nth, nth, nx2, 1.0, vn->Vxt, nx2, 1, Yth, nth, 1, vc->Vtt, 1, nth);
  }
  return 1;
}

/* ProximalRiccatiKernel::backwardImpl (riccati-kernel.hxx:104-129) */
int ora_backward_impl(const ora_knot *stages, int nstages, double mueq,
                      ora_stage_factor *datas) {
  if (nstages == 0)
    return 1;
  int N = nstages - 1;
  ora_terminal_solve(&stages[N], mueq, &datas[N]);
  if (N == 0)
    return 1;
  int ok = 1;
  for (int t = N - 1; t >= 0; --t)
    ok &= ora_stage_kernel_solve(&stages[t], &datas[t], &datas[t + 1].vm, mueq);
  return ok;
}

/* ProximalRiccatiKernel::forwardImpl (riccati-kernel.hxx:314-377) */
int ora_forward_impl(const ora_knot *stages, const ora_stage_factor *datas,
                     int nstages, double **xs, double **us, double **vs,
                     double **lbdas, const double *theta) {
  int N = nstages - 1;
  for (int t = 0; t <= N; ++t) {
    const ora_stage_factor *d = &datas[t];
    const ora_knot *m = &stages[t];
    const int nx = m->nx, nu = m->nu, nc = m->nc, nx2 = m->nx2, nth = m->nth;
    const double *K = d->fb, *Z = d->fb + (long)nu * nx;
    const double *kff = d->ff, *zff = d->ff + nu;
    if (nu > 0) { /* :332-336 */
      dcopy(nu, kff, us[t]);
      mm_acc(nu, 1, nx, 1.0, K, nx, 1, xs[t], 1, 1, us[t], 1, nu);
    }
    dcopy(nc, zff, vs[t]); /* :340-341 */
    mm_acc(nc, 1, nx, 1.0, Z, nx, 1, xs[t], 1, 1, vs[t], 1, nc);
    if (nth > 0 && theta) { /* :343-351 */
      const double *Kth = d->fth, *Zth = d->fth + (long)nu * nth;
      if (nu > 0)
        mm_acc(nu, 1, nth, 1.0, Kth, nth, 1, theta, 1, 1, us[t], 1, nu);
      mm_acc(nc, 1, nth, 1.0, Zth, nth, 1, theta, 1, 1, vs[t], 1, nc);
    }
    if (t == N)
      break;
    const double *Aff = d->fb + (long)(nu + nc) * nx;
    const double *yff = d->ff + nu + nc;
    dcopy(nx2, yff, xs[t + 1]); /* :360-361 */
    mm_acc(nx2, 1, nx, 1.0, Aff, nx, 1, xs[t], 1, 1, xs[t + 1], 1, nx2);
    if (nth > 0 && theta) { /* :363-367 */
      const double *Yth = d->fth + (long)(nu + nc) * nth;
      mm_acc(nx2, 1, nth, 1.0, Yth, nth, 1, theta, 1, 1, xs[t + 1], 1, nx2);
    }
    const ora_value *vn = &datas[t + 1].vm; /* :369-374 */
    dcopy(nx2, vn->vx, lbdas[t + 1]);
    mm_acc(nx2, 1, nx2, 1.0, vn->Vxx, 1, nx2, xs[t + 1], 1, 1, lbdas[t + 1], 1, nx2);
    if (nth > 0 && theta)
      mm_acc(nx2, 1, nth, 1.0, vn->Vxt, 1, nx2, theta, 1, 1, lbdas[t + 1], 1, nx2);
  }
  return 1;
}

/* ------------------------------------------------------------------------ */
/* ProximalRiccatiSolver (proximal-riccati.hxx)                              */
/* ------------------------------------------------------------------------ */
ora_prox_solver *ora_prox_new(const ora_problem *p) { /* :13-31 */
  ora_prox_solver *s = (ora_prox_solver *)calloc(1, sizeof(ora_prox_solver));
  s->problem = p;
  s->N = p->N;
  s->datas = (ora_stage_factor *)calloc((size_t)(p->N + 1), sizeof(ora_stage_factor));
  for (int t = 0; t <= p->N; ++t) {
    const ora_knot *k = &p->stages[t];
    factor_alloc(&s->datas[t], k->nx, k->nu, k->nc, k->nx2, k->nth);
  }
  int nx0 = p->stages[0].nx, nth = p->stages[0].nth;
  s->n0 = nx0 + p->nc0;
  s->kkt0_mat = dalloc((long)s->n0 * s->n0);
  s->kkt0_ff = dalloc(s->n0);
  s->kkt0_fth = dalloc((long)s->n0 * nth);
  s->kkt0_chol = ora_bk_new(s->n0);
  s->thGrad = dalloc(nth);
  s->thHess = dalloc((long)nth * nth);
  return s;
}
void ora_prox_free(ora_prox_solver *s) {
  if (!s)
    return;
  for (int t = 0; t <= s->N; ++t)
    factor_free(&s->datas[t]);
  free(s->datas);
  free(s->kkt0_mat); free(s->kkt0_ff); free(s->kkt0_fth);
  ora_bk_free(s->kkt0_chol);
  free(s->thGrad); free(s->thHess);
  free(s);
}

/* the initial-stage block of ProximalRiccatiSolver::backward (:42-60) */
static void prox_initial_stage(ora_prox_solver *s) {
  const ora_problem *p = s->problem;
  ora_stage_factor *d0 = &s->datas[0];
  ora_value *vinit = &d0->vm;
  const int nx = d0->nx, nc0 = p->nc0, nth = d0->nth, n0 = s->n0;
  double *M = s->kkt0_mat;
  for (int j = 0; j < nx; ++j)
    for (int i = 0; i < nx; ++i)
      CM(M, n0, i, j) = CM(vinit->Vxx, nx, i, j);
  for (int i = 0; i < nc0; ++i)
    for (int j = 0; j < nx; ++j) {
      CM(M, n0, nx + i, j) = CM(p->G0, nc0, i, j);
      CM(M, n0, j, nx + i) = CM(p->G0, nc0, i, j);
    }
  for (int j = 0; j < nc0; ++j)
    for (int i = 0; i < nc0; ++i)
      CM(M, n0, nx + i, nx + j) = 0.0;
  ora_bk_compute(s->kkt0_chol, M, n0);
  for (int i = 0; i < nx; ++i)
    s->kkt0_ff[i] = -vinit->vx[i];
  for (int i = 0; i < nc0; ++i)
    s->kkt0_ff[nx + i] = -p->g0[i];
  ora_bk_solve_in_place(s->kkt0_chol, s->kkt0_ff, 1, 1, 1);
  for (int i = 0; i < nx; ++i)
    for (int j = 0; j < nth; ++j)
      RM(s->kkt0_fth, nth, i, j) = -CM(vinit->Vxt, nx, i, j);
  for (int i = 0; i < nc0; ++i)
    for (int j = 0; j < nth; ++j)
      RM(s->kkt0_fth, nth, nx + i, j) = 0.0;
  if (nth > 0)
    ora_bk_solve_in_place(s->kkt0_chol, s->kkt0_fth, nth, 1, nth);
  /* thGrad = vt + Vxt^T x0 ; thHess = Vtt + Vxt^T fth_x  (:56-59) */
  dcopy(nth, vinit->vt, s->thGrad);
  mm_acc(nth, 1, nx, 1.0, vinit->Vxt, nx, 1, s->kkt0_ff, 1, 1, s->thGrad, 1, nth);
  dcopy((long)nth * nth, vinit->Vtt, s->thHess);
  mm_acc(nth, nth, nx, 1.0, vinit->Vxt, nx, 1, s->kkt0_fth, nth, 1, s->thHess, 1, nth);
}

int ora_prox_backward(ora_prox_solver *s, double mueq) { /* :34-62 */
  int ret = ora_backward_impl(s->problem->stages, s->N + 1, mueq, s->datas);
  prox_initial_stage(s);
  return ret;
}

/* computeInitial (riccati-kernel.hxx:195-207) + forwardImpl */
int ora_prox_forward(const ora_prox_solver *s, double **xs, double **us,
                     double **vs, double **lbdas, const double *theta) {
  const int nx = s->datas[0].nx, nc0 = s->problem->nc0, nth = s->datas[0].nth;
  dcopy(nx, s->kkt0_ff, xs[0]);
  dcopy(nc0, s->kkt0_ff + nx, lbdas[0]);
  if (theta && nth > 0) {
    mm_acc(nx, 1, nth, 1.0, s->kkt0_fth, nth, 1, theta, 1, 1, xs[0], 1, nx);
    mm_acc(nc0, 1, nth, 1.0, s->kkt0_fth + (long)nx * nth, nth, 1, theta, 1, 1, lbdas[0], 1, nc0);
  }
  return ora_forward_impl(s->problem->stages, s->datas, s->N + 1, xs, us, vs, lbdas, theta);
}

/* cycleAppend (proximal-riccati.hxx:79-86).  rotate_vec_left(datas, 0, 1)
 * (utils/mpc-util.hpp:16-22) keeps the LAST element (terminal factor) in
 * place and rotates datas[0..N-1] left by one; then datas[N-1] (N-1 =
 * "horizon() - 1", :82-83) is re-created for the new knot's dimensions. */
void ora_prox_cycle_append(ora_prox_solver *s, const ora_knot *knot) {
  int N = s->N;
  if (N >= 1) {
    ora_stage_factor first = s->datas[0];
    for (int t = 0; t + 1 < N; ++t)
      s->datas[t] = s->datas[t + 1];
    s->datas[N - 1] = first;
    factor_free(&s->datas[N - 1]);
    factor_alloc(&s->datas[N - 1], knot->nx, knot->nu, knot->nc, knot->nx2, knot->nth);
  }
  int nth = s->datas[0].nth;
  memset(s->thGrad, 0, sizeof(double) * (size_t)(nth > 0 ? nth : 1));
  memset(s->thHess, 0, sizeof(double) * (size_t)(nth > 0 ? (long)nth * nth : 1));
  memset(s->kkt0_mat, 0, sizeof(double) * (size_t)((long)s->n0 * s->n0));
}

/* ------------------------------------------------------------------------ */
/* block-tridiagonal (gar/block-tridiagonal.hpp)                             */
/* ------------------------------------------------------------------------ */
/* blockTridiagMatMul (:52-75): c <- beta c + A b */
void ora_blocktridiag_matmul(int nblk, const int *dims, double *const *sub,
                             double *const *diag, double *const *super,
                             double *const *b, double **c, double beta) {
  int N = nblk - 1;
  for (int i = 0; i <= N; ++i)
    for (int r = 0; r < dims[i]; ++r)
      c[i][r] *= beta;
  for (int i = 0; i <= N; ++i) {
    if (i > 0)
      mm_acc(dims[i], 1, dims[i - 1], 1.0, sub[i - 1], 1, dims[i], b[i - 1], 1, 1, c[i], 1, dims[i]);
    mm_acc(dims[i], 1, dims[i], 1.0, diag[i], 1, dims[i], b[i], 1, 1, c[i], 1, dims[i]);
    if (i < N)
      mm_acc(dims[i], 1, dims[i + 1], 1.0, super[i], 1, dims[i], b[i + 1], 1, 1, c[i], 1, dims[i]);
  }
}

/* symmetricBlockTridiagSolve (:82-138), up-looking */
int ora_blocktridiag_solve(int nblk, const int *dims, double **sub, double **diag,
                           double *const *super, double **rhs, ora_bk **facs) {
  int N = nblk - 1;
  if (N >= 1) {
    for (int i = N - 1; i >= 0; --i) {
      ora_bk *ldl = facs[i + 1];
      if (ora_bk_compute(ldl, diag[i + 1], dims[i + 1]) != BK_OK)
        return 0;
      ora_bk_solve_in_place(ldl, rhs[i + 1], 1, 1, 1);
      const double *Bip1 = super[i]; /* dims[i] x dims[i+1] */
      double *Cip1 = sub[i];         /* dims[i+1] x dims[i] */
      mm_acc(dims[i], 1, dims[i + 1], -1.0, Bip1, 1, dims[i], rhs[i + 1], 1, 1, rhs[i], 1, dims[i]);
      ora_bk_solve_in_place(ldl, Cip1, 1, dims[i + 1], dims[i]);
      mm_acc(dims[i], dims[i], dims[i + 1], -1.0, Bip1, 1, dims[i], Cip1, 1, dims[i + 1], diag[i], 1, dims[i]);
    }
  }
  {
    ora_bk *ldl = facs[0];
    if (ora_bk_compute(ldl, diag[0], dims[0]) != BK_OK)
      return 0;
    ora_bk_solve_in_place(ldl, rhs[0], 1, 1, 1);
  }
  for (int i = 0; i < N; ++i) /* :131-134 */
    mm_acc(dims[i + 1], 1, dims[i], -1.0, sub[i], 1, dims[i + 1], rhs[i], 1, 1, rhs[i + 1], 1, dims[i + 1]);
  return 1;
}

/* blockTridiagRefinementStep (:147-182) */
int ora_blocktridiag_refine(int nblk, const int *dims, double *const *upfacs,
                            double *const *super, ora_bk *const *facs, double **rhs) {
  int N = nblk - 1;
  for (int i = N - 1; i >= 0; --i) {
    ora_bk_solve_in_place(facs[i + 1], rhs[i + 1], 1, 1, 1);
    mm_acc(dims[i], 1, dims[i + 1], -1.0, super[i], 1, dims[i], rhs[i + 1], 1, 1, rhs[i], 1, dims[i]);
  }
  ora_bk_solve_in_place(facs[0], rhs[0], 1, 1, 1);
  for (int i = 0; i < N; ++i)
    mm_acc(dims[i + 1], 1, dims[i], -1.0, upfacs[i], 1, dims[i + 1], rhs[i], 1, 1, rhs[i + 1], 1, dims[i + 1]);
  return 1;
}

/* symmetricBlockTridiagSolveDownLooking (:189-243) */
int ora_blocktridiag_solve_down(int nblk, const int *dims, double *const *sub,
                                double **diag, double **super, double **rhs,
                                ora_bk **facs) {
  int N = nblk - 1;
  for (int i = 0; i < N; ++i) {
    ora_bk *ldl = facs[i];
    if (ora_bk_compute(ldl, diag[i], dims[i]) != BK_OK)
      return 0;
    ora_bk_solve_in_place(ldl, rhs[i], 1, 1, 1);
    double *Bip1 = super[i];     /* dims[i] x dims[i+1] */
    const double *Cip1 = sub[i]; /* dims[i+1] x dims[i] */
    mm_acc(dims[i + 1], 1, dims[i], -1.0, Cip1, 1, dims[i + 1], rhs[i], 1, 1, rhs[i + 1], 1, dims[i + 1]);
    ora_bk_solve_in_place(ldl, Bip1, 1, dims[i], dims[i + 1]);
    mm_acc(dims[i + 1], dims[i + 1], dims[i], -1.0, Cip1, 1, dims[i + 1], Bip1, 1, dims[i], diag[i + 1], 1, dims[i + 1]);
  }
  {
    ora_bk *ldl = facs[N];
    if (ora_bk_compute(ldl, diag[N], dims[N]) != BK_OK)
      return 0;
    ora_bk_solve_in_place(ldl, rhs[N], 1, 1, 1);
  }
  for (int i = N - 1; i >= 0; --i)
    mm_acc(dims[i], 1, dims[i + 1], -1.0, super[i], 1, dims[i], rhs[i + 1], 1, 1, rhs[i], 1, dims[i]);
  return 1;
}

/* ------------------------------------------------------------------------ */
/* ParallelRiccatiSolver (parallel-solver.hxx)                               */
/* ------------------------------------------------------------------------ */
void ora_get_work(int horz, int tid, int nthreads, int *beg, int *end) { /* :23-28 */
  *beg = (int)((long)tid * (horz + 1) / nthreads);
  *end = (int)((long)(tid + 1) * (horz + 1) / nthreads);
}

ora_par_solver *ora_par_new(ora_problem *p, int num_threads) { /* :32-82, 261-287 */
  if (num_threads < 2)
    return NULL; /* reference throws (:42-46) */
  ora_par_solver *s = (ora_par_solver *)calloc(1, sizeof(ora_par_solver));
  s->problem = p;
  s->N = p->N;
  s->num_threads = num_threads;
  s->condensedThreshold = 1e-10;
  s->maxRefinementSteps = 5;
  const int N = p->N;
  s->datas = (ora_stage_factor *)calloc((size_t)(N + 1), sizeof(ora_stage_factor));
  for (int i = 0; i < num_threads; ++i) { /* allocate_leg (:52-60) */
    int i0, i1;
    ora_get_work(N, i, num_threads, &i0, &i1);
    int last_leg = (i == num_threads - 1);
    int nth = p->stages[i1 - 1].nx2;
    for (int t = i0; t < i1; ++t) {
      ora_knot *k = &p->stages[t];
      if (!last_leg)
        ora_knot_add_parameterization(k, nth);
      factor_alloc(&s->datas[t], k->nx, k->nu, k->nc, k->nx2, k->nth);
    }
  }
  s->nblk = 2 * num_threads; /* rhsDims_ (:68-73) */
  s->dims = (int *)calloc((size_t)s->nblk, sizeof(int));
  s->dims[0] = p->nc0;
  s->dims[1] = p->stages[0].nx;
  for (int i = 0; i < num_threads - 1; ++i) {
    int i0, i1;
    ora_get_work(N, i, num_threads, &i0, &i1);
    s->dims[2 + 2 * i] = p->stages[i0].nx;
    s->dims[3 + 2 * i] = p->stages[i1 - 1].nx;
  }
  long total = 0;
  for (int i = 0; i < s->nblk; ++i)
    total += s->dims[i];
  s->rhs = dalloc(total);
  s->sol = dalloc(total);
  s->err = dalloc(total);
  s->rhs_blk = (double **)calloc((size_t)s->nblk, sizeof(double *));
  s->sol_blk = (double **)calloc((size_t)s->nblk, sizeof(double *));
  s->err_blk = (double **)calloc((size_t)s->nblk, sizeof(double *));
  long off = 0;
  for (int i = 0; i < s->nblk; ++i) {
    s->rhs_blk[i] = s->rhs + off;
    s->sol_blk[i] = s->sol + off;
    s->err_blk[i] = s->err + off;
    off += s->dims[i];
  }
  int nb = s->nblk;
  s->sub = (double **)calloc((size_t)nb, sizeof(double *));
  s->super = (double **)calloc((size_t)nb, sizeof(double *));
  s->diag = (double **)calloc((size_t)nb, sizeof(double *));
  s->diagFacs = (double **)calloc((size_t)nb, sizeof(double *));
  s->upFacs = (double **)calloc((size_t)nb, sizeof(double *));
  s->ldlt = (ora_bk **)calloc((size_t)nb, sizeof(ora_bk *));
  for (int i = 0; i < nb; ++i) { /* initializeTridiagSystem (:261-287) */
    s->diag[i] = dalloc((long)s->dims[i] * s->dims[i]);
    s->diagFacs[i] = dalloc((long)s->dims[i] * s->dims[i]);
    s->ldlt[i] = ora_bk_new(s->dims[i]);
    if (i + 1 < nb) {
      s->super[i] = dalloc((long)s->dims[i] * s->dims[i + 1]);
      s->sub[i] = dalloc((long)s->dims[i + 1] * s->dims[i]);
      s->upFacs[i] = dalloc((long)s->dims[i + 1] * s->dims[i]);
    }
  }
  return s;
}

void ora_par_free(ora_par_solver *s) {
  if (!s)
    return;
  for (int t = 0; t <= s->N; ++t)
    factor_free(&s->datas[t]);
  free(s->datas);
  for (int i = 0; i < s->nblk; ++i) {
    free(s->diag[i]); free(s->diagFacs[i]); ora_bk_free(s->ldlt[i]);
    free(s->super[i]); free(s->sub[i]); free(s->upFacs[i]);
  }
  free(s->diag); free(s->diagFacs); free(s->ldlt); free(s->super); free(s->sub); free(s->upFacs);
  free(s->rhs); free(s->sol); free(s->err);
  free(s->rhs_blk); free(s->sol_blk); free(s->err_blk);
  free(s->dims);
  free(s);
}

/* assembleCondensedSystem (:85-129) */
static void par_assemble(ora_par_solver *s, double mudyn) {
  const ora_problem *p = s->problem;
  const int N = s->N, J1 = s->num_threads;
  const int *dm = s->dims;
  memset(s->diag[0], 0, sizeof(double) * (size_t)((long)dm[0] * dm[0] > 0 ? (long)dm[0] * dm[0] : 1));
  for (int i = 0; i < dm[0]; ++i)
    CM(s->diag[0], dm[0], i, i) = -mudyn;
  dcopy((long)dm[0] * dm[1], p->G0, s->super[0]);
  dcopy((long)dm[1] * dm[1], s->datas[0].vm.Vxx, s->diag[1]);
  dcopy((long)dm[1] * dm[2], s->datas[0].vm.Vxt, s->super[1]);
  for (int i = 0; i < J1 - 1; ++i) {
    int i0, i1;
    ora_get_work(N, i, J1, &i0, &i1);
    int ip1 = i + 1;
    dcopy((long)dm[2 * ip1] * dm[2 * ip1], s->datas[i0].vm.Vtt, s->diag[2 * ip1]);
    dcopy((long)dm[2 * ip1 + 1] * dm[2 * ip1 + 1], s->datas[i1].vm.Vxx, s->diag[2 * ip1 + 1]);
    double *sup = s->super[2 * ip1]; /* -I (:108) */
    int r = dm[2 * ip1], c = dm[2 * ip1 + 1];
    memset(sup, 0, sizeof(double) * (size_t)((long)r * c));
    for (int k = 0; k < (r < c ? r : c); ++k)
      CM(sup, r, k, k) = -1.0;
    if (ip1 + 1 < J1)
      dcopy((long)dm[2 * ip1 + 1] * dm[2 * ip1 + 2], s->datas[i1].vm.Vxt, s->super[2 * ip1 + 1]);
  }
  for (int i = 0; i + 1 < s->nblk; ++i) { /* sub = super^T (:116-118) */
    int r = dm[i], c = dm[i + 1];
    for (int a = 0; a < r; ++a)
      for (int b = 0; b < c; ++b)
        CM(s->sub[i], c, b, a) = CM(s->super[i], r, a, b);
  }
  for (int i = 0; i < dm[0]; ++i)
    s->rhs_blk[0][i] = -p->g0[i];
  for (int i = 0; i < dm[1]; ++i)
    s->rhs_blk[1][i] = -s->datas[0].vm.vx[i];
  for (int i = 0; i < J1 - 1; ++i) {
    int i0, i1;
    ora_get_work(N, i, J1, &i0, &i1);
    int ip1 = i + 1;
    for (int k = 0; k < dm[2 * ip1]; ++k)
      s->rhs_blk[2 * ip1][k] = -s->datas[i0].vm.vt[k];
    for (int k = 0; k < dm[2 * ip1 + 1]; ++k)
      s->rhs_blk[2 * ip1 + 1][k] = -s->datas[i1].vm.vx[k];
  }
}

static void swap_ptr_arrays(double **a, double **b, int n) {
  for (int i = 0; i < n; ++i) {
    double *t = a[i];
    a[i] = b[i];
    b[i] = t;
  }
}

int ora_par_backward(ora_par_solver *s, double mueq) { /* :132-206 */
  ora_problem *p = s->problem;
  const int N = s->N, J1 = s->num_threads;
  for (int i = 0; i < J1 - 1; ++i) { /* configure_knot (:136-147) */
    int i0, i1;
    ora_get_work(N, i, J1, &i0, &i1);
    ora_knot *k = &p->stages[i1 - 1];
    for (int a = 0; a < k->nx; ++a)
      for (int b = 0; b < k->nth; ++b)
        CM(k->Gx, k->nx, a, b) = CM(k->A, k->nx2, b, a);
    for (int a = 0; a < k->nu; ++a)
      for (int b = 0; b < k->nth; ++b)
        CM(k->Gu, k->nu, a, b) = CM(k->B, k->nx2, b, a);
    memset(k->Gth, 0, sizeof(double) * (size_t)((long)k->nth * k->nth));
    dcopy(k->nth, k->f, k->gamma);
  }
  int ok = 1;
#pragma omp parallel for num_threads(J1) schedule(static, 1) reduction(& : ok)
  for (int i = 0; i < J1; ++i) { /* :150-164 */
    int beg, end;
    ora_get_work(N, i, J1, &beg, &end);
    ok &= ora_backward_impl(p->stages + beg, end - beg, mueq, s->datas + beg);
  }
  par_assemble(s, 0.0); /* :169 */
  long total = 0;
  for (int i = 0; i < s->nblk; ++i)
    total += s->dims[i];
  dcopy(total, s->rhs, s->sol);
  for (int i = 0; i < s->nblk; ++i) { /* :171-172 */
    dcopy((long)s->dims[i] * s->dims[i], s->diag[i], s->diagFacs[i]);
    if (i + 1 < s->nblk)
      dcopy((long)s->dims[i + 1] * s->dims[i], s->sub[i], s->upFacs[i]);
  }
  /* return value ignored by the reference (:176) */
  ora_blocktridiag_solve(s->nblk, s->dims, s->sub, s->diag, s->super, s->sol_blk, s->ldlt);
  swap_ptr_arrays(s->diagFacs, s->diag, s->nblk);    /* :180 */
  swap_ptr_arrays(s->upFacs, s->sub, s->nblk - 1);   /* :181 */
  s->last_refinement_steps = 0;
  for (int it = 0; it < s->maxRefinementSteps; ++it) { /* :184-202 */
    ora_blocktridiag_matmul(s->nblk, s->dims, s->sub, s->diag, s->super, s->sol_blk, s->err_blk, -1.0);
    for (long i = 0; i < total; ++i)
      s->err[i] *= -1.0;
    double resdl = inf_norm((int)total, s->err);
    s->last_residual = resdl;
    if (resdl <= s->condensedThreshold)
      return ok;
    ora_blocktridiag_refine(s->nblk, s->dims, s->upFacs, s->super, s->ldlt, s->err_blk);
    for (long i = 0; i < total; ++i)
      s->sol[i] += s->err[i];
    dcopy(total, s->rhs, s->err);
    s->last_refinement_steps = it + 1;
  }
  return ok;
}

int ora_par_forward(const ora_par_solver *s, double **xs, double **us, double **vs,
                    double **lbdas) { /* :209-243 */
  const int N = s->N, J1 = s->num_threads;
  for (int i = 0; i < J1; ++i) {
    int i0, i1;
    ora_get_work(N, i, J1, &i0, &i1);
    dcopy(s->dims[2 * i], s->sol_blk[2 * i], lbdas[i0]);
    dcopy(s->dims[2 * i + 1], s->sol_blk[2 * i + 1], xs[i0]);
  }
  const ora_knot *stages = s->problem->stages;
#pragma omp parallel for num_threads(J1) schedule(static, 1)
  for (int i = 0; i < J1; ++i) {
    int beg, end;
    ora_get_work(N, i, J1, &beg, &end);
    const double *theta = (i < J1 - 1) ? lbdas[end] : NULL;
    ora_forward_impl(stages + beg, s->datas + beg, end - beg, xs + beg, us + beg,
                     vs + beg, lbdas + beg, theta);
  }
  return 1;
}

/* collapseFeedback (parallel-solver.hpp:41-51): K0 -= Kth0 * subdiagonal[1].
 * After backward(), subdiagonal[1] holds the UNFACTORED Vxt(b0)^T because of
 * the swap at parallel-solver.hxx:180-181. */
void ora_par_collapse_feedback(ora_par_solver *s) {
  ora_stage_factor *d = &s->datas[0];
  const int nu = d->nu, nx = d->nx, nth = d->nth;
  /* sub[1] is dims[2] x dims[1] = nth x nx, col-major */
  mm_acc(nu, nx, nth, -1.0, d->fth, nth, 1, s->sub[1], 1, s->dims[2], d->fb, nx, 1);
}

/* ------------------------------------------------------------------------ */
/* lqrComputeKktError (gar/utils.hxx:88-182)                                 */
/* ------------------------------------------------------------------------ */
void ora_lqr_kkt_error(const ora_problem *p, double *const *xs, double *const *us,
                       double *const *vs, double *const *lbdas, double mueq,
                       const double *theta, double *out3) {
  const int N = p->N;
  double dynErr = 0.0, cstErr = 0.0, dualErr = 0.0;
  {
    int nc0 = p->nc0, nx0 = p->stages[0].nx;
    double *dyn = dalloc(nc0);
    dcopy(nc0, p->g0, dyn);
    mm_acc(nc0, 1, nx0, 1.0, p->G0, 1, nc0, xs[0], 1, 1, dyn, 1, nc0);
    dynErr = fmax(dynErr, inf_norm(nc0, dyn));
    free(dyn);
  }
  for (int t = 0; t <= N; ++t) {
    const ora_knot *k = &p->stages[t];
    const int nx = k->nx, nu = k->nu, nc = k->nc, nx2 = k->nx2, nth = k->nth;
    double *gx = dalloc(nx), *gu = dalloc(nu), *cst = dalloc(nc);
    /* :131-133 */
    for (int i = 0; i < nc; ++i)
      cst[i] = k->d[i] - mueq * vs[t][i];
    mm_acc(nc, 1, nx, 1.0, k->C, 1, nc, xs[t], 1, 1, cst, 1, nc);
    dcopy(nx, k->q, gx);
    mm_acc(nx, 1, nx, 1.0, k->Q, 1, nx, xs[t], 1, 1, gx, 1, nx);
    mm_acc(nx, 1, nc, 1.0, k->C, nc, 1, vs[t], 1, 1, gx, 1, nx);
    dcopy(nu, k->r, gu);
    mm_acc(nu, 1, nx, 1.0, k->S, nx, 1, xs[t], 1, 1, gu, 1, nu);
    mm_acc(nu, 1, nc, 1.0, k->D, nc, 1, vs[t], 1, 1, gu, 1, nu);
    if (nu > 0) { /* :135-139 */
      mm_acc(nc, 1, nu, 1.0, k->D, 1, nc, us[t], 1, 1, cst, 1, nc);
      mm_acc(nx, 1, nu, 1.0, k->S, 1, nx, us[t], 1, 1, gx, 1, nx);
      mm_acc(nu, 1, nu, 1.0, k->R, 1, nu, us[t], 1, 1, gu, 1, nu);
    }
    if (t == 0) { /* :141-145 */
      mm_acc(nx, 1, p->nc0, 1.0, p->G0, p->nc0, 1, lbdas[0], 1, 1, gx, 1, nx);
    } else {
      for (int i = 0; i < nx; ++i)
        gx[i] -= lbdas[t][i];
    }
    if (t < N) { /* :147-156 */
      double *dyn = dalloc(nx2);
      for (int i = 0; i < nx2; ++i)
        dyn[i] = k->f[i] - xs[t + 1][i];
      mm_acc(nx2, 1, nx, 1.0, k->A, 1, nx2, xs[t], 1, 1, dyn, 1, nx2);
      mm_acc(nx2, 1, nu, 1.0, k->B, 1, nx2, us[t], 1, 1, dyn, 1, nx2);
      mm_acc(nx, 1, nx2, 1.0, k->A, nx2, 1, lbdas[t + 1], 1, 1, gx, 1, nx);
      mm_acc(nu, 1, nx2, 1.0, k->B, nx2, 1, lbdas[t + 1], 1, 1, gu, 1, nu);
      dynErr = fmax(dynErr, inf_norm(nx2, dyn));
      free(dyn);
    }
    if (theta) { /* :158-167 (the theta-gradient _gt is computed but unused) */
      mm_acc(nx, 1, nth, 1.0, k->Gx, 1, nx, theta, 1, 1, gx, 1, nx);
      mm_acc(nu, 1, nth, 1.0, k->Gu, 1, nu, theta, 1, 1, gu, 1, nu);
    }
    dualErr = fmax(dualErr, fmax(inf_norm(nx, gx), inf_norm(nu, gu)));
    cstErr = fmax(cstErr, inf_norm(nc, cst));
    free(gx); free(gu); free(cst);
  }
  out3[0] = dynErr;
  out3[1] = cstErr;
  out3[2] = dualErr;
}

/* ------------------------------------------------------------------------ */
/* batched CPU baseline                                                      */
/* ------------------------------------------------------------------------ */
int ora_omp_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int ora_batch_sweep(ora_prox_solver **solvers, int nbatch, double mueq, double ***xs,
                    double ***us, double ***vs, double ***lbdas, int nthreads) {
  int fails = 0;
  if (nthreads < 1)
    nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1) reduction(+ : fails)
  for (int b = 0; b < nbatch; ++b) {
    int ok = ora_prox_backward(solvers[b], mueq);
    ora_prox_forward(solvers[b], xs[b], us[b], vs[b], lbdas[b], NULL);
    fails += !ok;
  }
  return fails;
}

/* The CPU baseline of bench.py, measured the way a tuned host code would run it:
 * every OpenMP thread deep-copies the problems it owns (static partition: its memory is first
 * touched, hence placed, on its own NUMA node), builds its solvers and solution vectors there, then
 * all threads sweep `reps` times between two barriers.  Returns the number of failed sweeps;
 * *seconds = wall time of the timed region (what the slowest thread needs). */
int ora_batch_sweep_local(const ora_problem *const *problems, int nbatch, double mueq, int nthreads,
                          int reps, double *seconds) {
  int fails = 0;
  double t_begin = 0.0, t_end = 0.0;
  if (nthreads < 1)
    nthreads = 1;
#pragma omp parallel num_threads(nthreads) reduction(+ : fails)
  {
    const int nt = omp_get_num_threads(), id = omp_get_thread_num();
    const int b0 = (int)((long)nbatch * id / nt), b1 = (int)((long)nbatch * (id + 1) / nt), nb = b1 - b0;
    ora_problem **P = (ora_problem **)calloc((size_t)(nb > 0 ? nb : 1), sizeof(*P));
    ora_prox_solver **S = (ora_prox_solver **)calloc((size_t)(nb > 0 ? nb : 1), sizeof(*S));
    double ***X = (double ***)calloc((size_t)(4 * (nb > 0 ? nb : 1)), sizeof(*X));
    for (int i = 0; i < nb; ++i) {
      P[i] = ora_problem_copy(problems[b0 + i]);
      S[i] = ora_prox_new(P[i]);
      const int N = P[i]->N;
      for (int k = 0; k < 4; ++k) {
        X[4 * i + k] = (double **)calloc((size_t)(N + 2), sizeof(double *));
        for (int t = 0; t <= N; ++t) {
          const ora_knot *kn = &P[i]->stages[t];
          const int n = k == 0 ? kn->nx : k == 1 ? kn->nu : k == 2 ? kn->nc
                        : (t == 0 ? P[i]->nc0 : P[i]->stages[t - 1].nx2);
          X[4 * i + k][t] = dalloc(n > 0 ? n : 1);
        }
      }
    }
#pragma omp barrier
#pragma omp master
    t_begin = omp_get_wtime();
    for (int r = 0; r < reps; ++r)
      for (int i = 0; i < nb; ++i) {
        const int ok = ora_prox_backward(S[i], mueq);
        ora_prox_forward(S[i], X[4 * i], X[4 * i + 1], X[4 * i + 2], X[4 * i + 3], NULL);
        fails += !ok;
      }
#pragma omp barrier
#pragma omp master
    t_end = omp_get_wtime();
    for (int i = 0; i < nb; ++i) {
      for (int k = 0; k < 4; ++k) {
        for (int t = 0; t <= P[i]->N; ++t)
          free(X[4 * i + k][t]);
        free(X[4 * i + k]);
      }
      ora_prox_free(S[i]);
      ora_problem_free(P[i]);
    }
    free(X);
    free(S);
    free(P);
  }
  if (seconds)
    *seconds = t_end - t_begin;
  return fails;
}
