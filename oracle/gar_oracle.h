/*
 * gar_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the reference's `gar` Riccati/LQR algorithm
 * (Simple-Robotics/aligator, include/aligator/gar/ and core/bunchkaufman.hpp).
 * Every function cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this library.  The product path (aligator_amd/, include/) never
 * links, imports or executes anything from oracle/.
 *
 * PARITY STATUS: PINNED against the reference's own code.  Eigen 3.4 is absent
 * from this image, so the reference *binary* (Eigen's GEMM kernels included)
 * cannot be produced; but the reference's OWN gar sources -- core/bunchkaufman.hpp,
 * gar/riccati-kernel.hxx, gar/proximal-riccati.hxx, gar/parallel-solver.hxx,
 * gar/block-tridiagonal.hpp, gar/lqr-problem.hxx, core/arena-matrix.hpp,
 * core/blk-matrix.hpp -- compile UNCHANGED from /root/reference over a minimal
 * Eigen-API stand-in (oracle/ref_shim, eager evaluation; oracle/ref_build.sh ->
 * oracle/_ref/libgar_ref.so) and this oracle is checked against them live
 * (tests/test_ref_pin.py: Bunch-Kaufman pivot sequences identical over 300
 * matrices incl. the blocked n > 32 path; solution, every StageFactor block,
 * kkt0, thGrad/thHess, leg-parallel factors, condensed solution, collapsed K0 to
 * rounding) and against the committed outputs of that build
 * (tests/golden/ref/*.npz, tests/test_golden.py).  What the stand-in cannot
 * reproduce is the association order of sums inside Eigen's product kernels --
 * the difference between any two BLAS.  Further pins: the reference's own test
 * thresholds (tests/test_oracle.py) and an independent LAPACK dense-KKT solve
 * (oracle/dense_kkt.py, tests/golden/*.npz).
 *
 * Data model mirrors the reference: every matrix is its own column-major
 * allocation (LqrKnotTpl, lqr-problem.hpp:34-103); ff/fb/fth, AtV and BtV are
 * row-major (riccati-kernel.hpp:88-98).
 */
#ifndef GAR_ORACLE_H
#define GAR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- LqrKnotTpl (lqr-problem.hpp:34-103) -------------------------------- */
typedef struct ora_knot {
  int nx, nu, nc, nx2, nth;
  double *Q, *S, *R, *q, *r; /* nx*nx, nx*nu, nu*nu, nx, nu        */
  double *A, *B, *f;         /* nx2*nx, nx2*nu, nx2               */
  double *C, *D, *d;         /* nc*nx, nc*nu, nc                  */
  double *Gth, *Gx, *Gu, *Gv, *gamma; /* nth*nth, nx*nth, nu*nth, nc*nth, nth */
} ora_knot;

/* ---- LqrProblemTpl (lqr-problem.hpp:105-195) ----------------------------- */
typedef struct ora_problem {
  int N;   /* horizon(): number of knots - 1                     */
  int nc0; /* rows of G0                                          */
  double *G0, *g0;
  ora_knot *stages; /* N+1 knots                                  */
} ora_problem;

/* dims5 = (N+1) x {nx,nu,nc,nx2,nth}; all blocks zero-initialised
 * (lqr-problem.hxx:28-72). */
ora_problem *ora_problem_new(int N, const int *dims5, int nc0);
ora_problem *ora_problem_copy(const ora_problem *p);
void ora_problem_free(ora_problem *p);
/* LqrKnotTpl::addParameterization (lqr-problem.hxx:232-241): re-sizes and
 * ZEROES Gth,Gx,Gu,Gv,gamma. */
void ora_knot_add_parameterization(ora_knot *k, int nth);
void ora_problem_add_parameterization(ora_problem *p, int nth);
/* raw block access for the Python wrapper: name in
 * {Q,S,R,q,r,A,B,f,C,D,d,Gth,Gx,Gu,Gv,gamma} */
double *ora_knot_block(ora_problem *p, int t, const char *name);
const int *ora_knot_dims(const ora_problem *p, int t); /* 5 ints */

/* ---- BunchKaufman (core/bunchkaufman.hpp) -------------------------------- */
typedef struct ora_bk {
  int n;
  double *L;       /* n*n col-major: unit-lower L, inverse D on the diagonal */
  double *subdiag; /* off-diagonal of inverse 2x2 D blocks                 */
  int *piv;        /* >=0: 1x1 with row index; <0: 2x2, -1-p               */
  double *W;       /* n x 32 workspace (blocked path, n > 32)              */
  int blocksize;
  int info; /* 0 = Success, 1 = NumericalIssue                             */
  int pivot_count;
} ora_bk;

ora_bk *ora_bk_new(int n);
void ora_bk_free(ora_bk *bk);
/* BunchKaufman::compute (bunchkaufman.hpp:653-676): copies the LOWER triangle
 * of a (col-major, leading dimension lda), factorises. Returns info. */
int ora_bk_compute(ora_bk *bk, const double *a, int lda);
/* bunch_kaufman_solve_in_place (bunchkaufman.hpp:451-518) on an n x ncols
 * right-hand side with arbitrary row/column strides. */
void ora_bk_solve_in_place(const ora_bk *bk, double *x, int rs, int cs, int ncols);

/* ---- StageFactor (riccati-kernel.hpp:30-102) ----------------------------- */
typedef struct ora_value {
  double *Vxx, *vx, *Vxt, *Vtt, *vt; /* nx*nx, nx, nx*nth, nth*nth, nth (col-major) */
} ora_value;

typedef struct ora_stage_factor {
  int nx, nu, nc, nx2, nth;
  double *Qhat, *Rhat, *Shat, *qhat, *rhat; /* col-major                 */
  double *AtV, *BtV;                        /* ROW-major nx*nx2, nu*nx2  */
  double *Gxhat, *Guhat;                    /* col-major nx*nth, nu*nth  */
  double *ff;     /* (nu+nc+nx2)            : [kff; zff; yff]            */
  double *fb;     /* (nu+nc+nx2) x nx   ROW-major: [K; Z; Aff]           */
  double *fth;    /* (nu+nc+nx2) x nth  ROW-major: [Kth; Zth; Yth]       */
  double *kktMat; /* (nu+nc)^2 col-major                                  */
  ora_bk *kktChol;
  ora_value vm;
} ora_stage_factor;

/* ---- ProximalRiccatiSolver (proximal-riccati.hpp/.hxx) ------------------- */
typedef struct ora_prox_solver {
  const ora_problem *problem; /* non-owning (proximal-riccati.hpp:46) */
  int N;
  ora_stage_factor *datas; /* N+1 */
  /* kkt0_t (riccati-kernel.hpp:113-123) */
  int n0;           /* nx0 + nc0 */
  double *kkt0_mat; /* n0*n0 col-major */
  double *kkt0_ff;  /* n0: [x0; lbd0] */
  double *kkt0_fth; /* n0 x nth row-major */
  ora_bk *kkt0_chol;
  double *thGrad, *thHess;
} ora_prox_solver;

ora_prox_solver *ora_prox_new(const ora_problem *p);
void ora_prox_free(ora_prox_solver *s);
/* ProximalRiccatiSolver::backward (proximal-riccati.hxx:34-62).
 * returns 1 on success, 0 if a stage LDL failed (the reference throws,
 * riccati-kernel.hxx:239-241). */
int ora_prox_backward(ora_prox_solver *s, double mueq);
/* ProximalRiccatiSolver::forward (proximal-riccati.hxx:65-77).  xs/us/vs/lbdas
 * are arrays of N+1 pointers to pre-sized vectors (us[N] may be NULL when the
 * last knot has nu = 0, utils.hpp:137-139); theta may be NULL. */
int ora_prox_forward(const ora_prox_solver *s, double **xs, double **us,
                     double **vs, double **lbdas, const double *theta);
/* ProximalRiccatiSolver::cycleAppend (proximal-riccati.hxx:79-86). */
void ora_prox_cycle_append(ora_prox_solver *s, const ora_knot *knot);

/* kernel-level entry points (riccati-kernel.hxx) */
void ora_terminal_solve(const ora_knot *model, double mueq, ora_stage_factor *d);
int ora_stage_kernel_solve(const ora_knot *model, ora_stage_factor *d,
                           ora_value *vn, double mueq);
int ora_backward_impl(const ora_knot *stages, int nstages, double mueq,
                      ora_stage_factor *datas);
int ora_forward_impl(const ora_knot *stages, const ora_stage_factor *datas,
                     int nstages, double **xs, double **us, double **vs,
                     double **lbdas, const double *theta);

/* ---- block-tridiagonal (gar/block-tridiagonal.hpp) ----------------------- */
/* blocks are col-major; dims[i] = size of diagonal block i, nblk blocks.
 * sub[i] is dims[i+1] x dims[i], super[i] is dims[i] x dims[i+1]. */
int ora_blocktridiag_solve(int nblk, const int *dims, double **sub,
                           double **diag, double *const *super, double **rhs,
                           ora_bk **facs); /* :82-138 up-looking */
int ora_blocktridiag_solve_down(int nblk, const int *dims, double *const *sub,
                                double **diag, double **super, double **rhs,
                                ora_bk **facs); /* :189-243 */
int ora_blocktridiag_refine(int nblk, const int *dims, double *const *upfacs,
                            double *const *super, ora_bk *const *facs,
                            double **rhs); /* :147-182 */
/* c <- beta c + A b (:52-75) */
void ora_blocktridiag_matmul(int nblk, const int *dims, double *const *sub,
                             double *const *diag, double *const *super,
                             double *const *b, double **c, double beta);

/* ---- ParallelRiccatiSolver (parallel-solver.hpp/.hxx) -------------------- */
typedef struct ora_par_solver {
  ora_problem *problem; /* non-owning, MUTATED (parallel-solver.hxx:52-60,136-147) */
  int N, num_threads;
  ora_stage_factor *datas;
  int nblk;  /* 2*num_threads */
  int *dims; /* rhsDims_ */
  double **sub, **diag, **super, **diagFacs, **upFacs;
  ora_bk **ldlt;
  double *rhs, *sol, *err; /* condensedKktRhs / Solution / Err (stacked) */
  double **rhs_blk, **sol_blk, **err_blk;
  double condensedThreshold; /* 1e-10 (parallel-solver.hpp:92) */
  int maxRefinementSteps;    /* 5     (parallel-solver.hpp:94) */
  int last_refinement_steps;
  double last_residual;
} ora_par_solver;

void ora_get_work(int horz, int tid, int nthreads, int *beg, int *end); /* :23-28 */
ora_par_solver *ora_par_new(ora_problem *p, int num_threads);
void ora_par_free(ora_par_solver *s);
int ora_par_backward(ora_par_solver *s, double mueq);                 /* :132-206 */
int ora_par_forward(const ora_par_solver *s, double **xs, double **us,
                    double **vs, double **lbdas);                     /* :209-243 */
void ora_par_collapse_feedback(ora_par_solver *s);                    /* .hpp:41-51 */

/* ---- utils (gar/utils.hxx:88-182) ---------------------------------------- */
/* out3 = {dynErr, cstErr, dualErr} */
void ora_lqr_kkt_error(const ora_problem *p, double *const *xs, double *const *us,
                       double *const *vs, double *const *lbdas, double mueq,
                       const double *theta, double *out3);

/* ---- batched CPU baseline: OpenMP parallel-for over
 * independent problems, one serial sweep (backward+forward) each.
 * sols: per problem 4 arrays of N+1 pointers. Returns #failures. */
int ora_batch_sweep(ora_prox_solver **solvers, int nbatch, double mueq,
                    double ***xs, double ***us, double ***vs, double ***lbdas,
                    int nthreads);
int ora_omp_max_threads(void);

/* first-touch-local batched sweep (bench.py's cpu_baseline): see gar_oracle.c */
int ora_batch_sweep_local(const ora_problem *const *problems, int nbatch, double mueq, int nthreads,
                          int reps, double *seconds);

#ifdef __cplusplus
}
#endif
#endif
