// TEST INFRASTRUCTURE ONLY -- a C-ABI driver around the REFERENCE's own gar sources, compiled unchanged from
// /root/reference over oracle/ref_shim (a minimal Eigen-API stand-in; Eigen is absent from this image) by
// oracle/ref_build.sh into oracle/_ref/libgar_ref.so.  It pins oracle/gar_oracle.c (tests/test_ref_pin.py) and
// generates the committed fixtures tests/golden/ref_*.npz (tests/golden/make_ref_golden.py).  Nothing of the product
// links, loads or includes it, and nothing here is copied from the reference: this file only calls
//   aligator::gar::LqrProblemTpl / LqrKnotTpl                 (gar/lqr-problem.hpp, .hxx)
//   aligator::gar::ProximalRiccatiSolver                      (gar/proximal-riccati.hpp, .hxx -> riccati-kernel.hxx)
//   aligator::gar::ParallelRiccatiSolver                      (gar/parallel-solver.hpp, .hxx -> block-tridiagonal.hpp)
//   aligator::gar::RiccatiSolverDense                         (gar/dense-riccati.hpp, .hxx -> dense-kernel.hpp)
//   aligator::BunchKaufman                                    (core/bunchkaufman.hpp)
//   aligator::gar::symmetricBlockTridiagSolve                 (gar/block-tridiagonal.hpp)
//   aligator::gar::lqrComputeKktError                         (gar/utils.hpp, .hxx)
//   aligator::rotate_vec_left + the solvers' cycleAppend      (utils/mpc-util.hpp; the MPC cycle of
//                                                              solvers/proxddp/workspace.hxx:122-126)
#define ALIGATOR_MULTITHREADING
#include <sched.h>
#ifndef ALIGATOR_TRACY_SET_THREAD_NAME
#define ALIGATOR_TRACY_SET_THREAD_NAME(x) delete[] (x)
#endif
#include "aligator/gar/lqr-problem.hxx"
#include "aligator/gar/riccati-kernel.hxx"
#include "aligator/gar/proximal-riccati.hxx"
#include "aligator/gar/parallel-solver.hxx"
#include "aligator/gar/block-tridiagonal.hpp"
#include "aligator/gar/dense-riccati.hxx"
#include "aligator/gar/utils.hxx"
#include "aligator/utils/mpc-util.hpp"

#include <cstring>
#include <memory>
#include <string>

using namespace aligator;
using Problem = gar::LqrProblemTpl<double>;
using Knot = gar::LqrKnotTpl<double>;
using Serial = gar::ProximalRiccatiSolver<double>;
using Parallel = gar::ParallelRiccatiSolver<double>;
using DenseSolver = gar::RiccatiSolverDense<double>;
using VectorXs = Eigen::Matrix<double, Eigen::Dynamic, 1>;
using MatrixXs = Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic>;

namespace {
thread_local std::string g_err;

template <class M> void put(const M &m, double *out) { // column-major copy out (fb / fth are row-major inside)
  if (!out)
    return;
  for (Eigen::Index j = 0; j < m.cols(); ++j)
    for (Eigen::Index i = 0; i < m.rows(); ++i)
      out[i + j * m.rows()] = m(i, j);
}
template <class M> void get(M &&m, const double *in) {
  if (!in)
    return;
  for (Eigen::Index j = 0; j < m.cols(); ++j)
    for (Eigen::Index i = 0; i < m.rows(); ++i)
      m(i, j) = in[i + j * m.rows()];
}

struct Sol {
  std::vector<VectorXs> xs, us, vs, lbdas;
  explicit Sol(const Problem &p) { // gar/utils.hpp:114-142 (lqrInitializeSolution), sizes only
    const int N = p.horizon();
    lbdas.emplace_back(VectorXs::Zero(p.nc0()));
    for (int t = 0; t <= N; ++t) {
      const Knot &k = p.stages[size_t(t)];
      xs.emplace_back(VectorXs::Zero(k.nx));
      if (!(t == N && k.nu == 0))
        us.emplace_back(VectorXs::Zero(k.nu));
      vs.emplace_back(VectorXs::Zero(k.nc));
      if (t < N)
        lbdas.emplace_back(VectorXs::Zero(k.nx2));
    }
  }
};
void flatten(const std::vector<VectorXs> &v, double *out) {
  if (!out)
    return;
  for (const auto &x : v)
    for (Eigen::Index i = 0; i < x.size(); ++i)
      *out++ = x(i);
}
template <class Solver> void factor_out(Solver &s, int t, int what, double *out) {
  auto &d = s.datas[size_t(t)];
  switch (what) {
  case 0: put(d.ff.matrix(), out); break;
  case 1: put(d.fb.matrix(), out); break;   // (nu+nc+nx2) x nx, written column-major here
  case 2: put(d.fth.matrix(), out); break;
  case 3: put(d.vm.Vxx, out); break;
  case 4: put(d.vm.vx, out); break;
  case 5: put(d.vm.Vxt, out); break;
  case 6: put(d.vm.Vtt, out); break;
  case 7: put(d.vm.vt, out); break;
  case 8: put(d.kktMat.matrix(), out); break;
  case 9: put(d.Rhat, out); break;
  case 10: { // pivots of the stage's Bunch-Kaufman (as doubles)
    const auto &p = d.kktChol.pivots();
    for (Eigen::Index i = 0; i < p.size(); ++i)
      out[i] = double(p(i));
    break;
  }
  default: break;
  }
}
} // namespace

extern "C" {

const char *ref_last_error() { return g_err.c_str(); }

void *ref_problem_new(int horizon, const int *dims5, int nc0) {
  Problem::KnotVector knots;
  for (int t = 0; t <= horizon; ++t) {
    const int *d = dims5 + 5 * t;
    knots.emplace_back(uint(d[0]), uint(d[1]), uint(d[2]), uint(d[3]), uint(d[4]));
  }
  auto *p = new Problem(knots, long(nc0));
  p->G0.setZero();
  p->g0.setZero();
  return p;
}
void ref_problem_free(void *p) { delete static_cast<Problem *>(p); }
// blocks of knot t, column-major; NULL leaves a block untouched
void ref_problem_set_knot(void *pp, int t, const double *Q, const double *S, const double *R, const double *q,
                          const double *r, const double *A, const double *B, const double *f, const double *C,
                          const double *D, const double *d, const double *Gth, const double *Gx, const double *Gu,
                          const double *Gv, const double *gamma) {
  Knot &k = static_cast<Problem *>(pp)->stages[size_t(t)];
  get(k.Q, Q); get(k.S, S); get(k.R, R); get(k.q, q); get(k.r, r); get(k.A, A); get(k.B, B); get(k.f, f);
  get(k.C, C); get(k.D, D); get(k.d, d); get(k.Gth, Gth); get(k.Gx, Gx); get(k.Gu, Gu); get(k.Gv, Gv);
  get(k.gamma, gamma);
}
void ref_problem_set_init(void *pp, const double *G0, const double *g0) {
  auto *p = static_cast<Problem *>(pp);
  get(p->G0, G0);
  get(p->g0, g0);
}
// what: 0 Q 1 S 2 R 3 q 4 r 5 A 6 B 7 f 8 C 9 D 10 d 11 Gth 12 Gx 13 Gu 14 Gv 15 gamma; returns nth of the knot
int ref_problem_get_knot(void *pp, int t, int what, double *out) {
  Knot &k = static_cast<Problem *>(pp)->stages[size_t(t)];
  switch (what) {
  case 0: put(k.Q, out); break; case 1: put(k.S, out); break; case 2: put(k.R, out); break;
  case 3: put(k.q, out); break; case 4: put(k.r, out); break; case 5: put(k.A, out); break;
  case 6: put(k.B, out); break; case 7: put(k.f, out); break; case 8: put(k.C, out); break;
  case 9: put(k.D, out); break; case 10: put(k.d, out); break; case 11: put(k.Gth, out); break;
  case 12: put(k.Gx, out); break; case 13: put(k.Gu, out); break; case 14: put(k.Gv, out); break;
  case 15: put(k.gamma, out); break; default: break;
  }
  return int(k.nth);
}

// ---- ProximalRiccatiSolver --------------------------------------------------------------------------------
void *ref_serial_new(void *pp) { return new Serial(*static_cast<Problem *>(pp)); }
void ref_serial_free(void *s) { delete static_cast<Serial *>(s); }
int ref_serial_backward(void *s, double mueq) {
  try {
    return static_cast<Serial *>(s)->backward(mueq) ? 0 : 1;
  } catch (const std::exception &e) {
    g_err = e.what();
    return -1;
  }
}
// packed outputs xs | us | vs | lbdas as separate arrays (stage after stage)
int ref_serial_forward(void *sp, void *pp, const double *theta, double *xs, double *us, double *vs, double *lbdas) {
  auto *s = static_cast<Serial *>(sp);
  const Problem &p = *static_cast<Problem *>(pp);
  Sol sol(p);
  bool ok;
  if (theta && p.ntheta() > 0) {
    VectorXs th(Eigen::Index(p.ntheta()));
    get(th, theta);
    Eigen::Ref<const VectorXs> thr(th);
    ok = s->forward(sol.xs, sol.us, sol.vs, sol.lbdas, thr);
  } else {
    ok = s->forward(sol.xs, sol.us, sol.vs, sol.lbdas);
  }
  flatten(sol.xs, xs); flatten(sol.us, us); flatten(sol.vs, vs); flatten(sol.lbdas, lbdas);
  return ok ? 0 : 1;
}
void ref_serial_factor(void *s, int t, int what, double *out) { factor_out(*static_cast<Serial *>(s), t, what, out); }
// what: 0 kkt0.ff, 1 kkt0.fth, 2 thGrad, 3 thHess
void ref_serial_initial(void *sp, int what, double *out) {
  auto *s = static_cast<Serial *>(sp);
  switch (what) {
  case 0: put(s->kkt0.ff.matrix(), out); break;
  case 1: put(s->kkt0.fth.matrix(), out); break;
  case 2: put(s->thGrad, out); break;
  case 3: put(s->thHess, out); break;
  default: break;
  }
}

// ---- ParallelRiccatiSolver (mutates the problem: knots of non-final legs get nth = nx) ----------------------
void *ref_parallel_new(void *pp, int num_threads) {
  try {
    return new Parallel(*static_cast<Problem *>(pp), uint(num_threads));
  } catch (const std::exception &e) {
    g_err = e.what();
    return nullptr;
  }
}
void ref_parallel_free(void *s) { delete static_cast<Parallel *>(s); }
void ref_parallel_set_refinement(void *s, double thr, int steps) {
  static_cast<Parallel *>(s)->condensedThreshold = thr;
  static_cast<Parallel *>(s)->maxRefinementSteps = uint(steps);
}
int ref_parallel_backward(void *s, double mueq) {
  try {
    return static_cast<Parallel *>(s)->backward(mueq) ? 0 : 1;
  } catch (const std::exception &e) {
    g_err = e.what();
    return -1;
  }
}
int ref_parallel_forward(void *sp, void *pp, double *xs, double *us, double *vs, double *lbdas) {
  auto *s = static_cast<Parallel *>(sp);
  Sol sol(*static_cast<Problem *>(pp));
  const bool ok = s->forward(sol.xs, sol.us, sol.vs, sol.lbdas);
  flatten(sol.xs, xs); flatten(sol.us, us); flatten(sol.vs, vs); flatten(sol.lbdas, lbdas);
  return ok ? 0 : 1;
}
void ref_parallel_factor(void *s, int t, int what, double *out) { factor_out(*static_cast<Parallel *>(s), t, what, out); }
void ref_parallel_collapse_feedback(void *s) { static_cast<Parallel *>(s)->collapseFeedback(); }
int ref_parallel_condensed_dim(void *s) { return int(static_cast<Parallel *>(s)->condensedKktSolution.size()); }
void ref_parallel_condensed_solution(void *s, double *out) { put(static_cast<Parallel *>(s)->condensedKktSolution, out); }

// ---- RiccatiSolverDense ------------------------------------------------------------------------------------------
void *ref_dense_new(void *pp) { return new DenseSolver(*static_cast<Problem *>(pp)); }
void ref_dense_free(void *s) { delete static_cast<DenseSolver *>(s); }
int ref_dense_backward(void *s, double mueq) {
  try {
    return static_cast<DenseSolver *>(s)->backward(mueq) ? 0 : 1;
  } catch (const std::exception &e) {
    g_err = e.what();
    return -1;
  }
}
int ref_dense_forward(void *sp, void *pp, const double *theta, double *xs, double *us, double *vs, double *lbdas) {
  auto *s = static_cast<DenseSolver *>(sp);
  const Problem &p = *static_cast<Problem *>(pp);
  Sol sol(p);
  bool ok;
  if (theta && p.ntheta() > 0) {
    VectorXs th(Eigen::Index(p.ntheta()));
    get(th, theta);
    Eigen::Ref<const VectorXs> thr(th);
    ok = s->forward(sol.xs, sol.us, sol.vs, sol.lbdas, thr);
  } else {
    ok = s->forward(sol.xs, sol.us, sol.vs, sol.lbdas);
  }
  flatten(sol.xs, xs); flatten(sol.us, us); flatten(sol.vs, vs); flatten(sol.lbdas, lbdas);
  return ok ? 0 : 1;
}
// what: 0 ff, 1 fb, 2 fth (rows [K; Z; L; Y]), 3 Pxx, 4 px, 5 Pxt, 6 Ptt, 7 pt
void ref_dense_factor(void *sp, int t, int what, double *out) {
  auto *s = static_cast<DenseSolver *>(sp);
  auto &d = s->stage_factors[size_t(t)];
  switch (what) {
  case 0: put(d.ff.matrix(), out); break;
  case 1: put(d.fb.matrix(), out); break;
  case 2: put(d.ft.matrix(), out); break;
  case 3: put(s->Pxx[size_t(t)], out); break;
  case 4: put(s->px[size_t(t)], out); break;
  case 5: put(s->Pxt[size_t(t)], out); break;
  case 6: put(s->Ptt[size_t(t)], out); break;
  case 7: put(s->pt[size_t(t)], out); break;
  default: break;
  }
}
void ref_dense_initial(void *sp, int what, double *out) {
  auto *s = static_cast<DenseSolver *>(sp);
  switch (what) {
  case 0: put(s->kkt0.ff.matrix(), out); break;
  case 1: put(s->kkt0.fth.matrix(), out); break;
  case 2: put(s->thGrad, out); break;
  case 3: put(s->thHess, out); break;
  default: break;
  }
}

// ---- the MPC cycle as WorkspaceTpl::cycleAppend drives it (solvers/proxddp/workspace.hxx:122-126,
// solver-proxddp.hxx:208): rotate the knots left keeping the terminal one, a fresh knot in the last-but-one slot
// (filled by the caller through ref_problem_set_knot), then the solver's own cycleAppend on that knot -----------------
void ref_problem_cycle(void *pp, const int *d5) {
  auto *p = static_cast<Problem *>(pp);
  const size_t N = size_t(p->horizon());
  rotate_vec_left(p->stages, 0, 1);
  p->stages[N - 1] = Knot(uint(d5[0]), uint(d5[1]), uint(d5[2]), uint(d5[3]), uint(d5[4]));
}
int ref_serial_cycle_append(void *s, void *pp) {
  auto *p = static_cast<Problem *>(pp);
  try {
    static_cast<Serial *>(s)->cycleAppend(p->stages[size_t(p->horizon()) - 1]);
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return -1;
  }
}
int ref_parallel_cycle_append(void *s, void *pp) {
  auto *p = static_cast<Problem *>(pp);
  try {
    static_cast<Parallel *>(s)->cycleAppend(p->stages[size_t(p->horizon()) - 1]);
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return -1;
  }
}
int ref_dense_cycle_append(void *s, void *pp) {
  auto *p = static_cast<Problem *>(pp);
  try {
    static_cast<DenseSolver *>(s)->cycleAppend(p->stages[size_t(p->horizon()) - 1]);
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return -1;
  }
}

// ---- lqrComputeKktError (gar/utils.hxx:88-182) on packed trajectories; out = {dynErr, cstErr, dualErr} ---------------
int ref_kkt_error(void *pp, const double *xs, const double *us, const double *vs, const double *lbdas, double mueq,
                  const double *theta, double out[3]) {
  const Problem &p = *static_cast<Problem *>(pp);
  Sol sol(p);
  auto fill = [](std::vector<VectorXs> &v, const double *in) {
    for (auto &x : v)
      for (Eigen::Index i = 0; i < x.size(); ++i)
        x(i) = *in++;
  };
  fill(sol.xs, xs); fill(sol.us, us); fill(sol.vs, vs); fill(sol.lbdas, lbdas);
  if (sol.us.size() < sol.xs.size()) // the function indexes us[t] up to t = N only when knot.nu > 0
    sol.us.emplace_back(VectorXs::Zero(0));
  try {
    std::array<double, 3> e;
    if (theta && p.ntheta() > 0) {
      VectorXs th(Eigen::Index(p.ntheta()));
      get(th, theta);
      std::optional<typename math_types<double>::ConstVectorRef> opt{std::in_place, th};
      e = gar::lqrComputeKktError<double>(p, sol.xs, sol.us, sol.vs, sol.lbdas, mueq, opt, false);
    } else {
      e = gar::lqrComputeKktError<double>(p, sol.xs, sol.us, sol.vs, sol.lbdas, mueq, std::nullopt, false);
    }
    out[0] = e[0]; out[1] = e[1]; out[2] = e[2];
    return 0;
  } catch (const std::exception &ex) {
    g_err = ex.what();
    return -1;
  }
}

// ---- BunchKaufman ---------------------------------------------------------------------------------------------
int ref_bk_compute(int n, const double *A, double *ldlt, double *subdiag, int *pivots) {
  MatrixXs a(n, n);
  get(a, A);
  Eigen::BunchKaufman<MatrixXs> bk(n);
  bk.compute(a);
  put(bk.matrixLDLT(), ldlt);
  for (int i = 0; i < n; ++i) {
    subdiag[i] = bk.subdiag()(i);
    pivots[i] = bk.pivots()(i);
  }
  return int(bk.info());
}
void ref_bk_solve(int n, const double *A, int nrhs, double *X) {
  MatrixXs a(n, n), x(n, nrhs);
  get(a, A);
  get(x, X);
  Eigen::BunchKaufman<MatrixXs> bk(a);
  bk.solveInPlace(x);
  put(x, X);
}

// ---- symmetricBlockTridiagSolve (+ the down-looking variant), blocks column-major back to back ---------------
// dims[nb]; diag: nb blocks, sub/super: nb-1 blocks; rhs: sum(dims); solves in place like the reference
int ref_block_tridiag_solve(int nb, const int *dims, double *sub, double *diag, double *super, double *rhs, int down) {
  std::vector<MatrixXs> S, D, U;
  std::vector<long> rd(dims, dims + nb);
  long tot = 0;
  const double *ps = sub, *pd = diag, *pu = super;
  for (int i = 0; i < nb; ++i) {
    D.emplace_back(dims[i], dims[i]);
    get(D.back(), pd);
    pd += dims[i] * dims[i];
    tot += dims[i];
    if (i + 1 < nb) {
      U.emplace_back(dims[i], dims[i + 1]);
      get(U.back(), pu);
      pu += dims[i] * dims[i + 1];
      S.emplace_back(dims[i + 1], dims[i]);
      get(S.back(), ps);
      ps += dims[i] * dims[i + 1];
    }
  }
  VectorXs r(tot);
  get(r, rhs);
  BlkMatrix<VectorXs, -1, 1> rv(r, rd);
  std::vector<Eigen::BunchKaufman<MatrixXs>> facs;
  for (int i = 0; i < nb; ++i)
    facs.emplace_back(dims[i]);
  const bool ok = down ? gar::symmetricBlockTridiagSolveDownLooking(S, D, U, rv, facs)
                       : gar::symmetricBlockTridiagSolve(S, D, U, rv, facs);
  put(rv.matrix(), rhs);
  return ok ? 0 : 1;
}

} // extern "C"
