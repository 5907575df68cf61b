// gar_hip.hpp -- C++17 host-side mirror of the reference's gar solver surface on top of the C ABI
// (gar_hip.h).  Header-only, dependency-free (the reference's Eigen/Boost are not needed): dense
// blocks are column-major std::vector<double>, exactly what LqrKnotTpl's ArenaMatrix blocks hand
// to `.data()`.
//
// Same names, argument meaning and error behaviour as the reference:
//   gar::LqrKnot                 <- gar::LqrKnotTpl<double>        (gar/lqr-problem.hpp:34-103)
//   gar::LqrProblem              <- gar::LqrProblemTpl<double>     (gar/lqr-problem.hpp:105-195)
//   gar::RiccatiSolverBase       <- gar::RiccatiSolverBase<double> (gar/riccati-base.hpp:13-37)
//   gar::ProximalRiccatiSolver   <- gar/proximal-riccati.hpp:17-47, .hxx:13-86
//   gar::ParallelRiccatiSolver   <- gar/parallel-solver.hpp:26-110, .hxx:32-258
//   gar::lqrInitializeSolution   <- gar/utils.hpp:114-142
//   gar::lqrComputeKktError      <- gar/utils.hxx:88-182
// A failed stage factorisation throws std::runtime_error where the reference throws
// aligator::RuntimeError (riccati-kernel.hxx:239-241); a solver cannot be created without a HIP
// device (there is no CPU path).  tests/cpp/test_gar.cpp uses this header the way
// tests/gar/riccati.cpp and tests/gar/parallel.cpp use the reference's.
#pragma once
#include "gar_hip.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

namespace aligator_hip::gar {

using uint = unsigned int;
using VectorXs = std::vector<double>;
using VectorOfVectors = std::vector<VectorXs>;

// column-major dense block
struct Matrix {
  int rows = 0, cols = 0;
  std::vector<double> v;
  Matrix() = default;
  Matrix(int r, int c) : rows(r), cols(c), v((size_t)r * c, 0.0) {}
  double &operator()(int i, int j) { return v[(size_t)j * rows + i]; }
  double operator()(int i, int j) const { return v[(size_t)j * rows + i]; }
  double *data() { return v.data(); }
  const double *data() const { return v.data(); }
  void setZero() { std::fill(v.begin(), v.end(), 0.0); }
  void setIdentity() {
    setZero();
    for (int i = 0; i < std::min(rows, cols); ++i)
      (*this)(i, i) = 1.0;
  }
};

// One stage of the constrained LQ problem (lqr-problem.hpp:16-33):
// cost 1/2 [x;u]^T [Q S; S^T R] [x;u] + q^T x + r^T u, x' = A x + B u + f, C x + D u + d = 0,
// optional parameterisation (Gth, Gx, Gu, Gv, gamma)
struct LqrKnot {
  uint nx, nu, nc, nx2, nth;
  Matrix Q, S, R;
  VectorXs q, r;
  Matrix A, B;
  VectorXs f;
  Matrix C, D;
  VectorXs d;
  Matrix Gth, Gx, Gu, Gv;
  VectorXs gamma;
  LqrKnot(uint nx, uint nu, uint nc, uint nx2, uint nth = 0)
      : nx(nx), nu(nu), nc(nc), nx2(nx2), nth(nth), Q(nx, nx), S(nx, nu), R(nu, nu), q(nx, 0.0),
        r(nu, 0.0), A(nx2, nx), B(nx2, nu), f(nx2, 0.0), C(nc, nx), D(nc, nu), d(nc, 0.0),
        Gth(nth, nth), Gx(nx, nth), Gu(nu, nth), Gv(nc, nth), gamma(nth, 0.0) {}
  LqrKnot(uint nx, uint nu, uint nc) : LqrKnot(nx, nu, nc, nx) {}
  // lqr-problem.hxx:232-241
  void addParameterization(uint n) {
    nth = n;
    Gth = Matrix(n, n);
    Gx = Matrix(nx, n);
    Gu = Matrix(nu, n);
    Gv = Matrix(nc, n);
    gamma.assign(n, 0.0);
  }
};

struct LqrProblem {
  using KnotVector = std::vector<LqrKnot>;
  Matrix G0;
  VectorXs g0;
  KnotVector stages;
  LqrProblem(KnotVector knots, long nc0)
      : G0((int)nc0, knots.empty() ? 0 : (int)knots[0].nx), g0((size_t)nc0, 0.0),
        stages(std::move(knots)) {}
  int horizon() const noexcept { return (int)stages.size() - 1; }
  uint nc0() const noexcept { return (uint)g0.size(); }
  bool isInitialized() const { return !stages.empty(); }
};

// (xs, us, vs, lbdas), zero-filled (gar/utils.hpp:114-142)
inline std::array<VectorOfVectors, 4> lqrInitializeSolution(const LqrProblem &problem) {
  const int N = problem.horizon();
  VectorOfVectors xs, us, vs, lbdas;
  lbdas.emplace_back(problem.nc0(), 0.0);
  for (int t = 0; t <= N; ++t) {
    const LqrKnot &k = problem.stages[t];
    xs.emplace_back(k.nx, 0.0);
    if (!(t == N && k.nu == 0))
      us.emplace_back(k.nu, 0.0);
    vs.emplace_back(k.nc, 0.0);
    if (t < N)
      lbdas.emplace_back(k.nx2, 0.0);
  }
  return {xs, us, vs, lbdas};
}

struct KktError {
  double dyn = 0, cstr = 0, dual = 0, max = 0;
};

// infinity norms of the KKT residuals (gar/utils.hxx:88-182)
inline KktError lqrComputeKktError(const LqrProblem &p, const VectorOfVectors &xs,
                                   const VectorOfVectors &us, const VectorOfVectors &vs,
                                   const VectorOfVectors &lbdas, double mueq = 0.0,
                                   const std::optional<VectorXs> &theta = std::nullopt) {
  auto inf = [](const VectorXs &v) {
    double m = 0;
    for (double x : v)
      m = std::max(m, std::fabs(x));
    return m;
  };
  auto gemv = [](const Matrix &M, const VectorXs &x, VectorXs &y, double sgn = 1.0) { // y += sgn M x
    for (int j = 0; j < M.cols; ++j)
      for (int i = 0; i < M.rows; ++i)
        y[i] += sgn * M(i, j) * x[j];
  };
  auto gemvT = [](const Matrix &M, const VectorXs &x, VectorXs &y) { // y += M^T x
    for (int j = 0; j < M.cols; ++j)
      for (int i = 0; i < M.rows; ++i)
        y[j] += M(i, j) * x[i];
  };
  KktError e;
  const int N = p.horizon();
  {
    VectorXs r0 = p.g0;
    gemv(p.G0, xs[0], r0);
    e.dyn = inf(r0);
  }
  for (int t = 0; t <= N; ++t) {
    const LqrKnot &k = p.stages[t];
    VectorXs cst = k.d, gx = k.q, gu = k.r;
    gemv(k.C, xs[t], cst);
    for (uint i = 0; i < k.nc; ++i)
      cst[i] -= mueq * vs[t][i];
    gemv(k.Q, xs[t], gx);
    gemvT(k.C, vs[t], gx);
    gemvT(k.S, xs[t], gu);
    gemvT(k.D, vs[t], gu);
    if (k.nu > 0) {
      gemv(k.D, us[t], cst);
      gemv(k.S, us[t], gx);
      gemv(k.R, us[t], gu);
    }
    if (t == 0) {
      gemvT(p.G0, lbdas[0], gx);
    } else {
      for (uint i = 0; i < k.nx; ++i)
        gx[i] -= lbdas[t][i];
    }
    if (t < N) {
      VectorXs dyn = k.f;
      gemv(k.A, xs[t], dyn);
      gemv(k.B, us[t], dyn);
      for (uint i = 0; i < k.nx2; ++i)
        dyn[i] -= xs[t + 1][i];
      gemvT(k.A, lbdas[t + 1], gx);
      gemvT(k.B, lbdas[t + 1], gu);
      e.dyn = std::max(e.dyn, inf(dyn));
    }
    if (theta && k.nth > 0) {
      gemv(k.Gx, *theta, gx);
      gemv(k.Gu, *theta, gu);
    }
    e.dual = std::max({e.dual, inf(gx), inf(gu)});
    e.cstr = std::max(e.cstr, inf(cst));
  }
  e.max = std::max({e.dyn, e.cstr, e.dual});
  return e;
}

// gar/riccati-base.hpp:13-37
class RiccatiSolverBase {
public:
  virtual bool backward(const double mueq) = 0;
  virtual bool forward(VectorOfVectors &xs, VectorOfVectors &us, VectorOfVectors &vs,
                       VectorOfVectors &lbdas,
                       const std::optional<VectorXs> &theta = std::nullopt) const = 0;
  virtual void cycleAppend(const LqrKnot &knot) = 0;
  virtual void collapseFeedback() {}
  virtual VectorXs getFeedforward(size_t) = 0;
  virtual Matrix getFeedback(size_t) = 0; // ROW-major in the reference; here (nr x nx) column-major copy
  virtual ~RiccatiSolverBase() = default;
};

namespace detail {

// (Kernel selection and padding onto a specialised shape happen inside the C ABI, gar_hip_solver_create: this
// mirror passes the knots' own dimensions and receives results in them.)

// the six virtuals over the C ABI, shared by the two solvers
class HipSolver : public RiccatiSolverBase {
public:
  ~HipSolver() override { gar_hip_solver_destroy(h_); }
  HipSolver(const HipSolver &) = delete;
  HipSolver &operator=(const HipSolver &) = delete;

  bool forward(VectorOfVectors &xs, VectorOfVectors &us, VectorOfVectors &vs,
               VectorOfVectors &lbdas,
               const std::optional<VectorXs> &theta = std::nullopt) const override {
    check(gar_hip_forward(h_, theta ? theta->data() : nullptr));
    size_t nxs = 0, nus = 0, nvs = 0, nls = problem_->nc0();
    const int N = problem_->horizon();
    for (int t = 0; t <= N; ++t) {
      const LqrKnot &k = problem_->stages[t];
      nxs += k.nx;
      nus += k.nu;
      nvs += k.nc;
      if (t < N)
        nls += k.nx2;
    }
    // ONE device-to-host copy of the whole solution record xs | us | vs | lbdas into the library's
    // pinned buffer (gar_hip_fetch_results), then the scatter into the caller's vectors
    check(gar_hip_fetch_results(h_, 0, 1));
    int64_t offs[3];
    const double *rec = gar_hip_host_results(h_, offs) + offs[0];
    const double *p = rec;
    p = scatter(p, xs, nxs);
    p = scatter(p, us, nus);
    p = scatter(p, vs, nvs);
    scatter(p, lbdas, nls);
    return true;
  }
  void cycleAppend(const LqrKnot &knot) override {
    gains_valid_ = false;
    const int32_t d[5] = {(int)knot.nx, (int)knot.nu, (int)knot.nc, (int)knot.nx2, (int)knot.nth};
    check(gar_hip_cycle_append(h_, d));
  }
  VectorXs getFeedforward(size_t i) override {
    const LqrKnot &k = problem_->stages[i];
    const double *src = gains(i, 0); // block rows [kff; zff; yff] (dense: [kff; zff; lff; yff], dense-riccati.hpp:49)
    return VectorXs(src, src + k.nu + k.nc + (dense_ ? 2 * k.nx2 : k.nx2));
  }
  Matrix getFeedback(size_t i) override {
    const LqrKnot &k = problem_->stages[i];
    const int nr = (int)(k.nu + k.nc + (dense_ ? 2 * k.nx2 : k.nx2));
    const double *rm = gains(i, 1); // row-major nr x nx like StageFactor::fb
    Matrix fb(nr, (int)k.nx);
    for (int r = 0; r < nr; ++r)
      for (uint j = 0; j < k.nx; ++j)
        fb(r, (int)j) = rm[(size_t)r * k.nx + j];
    return fb;
  }
  const char *kernelName() const { return gar_hip_kernel_name(h_); }
  /// the measured-best leg count for ONE problem of this shape on one device (gar_hip_suggest_num_legs)
  static int suggestedNumLegs(int horizon, int nx, int nu) { return gar_hip_suggest_num_legs(horizon, nx, nu); }
  /// leg mode: the fast solver of the condensed system ("cyclic", "reduced+cyclic", ...; gar_hip.h)
  const char *condensedSolverName() const { return gar_hip_condensed_solver_name(h_); }

protected:
  // devices: more than one id = the legs split over these devices inside this one object (gar_hip_multi_create)
  HipSolver(LqrProblem &problem, int num_legs, int device, bool dense = false, const std::vector<int> &devices = {})
      : problem_(&problem), dense_(dense) {
    const int N = problem.horizon();
    std::vector<int32_t> dims5;
    for (const LqrKnot &k : problem.stages) {
      const int32_t d[5] = {(int)k.nx, (int)k.nu, (int)k.nc, (int)k.nx2, num_legs > 1 ? 0 : (int)k.nth};
      dims5.insert(dims5.end(), d, d + 5);
    }
    h_ = dense ? gar_hip_solver_create_dense(device, N, dims5.data(), (int)problem.nc0(), 1)
         : devices.empty() ? gar_hip_solver_create(device, N, dims5.data(), (int)problem.nc0(), 1, num_legs)
                           : gar_hip_multi_create((int)devices.size(), devices.data(), N, dims5.data(), (int)problem.nc0(), 1, num_legs);
    if (!h_)
      throw std::runtime_error(gar_hip_last_error());
  }
  // the reference re-reads the caller's problem on every backward (proximal-riccati.hxx:37): the whole problem and
  // backward(mueq) in one call (gar_hip_backward_blocks)
  void backward_blocks(double mueq) const {
    const auto &st = problem_->stages;
    blocks_.resize(16 * st.size());
    for (size_t t = 0; t < st.size(); ++t) {
      const LqrKnot &k = st[t];
      const double *b[16] = {k.Q.data(), k.S.data(), k.R.data(), k.q.data(), k.r.data(), k.A.data(), k.B.data(), k.f.data(),
                             k.C.data(), k.D.data(), k.d.data(), k.Gth.data(), k.Gx.data(), k.Gu.data(), k.Gv.data(),
                             k.gamma.data()};
      std::copy(b, b + 16, blocks_.begin() + 16 * t);
    }
    check(gar_hip_backward_blocks(h_, blocks_.data(), problem_->G0.data(), problem_->g0.data(), mueq));
  }
  void upload() const {
    const auto &st = problem_->stages;
    for (int t = 0; t < (int)st.size(); ++t) {
      const LqrKnot &k = st[t];
      check(gar_hip_upload_stage(h_, 0, t, k.Q.data(), k.S.data(), k.R.data(), k.q.data(),
                                 k.r.data(), k.A.data(), k.B.data(), k.f.data(), k.C.data(),
                                 k.D.data(), k.d.data(), k.Gth.data(), k.Gx.data(), k.Gu.data(),
                                 k.Gv.data(), k.gamma.data()));
    }
    check(gar_hip_set_init(h_, 0, problem_->G0.data(), problem_->g0.data()));
  }
  static void check(int rc) {
    if (rc != GAR_HIP_OK)
      throw std::runtime_error(gar_hip_last_error());
  }
  // one part of the packed solution record (`total` doubles on the device) into the caller's vectors
  static const double *scatter(const double *packed, VectorOfVectors &out, size_t total) {
    size_t p = 0;
    for (VectorXs &v : out) {
      std::copy(packed + p, packed + p + v.size(), v.begin());
      p += v.size();
    }
    return packed + total;
  }
  // the gains of every stage through ONE device-side gather and ONE copy (gar_hip_fetch_results),
  // kept until the next backward / collapseFeedback / cycleAppend
  const double *gains(size_t i, int which) const {
    if (!gains_valid_) {
      check(gar_hip_fetch_results(h_, 0, 2));
      gains_valid_ = true;
    }
    int64_t offs[3], go[2];
    const double *base = gar_hip_host_results(h_, offs);
    check(gar_hip_gains_offsets(h_, (int)i, go));
    return base + offs[1 + which] + go[which];
  }
  LqrProblem *problem_;
  bool dense_ = false; // RiccatiSolverDense: nu+nc+2*nx2 gain rows
  mutable bool gains_valid_ = false;
  mutable std::vector<const double *> blocks_;
  gar_hip_solver *h_ = nullptr;
};

} // namespace detail

// gar/proximal-riccati.hpp:17-47
class ProximalRiccatiSolver : public detail::HipSolver {
public:
  explicit ProximalRiccatiSolver(LqrProblem &problem, int device = 0)
      : HipSolver(problem, 1, device) {}
  bool backward(const double mueq) override {
    gains_valid_ = false;
    upload();
    check(gar_hip_backward(h_, mueq));
    check(gar_hip_prefetch_gains(h_, 0)); // the gains start travelling now, under forward() // GAR_HIP_ERR_FACTOR -> "Failed stage LDL factorization"
    return true;
  }
};

// gar/dense-riccati.hpp:19-56: the stage-dense solver (one Bunch-Kaufman factorisation of the whole
// (nu+nc+2 nx2)^2 stage matrix per knot); getFeedforward / getFeedback return the block rows
// [K; Z; L; Y] as the reference's stage_factors[i].ff / .fb
class RiccatiSolverDense : public detail::HipSolver {
public:
  explicit RiccatiSolverDense(LqrProblem &problem, int device = 0)
      : HipSolver(problem, 1, device, true) {}
  bool backward(const double mueq) override {
    gains_valid_ = false;
    upload();
    check(gar_hip_backward(h_, mueq));
    check(gar_hip_prefetch_gains(h_, 0)); // the gains start travelling now, under forward()
    return true;
  }
};

// gar/parallel-solver.hpp:26-110
class ParallelRiccatiSolver : public detail::HipSolver {
public:
  ParallelRiccatiSolver(LqrProblem &problem, const uint num_threads, int device = 0)
      : HipSolver(problem, check_threads(num_threads), device), numThreads_(num_threads) {}
  /// the legs split over `devices` (device ids) inside this one object: one process, the boundary exchange inside
  /// backward() (include/gar_hip.h, gar_hip_multi_create)
  ParallelRiccatiSolver(LqrProblem &problem, const uint num_threads, const std::vector<int> &devices)
      : HipSolver(problem, check_threads(num_threads), devices.empty() ? 0 : devices[0], false, devices),
        numThreads_(num_threads) {}
  int numDevices() const { return gar_hip_num_devices(h_); }
  bool backward(const double mueq) override {
    check(gar_hip_set_refinement(h_, condensedThreshold, (int)maxRefinementSteps));
    gains_valid_ = false;
    backward_blocks(mueq);
    check(gar_hip_prefetch_gains(h_, 0)); // the gains start travelling now, under forward()
    return true;
  }
  void collapseFeedback() override {
    gains_valid_ = false;
    check(gar_hip_collapse_feedback(h_));
  }
  uint getNumThreads() const { return numThreads_; }
  double condensedThreshold = 1e-10; // parallel-solver.hpp:92
  uint maxRefinementSteps = 5;       // parallel-solver.hpp:94

private:
  static int check_threads(uint n) {
    if (n < 2) // parallel-solver.hxx:42-46
      throw std::runtime_error("(ParallelRiccatiSolver) numThreads should be greater than or equal to 2.");
    return (int)n;
  }
  uint numThreads_;
};

} // namespace aligator_hip::gar
