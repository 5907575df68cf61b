// TEST INFRASTRUCTURE -- BASELINE configs[0] / [4] as stated: the REFERENCE's own SolverProxDDP loop
// (include/aligator/solvers/proxddp/solver-proxddp.hxx, compiled unchanged from /root/reference over the Eigen-API
// stand-in oracle/ref_shim: oracle/ref_ddp_build.sh) calling the MI355X gar backend through the shipped
// include/aligator/gar/hip-riccati.hpp, next to the same loop on the reference's own ProximalRiccatiSolver /
// ParallelRiccatiSolver:
//
//   1. tests/lqr.cpp:29-75 ("lqr_proxddp": nx = 4, nu = 2, 100 steps, random dense LQR, mt19937_64{42}): converges,
//      `results_.num_iters == 1` -- with `linear_solver_` as SolverProxDDP::setup made it, and replaced by
//      HipRiccatiSolver (serial; 4 legs under LQSolverChoice::PARALLEL); trajectories and cost of the runs compared;
//   2. bench/lqr.cpp:23-57, 68-87 (BM_lqr_prox: dim = 56, nu = 22 -- the Talos-walk LQ shape --, TOL 1e-7, mu 1e-10,
//      max_iters 2, linear roll-out): wall-clock per `solver.run` and ProxDDP iterations per second, the reference's
//      SERIAL / PARALLEL solvers (OpenMP threads stated) and the HIP backend (serial, legs) on the same box.
//      The reference side runs over the naive stand-in products: its time is an UPPER bound on the Eigen build's.
//
// Linked with the wave-emulator build (CPU test) or, -DDDP_GPU, with the real aligator_amd/libgar_hip.so.
#ifndef ALIGATOR_MULTITHREADING
#define ALIGATOR_MULTITHREADING
#endif
#include "aligator/core/traj-opt-problem.hpp"
#include "aligator/solvers/proxddp/solver-proxddp.hpp"
#include "aligator/utils/rollout.hpp"
#include "aligator/modelling/costs/quad-costs.hpp"
#include "aligator/modelling/linear-discrete-dynamics.hpp"
#include "aligator/gar/hip-riccati.hpp" // this repository's include/

#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>

using namespace aligator;
using T = double;
using StageModel = StageModelTpl<T>;
using TrajOptProblem = TrajOptProblemTpl<T>;
using Solver = SolverProxDDPTpl<T>;
using LinearDynamics = dynamics::LinearDiscreteDynamicsTpl<T>;
using QuadraticCost = QuadraticCostTpl<T>;
using Eigen::MatrixXd;
using Eigen::VectorXd;

static std::mt19937_64 urng{42}; // tests/lqr.cpp:23
static std::normal_distribution<double> nrm;
static MatrixXd randn(int r, int c) { // MatrixXd::NullaryExpr(r, c, norm_gen): column by column
  MatrixXd m(r, c);
  for (int j = 0; j < c; ++j)
    for (int i = 0; i < r; ++i)
      m(i, j) = nrm(urng);
  return m;
}

// tests/lqr.cpp:29-57
static TrajOptProblem test_problem(size_t nsteps = 100, int nx = 4, int nu = 2) {
  MatrixXd A(nx, nx);
  A.setIdentity();
  A.bottomRightCorner(2, 2) = randn(2, 2);
  MatrixXd B = randn(nx, nu);
  VectorXd x0 = randn(nx, 1);
  auto dyn_model = LinearDynamics(A, B, VectorXd::Zero(nx));
  MatrixXd Q = randn(nx, nx);
  Q = Q.transpose() * Q;
  VectorXd q = randn(nx, 1);
  MatrixXd R = randn(nu, nu);
  R = R.transpose() * R;
  VectorXd r = VectorXd::Zero(nu);
  QuadraticCost cost = QuadraticCost(Q, R, q, r);
  QuadraticCost term_cost = QuadraticCost(Q * 10., MatrixXd());
  auto stage = StageModel(cost, dyn_model);
  std::vector<xyz::polymorphic<StageModel>> stages(nsteps, stage);
  return TrajOptProblem(x0, stages, term_cost);
}

// bench/lqr.cpp:23-57
static TrajOptProblem bench_problem(size_t nsteps, int dim = 56, int nu = 22) {
  MatrixXd A(dim, dim), B(dim, nu);
  VectorXd c_(dim);
  A.setIdentity();
  B.setIdentity();
  c_.setConstant(0.1);
  MatrixXd w_x(dim, dim), w_u(nu, nu);
  w_x.setIdentity();
  w_x(0, 0) = 2.;
  w_u.setIdentity();
  w_u *= 1e-2;
  auto dynptr = LinearDynamics(A, B, c_);
  auto space = dynptr.space_next_;
  auto rcost = QuadraticCost(w_x, w_u);
  auto stage = StageModel(rcost, dynptr);
  auto term_cost = rcost;
  VectorXd x0(dim);
  std::mt19937 g(7);
  std::uniform_real_distribution<double> u11(-1.0, 1.0);
  for (int i = 0; i < dim; ++i) // (x0.setRandom())
    x0(i) = u11(g);
  TrajOptProblem problem(x0, nu, space, term_cost);
  for (size_t i = 0; i < nsteps; i++)
    problem.addStage(stage);
  return problem;
}

// the share of a run spent INSIDE the linear solver (every virtual of RiccatiSolverBase, gains included): a
// pass-through around whichever solver `linear_solver_` holds -- what is left of the run is the reference's own loop
// (updateLQSubproblem, roll-out, merit function ...) over the Eigen stand-in, on BOTH sides of the comparison
struct TimedSolver : gar::RiccatiSolverBase<double> {
  using Base = gar::RiccatiSolverBase<double>;
  std::unique_ptr<Base> in;
  mutable double us = 0.0;
  explicit TimedSolver(std::unique_ptr<Base> s) : in(std::move(s)) {}
  struct Tick {
    double &acc;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit Tick(double &a) : acc(a) {}
    ~Tick() { acc += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
  };
  bool backward(const double mu) override { Tick t(us); return in->backward(mu); }
  bool forward(std::vector<VectorXs> &xs, std::vector<VectorXs> &us_, std::vector<VectorXs> &vs, std::vector<VectorXs> &lbdas,
               const std::optional<ConstVectorRef> &theta = std::nullopt) const override {
    Tick t(us);
    return in->forward(xs, us_, vs, lbdas, theta);
  }
  void cycleAppend(const LqrKnot &k) override { Tick t(us); in->cycleAppend(k); }
  void collapseFeedback() override { Tick t(us); in->collapseFeedback(); }
  VectorRef getFeedforward(size_t i) override { Tick t(us); return in->getFeedforward(i); }
  RowMatrixRef getFeedback(size_t i) override { Tick t(us); return in->getFeedback(i); }
};

enum class Backend { REF, HIP };
// setup as the reference does; HIP: the public member `linear_solver_` (solver-proxddp.hpp:181) is replaced by the
// MI355X backend on the SAME workspace_.lqr_problem (for legs: the problem ParallelRiccatiSolver's constructor has
// just parameterised, parallel-solver.hxx:51-82, with the same number of legs)
static void plug(Solver &s, const TrajOptProblem &problem, Backend be, int legs) {
  s.rollout_type_ = RolloutType::LINEAR;
  s.linear_solver_choice = legs > 1 ? LQSolverChoice::PARALLEL : LQSolverChoice::SERIAL;
  if (legs > 1)
    s.setNumThreads(size_t(legs));
  s.setup(problem);
  if (be == Backend::HIP)
    s.linear_solver_ = std::make_unique<gar::HipRiccatiSolver>(s.workspace_.lqr_problem, legs);
}
static TimedSolver &timed(Solver &s) { // wraps whatever plug() left in place
  auto t = std::make_unique<TimedSolver>(std::move(s.linear_solver_));
  TimedSolver &r = *t;
  s.linear_solver_ = std::move(t);
  return r;
}

static double maxdiff(const std::vector<VectorXd> &a, const std::vector<VectorXd> &b) {
  double m = 0;
  for (size_t i = 0; i < a.size(); ++i)
    for (Eigen::Index j = 0; j < a[i].size(); ++j)
      m = std::max(m, std::abs(a[i](j) - b[i](j)));
  return m;
}

int main(int argc, char **argv) {
  const bool json = argc > 1 && std::string(argv[1]) == "--json";
  const bool quick = argc > 1 && std::string(argv[1]) == "--quick"; // (the emulator build: part 1 and a tiny part 2)
  int bad = 0;
  // ---- 1. tests/lqr.cpp: one iteration, converged ---------------------------------------------------------------
  struct Run { const char *what; Backend be; int legs; };
  const Run runs[] = {{"reference SERIAL", Backend::REF, 1}, {"HIP serial", Backend::HIP, 1},
                      {"reference PARALLEL(4)", Backend::REF, 4}, {"HIP 4 legs", Backend::HIP, 4}};
  std::vector<VectorXd> xs0, us0;
  double cost0 = 0;
  std::string part1 = "";
  for (const Run &r : runs) {
    urng.seed(42);
    nrm.reset();
    TrajOptProblem problem = test_problem();
    Solver ddp(1e-6, 1e-8);
    ddp.max_iters = 2;
    ddp.verbose_ = QUIET;
    plug(ddp, problem, r.be, r.legs);
    const bool conv = ddp.run(problem);
    const size_t it = ddp.results_.num_iters;
    double dx = 0, du = 0;
    if (xs0.empty()) {
      xs0 = ddp.results_.xs;
      us0 = ddp.results_.us;
      cost0 = ddp.results_.traj_cost_;
    } else {
      dx = maxdiff(ddp.results_.xs, xs0);
      du = maxdiff(ddp.results_.us, us0);
    }
    const char *kern = r.be == Backend::HIP ? static_cast<gar::HipRiccatiSolver &>(*ddp.linear_solver_).kernelName() : "-";
    const bool ok = conv && it == 1 && dx <= 1e-8 && du <= 1e-8 && std::abs(ddp.results_.traj_cost_ - cost0) <= 1e-8 * std::abs(cost0);
    if (!json)
      std::printf("tests/lqr.cpp  %-22s conv %d num_iters %zu cost %.9e prim %.1e dual %.1e  |x - ref| %.1e |u - ref| %.1e  kernel %s  %s\n",
                  r.what, (int)conv, it, ddp.results_.traj_cost_, ddp.results_.prim_infeas, ddp.results_.dual_infeas, dx, du, kern,
                  ok ? "ok" : "MISMATCH");
    bad += !ok;
    char buf[256];
    std::snprintf(buf, sizeof buf, "%s\"%s\": {\"converged\": %s, \"num_iters\": %zu, \"max_dx_vs_reference_serial\": %.3e, \"max_du\": %.3e}",
                  part1.empty() ? "" : ", ", r.what, conv ? "true" : "false", it, dx, du);
    part1 += buf;
  }
  // ---- 2. bench/lqr.cpp: wall clock of solver.run ---------------------------------------------------------------
  struct Cfg { const char *what; Backend be; int legs; };
  std::string part2 = "";
  // horizons: 64 and 256 (rounds 5's rows) and the reference's own, bench/talos-walk.cpp:26-28: 165, 220, 275; the
  // reference's PARALLEL solver at the thread counts of its benchmark, :94-111: 2, 4, 6, 8
  const size_t sizes_full[] = {64, 165, 220, 256, 275}, sizes_quick[] = {16};
  const size_t *sizes = quick ? sizes_quick : sizes_full;
  const size_t nsizes = quick ? 1 : 5;
  for (size_t si = 0; si < nsizes; ++si) {
    const size_t nsteps = sizes[si];
    const int big = quick ? 2 : int(nsteps / 8);
    std::vector<Cfg> cfgs = {{"reference SERIAL", Backend::REF, 1}};
    if (quick) {
      cfgs.push_back({"reference PARALLEL(2 threads)", Backend::REF, 2});
    } else {
      cfgs.push_back({"reference PARALLEL(2 threads)", Backend::REF, 2});
      cfgs.push_back({"reference PARALLEL(4 threads)", Backend::REF, 4});
      cfgs.push_back({"reference PARALLEL(6 threads)", Backend::REF, 6});
      cfgs.push_back({"reference PARALLEL(8 threads)", Backend::REF, 8});
    }
    cfgs.push_back({"HIP serial", Backend::HIP, 1});
    cfgs.push_back({"HIP legs", Backend::HIP, big});
    std::vector<VectorXd> xref;
    for (const Cfg &c : cfgs) {
      TrajOptProblem problem = bench_problem(nsteps, quick ? 8 : 56, quick ? 4 : 22);
      const auto &dyn = *problem.stages_[0]->dynamics_;
      const VectorXd &x0 = problem.getInitState();
      std::vector<VectorXd> us_init;
      us_default_init(problem, us_init);
      std::vector<VectorXd> xs_init = rollout(dyn, x0, us_init);
      Solver solver(1e-7, 1e-10, 2, QUIET);
      solver.force_initial_condition_ = false;
      plug(solver, problem, c.be, c.legs);
      TimedSolver &tm = timed(solver);
      double best = 1e30, best_in = 0;
      size_t iters = 0;
      bool conv = true;
      const int reps = quick ? 2 : 5;
      for (int rep = 0; rep < reps + 1; ++rep) { // first run: warm-up
        tm.us = 0.0;
        const auto t0 = std::chrono::steady_clock::now();
        conv = solver.run(problem, xs_init, us_init) && conv;
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (rep > 0 && us < best) {
          best = us;
          best_in = tm.us;
        }
        iters = solver.results_.num_iters;
      }
      double dx = 0;
      if (xref.empty())
        xref = solver.results_.xs;
      else
        dx = maxdiff(solver.results_.xs, xref);
      const char *kern = c.be == Backend::HIP ? static_cast<gar::HipRiccatiSolver &>(*tm.in).kernelName() : "-";
      const bool ok = conv && dx <= 1e-7;
      if (!json)
        std::printf("bench/lqr.cpp  N=%-4zu %-30s legs/threads %-3d run %9.1f us (inside the linear solver %8.1f us = %4.1f %%)  %zu iteration(s)  => %8.1f ProxDDP iterations/s  |x - ref| %.1e kernel %s %s\n",
                    nsteps, c.what, c.legs, best, best_in, 100.0 * best_in / best, iters, 1e6 * double(iters) / best, dx, kern, ok ? "ok" : "MISMATCH");
      bad += !ok;
      char buf[400];
      std::snprintf(buf, sizeof buf, "%s\"N%zu %s\": {\"legs_or_threads\": %d, \"us_per_run\": %.1f, \"us_inside_the_linear_solver\": %.1f, \"iterations\": %zu, \"iterations_per_s\": %.1f, "
                    "\"converged\": %s, \"max_dx_vs_reference_serial\": %.2e, \"kernel\": \"%s\"}",
                    part2.empty() ? "" : ", ", nsteps, c.what, c.legs, best, best_in, iters, 1e6 * double(iters) / best, conv ? "true" : "false", dx, kern);
      part2 += buf;
    }
  }
  if (json)
    std::printf("{\"what\": \"the reference's own SolverProxDDP (compiled unchanged over the Eigen stand-in) with linear_solver_ = the reference's "
                "solvers / = HipRiccatiSolver; reference-side times are over naive products: an upper bound on the Eigen build's\", "
                "\"tests_lqr_cpp\": {%s}, \"bench_lqr_cpp_dim56_nu22\": {%s}, \"ok\": %s}\n", part1.c_str(), part2.c_str(), bad ? "false" : "true");
  else
    std::printf(bad ? "%d case(s) FAILED\n" : "proxddp ok\n", bad);
  return bad ? 1 : 0;
}
