// bench_lqr_loop.cpp -- what ONE Newton iteration of SolverProxDDP costs through the drop-in seam.
//
// BASELINE.json configs[4] (bench/talos-walk.cpp) needs Pinocchio; SURVEY section 8(d) prescribes the
// substitute: bench/lqr.cpp's pure-LQR ProxDDP loop (/root/reference/bench/lqr.cpp:25-57, max_iters = 2)
// restated on the host boundary.  Per iteration the reference does
//   updateLQSubproblem()                  rewrite every knot of the caller's LqrProblem on the host
//                                         (solvers/proxddp/solver-proxddp.hxx:734-805)
//   linear_solver_->backward(mu)          (:608)
//   linear_solver_->forward(dxs,dus,dvs,dlams)   (:610-611)
//   linear_solver_->collapseFeedback(); getFeedforward(i), getFeedback(i) for EVERY i   (:619-632)
// and that is what is timed here, through include/gar_hip.hpp (the C++ host mirror of the
// reference's classes over the C ABI): upload of N+1 knots (pinned staging, one H2D), the sweep, one
// device-side gather + one D2H for all gains, one D2H for the solution.  Beside it: the oracle
// (restated reference, one thread) on the same problem.  Shapes: bench/lqr.cpp's own (dim 56, nu 22:
// no specialised kernels yet) and the north star (36, 12), N = 256, serial and in leg mode (leg count: gar_hip_suggest_num_legs).
//
// build: make -C tests/cpp bench   (links libgar_hip.so and the oracle: test infrastructure)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "gar_hip.hpp"
#ifndef SEAM_NO_ORACLE
extern "C" {
#include "../../oracle/gar_oracle.h"
}
#endif

using namespace aligator_hip::gar;
using clk = std::chrono::steady_clock;

// bench/lqr.cpp:25-57: A = I, B = [I; 0], c = 0.1, w_x = I with w_x(0,0) = 2, w_u = 1e-2 I,
// terminal cost = the running cost's state part, x0 random; the LQ sub-problem ProxDDP hands to gar
static LqrProblem define_problem(int nsteps, int dim, int nu, unsigned seed) {
  std::vector<LqrKnot> knots;
  for (int t = 0; t <= nsteps; ++t) {
    LqrKnot k((uint)dim, t < nsteps ? (uint)nu : 0u, 0);
    for (int i = 0; i < dim; ++i) {
      k.Q(i, i) = (i == 0) ? 2.0 : 1.0;
      k.A(i, i) = 1.0;
      k.f[(size_t)i] = 0.1;
    }
    if (t < nsteps)
      for (int i = 0; i < nu; ++i) {
        k.R(i, i) = 1e-2;
        k.B(i, i) = 1.0;
      }
    knots.push_back(k);
  }
  LqrProblem p(knots, (uint)dim);
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  for (int i = 0; i < dim; ++i) {
    p.G0(i, i) = -1.0;
    p.g0[(size_t)i] = u(rng);
  }
  return p;
}

// the in-place rewrite of every knot a Newton iteration starts with (cost gradients change)
static void update_lq_subproblem(LqrProblem &p, int iter) {
  for (LqrKnot &k : p.stages) {
    for (double &v : k.q)
      v = 1e-3 * (iter + 1);
    for (double &v : k.r)
      v = -1e-3 * (iter + 1);
  }
}

#ifndef SEAM_NO_ORACLE
static ora_problem *to_oracle(const LqrProblem &p) {
  const int N = p.horizon();
  std::vector<int> d5;
  for (const LqrKnot &k : p.stages) {
    const int d[5] = {(int)k.nx, (int)k.nu, (int)k.nc, (int)k.nx2, (int)k.nth};
    d5.insert(d5.end(), d, d + 5);
  }
  ora_problem *o = ora_problem_new(N, d5.data(), (int)p.nc0());
  std::copy(p.G0.data(), p.G0.data() + p.nc0() * p.stages[0].nx, o->G0);
  std::copy(p.g0.begin(), p.g0.end(), o->g0);
  return o;
}
static void sync_oracle(const LqrProblem &p, ora_problem *o) { // the oracle re-reads the knots too
  for (size_t t = 0; t < p.stages.size(); ++t) {
    const LqrKnot &k = p.stages[t];
    ora_knot &q = o->stages[t];
    std::copy(k.Q.data(), k.Q.data() + k.nx * k.nx, q.Q);
    std::copy(k.q.begin(), k.q.end(), q.q);
    std::copy(k.A.data(), k.A.data() + k.nx2 * k.nx, q.A);
    std::copy(k.f.begin(), k.f.end(), q.f);
    if (k.nu > 0) {
      std::copy(k.S.data(), k.S.data() + k.nx * k.nu, q.S);
      std::copy(k.R.data(), k.R.data() + k.nu * k.nu, q.R);
      std::copy(k.r.begin(), k.r.end(), q.r);
      std::copy(k.B.data(), k.B.data() + k.nx2 * k.nu, q.B);
    }
  }
}

#endif

// ---- the same iteration, phase by phase (the `seam` object of bench.py's line) -------------------------------------
// The call sequence of include/aligator/gar/hip-riccati.hpp (the binding a maintainer adds to aligator), written
// out over the C ABI with a host clock around every call.  Gains are consumed the way solver-proxddp.hxx:619-632
// consumes them: copied, stage by stage, from the solver-owned views into the caller's own storage.
struct SeamPhases {
  double pack = 0, backward = 0, forward = 0, sol = 0, collapse = 0, gains = 0, consume = 0, total = 0; // us, host clock
  double dev_bwd = 0, dev_cond = 0, dev_fwd = 0;                                                        // ms, HIP events
  double h2d_bytes = 0, d2h_bytes = 0;
};
static SeamPhases time_phases_once(LqrProblem &p, int num_legs, int iters, double mu, std::string *kernel);
// best of three runs of `iters` iterations, each on a solver of its own: one run lasts about twenty milliseconds, and a
// host that is disturbed for that long (one bench line of round 6 read 1 330 us with normal device times, the run
// before and the run after it 769 and 796) would otherwise be the figure
static SeamPhases time_phases(LqrProblem &p, int num_legs, int iters, double mu, std::string *kernel) {
  SeamPhases best = time_phases_once(p, num_legs, iters, mu, kernel);
  for (int rep = 1; rep < 3; ++rep) {
    const SeamPhases q = time_phases_once(p, num_legs, iters, mu, kernel);
    if (q.total < best.total)
      best = q;
  }
  return best;
}
static SeamPhases time_phases_once(LqrProblem &p, int num_legs, int iters, double mu, std::string *kernel) {
  const int N = p.horizon();
  std::vector<int32_t> dims5;
  for (const LqrKnot &k : p.stages) {
    const int32_t d[5] = {(int)k.nx, (int)k.nu, (int)k.nc, (int)k.nx2, 0};
    dims5.insert(dims5.end(), d, d + 5);
  }
  gar_hip_solver *h = gar_hip_solver_create(0, N, dims5.data(), (int)p.nc0(), 1, num_legs);
  if (!h)
    throw std::runtime_error(gar_hip_last_error());
  *kernel = gar_hip_kernel_name(h);
  auto check = [](int rc) {
    if (rc != GAR_HIP_OK)
      throw std::runtime_error(gar_hip_last_error());
  };
  check(gar_hip_fetch_results(h, 0, 0));
  check(gar_hip_set_timing(h, 1));
  int64_t offs[3], gd[2];
  const double *base = gar_hip_host_results(h, offs);
  check(gar_hip_gains_doubles(h, gd));
  std::vector<double> my_gains((size_t)(gd[0] + gd[1])), my_sol((size_t)gar_hip_solution_doubles(h));
  std::vector<const double *> blocks(16 * (size_t)(N + 1));
  SeamPhases best;
  best.total = 1e30;
  auto us_since = [](clk::time_point t0) { return std::chrono::duration<double, std::micro>(clk::now() - t0).count(); };
  for (int it = 0; it < iters + 2; ++it) {
    update_lq_subproblem(p, it);
    SeamPhases ph;
    const auto t0 = clk::now();
    auto t = t0;
    for (int s = 0; s <= N; ++s) {
      const LqrKnot &k = p.stages[(size_t)s];
      const double *b16[16] = {k.Q.data(), k.S.data(), k.R.data(), k.q.data(), k.r.data(), k.A.data(), k.B.data(), k.f.data(),
                               k.C.data(), k.D.data(), k.d.data(), nullptr, nullptr, nullptr, nullptr, nullptr};
      std::copy(b16, b16 + 16, blocks.begin() + 16 * (size_t)s);
    }
    ph.pack = us_since(t);
    t = clk::now();
    check(gar_hip_backward_blocks(h, blocks.data(), p.G0.data(), p.g0.data(), mu)); // upload + sweep, one call
    check(gar_hip_prefetch_gains(h, 0)); // (the binding's backward(): the gains start travelling under forward())
    ph.backward = us_since(t);
    t = clk::now();
    check(gar_hip_forward(h, nullptr));
    ph.forward = us_since(t);
    t = clk::now();
    check(gar_hip_fetch_results(h, 0, 1));
    std::copy(base + offs[0], base + offs[0] + (int64_t)my_sol.size(), my_sol.begin());
    ph.sol = us_since(t);
    t = clk::now();
    check(gar_hip_collapse_feedback(h));
    ph.collapse = us_since(t);
    t = clk::now();
    check(gar_hip_fetch_results(h, 0, 2));
    ph.gains = us_since(t);
    t = clk::now();
    std::copy(base + offs[1], base + offs[1] + (int64_t)my_gains.size(), my_gains.begin());
    ph.consume = us_since(t);
    ph.total = us_since(t0);
    double km[3];
    check(gar_hip_last_kernel_ms(h, km));
    ph.dev_bwd = km[0], ph.dev_cond = km[1], ph.dev_fwd = km[2];
    if (it >= 2 && ph.total < best.total)
      best = ph;
  }
  best.h2d_bytes = 8.0 * (double)gar_hip_problem_doubles(h);
  best.d2h_bytes = 8.0 * (double)(my_gains.size() + my_sol.size());
  gar_hip_solver_destroy(h);
  return best;
}
static void print_phases_json(const char *name, const SeamPhases &q, const std::string &kernel, int legs, bool last) {
  std::printf("\"%s\": {\"kernel\": \"%s\", \"legs\": %d, \"us_per_newton_iteration\": %.1f, \"host_us\": {\"block_pointer_table\": %.1f, "
              "\"backward_blocks_pack_h2d_sweep_status_sync\": %.1f, \"forward\": %.1f, \"solution_d2h_and_scatter\": %.1f, "
              "\"collapse_feedback\": %.1f, \"gains_gather_and_d2h\": %.1f, \"gains_consumed_by_caller\": %.1f}, "
              "\"device_ms\": {\"backward_sweep\": %.4f, \"condensed_or_initial\": %.4f, \"forward_sweep\": %.4f}, "
              "\"h2d_bytes\": %.0f, \"d2h_bytes\": %.0f}%s",
              name, kernel.c_str(), legs, q.total, q.pack, q.backward, q.forward, q.sol, q.collapse, q.gains, q.consume, q.dev_bwd,
              q.dev_cond, q.dev_fwd, q.h2d_bytes, q.d2h_bytes, last ? "" : ", ");
}

template <class Solver> static double time_loop(Solver &solver, LqrProblem &p, int iters, double mu, double *checksum) {
  auto sol4 = lqrInitializeSolution(p);
  struct { VectorOfVectors &xs, &us, &vs, &lbdas; } sol{sol4[0], sol4[1], sol4[2], sol4[3]};
  const size_t N = (size_t)p.horizon();
  std::vector<VectorXs> ffs(N + 1);
  std::vector<Matrix> fbs(N + 1);
  double best = 1e30;
  for (int it = 0; it < iters + 2; ++it) { // two warm-up iterations
    update_lq_subproblem(p, it);
    const auto t0 = clk::now();
    solver.backward(mu);
    solver.forward(sol.xs, sol.us, sol.vs, sol.lbdas);
    solver.collapseFeedback();
    for (size_t i = 0; i <= N; ++i) {
      ffs[i] = solver.getFeedforward(i);
      fbs[i] = solver.getFeedback(i);
    }
    const double us = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
    if (it >= 2 && us < best)
      best = us;
  }
  double c = 0.0;
  for (const auto &x : sol.xs)
    for (double v : x)
      c += v;
  c += fbs[0](0, 0) + ffs[N / 2][0];
  *checksum = c;
  return best;
}

int main(int argc, char **argv) {
  if (gar_hip_device_count() <= 0) {
    std::printf("bench_lqr_loop: no HIP device (the backend has no CPU fallback)\n");
    return 77;
  }
  const bool json = argc > 1 && std::string(argv[1]) == "--json";
  const int json_legs = (json && argc > 2) ? std::atoi(argv[2]) : 0; // --json [legs]: the leg count (default: gar_hip_suggest_num_legs)
  const int N = (!json && argc > 1) ? std::atoi(argv[1]) : 256, iters = 20;
  const double mu = 1e-10; // bench/lqr.cpp: mu_init = 1e-10
  struct Shape { int dim, nu; const char *what; };
  const Shape shapes[] = {{36, 12, "north star (36, 12)"}, {56, 22, "bench/lqr.cpp (56, 22)"}};
  if (json) { // one JSON object: per shape, serial and in leg mode, phase by phase
    std::printf("{\"workload\": \"one Newton iteration of bench/lqr.cpp's ProxDDP loop through the RiccatiSolverBase seam "
                "(upload of %d knots, backward, forward, collapseFeedback, every stage's gains), N=%d, one problem\", ", N + 1, N);
    for (size_t i = 0; i < 2; ++i) {
      const Shape &sh = shapes[i];
      LqrProblem p = define_problem(N, sh.dim, sh.nu, 42), pp = define_problem(N, sh.dim, sh.nu, 42);
      std::string k1, k2;
      const int legs = json_legs > 1 ? json_legs : gar_hip_suggest_num_legs(N, sh.dim, sh.nu); // the library's table
      const SeamPhases a = time_phases(p, 1, iters, mu, &k1), b = time_phases(pp, legs, iters, mu, &k2);
      std::printf("\"nx%d_nu%d\": {", sh.dim, sh.nu);
      print_phases_json("serial", a, k1, 1, false);
#ifndef SEAM_NO_ORACLE
      print_phases_json("legs", b, k2, legs, false);
      { // the same iteration on this box's host, one thread: the restated reference (oracle/gar_oracle.c) re-reading
        // the knots, backward, forward -- the CPU figure to read beside us_per_newton_iteration
        ora_problem *op = to_oracle(p);
        ora_prox_solver *os = ora_prox_new(op);
        auto sol4 = lqrInitializeSolution(p);
        std::vector<double *> xs, us, vs, ls;
        for (auto &v : sol4[0]) xs.push_back(v.data());
        for (auto &v : sol4[1]) us.push_back(v.data());
        us.resize(xs.size(), nullptr);
        for (auto &v : sol4[2]) vs.push_back(v.data());
        for (auto &v : sol4[3]) ls.push_back(v.data());
        double bw = 1e30, it_us = 1e30;
        for (int it = 0; it < 5; ++it) {
          update_lq_subproblem(p, it);
          const auto t0 = clk::now();
          sync_oracle(p, op);
          ora_prox_backward(os, mu);
          const auto t1 = clk::now();
          ora_prox_forward(os, xs.data(), us.data(), vs.data(), ls.data(), nullptr);
          const auto t2 = clk::now();
          bw = std::min(bw, std::chrono::duration<double, std::micro>(t1 - t0).count());
          it_us = std::min(it_us, std::chrono::duration<double, std::micro>(t2 - t0).count());
        }
        ora_prox_free(os);
        ora_problem_free(op);
        std::printf("\"cpu_restated_reference_1_thread\": {\"us_per_newton_iteration\": %.1f, \"backward_us\": %.1f, "
                    "\"what\": \"oracle/gar_oracle.c (the reference's algorithm restated in C, -O3) on this box's host, one thread: "
                    "knots re-read, backward, forward; the gains are already on the host\"}", it_us, bw);
      }
#else
      print_phases_json("legs", b, k2, legs, true);
#endif
      std::printf("}%s", i == 0 ? ", " : "");
    }
    std::printf("}\n");
    return 0;
  }
  for (const Shape &sh : shapes) {
    LqrProblem p = define_problem(N, sh.dim, sh.nu, 42);
    double ora_us = 0.0;
#ifndef SEAM_NO_ORACLE
    // the oracle, one thread (the reference's BM_lqr_prox<SERIAL> role)
    ora_problem *op = to_oracle(p);
    ora_prox_solver *os = ora_prox_new(op);
    ora_us = 1e30;
    for (int it = 0; it < 5; ++it) {
      update_lq_subproblem(p, it);
      const auto t0 = clk::now();
      sync_oracle(p, op);
      ora_prox_backward(os, mu);
      const double us = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
      if (us < ora_us)
        ora_us = us;
    }
    ora_prox_free(os);
    ora_problem_free(op);
#endif
    double cs = 0.0, cp = 0.0;
    {
      ProximalRiccatiSolver s(p);
      const double us = time_loop(s, p, iters, mu, &cs);
      std::printf("%-24s N=%d  serial      %9.1f us / Newton iteration  (kernel %s)   oracle backward alone, 1 thread: %.0f us\n",
                  sh.what, N, us, s.kernelName(), ora_us);
    }
    const unsigned legs = (unsigned)(N / 8);
    try {
      LqrProblem pp = define_problem(N, sh.dim, sh.nu, 42);
      ParallelRiccatiSolver s(pp, legs);
      const double us = time_loop(s, pp, iters, mu, &cp);
      std::printf("%-24s N=%d  %3u legs    %9.1f us / Newton iteration  (kernel %s)   |checksum serial - legs| = %.2e\n",
                  sh.what, N, legs, us, s.kernelName(), std::fabs(cs - cp));
    } catch (const std::exception &e) {
      std::printf("%-24s N=%d  %3u legs    refused: %s\n", sh.what, N, legs, e.what());
    }
  }
  return 0;
}
