// C++ parity tests of the HIP gar backend through include/gar_hip.hpp, written the way the
// reference's own tests are (tests/gar/riccati.cpp, tests/gar/parallel.cpp): build a problem,
// `solver.backward(mu)`, `lqrInitializeSolution`, `solver.forward(...)`, check the KKT residual
// against the reference's thresholds.  No test framework (Catch2 is not in this image): a failed
// REQUIRE prints the line and makes the program exit 1.  Exit code 77 = no HIP device (the
// backend has no CPU path; the product must fail loudly, and it does: the constructor throws).
#include <gar_hip.hpp>

#include <cstdio>
#include <cstdlib>
#include <random>

using namespace aligator_hip::gar;

static int g_failed = 0;
// GAR_TEST_SMALL=1: the sizes the CPU emulator build (tests/emu) gets through in seconds
static const bool g_small = std::getenv("GAR_TEST_SMALL") != nullptr;
#define REQUIRE(cond)                                                          \
  do {                                                                         \
    if (!(cond)) {                                                             \
      std::printf("  REQUIRE failed at line %d: %s\n", __LINE__, #cond);       \
      ++g_failed;                                                              \
    }                                                                          \
  } while (0)

// a well-conditioned random knot (same distributions as aligator_amd/synth.py, generator "W",
// which restates tests/gar/test_util.cpp:14-76 with contractive dynamics)
static LqrKnot generate_knot(std::mt19937 &rng, uint nx, uint nu, uint nc = 0) {
  std::normal_distribution<double> n01(0.0, 1.0);
  std::uniform_real_distribution<double> u11(-1.0, 1.0);
  LqrKnot k(nx, nu, nc);
  const uint nw = nx + nu;
  Matrix root((int)nw, (int)nw + 1);
  for (double &v : root.v)
    v = n01(rng);
  auto W = [&](uint i, uint j) {
    double s = 0;
    for (uint c = 0; c <= nw; ++c)
      s += root((int)i, (int)c) * root((int)j, (int)c);
    return s / std::max(nx, std::max(nu, 1u));
  };
  for (uint i = 0; i < nx; ++i)
    for (uint j = 0; j < nx; ++j)
      k.Q((int)i, (int)j) = W(i, j);
  for (uint i = 0; i < nx; ++i)
    for (uint j = 0; j < nu; ++j)
      k.S((int)i, (int)j) = W(i, nx + j);
  for (uint i = 0; i < nu; ++i)
    for (uint j = 0; j < nu; ++j)
      k.R((int)i, (int)j) = W(nx + i, nx + j) * (i == j ? 1.0 + 1e-6 : 1.0);
  for (double &v : k.q) v = u11(rng);
  for (double &v : k.r) v = u11(rng);
  k.A.setIdentity();
  for (double &v : k.A.v) v += 0.1 * u11(rng);
  for (double &v : k.B.v) v = 0.5 * u11(rng);
  for (double &v : k.f) v = n01(rng);
  for (double &v : k.C.v) v = u11(rng);
  for (double &v : k.D.v) v = u11(rng);
  for (double &v : k.d) v = u11(rng);
  return k;
}

static LqrProblem generate_problem(std::mt19937 &rng, const VectorXs &x0, uint horz, uint nx, uint nu) {
  LqrProblem::KnotVector knots;
  for (uint t = 0; t < horz; ++t)
    knots.push_back(generate_knot(rng, nx, nu));
  knots.push_back(generate_knot(rng, nx, 0)); // terminal knot: no controls
  LqrProblem prob(std::move(knots), nx);
  prob.G0.setIdentity();
  for (double &v : prob.G0.v) v = -v;
  prob.g0 = x0; // G0 x0 + g0 = 0
  return prob;
}

static double maxdiff(const VectorOfVectors &a, const VectorOfVectors &b) {
  double m = 0;
  for (size_t t = 0; t < a.size(); ++t)
    for (size_t i = 0; i < a[t].size(); ++i)
      m = std::max(m, std::fabs(a[t][i] - b[t][i]));
  return m;
}

// tests/gar/riccati.cpp:24-86
static void riccati_short_horz_pb(uint horz) {
  std::printf("riccati_short_horz_pb horz=%u\n", horz);
  const double mueq = 1e-14;
  const uint nx = 2, nu = 2;
  std::mt19937 rng(42);
  std::uniform_real_distribution<double> u11(-1.0, 1.0);
  auto init_knot = [&](uint nc) {
    LqrKnot knot(nx, nu, nc);
    knot.A(0, 0) = 0.1; knot.A(0, 1) = 0.0; knot.A(1, 0) = -0.1; knot.A(1, 1) = 0.01;
    for (double &v : knot.B.v) v = u11(rng);
    for (double &v : knot.f) v = u11(rng);
    knot.Q.setIdentity();
    for (double &v : knot.Q.v) v *= 0.01;
    knot.R.setIdentity();
    for (double &v : knot.R.v) v *= 0.1;
    return knot;
  };
  LqrKnot base_knot = init_knot(0u);
  LqrKnot knot1 = base_knot;
  knot1.Q.setIdentity();
  knot1.q = {1.0, 1.0}; // -x1, x1 = -ones
  LqrProblem::KnotVector knots(horz + 1, base_knot);
  knots[4] = init_knot(nu);
  knots[4].D.setIdentity();
  knots[4].d.assign(nu, 0.1);
  knots[horz] = knot1;
  LqrProblem prob(std::move(knots), nx);
  prob.g0 = {-1.0, -1.0};
  prob.G0.setIdentity();
  ProximalRiccatiSolver solver{prob};
  REQUIRE(solver.backward(mueq));
  auto [xs, us, vs, lbdas] = lqrInitializeSolution(prob);
  REQUIRE(xs.size() == size_t(prob.horizon()) + 1);
  REQUIRE(vs.size() == size_t(prob.horizon()) + 1);
  REQUIRE(lbdas.size() == size_t(prob.horizon()) + 1);
  REQUIRE(solver.forward(xs, us, vs, lbdas));
  KktError err = lqrComputeKktError(prob, xs, us, vs, lbdas);
  std::printf("  kkt: dyn %.2e cstr %.2e dual %.2e\n", err.dyn, err.cstr, err.dual);
  REQUIRE(err.max <= 1e-9);
}

// tests/gar/riccati.cpp:88-105
static void riccati_one_knot_prob() {
  std::printf("riccati_one_knot_prob\n");
  std::mt19937 rng(7);
  auto problem = generate_problem(rng, VectorXs(2, 0.0), 0, 2, 2);
  ProximalRiccatiSolver solver(problem);
  auto [xs, us, vs, lbdas] = lqrInitializeSolution(problem);
  REQUIRE(xs.size() == 1);
  REQUIRE(us.size() == 0);
  REQUIRE(lbdas.size() == 1);
  solver.backward(1e-13);
  solver.forward(xs, us, vs, lbdas);
  REQUIRE(lqrComputeKktError(problem, xs, us, vs, lbdas).max <= 1e-10);
}

// tests/gar/riccati.cpp:107-139 (nx = 36, nu = 12)
static void riccati_random_large_problem() {
  std::printf("riccati_random_large_problem\n");
  std::mt19937 rng(11);
  const uint nx = 36, nu = 12, horz = g_small ? 6 : 100;
  VectorXs x0(nx, 1.0);
  auto prob = generate_problem(rng, x0, horz, nx, nu);
  ProximalRiccatiSolver solver{prob};
  const double mu = 1e-14;
  solver.backward(mu);
  auto [xs, us, vs, lbdas] = lqrInitializeSolution(prob);
  solver.forward(xs, us, vs, lbdas);
  KktError err = lqrComputeKktError(prob, xs, us, vs, lbdas, mu);
  std::printf("  kernel %s  kkt: dyn %.2e cstr %.2e dual %.2e\n", solver.kernelName(), err.dyn,
              err.cstr, err.dual);
  REQUIRE(err.max <= 1e-9);
  REQUIRE(solver.getFeedback(0).rows == (int)(nu + nx));
  REQUIRE(solver.getFeedforward(0).size() == nu + nx);
}

// tests/gar/parallel.cpp:185-245
static void parallel_solver_class(uint num_threads) {
  std::printf("parallel_solver_class threads=%u\n", num_threads);
  std::mt19937 rng(13);
  const uint nx = g_small ? 8 : 36, nu = g_small ? 4 : 12, horz = g_small ? 11 : 96;
  VectorXs x0(nx, 0.5);
  auto problem = generate_problem(rng, x0, horz, nx, nu);
  const double mu = 1e-12;
  ProximalRiccatiSolver refSolver{problem};
  refSolver.backward(mu);
  auto [xs_ref, us_ref, vs_ref, lbdas_ref] = lqrInitializeSolution(problem);
  refSolver.forward(xs_ref, us_ref, vs_ref, lbdas_ref);

  ParallelRiccatiSolver parSolver(problem, num_threads);
  parSolver.backward(mu);
  auto [xs, us, vs, lbdas] = lqrInitializeSolution(problem);
  parSolver.forward(xs, us, vs, lbdas);
  KktError err = lqrComputeKktError(problem, xs, us, vs, lbdas, mu);
  std::printf("  kernel %s  kkt max %.2e  |x - x_ref| %.2e  |lbda - lbda_ref| %.2e\n",
              parSolver.kernelName(), err.max, maxdiff(xs, xs_ref), maxdiff(lbdas, lbdas_ref));
  REQUIRE(err.max <= 1e-9);
  REQUIRE(maxdiff(xs, xs_ref) <= 1e-9);
  REQUIRE(maxdiff(lbdas, lbdas_ref) <= 1e-7);
  parSolver.collapseFeedback();
}

// BASELINE.json configs[2] shape (nx = 12, nu = 6) runs on the (12, 8) kernels with two dummy
// controls; (10, 3) on the (12, 4) kernels with two dummy states and one dummy control.  Neither
// may show in the gains or the solution.
static void padded_shape(uint nx, uint nu, const char *kernel) {
  std::printf("padded_shape (nx=%u, nu=%u)\n", nx, nu);
  std::mt19937 rng(17);
  const uint horz = g_small ? 6 : 64;
  auto problem = generate_problem(rng, VectorXs(nx, 0.3), horz, nx, nu);
  ProximalRiccatiSolver solver{problem};
  solver.backward(1e-12);
  auto [xs, us, vs, lbdas] = lqrInitializeSolution(problem);
  solver.forward(xs, us, vs, lbdas);
  REQUIRE(us[0].size() == nu);
  REQUIRE(lqrComputeKktError(problem, xs, us, vs, lbdas, 1e-12).max <= 1e-9);
  REQUIRE(solver.getFeedback(0).rows == (int)(nu + nx));
  REQUIRE(solver.getFeedback(0).cols == (int)nx);
  REQUIRE(solver.getFeedforward(0).size() == nu + nx);
  REQUIRE(std::string(solver.kernelName()).find(kernel) != std::string::npos);
  ParallelRiccatiSolver par(problem, 3);
  par.backward(1e-12);
  auto [xp, up, vp, lp] = lqrInitializeSolution(problem);
  par.forward(xp, up, vp, lp);
  std::printf("  kernels %s / %s  |x_par - x_serial| %.2e\n", solver.kernelName(), par.kernelName(),
              maxdiff(xp, xs));
  REQUIRE(maxdiff(xp, xs) <= 1e-9);
  REQUIRE(maxdiff(up, us) <= 1e-9);
  REQUIRE(maxdiff(lp, lbdas) <= 1e-8);
}

// The raw C ABI with the CALLER's dimensions (what the HipRiccatiSolver of INTEGRATION.md passes: knot.nx, knot.nu
// as they are): kernel selection and padding happen inside gar_hip_solver_create -- the Talos walk's (56, 22) must
// reach pair<56,24>, every result must come back in the caller's shapes, and the solution must satisfy the
// reference's residual (lqrComputeKktError, gar/utils.hxx:88-182) and equal the stage-dense solver's (which never
// pads) -- two kernel families, two algorithms, one answer.
static void raw_c_abi_wide_shape() {
  const uint nx = 56, nu = 22, horz = g_small ? 3 : 275;
  std::printf("raw_c_abi_wide_shape (nx=%u, nu=%u, N=%u)\n", nx, nu, horz);
  std::mt19937 rng(23);
  auto problem = generate_problem(rng, VectorXs(nx, 0.1), horz, nx, nu);
  std::vector<int32_t> dims5;
  for (const LqrKnot &k : problem.stages) {
    const int32_t d[5] = {(int)k.nx, (int)k.nu, (int)k.nc, (int)k.nx2, (int)k.nth};
    dims5.insert(dims5.end(), d, d + 5);
  }
  gar_hip_solver *h = gar_hip_solver_create(0, (int)horz, dims5.data(), (int)problem.nc0(), 1, 1);
  REQUIRE(h != nullptr);
  if (!h)
    return;
  std::printf("  kernel %s\n", gar_hip_kernel_name(h));
  REQUIRE(std::string(gar_hip_kernel_name(h)) == "pair<56,24>");
  for (int t = 0; t <= (int)horz; ++t) {
    const LqrKnot &k = problem.stages[(size_t)t];
    REQUIRE(gar_hip_upload_stage(h, 0, t, k.Q.data(), k.S.data(), k.R.data(), k.q.data(), k.r.data(), k.A.data(),
                                 k.B.data(), k.f.data(), k.C.data(), k.D.data(), k.d.data(), k.Gth.data(), k.Gx.data(),
                                 k.Gu.data(), k.Gv.data(), k.gamma.data()) == GAR_HIP_OK);
  }
  REQUIRE(gar_hip_set_init(h, 0, problem.G0.data(), problem.g0.data()) == GAR_HIP_OK);
  REQUIRE(gar_hip_backward(h, 1e-12) == GAR_HIP_OK);
  REQUIRE(gar_hip_forward(h, nullptr) == GAR_HIP_OK);
  auto [xs, us, vs, lbdas] = lqrInitializeSolution(problem);
  std::vector<double> X((size_t)(horz + 1) * nx), U((size_t)horz * nu), Lb((size_t)problem.nc0() + (size_t)horz * nx);
  REQUIRE(gar_hip_get_solution(h, 0, X.data(), U.data(), nullptr, Lb.data()) == GAR_HIP_OK);
  for (size_t t = 0; t <= horz; ++t)
    std::copy(X.begin() + (long)(t * nx), X.begin() + (long)((t + 1) * nx), xs[t].begin());
  for (size_t t = 0; t < horz; ++t)
    std::copy(U.begin() + (long)(t * nu), U.begin() + (long)((t + 1) * nu), us[t].begin());
  std::copy(Lb.begin(), Lb.begin() + (long)problem.nc0(), lbdas[0].begin());
  for (size_t t = 1; t <= horz; ++t)
    std::copy(Lb.begin() + (long)(problem.nc0() + (t - 1) * nx), Lb.begin() + (long)(problem.nc0() + t * nx), lbdas[t].begin());
  const double kkt = lqrComputeKktError(problem, xs, us, vs, lbdas, 1e-12).max;
  std::printf("  kkt %.2e\n", kkt);
  REQUIRE(kkt <= 1e-9);
  int64_t gd[2];
  REQUIRE(gar_hip_gains_doubles(h, gd) == GAR_HIP_OK);
  REQUIRE(gd[0] == (int64_t)horz * (nu + nx) + nx && gd[1] == (int64_t)horz * (nu + nx) * nx + (int64_t)nx * nx);
  std::vector<double> ff(nu + nx), fb((size_t)(nu + nx) * nx);
  REQUIRE(gar_hip_get_gains(h, 0, 0, ff.data(), fb.data(), nullptr) == GAR_HIP_OK);
  // u0 = kff + K x0 with the returned (22 x 56) gains reproduces the returned control
  double worst = 0.0;
  for (uint i = 0; i < nu; ++i) {
    double u = ff[i];
    for (uint j = 0; j < nx; ++j)
      u += fb[(size_t)i * nx + j] * xs[0][j];
    worst = std::max(worst, std::abs(u - us[0][i]));
  }
  REQUIRE(worst <= 1e-10);
  gar_hip_solver_destroy(h);
  if (g_small) { // and against the stage-dense solver, which runs at the caller's own dimensions
    RiccatiSolverDense dense{problem};
    dense.backward(1e-12);
    auto [xd, ud, vd, ld] = lqrInitializeSolution(problem);
    dense.forward(xd, ud, vd, ld);
    REQUIRE(maxdiff(xd, xs) <= 1e-9);
    REQUIRE(maxdiff(ud, us) <= 1e-9);
  }
}

// The pipelined sweep (gar_hip_set_pipeline, round 5) through the raw C ABI: a batch of three problems, plain then
// pipelined (two call pairs in a row), solutions bit for bit; and the terminal knot as SolverProxDDP builds it
// (nx2 = 0, solvers/proxddp/workspace.hxx:54-55): null A / f for that knot, the specialised kernels all the same, the
// nx2 = nx problem's solution
static void raw_c_abi_pipeline_and_terminal_knot() {
  const uint nx = 8, nu = 4, horz = 6;
  const int batch = 3;
  std::printf("raw_c_abi_pipeline_and_terminal_knot (nx=%u, nu=%u, N=%u, batch %d)\n", nx, nu, horz, batch);
  std::vector<LqrProblem> probs;
  for (int b = 0; b < batch; ++b) {
    std::mt19937 rng(100 + (unsigned)b);
    probs.push_back(generate_problem(rng, VectorXs(nx, 0.3 * (b + 1)), horz, nx, nu));
  }
  std::vector<int32_t> dims5;
  for (const LqrKnot &k : probs[0].stages) {
    const int32_t d[5] = {(int)k.nx, (int)k.nu, (int)k.nc, (int)k.nx2, (int)k.nth};
    dims5.insert(dims5.end(), d, d + 5);
  }
  std::vector<int32_t> dims_t0 = dims5;
  dims_t0[5 * horz + 3] = 0; // the terminal knot with nx2 = 0
  const size_t nX = (size_t)(horz + 1) * nx, nU = (size_t)horz * nu, nL = (size_t)nx + (size_t)horz * nx;
  auto solve = [&](const std::vector<int32_t> &d5, bool null_terminal, int pipeline, std::vector<double> &out) {
    gar_hip_solver *h = gar_hip_solver_create(0, (int)horz, d5.data(), (int)nx, batch, 1);
    REQUIRE(h != nullptr);
    if (!h)
      return;
    REQUIRE(std::string(gar_hip_kernel_name(h)).find("<8,4>") != std::string::npos);
    for (int b = 0; b < batch; ++b) {
      for (int t = 0; t <= (int)horz; ++t) {
        const LqrKnot &k = probs[(size_t)b].stages[(size_t)t];
        const bool nt = null_terminal && t == (int)horz;
        REQUIRE(gar_hip_upload_stage(h, b, t, k.Q.data(), k.S.data(), k.R.data(), k.q.data(), k.r.data(),
                                     nt ? nullptr : k.A.data(), k.B.data(), nt ? nullptr : k.f.data(), k.C.data(), k.D.data(),
                                     k.d.data(), k.Gth.data(), k.Gx.data(), k.Gu.data(), k.Gv.data(), k.gamma.data()) == GAR_HIP_OK);
      }
      REQUIRE(gar_hip_set_init(h, b, probs[(size_t)b].G0.data(), probs[(size_t)b].g0.data()) == GAR_HIP_OK);
    }
    const int rc = gar_hip_set_pipeline(h, pipeline);
    if (pipeline == 2 && batch >= 2 && std::string(gar_hip_kernel_name(h)) == "wave<8,4>")
      REQUIRE(rc == GAR_HIP_OK && gar_hip_pipeline(h) == 2);
    for (int rep = 0; rep < 2; ++rep) {
      REQUIRE(gar_hip_backward_async(h, 1e-12) == GAR_HIP_OK);
      REQUIRE(gar_hip_forward_async(h, nullptr) == GAR_HIP_OK);
    }
    REQUIRE(gar_hip_sync(h) == GAR_HIP_OK);
    REQUIRE(gar_hip_num_failed(h) == 0);
    out.assign((size_t)batch * (nX + nU + nL), 0.0);
    for (int b = 0; b < batch; ++b) {
      double *o = out.data() + (size_t)b * (nX + nU + nL);
      REQUIRE(gar_hip_get_solution(h, b, o, o + nX, nullptr, o + nX + nU) == GAR_HIP_OK);
    }
    if (null_terminal) { // the terminal knot's gains: the caller's 0 rows -- nothing is written
      double guard[2] = {-7.0, -7.0};
      REQUIRE(gar_hip_get_gains(h, 0, (int)horz, guard, guard + 1, nullptr) == GAR_HIP_OK);
      REQUIRE(guard[0] == -7.0 && guard[1] == -7.0);
      int32_t pd[5];
      REQUIRE(gar_hip_packed_stage_dims(h, (int)horz, pd) == GAR_HIP_OK && pd[3] == (int32_t)nx);
    }
    gar_hip_solver_destroy(h);
  };
  std::vector<double> plain, piped, term0;
  solve(dims5, false, 0, plain);
  solve(dims5, false, 2, piped);
  solve(dims_t0, true, 0, term0);
  REQUIRE(plain.size() == piped.size() && plain.size() == term0.size());
  double d1 = 0.0, d2 = 0.0, sc = 1.0;
  for (size_t i = 0; i < plain.size(); ++i) {
    d1 = std::max(d1, std::abs(plain[i] - piped[i]));
    d2 = std::max(d2, std::abs(plain[i] - term0[i]));
    sc = std::max(sc, std::abs(plain[i]));
  }
  std::printf("  |plain - pipelined| %.1e   |nx2 = nx - nx2 = 0| %.1e (scale %.1e)\n", d1, d2, sc);
  REQUIRE(d1 == 0.0);
  REQUIRE(d2 <= 1e-12 * sc);
}

// tests/gar/riccati.cpp:141-155 ("test dense solver"): KKT error <= 1e-8, and the same trajectory as
// the Riccati recursion
static void dense_solver() {
  std::printf("dense_solver\n");
  std::mt19937 rng(42);
  const uint nx = g_small ? 6 : 36, nu = g_small ? 3 : 12, horz = g_small ? 7 : 100;
  auto problem = generate_problem(rng, VectorXs(nx, 0.0), horz, nx, nu);
  const double mueq = 1e-14;
  RiccatiSolverDense denseSolver{problem};
  denseSolver.backward(mueq);
  auto [xsd, usd, vsd, lbdasd] = lqrInitializeSolution(problem);
  denseSolver.forward(xsd, usd, vsd, lbdasd);
  const KktError errd = lqrComputeKktError(problem, xsd, usd, vsd, lbdasd, mueq);
  ProximalRiccatiSolver solver{problem};
  solver.backward(mueq);
  auto [xs, us, vs, lbdas] = lqrInitializeSolution(problem);
  solver.forward(xs, us, vs, lbdas);
  std::printf("  kernel %s  kkt max %.2e  |x - x_riccati| %.2e\n", denseSolver.kernelName(), errd.max,
              maxdiff(xsd, xs));
  REQUIRE(errd.max <= 1e-8);
  REQUIRE(maxdiff(xsd, xs) <= 1e-8);
  REQUIRE(maxdiff(usd, us) <= 1e-8);
  REQUIRE(denseSolver.getFeedback(0).rows == (int)(nu + 2 * nx));
  REQUIRE(denseSolver.getFeedforward(0).size() == nu + 2 * nx);
  REQUIRE(std::string(denseSolver.kernelName()) == "dense");
}

// tests/mpc-cycle.cpp / proximal-riccati.hxx:79-86: cycleAppend rotates the solver's stages (a ring on
// the device: no record moves), the caller rotates its problem and writes the new last-but-one knot;
// the next sweep must solve the rotated problem -- through more cycles than there are stages
static void mpc_cycle() {
  std::printf("mpc_cycle\n");
  std::mt19937 rng(5);
  const uint nx = g_small ? 8 : 36, nu = g_small ? 4 : 12, horz = g_small ? 5 : 12;
  VectorXs x0(nx, 0.5);
  auto prob = generate_problem(rng, x0, horz, nx, nu);
  ProximalRiccatiSolver solver{prob};
  const double mu = 1e-12;
  solver.backward(mu);
  for (uint c = 0; c < horz + 2; ++c) {
    LqrKnot knot = generate_knot(rng, nx, nu);
    for (uint t = 0; t + 1 < horz; ++t)
      prob.stages[t] = prob.stages[t + 1];
    prob.stages[horz - 1] = knot;
    solver.cycleAppend(knot);
    solver.backward(mu);
    auto [xs, us, vs, lbdas] = lqrInitializeSolution(prob);
    solver.forward(xs, us, vs, lbdas);
    KktError err = lqrComputeKktError(prob, xs, us, vs, lbdas, mu);
    REQUIRE(err.max <= 1e-9);
    // a fresh solver on the rotated problem gives the same gains
    ProximalRiccatiSolver fresh{prob};
    fresh.backward(mu);
    const Matrix a = solver.getFeedback(horz / 2), b = fresh.getFeedback(horz / 2);
    double d = 0.0;
    for (int i = 0; i < a.rows; ++i)
      for (int j = 0; j < a.cols; ++j)
        d = std::max(d, std::fabs(a(i, j) - b(i, j)));
    REQUIRE(d == 0.0);
  }
  std::printf("  kernel %s, %u cycles\n", solver.kernelName(), horz + 2);
}

// ParallelRiccatiSolver::cycleAppend (parallel-solver.hxx:246-258: "just reinitialise everything"): in leg mode the
// C ABI rebuilds its layout and buffers; the next sweep must solve the caller's rotated problem like the serial solver
static void mpc_cycle_parallel() {
  std::printf("mpc_cycle_parallel\n");
  std::mt19937 rng(9);
  const uint nx = 8, nu = 4, horz = 11, legs = 3;
  auto prob = generate_problem(rng, VectorXs(nx, 0.0), horz, nx, nu);
  ParallelRiccatiSolver solver{prob, legs};
  const double mu = 1e-10;
  double worst = 0.0;
  solver.backward(mu);
  for (uint c = 0; c < 3; ++c) {
    LqrKnot knot = generate_knot(rng, nx, nu);
    for (uint t = 0; t + 1 < horz; ++t)
      prob.stages[t] = prob.stages[t + 1];
    prob.stages[horz - 1] = knot;
    solver.cycleAppend(knot);
    solver.backward(mu);
    auto [xs, us, vs, lbdas] = lqrInitializeSolution(prob);
    solver.forward(xs, us, vs, lbdas);
    KktError err = lqrComputeKktError(prob, xs, us, vs, lbdas, mu);
    REQUIRE(err.max <= 1e-9);
    ProximalRiccatiSolver serial{prob};
    serial.backward(mu);
    auto [xs2, us2, vs2, lbdas2] = lqrInitializeSolution(prob);
    serial.forward(xs2, us2, vs2, lbdas2);
    REQUIRE(maxdiff(xs, xs2) <= 1e-9);
    REQUIRE(maxdiff(us, us2) <= 1e-9);
    worst = std::max({worst, err.max, maxdiff(xs, xs2), maxdiff(us, us2)});
  }
  std::printf("  kernel %s, 3 cycles, kkt / |x_par - x_serial| <= %.2e\n", solver.kernelName(), worst);
}

static void error_behaviour() {
  std::printf("error_behaviour\n");
  std::mt19937 rng(3);
  auto prob = generate_problem(rng, VectorXs(8, 0.0), 3, 8, 4);
  for (int t = 0; t < prob.horizon(); ++t) {
    prob.stages[t].R.setZero();
    prob.stages[t].S.setZero();
    prob.stages[t].B.setZero();
  }
  bool threw = false;
  try {
    ProximalRiccatiSolver s{prob};
    s.backward(1e-10);
  } catch (const std::runtime_error &e) { // riccati-kernel.hxx:239-241
    threw = std::string(e.what()).find("LDL") != std::string::npos;
  }
  REQUIRE(threw);
  threw = false;
  try {
    ParallelRiccatiSolver s(prob, 1); // parallel-solver.hxx:42-46
  } catch (const std::runtime_error &) {
    threw = true;
  }
  REQUIRE(threw);
}

int main() {
  std::printf("%s, %d HIP device(s)\n", gar_hip_version(), gar_hip_device_count());
  try {
    for (uint horz : {4u, 8u, 16u})
      if (!(g_small && horz == 16u))
        riccati_short_horz_pb(horz);
  } catch (const std::runtime_error &e) {
    if (gar_hip_device_count() == 0) {
      std::printf("no HIP device: %s\n", e.what());
      return 77;
    }
    throw;
  }
  riccati_one_knot_prob();
  riccati_random_large_problem();
  for (uint th : {2u, 4u, 8u})
    if (!(g_small && th == 8u))
      parallel_solver_class(th);
  dense_solver();
  mpc_cycle();
  mpc_cycle_parallel();
  padded_shape(12, 6, "12,8");
  padded_shape(10, 3, "12,4");
  raw_c_abi_wide_shape();
  raw_c_abi_pipeline_and_terminal_knot();
  error_behaviour();
  std::printf(g_failed ? "%d REQUIRE(s) FAILED\n" : "all passed\n", g_failed);
  return g_failed ? 1 : 0;
}
