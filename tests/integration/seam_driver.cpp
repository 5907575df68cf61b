// TEST INFRASTRUCTURE -- the drop-in seam executed with the REFERENCE's own types: the HipRiccatiSolver this
// repository ships as include/aligator/gar/hip-riccati.hpp (the file a maintainer adds to aligator) is
// compiled against /root/reference/include (over the Eigen-API stand-in oracle/ref_shim, as oracle/_ref is) and
// driven through gar::RiccatiSolverBase<double>* exactly as SolverProxDDP drives linear_solver_
// (solver-proxddp.hxx:208, 608-611, 619, 624-625, 631-632), next to the reference's own ProximalRiccatiSolver /
// ParallelRiccatiSolver on the same LqrProblemTpl.  Linked with the wave-emulator build of the library on CPU
// (tests/test_integration_binding.py) -- or, with -DSEAM_GPU, with the REAL aligator_amd/libgar_hip.so: that
// executable is built in the container that has /root/reference (tests/integration/build_gpu_driver.sh, from
// __graft_entry__.build()), travels to the GPU box like oracle/_ref/libgar_ref.so and is run there by the -m gpu test
// of the same file: the reference's own solvers and the MI355X backend, same LqrProblemTpl, same process.
#define ALIGATOR_MULTITHREADING
#include <sched.h>
#define ALIGATOR_TRACY_SET_THREAD_NAME(x) delete[] (x)
#include "aligator/gar/lqr-problem.hxx"
#include "aligator/gar/riccati-kernel.hxx"
#include "aligator/gar/proximal-riccati.hxx"
#include "aligator/gar/parallel-solver.hxx"
#include "aligator/gar/hip-riccati.hpp" // this repository's include/, next to the reference's include/aligator/gar

#include <cstdio>
#include <memory>
#include <random>

using namespace aligator;
using Problem = gar::LqrProblemTpl<double>;
using Knot = gar::LqrKnotTpl<double>;
using Base = gar::RiccatiSolverBase<double>;
using VectorXs = Eigen::Matrix<double, Eigen::Dynamic, 1>;

static Problem make_problem(uint nx, uint nu, uint nc, uint N, unsigned seed, bool coupled = false) {
  std::mt19937 rng(seed);
  std::normal_distribution<double> n01(0.0, 1.0);
  std::uniform_real_distribution<double> u11(-1.0, 1.0);
  Problem::KnotVector knots;
  for (uint t = 0; t <= N; ++t) {
    const uint nut = t < N ? nu : 0;
    Knot k(nx, nut, nc);
    const uint nw = nx + nut;
    Eigen::Matrix<double, -1, -1> G(nw, nw + 1);
    for (Eigen::Index j = 0; j < G.cols(); ++j)
      for (Eigen::Index i = 0; i < G.rows(); ++i)
        G(i, j) = n01(rng);
    Eigen::Matrix<double, -1, -1> W = G * G.transpose();
    W *= 1.0 / double(nx);
    k.Q = W.block(0, 0, nx, nx);
    if (nut > 0) {
      k.S = W.block(0, nx, nx, nut);
      k.R = W.block(nx, nx, nut, nut);
    }
    for (uint i = 0; i < nx; ++i) {
      k.q(i) = u11(rng);
      k.f(i) = n01(rng);
      for (uint j = 0; j < nx; ++j)
        k.A(i, j) = (i == j ? 1.0 : 0.0) + 0.1 * u11(rng);
      for (uint j = 0; j < nut; ++j)
        k.B(i, j) = 0.5 * u11(rng);
    }
    for (uint j = 0; j < nut; ++j)
      k.r(j) = u11(rng);
    for (uint c = 0; c < nc; ++c) {
      k.d(c) = u11(rng);
      for (uint j = 0; j < nx; ++j)
        k.C(c, j) = u11(rng);
      if (coupled) // (D != 0: the reduced KKT matrix [Rhat D^T; D -mu I] is really coupled, riccati-kernel.hxx:232-241)
        for (uint j = 0; j < nut; ++j)
          k.D(c, j) = u11(rng);
    }
    knots.push_back(std::move(k));
  }
  Problem p(knots, long(nx));
  p.G0.setIdentity();
  p.G0 *= -1.0;
  for (uint i = 0; i < nx; ++i)
    p.g0(i) = u11(rng);
  return p;
}

struct Sol {
  std::vector<VectorXs> xs, us, vs, lbdas;
  explicit Sol(const Problem &p) {
    const int N = p.horizon();
    lbdas.emplace_back(VectorXs::Zero(p.nc0()));
    for (int t = 0; t <= N; ++t) {
      const Knot &k = p.stages[size_t(t)];
      xs.emplace_back(VectorXs::Zero(k.nx));
      if (t < N)
        us.emplace_back(VectorXs::Zero(k.nu));
      vs.emplace_back(VectorXs::Zero(k.nc));
      if (t < N)
        lbdas.emplace_back(VectorXs::Zero(k.nx2));
    }
  }
};
static double diff(const std::vector<VectorXs> &a, const std::vector<VectorXs> &b) {
  double m = 0;
  for (size_t i = 0; i < a.size(); ++i)
    for (Eigen::Index j = 0; j < a[i].size(); ++j)
      m = std::max(m, std::abs(a[i](j) - b[i](j)));
  return m;
}

// what SolverProxDDP does with linear_solver_ between two Newton iterations
static double run(Base &solver, Problem &p, Sol &s, double mu, std::vector<double> &gains) {
  solver.backward(mu);
  solver.forward(s.xs, s.us, s.vs, s.lbdas);
  solver.collapseFeedback();
  gains.clear();
  for (size_t t = 0; t + 1 < p.stages.size(); ++t) {
    auto ff = solver.getFeedforward(t);
    auto fb = solver.getFeedback(t);
    for (Eigen::Index i = 0; i < ff.size(); ++i)
      gains.push_back(ff(i));
    for (Eigen::Index i = 0; i < fb.rows(); ++i)
      for (Eigen::Index j = 0; j < fb.cols(); ++j)
        gains.push_back(fb(i, j));
  }
  double scale = 1.0;
  for (const auto &l : s.lbdas)
    for (Eigen::Index j = 0; j < l.size(); ++j)
      scale = std::max(scale, std::abs(l(j)));
  return scale;
}

#ifndef SEAM_GPU
extern "C" void emu_set_device_count(int n); // the emulator build's virtual devices (tests/emu/emu_runtime.cpp)
#else
#include <chrono>
#endif

int main() {
  int bad = 0;
  // ndev > 1: the legs split over that many devices behind the ONE RiccatiSolverBase object (gar_hip_multi_create)
  struct Case { uint nx, nu, nc, N; int legs; double mu; const char *want; int ndev; bool coupled; };
#ifndef SEAM_GPU
  emu_set_device_count(3);
  const Case cases[] = {{8, 4, 0, 12, 1, 1e-10, "<8,4>", 1},        {7, 3, 0, 9, 1, 1e-10, "<8,4>", 1}, // padded inside the C ABI
                        {8, 4, 3, 10, 1, 1e-6, "generic", 1},       {12, 6, 0, 14, 3, 1e-10, "wave_leg<12,8>", 1},
                        {8, 4, 2, 11, 2, 1e-6, "wave_leg<8,4>+fold", 1},
                        // D != 0 in leg mode: the constrained segment legs (gar_cstr_seg.hpp) against the reference's
                        // ParallelRiccatiSolver, one device and the legs over two
                        {8, 4, 4, 13, 3, 1e-6, "wave_seg<8,4,4>", 1, true}, {8, 4, 4, 17, 4, 1e-6, "wave_seg<8,4,4>", 2, true},
                        {12, 6, 0, 14, 3, 1e-10, "wave_leg<12,8>", 3}, {8, 4, 0, 17, 5, 1e-10, "wave_leg<8,4>", 2},
                        {5, 2, 1, 11, 4, 1e-6, "generic", 3}};
#else
  // on the MI355X: the north star serial and in leg mode, the Talos-walk LQ shape (padded inside the C ABI) serial and
  // in leg mode, the reference's own benchmark shape nc = 32 (bench/gar-riccati.cpp:19-22) serial and folded into
  // legs, a generic shape, and the legs behind ONE object split over two sub-solvers (ids {0, 0}: one GPU here)
  const Case cases[] = {{36, 12, 0, 64, 1, 1e-10, "mfma<36,12>", 1},   {36, 12, 0, 96, 6, 1e-10, "wave_leg<36,12>", 1},
                        {56, 22, 0, 40, 1, 1e-10, "pair<56,24>", 1},    {56, 22, 0, 48, 6, 1e-10, "pair_leg<56,24>", 1},
                        {36, 12, 32, 32, 1, 1e-8, "wave<36,12,32>", 1}, {36, 12, 32, 36, 4, 1e-8, "fold", 1},
                        // (D != 0: the constrained segment legs.  mu = 1e-6: at 1e-8 the coupled gains of ANY two solvers of
                        // this problem differ by 5e-8 of their scale -- the any-dimension leg kernels' from the reference's
                        // by 4.9e-8, the segment legs' by 4.3e-8 -- where the fold's are C / mu to the last bit)
                        {36, 12, 32, 36, 4, 1e-6, "wave_seg<36,12,32>", 1, true},
                        {7, 3, 0, 9, 1, 1e-10, "<8,4>", 1},             {5, 2, 1, 11, 4, 1e-6, "generic", 1},
                        {36, 12, 0, 96, 6, 1e-10, "wave_leg<36,12>", 2}, {12, 6, 0, 30, 5, 1e-10, "wave_leg<12,8>", 2}};
#endif
  for (const Case &c : cases) {
    Problem pr = make_problem(c.nx, c.nu, c.nc, c.N, 7 + c.nx, c.coupled), ph = pr;
    Sol sr(pr), sh(ph);
    std::vector<double> gr, gh;
    std::unique_ptr<Base> ref, hip;
    if (c.legs == 1) {
      ref = std::make_unique<gar::ProximalRiccatiSolver<double>>(pr);
      hip = std::make_unique<gar::HipRiccatiSolver>(ph, 1);
    } else { // both re-parameterise their problem in place (parallel-solver.hxx:51-82)
      ref = std::make_unique<gar::ParallelRiccatiSolver<double>>(pr, uint(c.legs));
      for (int i = 0; i + 1 < c.legs; ++i) {
        const uint b = uint(i) * (c.N + 1) / uint(c.legs), e = uint(i + 1) * (c.N + 1) / uint(c.legs);
        for (uint t = b; t < e; ++t)
          ph.stages[t].addParameterization(ph.stages[e - 1].nx2);
      }
      if (c.ndev == 1) {
        hip = std::make_unique<gar::HipRiccatiSolver>(ph, c.legs);
      } else {
        std::vector<int> devs;
        for (int d = 0; d < c.ndev; ++d)
#ifndef SEAM_GPU
          devs.push_back(d);
#else
          devs.push_back(0);
#endif
        hip = std::make_unique<gar::HipRiccatiSolver>(ph, c.legs, devs);
      }
    }
    const double scale = run(*ref, pr, sr, c.mu, gr);
    run(*hip, ph, sh, c.mu, gh);
#ifdef SEAM_GPU
    { // the iteration again, timed (best of 5): the MI355X backend and, beside it, the reference's own solver as
      // compiled HERE (over the naive Eigen stand-in: an upper bound on its time, not a fair timing of Eigen)
      auto best = [&](Base &sv, Problem &p, Sol &sl) {
        double b = 1e30;
        std::vector<double> g;
        for (int r = 0; r < 5; ++r) {
          const auto t0 = std::chrono::steady_clock::now();
          run(sv, p, sl, c.mu, g);
          b = std::min(b, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
        return b;
      };
      const double th = best(*hip, ph, sh), tr = best(*ref, pr, sr);
      std::printf("    one iteration (backward + forward + collapse + every gain): HIP %.0f us, reference over the stand-in %.0f us\n", th, tr);
    }
#endif
    double dg = 0, gs = 1;
    for (size_t i = 0; i < std::min(gr.size(), gh.size()); ++i) {
      dg = std::max(dg, std::abs(gr[i] - gh[i]));
      gs = std::max(gs, std::abs(gr[i]));
    }
    const double dx = diff(sr.xs, sh.xs), du = diff(sr.us, sh.us), dv = diff(sr.vs, sh.vs), dl = diff(sr.lbdas, sh.lbdas);
    const auto &hs = static_cast<gar::HipRiccatiSolver &>(*hip);
    const char *name = hs.kernelName();
    // 1e-8 of their scale for the solution and for EVERY stage's gains, in every case but one: constrained knots folded
    // into legs with a small mu carry multipliers of order 1 / mu whose trailing digits no two factorisations share (the
    // reference's leg-parallel solve against the folded leg kernels); there the reference's own bar at this size is
    // 1e-6 (tests/gar/riccati.cpp:138) and x, u are held to 1e-5 of their scale, v and lbd to 1e-5 of theirs -- the
    // gains stay at 1e-8
    double xs_scale = 1.0;
    for (const auto &x : sr.xs)
      for (Eigen::Index j = 0; j < x.size(); ++j)
        xs_scale = std::max(xs_scale, std::abs(x(j)));
    const bool folded = std::string(name).find("fold") != std::string::npos;
    const double tol = folded ? 1e-5 : 1e-8, tol_gains = 1e-8;
    const bool ok = hs.numDevices() == c.ndev && gr.size() == gh.size() && std::max(dx, du) <= tol * (c.nc > 0 ? xs_scale : scale) &&
                    std::max(dv, dl) <= tol * scale && dg <= tol_gains * gs && std::string(name).find(c.want) != std::string::npos;
    std::printf("nx=%u nu=%u nc=%u N=%u legs=%d devices=%d kernel %-22s |x| %.1e |u| %.1e |v| %.1e |lbd| %.1e |gains| %.1e (rel %.1e)  %s\n", c.nx, c.nu,
                c.nc, c.N, c.legs, c.ndev, name, dx, du, dv, dl, dg, dg / gs, ok ? "ok" : "MISMATCH");
    bad += !ok;
  }
  std::printf(bad ? "%d case(s) FAILED\n" : "seam ok\n", bad);
  return bad ? 1 : 0;
}
