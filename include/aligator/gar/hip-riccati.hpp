/// @file hip-riccati.hpp
/// The reference-side binding of the MI355X gar backend: the ONE file a maintainer adds to aligator
/// (as include/aligator/gar/hip-riccati.hpp) to make the HIP library a `gar::RiccatiSolverBase<double>`.
///
/// It forwards the six virtuals SolverProxDDPTpl calls on `linear_solver_`
/// (include/aligator/solvers/proxddp/solver-proxddp.hxx:208 cycleAppend, :608 backward, :610 forward,
/// :619 collapseFeedback, :624-625 / :631-632 getFeedforward / getFeedback) to the C ABI of gar_hip.h
/// (link with -lgar_hip; no HIP headers are needed on the aligator side).
///
/// tests/test_integration_binding.py compiles THIS file against /root/reference/include and drives it through
/// `gar::RiccatiSolverBase<double>*` next to the reference's own ProximalRiccatiSolver / ParallelRiccatiSolver on
/// the same LqrProblemTpl (tests/integration/seam_driver.cpp).
#pragma once
#include "aligator/gar/riccati-base.hpp"
#include "aligator/gar/lqr-problem.hpp"
#include "aligator/utils/exceptions.hpp"
#include <gar_hip.h>   // this repository: include/gar_hip.h
#include <algorithm>
#include <vector>

namespace aligator::gar {

/// MI355X backend behind RiccatiSolverBase: serial in time if num_legs == 1 (ProximalRiccatiSolver semantics),
/// ParallelRiccatiSolver(problem, num_legs) semantics otherwise -- on one device, or with the legs split over
/// several devices of the node (`devices`): still ONE solver object in ONE process, the boundary exchange the
/// reference performs at the end of its OpenMP region (parallel-solver.hxx:150-169) happens inside backward().
class HipRiccatiSolver : public RiccatiSolverBase<double> {
public:
  using Base = RiccatiSolverBase<double>;
  using Problem = LqrProblemTpl<double>;
  using Knot = LqrKnotTpl<double>;
  using VectorXs = Base::VectorXs;
  using VectorMap = Eigen::Map<VectorXs>;
  using RowMatrixMap = Eigen::Map<Eigen::Matrix<double, -1, -1, Eigen::RowMajor>>;

  HipRiccatiSolver(Problem &problem, int num_legs = 1, int device = 0)
      : problem_(&problem), num_legs_(num_legs), devices_{device} { create(); }
  /// horizon sharded over `devices` (device ids; at most num_legs of them): gar_hip_multi_create
  HipRiccatiSolver(Problem &problem, int num_legs, std::vector<int> devices)
      : problem_(&problem), num_legs_(num_legs), devices_(std::move(devices)) { create(); }
  ~HipRiccatiSolver() { gar_hip_solver_destroy(h_); }

  bool backward(const double mueq) override {
    // the reference re-reads the caller's problem on every call (proximal-riccati.hxx:37):
    // 16 memcpy's per knot into the library's pinned staging area (of the device that owns the stage), ONE
    // host-to-device copy per device in gar_hip_backward
    // -- handed over in ONE call (gar_hip_backward_blocks)
    const auto &st = problem_->stages;
    blocks_.resize(16 * st.size());
    for (size_t t = 0; t < st.size(); ++t) {
      const Knot &k = st[t];
      const double *b[16] = {k.Q.data(), k.S.data(), k.R.data(), k.q.data(), k.r.data(), k.A.data(), k.B.data(), k.f.data(),
                             k.C.data(), k.D.data(), k.d.data(), k.Gth.data(), k.Gx.data(), k.Gu.data(), k.Gv.data(),
                             k.gamma.data()};
      std::copy(b, b + 16, blocks_.begin() + 16 * t);
    }

    const int rc = gar_hip_backward_blocks(h_, blocks_.data(), problem_->G0.data(), problem_->g0.data(), mueq);
    if (rc == GAR_HIP_ERR_FACTOR)                       // riccati-kernel.hxx:239-241
      ALIGATOR_RUNTIME_ERROR("Failed stage LDL factorization");
    check(rc);
    check(gar_hip_prefetch_gains(h_, 0));               // the gains start travelling now, under forward()
    gains_stale_ = true;                                // waited for once, on first use (after collapseFeedback, :619)
    return true;
  }

  bool forward(std::vector<VectorXs> &xs, std::vector<VectorXs> &us, std::vector<VectorXs> &vs,
               std::vector<VectorXs> &lbdas,
               const std::optional<ConstVectorRef> &theta = std::nullopt) const override {
    check(gar_hip_forward(h_, theta ? theta->data() : nullptr));
    check(gar_hip_fetch_results(h_, 0, /*solution*/ 1));        // ONE D2H (per device) into the pinned buffer
    int64_t offs[3];
    const double *p = gar_hip_host_results(h_, offs) + offs[0];  // xs | us | vs | lbdas, packed
    p = scatter(p, xs, nxs_); p = scatter(p, us, nus_); p = scatter(p, vs, nvs_); scatter(p, lbdas, nls_);
    return true;
  }

  void cycleAppend(const Knot &knot) override {
    const int32_t d[5] = {(int)knot.nx, (int)knot.nu, (int)knot.nc, (int)knot.nx2, (int)knot.nth};
    check(gar_hip_cycle_append(h_, d));
    map_gains();                                        // dimensions of the last-but-one stage may differ
  }
  void collapseFeedback() override { check(gar_hip_collapse_feedback(h_)); gains_stale_ = true; }
  // views into solver-owned host memory, valid until the next backward / cycleAppend (the reference's contract);
  // every stage's ff / fb comes over in ONE gather + ONE device-to-host copy, when the caller first asks
  VectorRef getFeedforward(size_t i) override { fetch_gains(); return ff_[i]; }
  RowMatrixRef getFeedback(size_t i) override { fetch_gains(); return fb_[i]; }   // row-major [K; Z; Aff]
  const char *kernelName() const { return gar_hip_kernel_name(h_); }  // the family that runs (padding is the library's business)
  int numDevices() const { return gar_hip_num_devices(h_); }

private:
  void create() {
    const auto &st = problem_->stages;
    std::vector<int32_t> dims5;
    for (const Knot &k : st) {
      // (the terminal knot SolverProxDDP builds has nx2 = 0, solvers/proxddp/workspace.hxx:54-55: passed as it is --
      // the library takes such a knot in as nx2 = nx with zeros for its unread A, f: gar_hip.h, gar_hip_solver_create)
      const int32_t d[5] = {(int)k.nx, (int)k.nu, (int)k.nc, (int)k.nx2, num_legs_ > 1 ? 0 : (int)k.nth};
      dims5.insert(dims5.end(), d, d + 5);
      nxs_ += k.nx; nus_ += k.nu; nvs_ += k.nc;
    }
    nls_ = problem_->nc0();
    for (size_t t = 0; t + 1 < st.size(); ++t) nls_ += st[t].nx2;
    // one device: gar_hip_multi_create hands back the plain solver of gar_hip_solver_create
    h_ = gar_hip_multi_create((int)devices_.size(), devices_.data(), (int)st.size() - 1, dims5.data(),
                              (int)problem_->nc0(), 1, num_legs_);
    if (!h_) ALIGATOR_RUNTIME_ERROR(gar_hip_last_error());   // no HIP device: there is no CPU fallback
    blocks_.reserve(16 * st.size());                    // (backward() must not allocate: ALIGATOR_NOMALLOC_SCOPED covers every call)
    map_gains();
  }
  // ff_[t] / fb_[t]: Eigen::Map views onto the library's pinned host buffer (solver-owned host memory,
  // valid until the next backward / cycleAppend, exactly the reference's contract)
  void map_gains() {
    check(gar_hip_fetch_results(h_, 0, 0));             // allocates the pinned buffer, copies nothing
    int64_t offs[3], go[2];
    double *base = const_cast<double *>(gar_hip_host_results(h_, offs));
    ff_.clear(); fb_.clear();
    const auto &st = problem_->stages;
    for (size_t t = 0; t < st.size(); ++t) {
      check(gar_hip_gains_offsets(h_, (int)t, go));
      const long nr = st[t].nu + st[t].nc + st[t].nx2;
      ff_.emplace_back(base + offs[1] + go[0], nr);
      fb_.emplace_back(base + offs[2] + go[1], nr, (long)st[t].nx);
    }
  }
  // every stage's ff / fb: one device-side gather, ONE D2H, one synchronisation (per device, concurrently)
  void fetch_gains() {
    if (!gains_stale_) return;
    check(gar_hip_fetch_results(h_, 0, /*gains*/ 2));
    gains_stale_ = false;
  }
  static void check(int rc) { if (rc != GAR_HIP_OK) ALIGATOR_RUNTIME_ERROR(gar_hip_last_error()); }
  static const double *scatter(const double *p, std::vector<VectorXs> &out, size_t total) {
    const double *q = p;
    for (VectorXs &v : out) { v = VectorMap(const_cast<double *>(q), v.size()); q += v.size(); }
    return p + total;
  }
  Problem *problem_; int num_legs_; std::vector<int> devices_; gar_hip_solver *h_ = nullptr;
  size_t nxs_ = 0, nus_ = 0, nvs_ = 0, nls_ = 0;        // doubles per part of the packed solution
  std::vector<VectorMap> ff_; std::vector<RowMatrixMap> fb_;
  std::vector<const double *> blocks_;                  // the 16 block pointers of every knot, rebuilt per backward
  bool gains_stale_ = false;
};

} // namespace aligator::gar
