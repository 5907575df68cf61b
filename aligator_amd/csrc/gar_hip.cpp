// gar_hip.cpp -- C-ABI implementation (include/gar_hip.h) over the HIP kernels.
// Host C++ only packs per-stage blocks into the contiguous device records,
// launches kernels on a HIP stream and copies results back; all arithmetic of
// the Riccati path runs in the kernels (gar_generic.hpp, gar_mfma.hpp).
#include "../../include/gar_hip.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "gar_generic.hpp"
#include "gar_layout.h"
#include "gar_mfma.hpp"
#include "gar_forward_lean.hpp"
#include "gar_wave.hpp"
#include "gar_wave_leg.hpp"
#include "gar_wave_pair.hpp"
#include "gar_cyclic.hpp"
#include "gar_condensed_cr.hpp"
#include "gar_dense.hpp"
#include "gar_fold.hpp"
#include "gar_cstr_seg_api.hpp"
#include "gar_leg_seg.hpp"

namespace gar { // instantiated in gar_wave_sweep.cpp (its own translation unit, its own code-generation flags)
#define GAR_SWEEP_EXTERN(NX, NU)                                                                                        \
  extern template __global__ void gar_backward_wave<NX, NU, 0>(MfmaParams, int);                                       \
  extern template __global__ void gar_backward_wave_half<NX, NU>(MfmaParams, int);
GAR_SWEEP_SHAPES(GAR_SWEEP_EXTERN)
  extern template __global__ void gar_backward_wave<56, 24, 0>(MfmaParams, int); // gar_wave_wide.cpp
#undef GAR_SWEEP_EXTERN
} // namespace gar

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess)                                                                \
      return fail(GAR_HIP_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

inline int align2(int x) { return (x + 1) & ~1; }

// Every device / pinned-host allocation of the library goes through these two and is counted
// (gar_hip_debug_alloc_count): the reference runs backward / forward under ALIGATOR_NOMALLOC_SCOPED
// (gar/proximal-riccati.hxx:35, tests/nomalloc.cpp); tests/test_nomalloc.py asserts the same here -- the count
// does not move across repeated backward + forward calls.
std::atomic<long long> g_alloc_count{0};
inline hipError_t gar_dev_malloc(void **p, size_t bytes) {
  g_alloc_count.fetch_add(1, std::memory_order_relaxed);
  return hipMalloc(p, bytes);
}
inline hipError_t gar_host_malloc(void **p, size_t bytes, unsigned flags) {
  g_alloc_count.fetch_add(1, std::memory_order_relaxed);
  return hipHostMalloc(p, bytes, flags);
}

// Tracing hooks (the reference brackets the same places with Tracy zones: backwardImpl
// riccati-kernel.hxx:108, "factor_initial" proximal-riccati.hxx:43, forwardImpl riccati-kernel.hxx:320,
// "parallel_backward" / "parallel_forward" parallel-solver.hxx:134,213, assembleCondensedSystem :87):
// roctx ranges around the launches, visible in rocprofv3 --marker-trace.  Opt-in (GAR_HIP_ROCTX=1) and
// resolved at run time, so the library carries no link dependency on the tracer.
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char *e = std::getenv("GAR_HIP_ROCTX"); // (read once, when the first range opens: environment only)
    if (!e || e[0] != '1')
      return;
    if (void *h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL)) {
      push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
      pop = (int (*)())dlsym(h, "roctxRangePop");
      if (!push || !pop)
        push = nullptr, pop = nullptr;
    }
  }
};
inline const Roctx &roctx() {
  static const Roctx r;
  return r;
}
struct RoctxRange {
  explicit RoctxRange(const char *name) {
    if (roctx().push)
      roctx().push(name);
  }
  ~RoctxRange() {
    if (roctx().pop)
      roctx().pop();
  }
};

// Every entry point runs on the solver's own device, whatever the caller's current device is, and
// leaves the caller's current device as it found it (two solvers on two GPUs in one process;
// torch's notion of the current device).
// Behaviour switches (kernel family, padding, condensed solver, ...): `GAR_HIP_*` names, looked up in the overrides
// set through gar_hip_set_option first, in the environment second.  Most are read when a solver is created
// (family selection), some per launch (GAR_HIP_SPD_ACCEPT) -- include/gar_hip.h lists them.
std::mutex &option_mutex() {
  static std::mutex m;
  return m;
}
std::map<std::string, std::string> &option_overrides() {
  static std::map<std::string, std::string> o;
  return o;
}
const char *gar_option(const char *name) {
  // (the value is copied out under the lock into a per-thread slot: a concurrent gar_hip_set_option cannot pull the
  // string from under the caller; eight slots cover every use that holds more than one option at a time)
  thread_local std::string slot[8];
  thread_local unsigned next = 0;
  {
    std::lock_guard<std::mutex> g(option_mutex());
    auto it = option_overrides().find(name);
    if (it != option_overrides().end()) {
      std::string &v = slot[next++ & 7u];
      v = it->second;
      return v.c_str();
    }
  }
  return std::getenv(name);
}

void pipe_autojoin(const gar_hip_solver *s);
struct DeviceGuard {
  int prev = -1, dev;
  explicit DeviceGuard(int device) : dev(device) { // device < 0 (null solver): no-op
    if (dev < 0)
      return;
    if (hipGetDevice(&prev) != hipSuccess)
      prev = -1;
    if (prev != dev)
      (void)hipSetDevice(dev);
  }
  ~DeviceGuard() {
    if (dev >= 0 && prev >= 0 && prev != dev)
      (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};
// (every entry point but the two that feed the pipelined sweep first orders the caller's stream behind the half
// streams: pipe_join, below)
#define GAR_GUARD_NOJOIN(s) DeviceGuard guard_((s) ? (s)->device : -1)
#define GAR_GUARD(s)                                                                                                   \
  GAR_GUARD_NOJOIN(s);                                                                                                 \
  pipe_autojoin(s)
// one process, several devices (gar_multi.hpp): the handle owns one ranked solver per device and serves every
// entry point by routing to them
#define GAR_MULTI(s, expr)                                                                                             \
  do {                                                                                                                 \
    if ((s) && (s)->multi)                                                                                             \
      return (expr);                                                                                                   \
  } while (0)

} // namespace

struct gar_multi;

struct gar_hip_solver {
  // gar_hip_multi_create: this object holds the layout only (no device memory); `multi` owns the per-device solvers
  gar_multi *multi = nullptr;
  int device = 0, horizon = 0, nc0 = 0, batch = 0;
  int num_legs = 1, leg_begin = 0, leg_end = 1;
  int world = 1, rank = 0; // horizon sharding: this solver owns legs [rank J / W, (rank + 1) J / W)
  std::vector<gar_stage_meta> meta;
  std::vector<int32_t> dims5; // dimensions of the DEVICE records (= the caller's unless padded)
  // Padding onto a specialised kernel family happens HERE, behind the C ABI (riccati-base.hpp:13-37 is what
  // binds: the caller passes the knots' own dimensions).  A uniform, unconstrained, unparameterised problem whose
  // (nx, nu) has no kernel of its own runs on the smallest specialised shape (NX >= nx, NU >= nu) with DUMMY
  // controls (R = I, S = 0, B = 0, r = 0) and DUMMY states (Q = I, A = 0, f = 0, pinned to zero by extra rows
  // [0 -I] x0 = 0 of the initial constraint): both solve to exactly zero, decouple from the real variables, and
  // are stripped from every result.  user_* = what the caller passed; `ulay` = the layout of the caller-facing
  // records (packed problem, solution, gains) when they differ from the device's.
  std::vector<int32_t> user_dims5;
  int user_nc0 = 0;
  bool padded = false;
  int unx = 0, unu = 0, pnx = 0, pnu = 0; // caller's / device (nx, nu) of the uniform stages
  gar_hip_solver *ulay = nullptr;         // host-only layout object (no device memory), owned
  // Constrained knots (nc > 0) in leg mode on the unconstrained wave-leg kernels (gar_fold.hpp): `flay` = the
  // layout of the folded problem (same knots, nc = 0), d_prob2 / d_fac2 / d_meta2 its device records.  Problems
  // with D != 0 are flagged on the device (d_status + batch + 4) and taken by the generic leg kernels.
  // Segment legs (gar_leg_seg.hpp): leg mode for shapes with a serial stage kernel but no wave-leg family -- the
  // plain part of every leg by that kernel into scratch records (flay: the same knots, nth = 0; d_fac2), the
  // parameter part by the generic matrix recursion, which writes the caller-visible records and the tuples
  void (*seg_bwd_kernel)(gar::MfmaParams, int, int) = nullptr;
  void (*seg_fwd_kernel)(gar::GenericParams) = nullptr; // its roll-out (gar_forward_wide_leg), leg mode
  int seg_lds_doubles = 0;
  bool fold = false, fold_expanded = false, coupled_known = false;
  // ... unless the shape has the constrained segment legs (gar_cstr_seg.hpp, round 6): then the flagged problems run
  // on the serial constrained chain's stage kernels, leg by leg, + a parameter recursion; the knots keep Q, R packed
  // (qr_packed), the plain part's records go to the flagged problem's slice of d_fac2, d_cseg_resume holds the chain's
  // hand-over knot per (problem, local leg)
  bool cseg_on = false;
  bool mu_divides = false; // some knot's solve divides by mueq outright (see gar_hip_backward_legs_async)
  gar::CsegKernels cseg;
  int *d_cseg_resume = nullptr;
  gar_hip_solver *flay = nullptr;
  double *d_prob2 = nullptr, *d_fac2 = nullptr;
  gar_stage_meta *d_meta2 = nullptr;
  double fold_mueq = 0.0;
  std::vector<int> h_coupled;
  gar_stage_meta *d_meta = nullptr;
  int64_t prob_doubles = 0, fac_doubles = 0, sol_doubles = 0, init_doubles = 0;
  int64_t G0_off = 0, g0_off = 0;
  int64_t sol_x = 0, sol_u = 0, sol_v = 0, sol_l = 0; // base offsets of xs/us/vs/lbdas
  int nx0 = 0, nth0 = 0, n0 = 0;
  // MPC cycling as a ring (uniform serial problems): logical stage t < horizon lives in record slot
  // (t + ring0) mod horizon; meta[t].in_off / fac_off follow, the records never move
  int ring0 = 0;
  int64_t uni_in0 = 0, uni_in_rec = 0, uni_fac_rec = 0; // slot 0 and the record pitches (layout time)
  double *d_prob = nullptr, *d_fac = nullptr, *d_sol = nullptr, *d_init = nullptr;
  double *d_theta = nullptr;
  int *d_status = nullptr;
  double *d_kkt = nullptr; // gar_hip_get_kkt's staging ((nu+nc)^2 doubles, allocated on first use)
  int64_t kkt_doubles = 0;
  // leg mode
  int nxb = 0;
  int64_t tuple_doubles = 0, cscratch_doubles = 0;
  double *d_bound_local = nullptr, *d_bound_all = nullptr, *d_csol = nullptr,
         *d_cscratch = nullptr;
  bool bound_all_owned = false;
  int legs_per_rank = 0;
  double cond_threshold = 1e-10; // parallel-solver.hpp:92
  double cond_backward_ok = GAR_CONDENSED_BACKWARD_OK; // gar_hip_set_condensed_backward_ok
  bool cond_reduced = false; // generic condensed solve: leg states eliminated leg-parallel first (GAR_HIP_CONDENSED_REDUCED=0: off)
  bool cond_cr = false;      // ... and the J remaining blocks by block cyclic reduction, a workgroup per block and level
                             // (gar_condensed_cr.hpp; GAR_HIP_CONDENSED_CR=0: the one-workgroup chain, =<k>: from k legs on)
  int max_refinement = 5;        // parallel-solver.hpp:94
  // host staging
  double *h_prob = nullptr; // pinned, batch * prob_doubles (when small enough)
  bool staged = false, dirty = false;
  bool stage_nt = false; // pack with non-temporal stores (problems of >= 12 MiB; GAR_HIP_STAGE_NT=0/1 overrides)
  // what the host wrote into the staging area since the last flush: per problem, a sorted list of
  // disjoint [lo, hi) ranges (doubles).  commit() copies exactly these, so knots a device-resident
  // producer wrote in place (gar_hip_device_problems) survive a later set_init / upload_stage
  std::vector<std::vector<std::pair<int64_t, int64_t>>> dirty_iv;
  hipStream_t own_stream = nullptr, stream = nullptr;
  gar::LdsPlan lds{};
  // RiccatiSolverDense (gar_dense.hpp): factor records carry nu+nc+2*nx2 gain rows
  bool dense = false;
  gar::DensePlan dense_lds{};
  int cond_lds_doubles = 0;
  std::string kernel_name = "generic";
  int last_failed = 0;
  // specialised backward kernel (gar_mfma.hpp), null = generic
  void (*mfma_kernel)(gar::MfmaParams) = nullptr;
  void (*mfma_fwd_kernel)(gar::MfmaFwdParams) = nullptr;
  size_t mfma_fwd_lds_bytes = 0; // gar_forward_mfma: the packed Vxx' of a stage goes through LDS
  int mfma_lds_doubles = 0;
  // one-wave-per-problem backward kernel (gar_wave.hpp), preferred when bound
  void (*wave_kernel)(gar::MfmaParams, int) = nullptr;
  void (*wave_coupled_kernel)(gar::MfmaParams, int) = nullptr; // constrained sweeps: the second ...
  void (*wave_bk_kernel)(gar::MfmaParams, int) = nullptr;      // ... and the third kernel of the chain
  int wave_lds_doubles = 0, waves_per_block = 1;
  int wave_block_threads = 64; // 128: two waves per problem (gar_wave_pair.hpp)
  bool fb_t2 = false;      // factor records keep fb / fth in the fbT2 device order (gar_mfma.hpp)
  bool vxx_packed = false; // ... and the lower triangle of Vxx, packed (gar_layout.h: the serial one-wave family)
  bool wide_vxx_packed = false; // (set by bind_wide: the serial two-wave family with packed records)
  bool qr_packed = false;  // knots t < N keep Q and R as packed lower triangles (gar_layout.h: the headline sweep)
  std::string lds_error;   // the generic kernels do not fit a CU's LDS (fatal unless a specialised family serves the shape)
  bool wave_fused_init = false;
  bool init_closed = true; // closed-form initial stage when G0 = +-I (GAR_HIP_INIT=bk: always factorise)
  // one-wave-per-(problem, leg) kernels (gar_wave_leg.hpp), bound for uniform leg-mode problems
  void (*leg_bwd_kernel)(gar::LegParams) = nullptr;
  void (*leg_tuple_kernel)(gar::LegParams) = nullptr;
  void (*leg_fwd_kernel)(gar::LegParams) = nullptr;
  void (*leg_collapse_kernel)(const gar_stage_meta *, double *, long long, int, const int *, int) = nullptr;
  int leg_lds_doubles = 0, leg_waves = 1;
  void (*cond_wave_kernel)(gar::CondensedParams) = nullptr;
  int cond_wave_lds_doubles = 0;
  // block cyclic reduction of the condensed system (gar_cyclic.hpp), preferred when bound
  void (*cyc_setup_kernel)(gar::CyclicParams) = nullptr;
  void (*cyc_reduce_kernel)(gar::CyclicParams) = nullptr;
  void (*cyc_top_kernel)(gar::CyclicParams) = nullptr;
  void (*cyc_backlevel_kernel)(gar::CyclicParams) = nullptr;
  void (*cyc_recover_kernel)(gar::CyclicParams) = nullptr;
  int cyc_lds_doubles = 0;
  int cyc_block_doubles = 0; // one NX x NX block of the cyclic-reduction kernels (gar_cyclic_recover's LDS)
  long long *d_trace = nullptr; // 64 cycle stamps (debug)
  // optional per-kernel timing of the sweep (bench.py's roofline figure): HIP events recorded on
  // the launch stream around the backward sweep kernel, the initial-stage kernel and the forward
  // sweep kernel of the LAST backward/forward calls
  // device-resident updateLQSubproblem: layout of one problem's derivative buffer
  std::vector<long long> deriv_off; // per stage
  long long deriv_doubles = 0, d_G0 = 0, d_g0 = 0, d_iH = 0;
  long long *d_deriv_off = nullptr;
  // bulk read-back (gar_hip_fetch_results): per-stage offsets inside ff_all / fb_all, the device
  // gather buffer and the pinned host buffer [solution | ff_all | fb_all] of one problem
  std::vector<long long> gain_off; // 2 per stage
  long long ff_all_doubles = 0, fb_all_doubles = 0;
  long long *d_gain_off = nullptr;
  double *d_gains = nullptr, *h_results = nullptr;
  bool timing = false;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  // gar_hip_prefetch_gains: the bulk read-back of the gains started right behind the backward sweep on a second
  // stream, so that it overlaps the forward sweep and the solution read-back
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_main = nullptr, ev_pref = nullptr;
  int pref_b = -1;             // problem whose gains are in flight / in h_results (-1: none)
  // gar_hip_backward_blocks on a problem without parameter: the roll-out and the solution's copy are enqueued BEHIND the
  // sweep before the host waits for the status word, so that gar_hip_forward / the solution fetch find them done
  int *h_status = nullptr;     // pinned
  hipEvent_t ev_status = nullptr, ev_sol = nullptr;
  bool eager_fwd = false;      // d_sol and h_results hold the roll-out of the last sweep (theta = none)
  bool pref_collapsed = false; // collapseFeedback ran since: stage 0's gains are fetched again
  // ---- pipelined sweep (gar_hip_set_pipeline; the serial one-wave family, batch >= 2) --------------------------
  // The batch is cut in two halves with a stream each; backward sweeps alternate between the halves (events), the
  // forward sweep of a half is gar_forward_lean, which fits in the registers and the LDS the backward wave of the
  // OTHER half leaves free on every SIMD (gar_forward_lean.hpp): B(h0) | F(h0) + B(h1) | F(h1) + B'(h0) | ...
  int pipe_halves = 0;                 // 0: off
  int pipe_requested = 0;              // what the caller last asked gar_hip_set_pipeline for (a rebuild re-validates it)
  void (*lean_fwd_kernel)(gar::MfmaFwdParams, int) = nullptr;
  void (*wave_half_kernel)(gar::MfmaParams, int) = nullptr; // the backward sweep under its half-batch launch name
  size_t lean_fwd_used = 0;            // LDS the kernel uses
  size_t lean_fwd_lds_bytes = 0;       // what the launch ASKS for (> half a CU: one workgroup per CU), see pipe_plan
  int wave_lds_doubles_small = 0;      // the backward launch without the fused initial stage's kkt0 overlay
  hipStream_t pipe_stream[2] = {nullptr, nullptr};
  hipEvent_t pipe_evB[2] = {nullptr, nullptr}, pipe_evF[2] = {nullptr, nullptr}, pipe_evFork = nullptr;
  hipEvent_t pipe_evT[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}}; // timing
  bool pipe_evB_valid[2] = {false, false};
  bool pipe_forked = false;            // the half streams hold work the caller's stream has not been ordered behind
  // SolverProxDDP builds the terminal knot with nx2 = 0 (solvers/proxddp/workspace.hxx:54-55); nothing of the
  // algorithm reads a terminal knot's A, f (riccati-kernel.hxx:130-193).  Such a knot is taken in as nx2 = nx -- the
  // uniform record every kernel family addresses -- with zeros for the two blocks; the gains of that knot go back
  // with the caller's row count (normalise_terminal, gar_hip_upload_stage, gar_hip_get_gains)
  bool term_grown = false;
  std::vector<double> term_zeros;
};

namespace {

// ---- layout ---------------------------------------------------------------
int build_layout(gar_hip_solver *s) {
  const int N = s->horizon;
  s->meta.assign(N + 1, gar_stage_meta{});
  std::vector<int> leg_of(N + 1, 0), nth_eff(N + 1, 0), flags(N + 1, 0);
  for (int t = 0; t <= N; ++t) {
    const int32_t *d = &s->dims5[5 * t];
    if (d[0] < 0 || d[1] < 0 || d[2] < 0 || d[3] < 0 || d[4] < 0 || d[0] == 0)
      return fail(GAR_HIP_ERR_ARG, "negative or zero state dimension");
    nth_eff[t] = d[4];
    flags[t] = d[4] > 0 ? GAR_KNOT_HAS_PARAM : 0;
  }
  if (s->num_legs > 1) {
    // ParallelRiccatiSolver::initialize (parallel-solver.hxx:51-82)
    for (int i = 0; i < s->num_legs; ++i) {
      int i0, i1;
      gar_get_work(N, i, s->num_legs, &i0, &i1);
      if (i1 <= i0)
        return fail(GAR_HIP_ERR_ARG, "more legs than stages");
      const bool last_leg = (i == s->num_legs - 1);
      const int nth = s->dims5[5 * (i1 - 1) + 3]; // nx2 of the leg's last knot
      for (int t = i0; t < i1; ++t) {
        if (s->dims5[5 * t + 4] != 0)
          return fail(GAR_HIP_ERR_UNSUPPORTED,
                      "leg mode on a user-parameterised problem is not supported");
        leg_of[t] = i;
        nth_eff[t] = last_leg ? 0 : nth;
        flags[t] = last_leg ? 0 : (t == i1 - 1 ? GAR_KNOT_LEG_END : GAR_KNOT_LEG_PARAM);
      }
    }
  }
  s->nx0 = s->dims5[0];
  s->nth0 = nth_eff[0];
  s->n0 = s->nx0 + s->nc0;
  int64_t in = 0, fo = 0;
  s->G0_off = in;
  in += (int64_t)s->nc0 * s->nx0;
  s->g0_off = in;
  in += s->nc0;
  in = (in + 1) & ~(int64_t)1;
  int64_t x = 0, u = 0, v = 0, l = s->nc0;
  for (int t = 0; t <= N; ++t) {
    const int32_t *d = &s->dims5[5 * t];
    gar_stage_meta &m = s->meta[t];
    m.nx = d[0]; m.nu = d[1]; m.nc = d[2]; m.nx2 = d[3];
    m.nth = nth_eff[t];
    m.flags = flags[t];
    m.leg = leg_of[t];
    m.in_off = in;
    in += gar_knot_doubles(d[0], d[1], d[2], d[3], (flags[t] & GAR_KNOT_HAS_PARAM) ? d[4] : 0);
    in = (in + 1) & ~(int64_t)1; // keep records 16-byte aligned
    m.fac_off = fo;
    fo += gar_factor_doubles(d[0], d[1], d[2], s->dense ? 2 * d[3] : d[3], nth_eff[t]);
    fo = (fo + 1) & ~(int64_t)1;
    m.x_off = (int32_t)x; x += d[0];
    m.u_off = (int32_t)u; u += d[1];
    m.v_off = (int32_t)v; v += d[2];
    m.l_off = (int32_t)(t == 0 ? 0 : l);
    if (t > 0)
      l += s->dims5[5 * (t - 1) + 3];
  }
  // lbdas[t] (t>=1) has nx2 of stage t-1: recompute l offsets cleanly
  {
    int64_t lo = s->nc0;
    s->meta[0].l_off = 0;
    for (int t = 1; t <= N; ++t) {
      s->meta[t].l_off = (int32_t)lo;
      lo += s->dims5[5 * (t - 1) + 3];
    }
    l = lo;
  }
  s->prob_doubles = in;
  s->fac_doubles = fo;
  s->sol_x = 0;
  s->sol_u = x;
  s->sol_v = x + u;
  s->sol_l = x + u + v;
  for (int t = 0; t <= N; ++t) {
    s->meta[t].u_off += (int32_t)s->sol_u;
    s->meta[t].v_off += (int32_t)s->sol_v;
    s->meta[t].l_off += (int32_t)s->sol_l;
  }
  s->sol_doubles = (x + u + v + l + 1) & ~(int64_t)1;
  { // derivative buffer: header G0 | g0 | init Hxx, then one record per stage (even offsets)
    long long p = 0;
    s->d_G0 = p; p += (long long)s->nc0 * s->nx0;
    s->d_g0 = p; p += s->nc0;
    s->d_iH = p; p += (long long)s->nx0 * s->nx0;
    p = (p + 1) & ~1ll;
    s->deriv_off.assign(N + 1, 0);
    for (int t = 0; t <= N; ++t) {
      const int32_t *d = &s->dims5[5 * t];
      s->deriv_off[t] = p;
      p += gar_deriv_layout(d[0], d[1], d[2], d[3]).total;
      p = (p + 1) & ~1ll;
    }
    s->deriv_doubles = p;
  }
  {
    long long pf = 0, pb = 0;
    s->gain_off.assign(2 * (size_t)(N + 1), 0);
    for (int t = 0; t <= N; ++t) {
      const int32_t *d = &s->dims5[5 * t];
      const long long nr = (long long)d[1] + d[2] + (s->dense ? 2 * d[3] : d[3]);
      s->gain_off[2 * t] = pf;
      s->gain_off[2 * t + 1] = pb;
      pf += nr;
      pb += nr * d[0];
    }
    s->ff_all_doubles = pf;
    s->fb_all_doubles = pb;
  }
  s->init_doubles = ((int64_t)s->n0 + (int64_t)s->n0 * s->nth0 + s->nth0 +
                     (int64_t)s->nth0 * s->nth0 + 1) & ~(int64_t)1;
  s->ring0 = 0;
  s->uni_in0 = s->meta[0].in_off;
  s->uni_in_rec = N > 1 ? s->meta[1].in_off - s->meta[0].in_off : s->meta[N].in_off - s->meta[0].in_off;
  s->uni_fac_rec = N > 1 ? s->meta[1].fac_off - s->meta[0].fac_off : s->meta[N].fac_off;
  return GAR_HIP_OK;
}

int plan_lds(gar_hip_solver *s) {
  int nxM = 0, nuM = 0, ncM = 0, nthM = 0, nwM = 0, nkM = 0;
  for (const auto &m : s->meta) {
    nxM = std::max(nxM, std::max(m.nx, m.nx2));
    nuM = std::max(nuM, m.nu);
    ncM = std::max(ncM, m.nc);
    nthM = std::max(nthM, m.nth);
    nwM = std::max(nwM, m.nx + m.nu);
    nkM = std::max(nkM, m.nu + m.nc);
  }
  gar::LdsPlan &L = s->lds;
  int p = 0;
  auto take = [&](int n) { int o = p; p += align2(std::max(n, 0)); return o; };
  L.lean = 0;
replan:
  p = 0;
  for (int k = 0; k < 2; ++k) {
    // Vxx', vx' are dead once P = V'[A B] and vplus are formed (S1), Vxx, vx are written in S5: one
    // buffer serves both (25 KB at nx = 56, what lets the Talos shape fit a CU's LDS); the
    // parameter blocks are read and written in the same phase and keep two
    L.V[k] = k == 0 ? take(nxM * nxM) : L.V[0];
    L.v[k] = k == 0 ? take(nxM) : L.v[0];
    L.Vxt[k] = (k == 1 && L.lean) ? L.Vxt[0] : take(nxM * nthM);
    L.Vtt[k] = (k == 1 && L.lean) ? L.Vtt[0] : take(nthM * nthM);
    L.vt[k] = (k == 1 && L.lean) ? L.vt[0] : take(nthM);
  }
  const int stage_begin = p;
  L.H = take(nwM * nwM);
  L.h = take(nwM);
  L.F = take(nxM * nwM);
  L.fv = take(nxM);
  L.P = take(std::max(nxM * nwM, nxM * nxM));
  L.vp = take(nxM);
  L.CD = take(ncM * nwM);
  L.dd = take(ncM);
  L.Gu = take(nuM * nthM);
  L.Guh = take(nuM * nthM);
  L.Gv = take(ncM * nthM);
  L.M = take(nkM * nkM);
  L.msub = take(nkM);
  L.piv = take(264); // 512 pivots + 8 control ints
  L.G = take(nkM * (1 + nxM + nthM));
  L.Yth = take(nxM * nthM);
  L.yff = take(nxM);
  const int stage_end = p;
  // initial-stage KKT aliases the per-stage buffers
  p = stage_begin;
  const int n0 = s->n0, nth0 = s->nth0;
  L.k0mat = take(n0 * n0);
  L.k0rhs = take(n0 * (1 + nth0));
  L.k0sub = take(n0);
  L.k0piv = take(264);
  L.total = std::max(stage_end, p);
  // panel workspace of the blocked Bunch-Kaufman, behind everything -- where there is room for it
  L.bkw = -1;
  if (nkM <= 128 && (size_t)(L.total + align2(nkM * GAR_BK_PANEL)) * sizeof(double) <= 160 * 1024) {
    L.bkw = L.total;
    L.total += align2(nkM * GAR_BK_PANEL);
  }
  if (!L.lean && nthM > 0 && (size_t)L.total * sizeof(double) > 160 * 1024) {
    L.lean = 1; // one generation of the parameter blocks in LDS, the other read back from the records
    goto replan;
  }
  // forward kernel
  p = 0;
  L.fx = take(nxM);
  L.fxn = take(nxM);
  L.fth = take(nthM);
  L.ftotal = p;
  if (s->dense) {
    int nM = 0, rldM = 0;
    for (const auto &m : s->meta) {
      nM = std::max(nM, m.nu + m.nc + 2 * m.nx2);
      rldM = std::max(rldM, 1 + m.nx + m.nth);
    }
    gar::DensePlan &D = s->dense_lds;
    p = 0;
    D.K = take(nM * (nM + 1) / 2); // packed lower triangle
    D.R = take(nM * rldM);
    D.sub = take(nM);
    D.piv = take(264);
    D.wk = -1;
    if (nM <= 128 && (int64_t)(p + nM * GAR_BK_PANEL) * 8 <= 160 * 1024)
      D.wk = take(nM * GAR_BK_PANEL);
    const int stage_total = p;
    p = 0; // the initial stage reuses the buffer from its start (gar_backward_dense)
    take(n0 * n0);
    take(n0 * (1 + nth0));
    take(n0);
    take(264);
    D.total = std::max(stage_total, p);
    if (n0 > 512 || nM > 512)
      return fail(GAR_HIP_ERR_UNSUPPORTED, "KKT dimension above 512");
    if ((int64_t)D.total * 8 > 160 * 1024)
      return fail(GAR_HIP_ERR_UNSUPPORTED,
                  "stage-dense solver: stage dimensions need " + std::to_string((int64_t)D.total * 8) +
                      " B of LDS (> 160 KiB per CU)");
    return GAR_HIP_OK;
  }
  if (n0 > 512 || nkM > 512)
    return fail(GAR_HIP_ERR_UNSUPPORTED, "KKT dimension above 512");
  s->lds_error.clear();
  if ((int64_t)L.total * 8 > 160 * 1024) // fatal only if no specialised family serves the shape (allocate)
    s->lds_error = "stage dimensions need " + std::to_string((int64_t)L.total * 8) +
                   " B of LDS (> 160 KiB per CU)";
  return GAR_HIP_OK;
}

#include "gar_select.hpp"

gar::GenericParams make_params(gar_hip_solver *s, double mueq) {
  gar::GenericParams P{};
  P.meta = s->d_meta;
  P.prob = s->d_prob;
  P.fac = s->d_fac;
  P.sol = s->d_sol;
  P.init = s->d_init;
  P.status = s->d_status;
  P.boundary = s->d_bound_local;
  P.csol = s->d_csol;
  P.theta = nullptr;
  P.prob_stride = s->prob_doubles;
  P.fac_stride = s->fac_doubles;
  P.sol_stride = s->sol_doubles;
  P.init_stride = s->init_doubles;
  P.boundary_stride = (long long)s->legs_per_rank * s->tuple_doubles;
  P.G0_off = s->G0_off;
  P.g0_off = s->g0_off;
  P.horizon = s->horizon;
  P.nc0 = s->nc0;
  P.nx0 = s->nx0;
  P.nth0 = s->nth0;
  P.num_legs = s->num_legs;
  P.leg_begin = s->leg_begin;
  P.local_legs = s->leg_end - s->leg_begin;
  P.tuple_doubles = (int)s->tuple_doubles;
  P.nxb = s->nxb;
  P.mueq = mueq;
  P.lds = s->lds;
  P.dense = s->dense_lds;
  P.init_closed = s->init_closed ? 1 : 0;
  P.vxx_packed = s->vxx_packed ? 1 : 0;
  return P;
}

// non-temporal stores (stage_copy) are weakly ordered: make them globally visible before a DMA engine reads them
inline void stage_fence() {
#if defined(__SSE2__) && !defined(__HIP_DEVICE_COMPILE__)
  _mm_sfence();
#endif
}

void mark_dirty(gar_hip_solver *s, int b, int64_t lo, int64_t hi) {
  auto &iv = s->dirty_iv[(size_t)b];
  // the common pattern is "append right after the last range" (knot after knot): O(1); gaps of
  // one double are record-alignment padding and merge as well
  if (!iv.empty() && lo >= iv.back().first && lo <= iv.back().second + 1) {
    iv.back().second = std::max(iv.back().second, hi);
  } else {
    iv.emplace_back(lo, hi);
    if (iv.size() > 1 && iv[iv.size() - 2].first > lo) { // out of order: sort and merge
      std::sort(iv.begin(), iv.end());
      size_t w = 0;
      for (size_t r = 1; r < iv.size(); ++r) {
        if (iv[r].first <= iv[w].second + 1)
          iv[w].second = std::max(iv[w].second, iv[r].second);
        else
          iv[++w] = iv[r];
      }
      iv.resize(w + 1);
    }
  }
  s->dirty = true;
}

int commit(gar_hip_solver *s) {
  if (!(s->staged && s->dirty))
    return GAR_HIP_OK;
  stage_fence();
  const int64_t P = s->prob_doubles;
  // whole problems, back to back: one copy per run of fully rewritten problems
  int b = 0;
  while (b < s->batch) {
    auto &iv = s->dirty_iv[(size_t)b];
    if (iv.empty()) {
      ++b;
      continue;
    }
    const bool whole = iv.size() == 1 && iv[0].first == 0 && iv[0].second >= P - 1;
    if (whole) {
      int e = b + 1;
      while (e < s->batch && s->dirty_iv[(size_t)e].size() == 1 && s->dirty_iv[(size_t)e][0].first == 0 &&
             s->dirty_iv[(size_t)e][0].second >= P - 1)
        ++e;
      HIP_TRY(hipMemcpyAsync(s->d_prob + (int64_t)b * P, s->h_prob + (int64_t)b * P,
                             sizeof(double) * (size_t)P * (size_t)(e - b), hipMemcpyHostToDevice,
                             s->stream));
      for (int k = b; k < e; ++k)
        s->dirty_iv[(size_t)k].clear();
      b = e;
      continue;
    }
    for (const auto &r : iv) {
      const int64_t hi = std::min(r.second, P);
      HIP_TRY(hipMemcpyAsync(s->d_prob + (int64_t)b * P + r.first, s->h_prob + (int64_t)b * P + r.first,
                             sizeof(double) * (size_t)(hi - r.first), hipMemcpyHostToDevice,
                             s->stream));
    }
    iv.clear();
    ++b;
  }
  s->dirty = false;
  return GAR_HIP_OK;
}

// Host copy into the pinned staging area with non-temporal stores: the destination is written once and next read by
// the DMA engine, so it should not be pulled into (read-for-ownership) nor left in the caches of the packing core --
// a third less memory traffic than memcpy's cached stores on the 19 MB of a (56, 22), N = 256 problem (measured on
// the GPU box's host: 920 -> 560 us).  Only for problems that do not fit the last-level cache next to their source
// (gar_hip_solver::stage_nt): the 7.6 MB of the north-star problem pack at cache speed with plain stores (150 us
// against 440 us with non-temporal ones).
inline void stage_copy(double *dst, const double *src, size_t n, bool nt) {
#if defined(__SSE2__) && !defined(__HIP_DEVICE_COMPILE__)
  if (nt && n >= 512) {
    size_t i = 0;
    if (reinterpret_cast<uintptr_t>(dst) & 15) // 8-byte aligned at least: one scalar brings it to 16
      dst[i] = src[i], ++i;
    for (; i + 8 <= n; i += 8) {
      const __m128d a = _mm_loadu_pd(src + i), b = _mm_loadu_pd(src + i + 2), c = _mm_loadu_pd(src + i + 4),
                    d = _mm_loadu_pd(src + i + 6);
      _mm_stream_pd(dst + i, a);
      _mm_stream_pd(dst + i + 2, b);
      _mm_stream_pd(dst + i + 4, c);
      _mm_stream_pd(dst + i + 6, d);
    }
    for (; i < n; ++i)
      dst[i] = src[i];
    return;
  }
#endif
  std::memcpy(dst, src, sizeof(double) * n);
}

// Lower triangle of an r x r column-major block `src` (null: zeros), packed (gar_layout.h: gar_lower_index) as
// the lower triangle of an R x R block, R >= r, with `diag` on the diagonal beyond r: `dst` gets R (R + 1) / 2 doubles
inline void pack_lower(double *dst, const double *src, int r, int R, double diag) {
  for (int j = 0; j < R; ++j) {
    double *col = dst + gar_lower_index(R, j, j);
    const int n = R - j;
    if (j < r && src) {
      std::memcpy(col, src + (size_t)j * r + j, sizeof(double) * (size_t)(r - j));
      if (R > r)
        std::memset(col + (r - j), 0, sizeof(double) * (size_t)(R - r));
    } else {
      std::memset(col, 0, sizeof(double) * (size_t)n);
      if (j >= r)
        col[0] = diag;
    }
  }
}

// pipeline the upload: once the range being appended to has grown past 1 MiB it goes out
// (asynchronously, pinned -> HBM) while the caller packs the next knots -- one Newton
// iteration's 7.6 MB of knots then costs max(host packing, PCIe), not their sum
int flush_if_grown(gar_hip_solver *s, int b) {
  auto &iv = s->dirty_iv[(size_t)b];
  if (!iv.empty() && iv.back().second - iv.back().first >= (int64_t)(1 << 17)) {
    stage_fence();
    const int64_t lo = iv.back().first, hi = std::min(iv.back().second, s->prob_doubles);
    HIP_TRY(hipMemcpyAsync(s->d_prob + (int64_t)b * s->prob_doubles + lo,
                           s->h_prob + (int64_t)b * s->prob_doubles + lo,
                           sizeof(double) * (size_t)(hi - lo), hipMemcpyHostToDevice, s->stream));
    iv.pop_back();
  }
  return GAR_HIP_OK;
}

int write_block(gar_hip_solver *s, int b, int64_t off, const double *src, int64_t n) {
  if (n <= 0)
    return GAR_HIP_OK;
  if (s->staged) {
    double *dst = s->h_prob + (int64_t)b * s->prob_doubles + off;
    if (src)
      stage_copy(dst, src, (size_t)n, s->stage_nt);
    else
      std::memset(dst, 0, sizeof(double) * (size_t)n);
    mark_dirty(s, b, off, off + n);
    return flush_if_grown(s, b);
  }
  double *dst = s->d_prob + (int64_t)b * s->prob_doubles + off;
  if (src)
    HIP_TRY(hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s->stream));
  else
    HIP_TRY(hipMemsetAsync(dst, 0, sizeof(double) * (size_t)n, s->stream));
  return GAR_HIP_OK;
}

gar::LegParams make_leg_params(gar_hip_solver *s) {
  gar::LegParams Q{};
  const int N = s->horizon;
  Q.M.prob = s->d_prob;
  Q.M.fac = s->d_fac;
  Q.M.status = s->d_status;
  Q.M.slow = s->d_status + s->batch;
  Q.M.prob_stride = s->prob_doubles;
  Q.M.fac_stride = s->fac_doubles;
  Q.M.in_off0 = s->uni_in0;
  Q.M.in_rec = s->uni_in_rec;
  Q.M.in_offN = s->meta[N].in_off;
  Q.M.horizon = N;
  Q.M.trace = s->d_trace;
  Q.meta = s->d_meta;
  Q.skip = nullptr;
  if (s->fold) { // the wave-leg family sweeps the folded knots and keeps its own (nc = 0) factor records
    const gar_hip_solver *f = s->flay;
    Q.M.prob = s->d_prob2;
    Q.M.fac = s->d_fac2;
    Q.M.prob_stride = f->prob_doubles;
    Q.M.fac_stride = f->fac_doubles;
    Q.M.in_off0 = f->uni_in0;
    Q.M.in_rec = f->uni_in_rec;
    Q.M.in_offN = f->meta[N].in_off;
    Q.meta = s->d_meta2;
    Q.skip = s->d_status + s->batch + 4; // problems with D != 0: the generic leg kernels take them
  }
  Q.num_legs = s->num_legs;
  Q.leg_begin = s->leg_begin;
  Q.csol = s->d_csol;
  Q.sol = s->d_sol;
  Q.sol_stride = s->sol_doubles;
  Q.sol_u = (int)s->sol_u;
  Q.sol_l = (int)s->sol_l;
  Q.nc0 = s->nc0;
  Q.boundary = s->d_bound_local;
  Q.boundary_stride = (long long)s->legs_per_rank * s->tuple_doubles;
  Q.tuple_doubles = (int)s->tuple_doubles;
  {
    const int64_t nblk = 2 * s->num_legs, bs = (int64_t)s->nxb * s->nxb;
    Q.cinfo = s->d_cscratch ? s->d_cscratch + 4 * nblk * bs + 4 * nblk * s->nxb : nullptr;
    Q.cinfo_stride = s->cscratch_doubles;
  }
  return Q;
}

// gar_fold_constraints stages C and C / mu in LDS where they fit the default dynamic allocation (0: they do not)
size_t fold_lds_bytes(const gar_hip_solver *s) {
  int nxm = 0, ncm = 0;
  for (const auto &m : s->meta)
    nxm = std::max(nxm, (int)m.nx), ncm = std::max(ncm, (int)m.nc);
  const size_t b = sizeof(double) * (size_t)gar::fold_lds_doubles(nxm, ncm);
  return b <= 48 * 1024 ? b : 0;
}
gar::FoldParams make_fold_params(gar_hip_solver *s) {
  gar::FoldParams F{};
  const gar_hip_solver *f = s->flay;
  F.meta = s->d_meta;
  F.meta2 = s->d_meta2;
  F.prob = s->d_prob;
  F.prob2 = s->d_prob2;
  F.fac = s->d_fac;
  F.fac2 = s->d_fac2;
  F.sol = s->d_sol;
  F.prob_stride = s->prob_doubles;
  F.prob2_stride = f->prob_doubles;
  F.fac_stride = s->fac_doubles;
  F.fac2_stride = f->fac_doubles;
  F.sol_stride = s->sol_doubles;
  F.coupled = s->d_status + s->batch + 4;
  F.horizon = s->horizon;
  F.t2 = s->fb_t2 ? 1 : 0;
  int lo, hi, dummy;
  gar_get_work(s->horizon, s->leg_begin, s->num_legs, &lo, &dummy);
  gar_get_work(s->horizon, s->leg_end - 1, s->num_legs, &dummy, &hi);
  F.t_lo = lo;
  F.t_hi = hi;
  F.mueq = s->fold_mueq;
  F.qr_packed = s->qr_packed ? 1 : 0;
  F.lds = fold_lds_bytes(s) > 0 ? 1 : 0;
  return F;
}

// Folded solvers (gar_hip_solver::fold): the caller-visible factor records are formed on first request after a
// backward (nobody who only reads the solution pays for them), and which problems the generic kernels took
// (D != 0: row-major fb in their records instead of fbT2) is read back once.
int ensure_expanded(gar_hip_solver *s) {
  if (!s->fold)
    return GAR_HIP_OK;
  if (!s->fold_expanded) {
    hipLaunchKernelGGL(gar::gar_expand_constrained, dim3((unsigned)(s->horizon + 1), (unsigned)s->batch), dim3(256), 0,
                       s->stream, make_fold_params(s));
    HIP_TRY(hipGetLastError());
    s->fold_expanded = true;
  }
  if (!s->coupled_known) {
    s->h_coupled.assign((size_t)s->batch, 0);
    HIP_TRY(hipMemcpyAsync(s->h_coupled.data(), s->d_status + s->batch + 4, sizeof(int) * (size_t)s->batch,
                           hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->coupled_known = true;
  }
  return GAR_HIP_OK;
}
// fb / fth of problem b in the fbT2 device order?
inline bool records_t2(const gar_hip_solver *s, int b) {
  return s->fb_t2 && !(s->fold && s->coupled_known && s->h_coupled[(size_t)b] != 0);
}

#include "gar_launch.hpp"

void free_device(gar_hip_solver *s) {
  (void)hipFree(s->d_meta);
  (void)hipFree(s->d_prob);
  (void)hipFree(s->d_fac);
  (void)hipFree(s->d_sol);
  (void)hipFree(s->d_init);
  (void)hipFree(s->d_theta);
  (void)hipFree(s->d_status);
  if (s->d_kkt)
    (void)hipFree(s->d_kkt);
  s->d_kkt = nullptr;
  s->kkt_doubles = 0;
  (void)hipFree(s->d_bound_local);
  if (s->bound_all_owned)
    (void)hipFree(s->d_bound_all);
  (void)hipFree(s->d_csol);
  (void)hipFree(s->d_cscratch);

  (void)hipFree(s->d_prob2);
  (void)hipFree(s->d_fac2);
  (void)hipFree(s->d_cseg_resume);
  s->d_cseg_resume = nullptr;
  (void)hipFree(s->d_meta2);
  s->d_prob2 = s->d_fac2 = nullptr;
  s->d_meta2 = nullptr;
  (void)hipFree(s->d_trace);
  s->d_trace = nullptr;
  (void)hipFree(s->d_deriv_off);
  s->d_deriv_off = nullptr;
  (void)hipFree(s->d_gain_off);
  (void)hipFree(s->d_gains);
  if (s->h_results)
    (void)hipHostFree(s->h_results);
  s->d_gain_off = nullptr;
  s->d_gains = nullptr;
  s->h_results = nullptr;
  if (s->h_prob)
    (void)hipHostFree(s->h_prob);
  s->d_meta = nullptr;
  s->d_prob = s->d_fac = s->d_sol = s->d_init = s->d_theta = nullptr;
  s->d_status = nullptr;
  s->d_bound_local = s->d_bound_all = s->d_csol = s->d_cscratch = nullptr;
  s->h_prob = nullptr;
}

int allocate(gar_hip_solver *s) {
  const size_t B = (size_t)s->batch;
  HIP_TRY(gar_dev_malloc((void **)&s->d_meta, sizeof(gar_stage_meta) * s->meta.size()));
  HIP_TRY(hipMemcpy(s->d_meta, s->meta.data(), sizeof(gar_stage_meta) * s->meta.size(),
                    hipMemcpyHostToDevice));
  HIP_TRY(gar_dev_malloc((void **)&s->d_prob, sizeof(double) * (size_t)s->prob_doubles * B));
  HIP_TRY(hipMemset(s->d_prob, 0, sizeof(double) * (size_t)s->prob_doubles * B));
  HIP_TRY(gar_dev_malloc((void **)&s->d_fac, sizeof(double) * (size_t)s->fac_doubles * B));
  HIP_TRY(hipMemset(s->d_fac, 0, sizeof(double) * (size_t)s->fac_doubles * B));
  HIP_TRY(gar_dev_malloc((void **)&s->d_sol, sizeof(double) * (size_t)s->sol_doubles * B));
  HIP_TRY(hipMemset(s->d_sol, 0, sizeof(double) * (size_t)s->sol_doubles * B));
  HIP_TRY(gar_dev_malloc((void **)&s->d_init, sizeof(double) * (size_t)s->init_doubles * B));
  HIP_TRY(hipMemset(s->d_init, 0, sizeof(double) * (size_t)s->init_doubles * B));
  HIP_TRY(gar_dev_malloc((void **)&s->d_theta, sizeof(double) * (size_t)std::max(s->nth0, 1) * B));
  // per-problem failure flags, then the four slow-path counters (MfmaParams::slow), then MfmaParams::resume
  HIP_TRY(gar_dev_malloc((void **)&s->d_status, sizeof(int) * (2 * B + 4)));
  HIP_TRY(hipMemset(s->d_status, 0, sizeof(int) * (2 * B + 4)));
  if (s->num_legs > 1) {
    const int chunk = s->legs_per_rank; // >= this rank's own leg count; equal-sized chunks for the all-gather
    const int nblk = 2 * s->num_legs;
    const size_t bs = (size_t)s->nxb * s->nxb;
    HIP_TRY(gar_dev_malloc((void **)&s->d_bound_local, sizeof(double) * s->tuple_doubles * chunk * B));
    HIP_TRY(hipMemset(s->d_bound_local, 0, sizeof(double) * s->tuple_doubles * chunk * B));
    if (s->world == 1) {
      s->d_bound_all = s->d_bound_local;
      s->bound_all_owned = false;
    } else {
      HIP_TRY(gar_dev_malloc((void **)&s->d_bound_all,
                        sizeof(double) * s->tuple_doubles * chunk * s->world * B));
      s->bound_all_owned = true;
    }
    HIP_TRY(gar_dev_malloc((void **)&s->d_csol, sizeof(double) * (size_t)nblk * s->nxb * B));
    s->cscratch_doubles = (int64_t)(4 * nblk * bs + 4 * (size_t)nblk * s->nxb + 4);
    HIP_TRY(gar_dev_malloc((void **)&s->d_cscratch, sizeof(double) * (size_t)s->cscratch_doubles * B));
    // (the info slots behind the blocks -- residual, steps, scale, resolved -- are read by the gar_hip_condensed_*
    // getters: defined before the first solve)
    HIP_TRY(hipMemset(s->d_cscratch, 0, sizeof(double) * (size_t)s->cscratch_doubles * B));
    s->cond_lds_doubles = (int)(3 * bs + 4 * s->nxb + 2 + (s->nxb + 16) / 2 + 2 + (s->nxb < 9 ? 9 * s->nxb : 0));
    {
      const char *cr = gar_option("GAR_HIP_CONDENSED_REDUCED");
      s->cond_reduced = !(cr && cr[0] == '0') && s->nx0 == s->nxb &&
                        (size_t)gar::gar_condensed_leg_lds_doubles(s->nxb) * sizeof(double) <= 160 * 1024;
      // cyclic reduction of the reduced system: log2 J dependent steps instead of J; below 4 legs the chain is as short
      const char *cc = gar_option("GAR_HIP_CONDENSED_CR");
      const int cr_min = cc && cc[0] ? std::atoi(cc) : 4;
      s->cond_cr = s->cond_reduced && cr_min >= 1 && s->num_legs >= std::max(cr_min, 2) && s->nc0 <= s->nxb &&
                   (size_t)gar::gar_condensed_cr_back_lds_doubles(s->nxb, s->num_legs) * sizeof(double) <= 160 * 1024;
    }
  }
  if (s->seg_bwd_kernel) {
    const gar_hip_solver *f = s->flay;
    HIP_TRY(gar_dev_malloc((void **)&s->d_fac2, sizeof(double) * (size_t)f->fac_doubles * B));
    HIP_TRY(hipMemset(s->d_fac2, 0, sizeof(double) * (size_t)f->fac_doubles * B));
    HIP_TRY(gar_dev_malloc((void **)&s->d_meta2, sizeof(gar_stage_meta) * f->meta.size()));
    HIP_TRY(hipMemcpy(s->d_meta2, f->meta.data(), sizeof(gar_stage_meta) * f->meta.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipFuncSetAttribute((const void *)s->seg_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(s->seg_lds_doubles * sizeof(double))));
    HIP_TRY(hipFuncSetAttribute((const void *)gar::gar_leg_param_chain, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(gar::leg_chain_lds_doubles(s->dims5[0]) * sizeof(double))));
    HIP_TRY(hipFuncSetAttribute((const void *)gar::gar_leg_param_stage, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(gar::leg_stage_lds_doubles(s->dims5[0], s->dims5[1]) * sizeof(double))));
  }
  if (s->fold) {
    const gar_hip_solver *f = s->flay;
    HIP_TRY(gar_dev_malloc((void **)&s->d_prob2, sizeof(double) * (size_t)f->prob_doubles * B));
    HIP_TRY(hipMemset(s->d_prob2, 0, sizeof(double) * (size_t)f->prob_doubles * B));
    HIP_TRY(gar_dev_malloc((void **)&s->d_fac2, sizeof(double) * (size_t)f->fac_doubles * B));
    HIP_TRY(hipMemset(s->d_fac2, 0, sizeof(double) * (size_t)f->fac_doubles * B));
    HIP_TRY(gar_dev_malloc((void **)&s->d_meta2, sizeof(gar_stage_meta) * f->meta.size()));
    HIP_TRY(hipMemcpy(s->d_meta2, f->meta.data(), sizeof(gar_stage_meta) * f->meta.size(), hipMemcpyHostToDevice));
    if (s->cseg_on) {
      const size_t units = B * (size_t)s->legs_per_rank;
      HIP_TRY(gar_dev_malloc((void **)&s->d_cseg_resume, sizeof(int) * units));
      HIP_TRY(hipMemset(s->d_cseg_resume, 0xff, sizeof(int) * units));
      for (auto k : s->cseg.backward)
        HIP_TRY(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(s->cseg.backward_lds_doubles * sizeof(double))));
      HIP_TRY(hipFuncSetAttribute((const void *)s->cseg.leg_end, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(s->cseg.leg_end_lds_doubles * sizeof(double))));
      HIP_TRY(hipFuncSetAttribute((const void *)s->cseg.chain, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(s->cseg.chain_lds_doubles * sizeof(double))));
      HIP_TRY(hipFuncSetAttribute((const void *)s->cseg.stage, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(s->cseg.stage_lds_doubles * sizeof(double))));
    }
  }
  s->fold_expanded = s->coupled_known = false;
  const size_t staging = sizeof(double) * (size_t)s->prob_doubles * B;
  if (staging <= ((size_t)1 << 30)) {
    HIP_TRY(gar_host_malloc((void **)&s->h_prob, staging, hipHostMallocDefault));
    std::memset(s->h_prob, 0, staging);
    s->staged = true;
    const char *nt = gar_option("GAR_HIP_STAGE_NT");
    s->stage_nt = nt && nt[0] ? nt[0] == '1' : sizeof(double) * (size_t)s->prob_doubles >= ((size_t)12 << 20);
  }
  s->dirty_iv.assign(B, {});
  s->dirty = false;
  if (s->dense) {
    HIP_TRY(hipFuncSetAttribute((const void *)gar::gar_backward_dense,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(s->dense_lds.total * sizeof(double))));
    return GAR_HIP_OK;
  }
  if (s->mfma_kernel)
    HIP_TRY(hipFuncSetAttribute((const void *)s->mfma_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(s->mfma_lds_doubles * sizeof(double))));
  if (s->cyc_setup_kernel) {
    const int lds = (int)(s->cyc_lds_doubles * sizeof(double));
    HIP_TRY(hipFuncSetAttribute((const void *)s->cyc_setup_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds + (int)(s->cyc_block_doubles * sizeof(double))));
    HIP_TRY(hipFuncSetAttribute((const void *)s->cyc_reduce_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * lds + 512 + (int)(s->cyc_block_doubles * sizeof(double))));
    HIP_TRY(hipFuncSetAttribute((const void *)s->cyc_top_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  if (s->cond_wave_kernel)
    HIP_TRY(hipFuncSetAttribute((const void *)s->cond_wave_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(s->cond_wave_lds_doubles * sizeof(double))));
  if (s->leg_bwd_kernel)
    HIP_TRY(hipFuncSetAttribute((const void *)s->leg_bwd_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(s->leg_lds_doubles * sizeof(double))));
  if (s->wave_kernel)
    HIP_TRY(hipFuncSetAttribute((const void *)s->wave_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(s->wave_lds_doubles * s->waves_per_block * sizeof(double))));
  for (auto k : {s->wave_coupled_kernel, s->wave_bk_kernel})
    if (k)
      HIP_TRY(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(s->wave_lds_doubles * s->waves_per_block * sizeof(double))));
  if (s->lds_error.empty())
    HIP_TRY(hipFuncSetAttribute((const void *)gar::gar_initial_generic,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(s->lds.total * sizeof(double))));
  if (s->n0 <= 128)
    HIP_TRY(hipFuncSetAttribute(
        (const void *)gar::gar_initial_wave, hipFuncAttributeMaxDynamicSharedMemorySize,
        (int)(gar::gar_initial_wave_lds_doubles(s->n0, s->nth0) * sizeof(double))));
  // > 64 KiB of dynamic LDS needs the opt-in attribute
  if (s->lds_error.empty())
    HIP_TRY(hipFuncSetAttribute((const void *)gar::gar_backward_generic,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(s->lds.total * sizeof(double))));
  if (s->num_legs > 1)
    HIP_TRY(hipFuncSetAttribute((const void *)gar::gar_condensed_generic,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(s->cond_lds_doubles * sizeof(double))));
  if (s->num_legs > 1 && s->cond_reduced)
    HIP_TRY(hipFuncSetAttribute((const void *)gar::gar_condensed_leg_eliminate,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(gar::gar_condensed_leg_lds_doubles(s->nxb) * sizeof(double))));
  if (s->num_legs > 1 && s->cond_cr) {
    HIP_TRY(hipFuncSetAttribute((const void *)gar::gar_condensed_cr_eliminate,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(gar::gar_condensed_leg_lds_doubles(s->nxb) * sizeof(double))));
    HIP_TRY(hipFuncSetAttribute((const void *)gar::gar_condensed_cr_update, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)std::min<size_t>(160 * 1024, (size_t)gar::gar_condensed_cr_update_lds_doubles(s->nxb, 1) *
                                                                      sizeof(double))));
    HIP_TRY(hipFuncSetAttribute((const void *)gar::gar_condensed_cr_assemble, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((size_t)s->nxb * s->nxb * sizeof(double))));
    HIP_TRY(hipFuncSetAttribute((const void *)gar::gar_condensed_cr_back, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(gar::gar_condensed_cr_back_lds_doubles(s->nxb, s->num_legs) * sizeof(double))));
  }
  return GAR_HIP_OK;
}

} // namespace

// ---- padding helpers (gar_hip_solver::padded) ------------------------------------------------------------------
namespace {
// r x c column-major `src` (null: zeros) into the top-left corner of an R x C column-major block; `diag` on the
// part of the diagonal beyond (r, c)
const double *padded_block(std::vector<double> &buf, const double *src, int r, int c, int R, int C, double diag) {
  buf.assign((size_t)R * C, 0.0);
  if (src)
    for (int j = 0; j < c; ++j)
      std::memcpy(&buf[(size_t)j * R], src + (size_t)j * r, sizeof(double) * (size_t)r);
  if (diag != 0.0)
    for (int i = std::min(r, c); i < std::min(R, C); ++i)
      buf[(size_t)i * R + i] = diag;
  return buf.data();
}
// the leading r x r part of a symmetric R x R block stored as its packed lower triangle (gar_layout.h), as a full
// column-major r x r block (both triangles)
void unpack_lower(double *dst, const double *src, int r, int R) {
  for (int j = 0; j < r; ++j)
    for (int i = j; i < r; ++i)
      dst[(size_t)j * r + i] = dst[(size_t)i * r + j] = src[gar_lower_index(R, i, j)];
}
// rows [0, unu) and [NU, NU + unx) of a gain block with NU + NX (+...) rows: the real controls and states
inline int gain_row(const gar_hip_solver *s, int r, int nu_dev) {
  const int unu = nu_dev > 0 ? s->unu : 0;
  return r < unu ? r : r - unu + nu_dev;
}
// the caller's solution arrays out of one device solution record
void strip_solution(const gar_hip_solver *s, const double *rec, double *xs, double *us, double *vs, double *lbdas) {
  const int N = s->horizon, nx = s->unx;
  (void)vs; // padding applies to unconstrained problems only
  for (int t = 0; t <= N; ++t) {
    const gar_stage_meta &m = s->meta[t];
    if (xs)
      std::memcpy(xs + (size_t)t * nx, rec + m.x_off, sizeof(double) * (size_t)nx);
    if (us && m.nu > 0)
      std::memcpy(us + (size_t)t * s->unu, rec + m.u_off, sizeof(double) * (size_t)s->unu);
    if (lbdas) {
      if (t == 0)
        std::memcpy(lbdas, rec + m.l_off, sizeof(double) * (size_t)s->user_nc0);
      else
        std::memcpy(lbdas + s->user_nc0 + (size_t)(t - 1) * nx, rec + m.l_off, sizeof(double) * (size_t)nx);
    }
  }
}
// one device solution record -> the caller's record layout (ulay)
void strip_solution_rec(const gar_hip_solver *s, const double *dev, double *rec) {
  const gar_hip_solver *u = s->ulay;
  strip_solution(s, dev, rec + u->sol_x, rec + u->sol_u, rec + u->sol_v, rec + u->sol_l);
}
} // namespace

static int fetch_results_impl(gar_hip_solver *s, int b, int what, int t_lo, int t_hi, double *gains_base, bool sync);
static int prefetch_impl(gar_hip_solver *s, int b);
static int upload_stage_impl(gar_hip_solver *s, int b, int t, const double *Q, const double *S, const double *R,
                             const double *q, const double *r, const double *A, const double *B, const double *f,
                             const double *C, const double *D, const double *d, const double *Gth, const double *Gx,
                             const double *Gu, const double *Gv, const double *gamma);
// (see gar_hip_solver::term_grown) the caller's dimensions as the library keeps them
void normalise_terminal(gar_hip_solver *s) {
  const int N = s->horizon;
  if (s->dense || N < 1)
    return;
  int32_t *d = &s->user_dims5[5 * (size_t)N];
  const int32_t *prev = &s->user_dims5[5 * (size_t)(N - 1)];
  if (d[1] == 0 && d[3] == 0 && d[4] == 0 && d[0] > 0 && d[0] == prev[3]) {
    d[3] = d[0];
    s->term_grown = true;
    s->term_zeros.assign((size_t)d[0] * (size_t)d[0], 0.0);
  }
}

#include "gar_multi.hpp"

extern "C" {

const char *gar_hip_version(void) { return "gar-hip 0.1 (gfx950)"; }
const char *gar_hip_last_error(void) { return g_last_error.c_str(); }

double gar_hip_stream_ceiling_ms(int device, int batch, int horizon, int64_t in_bytes_per_stage,
                                 int64_t out_bytes_per_stage, int reps) {
  // reps < 0: the walk with TWO records requested ahead (gar_stream_sweep2), |reps| repetitions
  const bool reps_ahead2 = reps < 0;
  if (reps < 0)
    reps = -reps;
  const char *lay = std::getenv("GAR_STREAM_LAYOUT"); // "stage": the layout probe of gar_stream_sweep
  const int stage_major = (lay && std::string(lay) == "stage") ? 1 : 0;
  // (north-star records fit the compiled piece counts: <= 32 x 64 x 16 B = 32 KB in, 28 KB out per stage)
  const int in_pieces = (int)((in_bytes_per_stage + 15) / 16), out_pieces = (int)((out_bytes_per_stage + 15) / 16);
  if (batch <= 0 || horizon <= 0 || in_pieces <= 0 || out_pieces <= 0 || in_pieces > 32 * 64 || out_pieces > 28 * 64 ||
      reps <= 0)
    return -1.0;
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess)
    return -1.0;
  gar::gar_double2 *in = nullptr, *out = nullptr;
  double *sink = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  double best = -1.0;
  const size_t nin = (size_t)batch * horizon * in_pieces * 16, nout = (size_t)batch * horizon * out_pieces * 16;
  if (gar_dev_malloc((void **)&in, nin) == hipSuccess && gar_dev_malloc((void **)&out, nout) == hipSuccess &&
      gar_dev_malloc((void **)&sink, (size_t)batch * 64 * 8) == hipSuccess && hipMemset(in, 0, nin) == hipSuccess &&
      hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
    for (int r = 0; r < reps + 1; ++r) { // first launch: warm-up
      (void)hipEventRecord(e0, nullptr);
      if (reps_ahead2)
        hipLaunchKernelGGL((gar::gar_stream_sweep2<32, 28>), dim3((unsigned)batch), dim3(64), 0, nullptr, in, out, sink,
                           horizon, in_pieces, out_pieces);
      else
        hipLaunchKernelGGL((gar::gar_stream_sweep<32, 28>), dim3((unsigned)batch), dim3(64), 0, nullptr, in, out, sink,
                           horizon, in_pieces, out_pieces, stage_major);
      (void)hipEventRecord(e1, nullptr);
      float ms = 0.f;
      if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
        best = -1.0;
        break;
      }
      if (r > 0 && (best < 0.0 || ms < best))
        best = ms;
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(in);
  (void)hipFree(out);
  (void)hipFree(sink);
  (void)hipSetDevice(prev);
  return best;
}

double gar_hip_copy_ceiling_ms(int device, int64_t bytes_moved, int reps) {
  // bytes_moved = read + written: bytes_moved / 2 are copied
  const long long n = (long long)(bytes_moved / 2 / 16);
  if (n <= 0 || reps <= 0)
    return -1.0;
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess)
    return -1.0;
  gar::gar_double2 *src = nullptr, *dst = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  double best = -1.0;
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
  if (gar_dev_malloc((void **)&src, (size_t)n * 16) == hipSuccess && gar_dev_malloc((void **)&dst, (size_t)n * 16) == hipSuccess &&
      hipMemset(src, 0, (size_t)n * 16) == hipSuccess && hipEventCreate(&e0) == hipSuccess &&
      hipEventCreate(&e1) == hipSuccess) {
    for (int r = 0; r < reps + 1; ++r) { // first launch: warm-up
      (void)hipEventRecord(e0, nullptr);
      hipLaunchKernelGGL(gar::gar_plain_copy, dim3((unsigned)(cus * 64)), dim3(256), 0, nullptr, src, dst, n);
      (void)hipEventRecord(e1, nullptr);
      float ms = 0.f;
      if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
        best = -1.0;
        break;
      }
      if (r > 0 && (best < 0.0 || ms < best))
        best = ms;
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(src);
  (void)hipFree(dst);
  (void)hipSetDevice(prev);
  return best;
}

int gar_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

int64_t gar_hip_knot_doubles(const int32_t d[5]) {
  return gar_knot_doubles(d[0], d[1], d[2], d[3], d[4]);
}
int64_t gar_hip_factor_doubles(const int32_t d[5]) {
  return gar_factor_doubles(d[0], d[1], d[2], d[3], d[4]);
}

namespace {
gar_hip_solver *create_impl(int device, int horizon, const int32_t *dims5, int nc0, int batch,
                            int num_legs, int rank, int world, bool dense) {
  if (horizon < 0 || !dims5 || nc0 < 0 || batch < 1 || num_legs < 1 || world < 1 || rank < 0 || rank >= world ||
      (num_legs == 1 && world != 1)) {
    fail(GAR_HIP_ERR_ARG, "gar_hip_solver_create: bad argument");
    return nullptr;
  }
  if (gar_hip_device_count() <= device) {
    fail(GAR_HIP_ERR_DEVICE, "gar_hip_solver_create: no such HIP device (the HIP "
                             "backend has no CPU fallback)");
    return nullptr;
  }
  gar_hip_solver *s = new gar_hip_solver();
  s->device = device;
  s->horizon = horizon;
  s->user_nc0 = nc0;
  s->batch = batch;
  s->num_legs = num_legs;
  s->rank = rank;
  s->world = world;
  s->dense = dense;
  s->user_dims5.assign(dims5, dims5 + 5 * (horizon + 1));
  normalise_terminal(s);
  DeviceGuard guard_(device); // the caller's current device is restored on return
  {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != device) {
      fail(GAR_HIP_ERR_DEVICE, "hipSetDevice failed");
      delete s;
      return nullptr;
    }
  }
  if (configure(s) != GAR_HIP_OK) {
    delete s->ulay;
    delete s->flay;
    delete s;
    return nullptr;
  }
  if (hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking) != hipSuccess) {
    fail(GAR_HIP_ERR_DEVICE, "hipStreamCreate failed");
    delete s->ulay;
    delete s->flay;
    delete s;
    return nullptr;
  }
  s->stream = s->own_stream;
  if (allocate(s) != GAR_HIP_OK) {
    free_device(s);
    (void)hipStreamDestroy(s->own_stream);
    delete s->ulay;
    delete s->flay;
    delete s;
    return nullptr;
  }
  { // the default schedule is the library's own choice (GAR_HIP_PIPELINE = auto | 0 | 2; gar_hip_set_pipeline overrides)
    const char *pl = gar_option("GAR_HIP_PIPELINE");
    const int want = pl ? (pl[0] == '0' ? 0 : pl[0] == '2' ? 2 : -1) : -1;
    if (want != 0 && gar_hip_set_pipeline(s, want) != GAR_HIP_OK)
      s->pipe_halves = s->pipe_requested = 0; // (an explicit 2 on a shape without the family: plain, not an error at create)
  }
  return s;
}
} // namespace

gar_hip_solver *gar_hip_solver_create_ranked(int device, int horizon, const int32_t *dims5, int nc0, int batch,
                                             int num_legs, int rank, int world) {
  return create_impl(device, horizon, dims5, nc0, batch, num_legs, rank, world, false);
}

// legs [leg_begin, leg_end) of an EVEN split (every rank the same number of legs); any split: _ranked
gar_hip_solver *gar_hip_solver_create_sharded(int device, int horizon, const int32_t *dims5,
                                              int nc0, int batch, int num_legs, int leg_begin,
                                              int leg_end) {
  const int local = leg_end - leg_begin;
  if (num_legs < 1 || local < 1 || leg_begin < 0 || leg_end > num_legs || num_legs % local != 0 || leg_begin % local != 0) {
    fail(GAR_HIP_ERR_ARG, "gar_hip_solver_create_sharded: legs must be split evenly over ranks "
                          "(use gar_hip_solver_create_ranked for any split)");
    return nullptr;
  }
  return create_impl(device, horizon, dims5, nc0, batch, num_legs, leg_begin / local, num_legs / local, false);
}

// RiccatiSolverDense (gar/dense-riccati.hpp:19-56): serial in time, any dimensions
gar_hip_solver *gar_hip_solver_create_dense(int device, int horizon, const int32_t *dims5, int nc0,
                                            int batch) {
  return create_impl(device, horizon, dims5, nc0, batch, 1, 0, 1, true);
}

gar_hip_solver *gar_hip_solver_create(int device, int horizon, const int32_t *dims5, int nc0,
                                      int batch, int num_legs) {
  return create_impl(device, horizon, dims5, nc0, batch, num_legs, 0, 1, false);
}

void gar_hip_solver_destroy(gar_hip_solver *s) {
  if (!s)
    return;
  if (s->multi) {
    multi_destroy(s);
    return;
  }
  GAR_GUARD(s);
  (void)hipStreamSynchronize(s->stream);
  for (int h = 0; h < 2; ++h)
    if (s->pipe_stream[h]) {
      (void)hipStreamSynchronize(s->pipe_stream[h]);
      (void)hipStreamDestroy(s->pipe_stream[h]);
      (void)hipEventDestroy(s->pipe_evB[h]);
      (void)hipEventDestroy(s->pipe_evF[h]);
      for (auto &e : s->pipe_evT[h])
        (void)hipEventDestroy(e);
    }
  if (s->pipe_evFork)
    (void)hipEventDestroy(s->pipe_evFork);
  free_device(s);
  if (s->own_stream)
    (void)hipStreamDestroy(s->own_stream);
  for (auto &e : s->ev)
    if (e)
      (void)hipEventDestroy(e);
  if (s->aux_stream) {
    (void)hipStreamSynchronize(s->aux_stream);
    (void)hipStreamDestroy(s->aux_stream);
    (void)hipEventDestroy(s->ev_main);
    (void)hipEventDestroy(s->ev_pref);
  }
  if (s->h_status) {
    (void)hipHostFree(s->h_status);
    (void)hipEventDestroy(s->ev_status);
    (void)hipEventDestroy(s->ev_sol);
  }
  delete s->ulay;
  delete s->flay;
  delete s;
}

int gar_hip_set_stream(gar_hip_solver *s, void *hip_stream) {
  GAR_GUARD(s);
  if (!s)
    return fail(GAR_HIP_ERR_ARG, "null solver");
  GAR_MULTI(s, fail(GAR_HIP_ERR_UNSUPPORTED, "a multi-device solver runs on one stream per device, its own"));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->stream = hip_stream ? (hipStream_t)hip_stream : s->own_stream;
  return GAR_HIP_OK;
}

int gar_hip_sync(gar_hip_solver *s) {
  GAR_GUARD(s);
  if (!s)
    return fail(GAR_HIP_ERR_ARG, "null solver");
  GAR_MULTI(s, multi_sync(s));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return GAR_HIP_OK;
}

// (the caller's records: under padding they are laid out by the caller's dimensions, `ulay`)
int64_t gar_hip_problem_doubles(const gar_hip_solver *s) { return s ? (s->ulay ? s->ulay->prob_doubles : s->prob_doubles) : 0; }
int64_t gar_hip_factors_doubles(const gar_hip_solver *s) { return s ? s->fac_doubles : 0; }
int64_t gar_hip_solution_doubles(const gar_hip_solver *s) { return s ? (s->ulay ? s->ulay->sol_doubles : s->sol_doubles) : 0; }
int gar_hip_batch(const gar_hip_solver *s) { return s ? s->batch : 0; }
int gar_hip_horizon(const gar_hip_solver *s) { return s ? s->horizon : -1; }
const char *gar_hip_kernel_name(const gar_hip_solver *s) {
  return s ? s->kernel_name.c_str() : "";
}

int gar_hip_suggest_num_legs(int horizon, int nx, int nu) {
  (void)nu;
  if (horizon < 1)
    return 1;
  const int per_leg = nx <= 36 ? 4 : 8; // stages per leg (measured at N = 256: include/gar_hip.h)
  const int legs = (horizon + per_leg - 1) / per_leg;
  return legs < 2 ? 2 : (legs > horizon + 1 ? horizon + 1 : legs);
}

const char *gar_hip_condensed_solver_name(const gar_hip_solver *s) {
  if (!s || s->num_legs < 2)
    return "";
  GAR_MULTI(s, gar_hip_condensed_solver_name(s->multi->subs[0]));
  if (s->cyc_setup_kernel)
    return "cyclic"; // gar_cyclic.hpp (specialised leg families), the wave-scope chain gated behind it
  if (s->cond_wave_kernel)
    return "chain";
  if (s->cond_reduced)
    return s->cond_cr ? "reduced+cyclic" : "reduced+chain"; // gar_generic.hpp / gar_condensed_cr.hpp
  return "generic-chain";
}

int gar_hip_stage_offsets(const gar_hip_solver *s, int t, int64_t out[6]) {
  if (int rc = check_bt(s, 0, t))
    return rc;
  const gar_stage_meta &m = s->ulay ? s->ulay->meta[t] : s->meta[t];
  out[0] = m.in_off;
  out[1] = s->meta[t].fac_off; // factor records exist on the device only
  out[2] = m.x_off;
  out[3] = m.u_off;
  out[4] = m.v_off;
  out[5] = m.l_off;
  return GAR_HIP_OK;
}

int gar_hip_init_offsets(const gar_hip_solver *s, int64_t out[2]) {
  if (!s)
    return fail(GAR_HIP_ERR_ARG, "null solver");
  out[0] = s->ulay ? s->ulay->G0_off : s->G0_off;
  out[1] = s->ulay ? s->ulay->g0_off : s->g0_off;
  return GAR_HIP_OK;
}

static int upload_stage_dev(gar_hip_solver *s, int b, int t, const double *Q, const double *S,
                         const double *R, const double *q, const double *r, const double *A,
                         const double *B, const double *f, const double *C, const double *D,
                         const double *d, const double *Gth, const double *Gx,
                         const double *Gu, const double *Gv, const double *gamma) {
  const gar_stage_meta &m = s->meta[t];
  const int nth_st = (m.flags & GAR_KNOT_HAS_PARAM) ? m.nth : 0;
  const gar_knot_offsets o = gar_knot_layout(m.nx, m.nu, m.nc, m.nx2, nth_st);
  const int64_t base = m.in_off;
  const int nx = m.nx, nu = m.nu, nc = m.nc, nx2 = m.nx2;
  if (!Q || !q || !A || !f || (nu > 0 && (!S || !R || !r || !B)))
    return fail(GAR_HIP_ERR_ARG, "gar_hip_upload_stage: null block");
  int rc = 0;
  // staged: the blocks are copied into the pinned record and the KNOT becomes one dirty range (the holes the packed
  // triangles of Q and R leave inside their blocks included: block-by-block ranges would not merge, and a problem
  // would go out as two small copies per knot instead of 1 MiB pieces)
  auto put = [&](int64_t off, const double *src, int64_t n) {
    if (n <= 0)
      return;
    if (!s->staged) {
      rc |= write_block(s, b, off, src, n);
      return;
    }
    double *dst = s->h_prob + (int64_t)b * s->prob_doubles + off;
    if (src)
      stage_copy(dst, src, (size_t)n, s->stage_nt);
    else
      std::memset(dst, 0, sizeof(double) * (size_t)n);
  };
  if (s->qr_packed && t < s->horizon) { // Q, R: their lower triangles, packed, in the first n (n + 1) / 2 doubles of the block
    if (s->staged) {
      double *rec = s->h_prob + (int64_t)b * s->prob_doubles + base;
      pack_lower(rec + o.Q, Q, nx, nx, 0.0);
      pack_lower(rec + o.R, R, nu, nu, 0.0);
    } else {
      thread_local std::vector<double> pq, pr;
      pq.resize((size_t)nx * (nx + 1) / 2);
      pr.resize((size_t)nu * (nu + 1) / 2);
      pack_lower(pq.data(), Q, nx, nx, 0.0);
      pack_lower(pr.data(), R, nu, nu, 0.0);
      put(base + o.Q, pq.data(), (int64_t)pq.size());
      put(base + o.R, pr.data(), (int64_t)pr.size());
    }
  } else {
    put(base + o.Q, Q, (int64_t)nx * nx);
    put(base + o.R, R, (int64_t)nu * nu);
  }
  put(base + o.S, S, (int64_t)nx * nu);
  put(base + o.q, q, nx);
  put(base + o.r, r, nu);
  put(base + o.A, A, (int64_t)nx2 * nx);
  put(base + o.B, B, (int64_t)nx2 * nu);
  put(base + o.f, f, nx2);
  put(base + o.C, C, (int64_t)nc * nx);
  put(base + o.D, D, (int64_t)nc * nu);
  put(base + o.d, d, nc);
  if (nth_st > 0) {
    put(base + o.Gth, Gth, (int64_t)nth_st * nth_st);
    put(base + o.Gx, Gx, (int64_t)nx * nth_st);
    put(base + o.Gu, Gu, (int64_t)nu * nth_st);
    put(base + o.Gv, Gv, (int64_t)nc * nth_st);
    put(base + o.gamma, gamma, nth_st);
  }
  if (s->staged) {
    mark_dirty(s, b, base, base + gar_knot_doubles(nx, nu, nc, nx2, nth_st));
    return flush_if_grown(s, b);
  }
  return rc ? GAR_HIP_ERR_DEVICE : GAR_HIP_OK;
}

static int set_init_dev(gar_hip_solver *s, int b, const double *G0, const double *g0) {
  if (s->nc0 > 0 && (!G0 || !g0))
    return fail(GAR_HIP_ERR_ARG, "gar_hip_set_init: null block");
  int rc = write_block(s, b, s->G0_off, G0, (int64_t)s->nc0 * s->nx0);
  rc |= write_block(s, b, s->g0_off, g0, s->nc0);
  return rc ? GAR_HIP_ERR_DEVICE : GAR_HIP_OK;
}

int gar_hip_upload_packed(gar_hip_solver *s, int b0, int nb, const double *packed) {
  GAR_GUARD(s);
  GAR_MULTI(s, multi_upload_packed(s, b0, nb, packed));
  if (!s || !packed || b0 < 0 || nb < 0 || b0 + nb > s->batch)
    return fail(GAR_HIP_ERR_ARG, "gar_hip_upload_packed: bad argument");
  if (s->padded || s->qr_packed) { // the caller's records: knot by knot through the padding / packing path
    const gar_hip_solver *u = s->ulay ? s->ulay : s;
    for (int b = b0; b < b0 + nb; ++b) {
      const double *rec = packed + (int64_t)(b - b0) * u->prob_doubles;
      for (int t = 0; t <= s->horizon; ++t) {
        // (a padded solver is unconstrained and unparameterised; one that only packs Q / R may carry both)
        const gar_stage_meta &m = u->meta[t];
        const int nth_st = (m.flags & GAR_KNOT_HAS_PARAM) ? m.nth : 0;
        const gar_knot_offsets o = gar_knot_layout(m.nx, m.nu, m.nc, m.nx2, nth_st);
        const double *k = rec + m.in_off;
        if (int rc = gar_hip_upload_stage(s, b, t, k + o.Q, k + o.S, k + o.R, k + o.q, k + o.r, k + o.A, k + o.B, k + o.f,
                                          k + o.C, k + o.D, k + o.d, k + o.Gth, k + o.Gx, k + o.Gu, k + o.Gv, k + o.gamma))
          return rc;
      }
      if (int rc = gar_hip_set_init(s, b, rec + u->G0_off, rec + u->g0_off))
        return rc;
    }
    return GAR_HIP_OK;
  }
  const size_t bytes = sizeof(double) * (size_t)s->prob_doubles * nb;
  if (s->staged) {
    std::memcpy(s->h_prob + (int64_t)b0 * s->prob_doubles, packed, bytes);
    for (int b = b0; b < b0 + nb; ++b) {
      s->dirty_iv[(size_t)b].clear();
      mark_dirty(s, b, 0, s->prob_doubles);
    }
    return GAR_HIP_OK;
  }
  HIP_TRY(hipMemcpyAsync(s->d_prob + (int64_t)b0 * s->prob_doubles, packed, bytes,
                         hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return GAR_HIP_OK;
}

int gar_hip_upload_packed_device(gar_hip_solver *s, int b0, int nb, const double *packed_dev) {
  GAR_GUARD(s);
  GAR_MULTI(s, fail(GAR_HIP_ERR_UNSUPPORTED, "device-resident upload into a multi-device solver: the records live on several devices"));
  if (!s || !packed_dev || b0 < 0 || nb < 0 || b0 + nb > s->batch)
    return fail(GAR_HIP_ERR_ARG, "gar_hip_upload_packed_device: bad argument");
  if (int rc = commit(s)) // staged host data first, then the device copy wins
    return rc;
  HIP_TRY(hipMemcpyAsync(s->d_prob + (int64_t)b0 * s->prob_doubles, packed_dev,
                         sizeof(double) * (size_t)s->prob_doubles * nb, hipMemcpyDeviceToDevice,
                         s->stream));
  // (the staging area stays in use: commit() flushes only the ranges the host writes later)
  return GAR_HIP_OK;
}

int gar_hip_upload_packed_device_fmt(gar_hip_solver *s, int b0, int nb, const double *packed_dev, int record_format) {
  if (s && ((record_format & GAR_HIP_FMT_QR_PACKED) != 0) != s->qr_packed)
    return fail(GAR_HIP_ERR_ARG, std::string("gar_hip_upload_packed_device_fmt: the records were written with ") +
                                     ((record_format & GAR_HIP_FMT_QR_PACKED) ? "packed lower triangles" : "full blocks") +
                                     " of Q / R, this solver's sweep (" + s->kernel_name + ") reads " +
                                     (s->qr_packed ? "packed lower triangles" : "full blocks") +
                                     " (gar_hip_device_record_format)");
  return gar_hip_upload_packed_device(s, b0, nb, packed_dev);
}

int gar_hip_commit(gar_hip_solver *s) {
  GAR_GUARD(s);
  if (!s)
    return fail(GAR_HIP_ERR_ARG, "null solver");
  GAR_MULTI(s, multi_all(s, [](gar_hip_solver *q) { return gar_hip_commit(q); }));
  return commit(s);
}

double *gar_hip_device_problems(gar_hip_solver *s) {
  pipe_autojoin(s);
  return s ? s->d_prob : nullptr;
}
double *gar_hip_device_factors(gar_hip_solver *s) {
  if (s && s->multi) // (the records live on several devices)
    return nullptr;
  if (s && s->fold) { // the caller-visible records of a folded solver are formed on request
    GAR_GUARD(s);
    (void)ensure_expanded(s);
  } else if (s && s->pipe_forked) { // (a pipelined sweep: the caller's stream is ordered behind it before the pointer leaves)
    GAR_GUARD(s);
  }
  return s ? s->d_fac : nullptr;
}
double *gar_hip_device_solutions(gar_hip_solver *s) {
  pipe_autojoin(s); // (a pipelined sweep: the caller's stream is ordered behind it before the pointer leaves)
  return s ? s->d_sol : nullptr;
}

int gar_hip_backward_legs_async(gar_hip_solver *s, double mueq) {
  GAR_GUARD(s);
  if (!s)
    return fail(GAR_HIP_ERR_ARG, "null solver");
  // A knot whose solve divides by mueq outright -- terminalSolve without controls, Z = C / mu (riccati-kernel.hxx:146-149);
  // the decoupled constrained stage and the fold (gar_wave2.hpp, gar_fold.hpp: D = 0 makes kktMat singular at mu = 0, the
  // reference throws there, :239-241) -- turns mueq = 0 (or a NaN) into infinities that the factorisations downstream
  // index with: reported as the failed stage it is, before anything is launched
  if (s->mu_divides && !(std::fabs(mueq) >= 1e-290))
    return fail(GAR_HIP_ERR_FACTOR, "Failed stage LDL factorization (mueq = " + std::to_string(mueq) +
                                        " on constrained knots whose solve divides by it)");
  GAR_MULTI(s, multi_backward_legs(s, mueq));
  s->eager_fwd = false;
  if (s->ev_pref) { // a read-back of the previous sweep's gains may still be in flight on the second stream
    HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_pref, 0));
    s->pref_b = -1;
  }
  if (int rc = commit(s))
    return rc;
  HIP_TRY(hipMemsetAsync(s->d_status, 0, sizeof(int) * ((size_t)s->batch + 4 + (s->fold ? (size_t)s->batch : 0)), s->stream));
  return launch_backward(s, mueq);
}

int gar_hip_condensed_solve_async(gar_hip_solver *s) {
  GAR_GUARD(s);
  if (!s || s->num_legs < 2)
    return fail(GAR_HIP_ERR_ARG, "condensed solve needs leg mode");
  GAR_MULTI(s, multi_exchange_and_condensed(s)); // the boundary exchange happens HERE, inside the library
  return launch_condensed(s);
}

int gar_hip_forward_legs_async(gar_hip_solver *s) {
  GAR_GUARD(s);
  if (!s)
    return fail(GAR_HIP_ERR_ARG, "null solver");
  GAR_MULTI(s, multi_forward(s));
  return launch_forward(s, nullptr);
}

int gar_hip_set_option(const char *name, const char *value) {
  if (!name || !*name)
    return fail(GAR_HIP_ERR_ARG, "gar_hip_set_option: empty name");
  std::string key = name;
  if (key.rfind("GAR_HIP_", 0) != 0)
    key = "GAR_HIP_" + key;
  static const char *known[] = {"BACKWARD", "WIDE", "LEG_WAVES", "CONDENSED", "CONDENSED_REDUCED", "CONDENSED_CR", "LEGS",
                                "FOLD", "SEG_LEGS", "INIT", "FORCE_GENERIC", "PAD", "SPD_ACCEPT", "STAGE_NT", "EAGER",
                                "MULTI_EXCHANGE", "PIPE_PRIORITY", "FORWARD", "PIPELINE"};
  bool ok = false;
  for (const char *k : known)
    ok |= key == std::string("GAR_HIP_") + k;
  if (!ok)
    return fail(GAR_HIP_ERR_ARG, "gar_hip_set_option: unknown switch " + key + " (include/gar_hip.h lists them)");
  std::lock_guard<std::mutex> g(option_mutex());
  if (value)
    option_overrides()[key] = value;
  else
    option_overrides().erase(key);
  return GAR_HIP_OK;
}
const char *gar_hip_get_option(const char *name) {
  if (!name)
    return nullptr;
  std::string key = name;
  if (key.rfind("GAR_HIP_", 0) != 0)
    key = "GAR_HIP_" + key;
  return gar_option(key.c_str());
}

namespace {
bool pipe_eligible(const gar_hip_solver *s) {
  return !(s->multi || s->world > 1 || s->num_legs != 1 || s->nth0 != 0 || s->batch < 2 || !s->wave_kernel || !s->lean_fwd_kernel ||
           !s->wave_half_kernel || s->wave_coupled_kernel || s->wave_block_threads != 64 || s->waves_per_block != 1 || !s->vxx_packed ||
           s->dense);
}
// The library's own choice of schedule (GAR_HIP_PIPELINE = auto, the default): the pipelined sweep pays by the tails
// it hides -- the forward sweep of one half starts while the other half's backward sweep still runs -- and only once
// EACH half fills every SIMD of the chip with a backward wave (measured, round 5: +0.4 ... +3.7 % at 4 096 problems on
// 256 CUs over four boxes, -0.2 % on one; below two full waves of problems per half the plain schedule's whole-chip
// launches win).  So: 2 halves iff the solver is eligible and batch >= 2 x (4 SIMDs x #CUs); plain otherwise.
int pipe_auto_halves(const gar_hip_solver *s) {
  if (!pipe_eligible(s))
    return 0;
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device);
  if (cus <= 0) // (a runtime that does not know: the MI355X's count)
    cus = 256;
  return s->batch >= 8 * cus ? 2 : 0;
}
} // namespace

int gar_hip_set_pipeline(gar_hip_solver *s, int halves) {
  GAR_GUARD(s); // (orders the caller's stream behind anything the half streams still hold)
  if (!s)
    return fail(GAR_HIP_ERR_ARG, "null solver");
  bool automatic = false;
  if (halves < 0) { // the library's choice: never an error
    automatic = true;
    halves = pipe_auto_halves(s);
    if (halves <= 1) {
      s->pipe_halves = 0;
      s->pipe_requested = -1;
      return GAR_HIP_OK;
    }
  }
  if (halves <= 1) {
    s->pipe_halves = s->pipe_requested = 0;
    return GAR_HIP_OK;
  }
  if (halves != 2)
    return fail(GAR_HIP_ERR_ARG, "gar_hip_set_pipeline: 0 (off) or 2 (two half-batches)");
  if (!pipe_eligible(s))
    return fail(GAR_HIP_ERR_UNSUPPORTED, "pipelined sweep: serial, unconstrained, unparameterised batches (>= 2 problems) "
                                         "on the one-wave-per-problem kernel family only (this solver runs " +
                                             s->kernel_name + ")");
  if (int rc = pipe_plan(s, s->lean_fwd_used))
    return rc;
  if (!s->pipe_stream[0]) {
    HIP_TRY(hipFuncSetAttribute((const void *)s->lean_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)s->lean_fwd_lds_bytes));
    HIP_TRY(hipFuncSetAttribute((const void *)s->wave_half_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((size_t)s->wave_lds_doubles_small * sizeof(double))));
    for (int h = 0; h < 2; ++h) {
      {
        // two streams that share a hardware queue run their kernels one after the other, in submission order: the
        // runtime hands out at most GPU_MAX_HW_QUEUES (4) queues PER PRIORITY, so the halves ask for different ones
        // (with GPU_MAX_HW_QUEUES >= 8 in the environment -- bench.py sets it before the runtime starts -- plain
        // streams get queues of their own; GAR_HIP_PIPE_PRIORITY=0 / 1 forces the choice)
        const char *pp = gar_option("GAR_HIP_PIPE_PRIORITY"), *mq = std::getenv("GPU_MAX_HW_QUEUES");
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        const bool plain = pp ? pp[0] == '0' : (mq && std::atoi(mq) >= 8);
        if (plain)
          HIP_TRY(hipStreamCreateWithFlags(&s->pipe_stream[h], hipStreamNonBlocking));
        else
          HIP_TRY(hipStreamCreateWithPriority(&s->pipe_stream[h], hipStreamNonBlocking, h == 0 ? 0 : hi));
      }
      HIP_TRY(hipEventCreateWithFlags(&s->pipe_evB[h], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&s->pipe_evF[h], hipEventDisableTiming));
      for (auto &e : s->pipe_evT[h])
        HIP_TRY(hipEventCreate(&e));
    }
    HIP_TRY(hipEventCreateWithFlags(&s->pipe_evFork, hipEventDisableTiming));
  }
  s->pipe_halves = 2;
  s->pipe_requested = automatic ? -1 : 2;
  return GAR_HIP_OK;
}
int gar_hip_pipeline(const gar_hip_solver *s) { return s ? s->pipe_halves : 0; }

int gar_hip_backward_async(gar_hip_solver *s, double mueq) {
  GAR_GUARD_NOJOIN(s);
  if (pipe_on(s))
    return pipe_backward(s, mueq);
  pipe_autojoin(s);
  if (int rc = gar_hip_backward_legs_async(s, mueq))
    return rc;
  GAR_MULTI(s, multi_exchange_and_condensed(s));
  if (s->num_legs > 1) {
    if (s->world > 1)
      return fail(GAR_HIP_ERR_ARG, "sharded solver: exchange boundaries, then call "
                                   "gar_hip_condensed_solve_async");
    return launch_condensed(s);
  }
  return GAR_HIP_OK;
}

// The caller's whole problem and backward(mueq) in ONE call (what the binding's backward() has in hand): the
// sequence gar_hip_upload_stage x (N+1), gar_hip_set_init, gar_hip_backward behind one crossing of the ABI (the
// staged knots go out in 1 MiB pieces while the next ones are packed, gar_hip_upload_stage).
int gar_hip_backward_blocks(gar_hip_solver *s, const double *const *blocks, const double *G0, const double *g0,
                            double mueq) {
  GAR_GUARD(s);
  if (!s || !blocks)
    return fail(GAR_HIP_ERR_ARG, "gar_hip_backward_blocks: bad argument");
  if (s->batch != 1)
    return fail(GAR_HIP_ERR_ARG, "gar_hip_backward_blocks serves one problem (batch = 1): use gar_hip_upload_stage / "
                                 "gar_hip_upload_packed + gar_hip_backward for a batch");
  const int N = s->horizon;
  auto upload = [&](int t) {
    const double *const *k = blocks + 16 * (size_t)t;
    return gar_hip_upload_stage(s, 0, t, k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7], k[8], k[9], k[10], k[11], k[12],
                                k[13], k[14], k[15]);
  };
  // (Measured and not kept, round 4: sweeping the legs chunk by chunk as their knots arrive, copies on a stream of
  // their own -- the sweep of a chunk takes as long as the sweep of all legs, a leg being a sequential chain over its
  // stages, so the critical path stays packing + one leg + condensed solve; chunks on one stream add their sweeps:
  // (56, 22), 32 legs: 2.35 -> 2.95 ms per Newton iteration.)
  // (Measured and not kept, round 4: a second host thread packing alternate 1 MiB chunks of a >= 12 MiB problem --
  // (56, 22), 32 legs: 1.53-1.57 -> 1.50 ms for this call, the link and the host's memory system are the bound, not
  // the packing core; profiles/r04_seam_two_packing_threads_not_kept.json.)
  for (int t = 0; t <= N; ++t)
    if (int rc = upload(t))
      return rc;
  if (int rc = gar_hip_set_init(s, 0, G0, g0))
    return rc;
  // Without a parameter the roll-out depends on nothing the caller still has to say: it is enqueued right behind the
  // sweep, with the solution's copy and the gains' read-back (second stream), BEFORE the host waits for the status
  // word -- the device runs sweep, roll-out and copies back to back instead of waiting for the host between them.
  // gar_hip_forward / gar_hip_prefetch_gains / the solution fetch then find their work done.
  const bool eager_on = [] { const char *e = gar_option("GAR_HIP_EAGER"); return !(e && e[0] == '0'); }();
  const bool eager = eager_on && !s->multi && !s->fold && !s->dense && (s->nth0 == 0 || s->num_legs > 1) && s->world == 1;
  if (!eager)
    return gar_hip_backward(s, mueq);
  if (int rc = gar_hip_backward_legs_async(s, mueq))
    return rc;
  if (s->num_legs > 1) {
    // the gains are final once the leg sweeps are (the condensed solve reads the boundary tuples and writes the
    // boundary solution): their read-back starts HERE, under the condensed solve, not behind it
    if (int rc = prefetch_impl(s, 0)) // (allocates the read-back buffers on first use)
      return rc;
    if (int rc = launch_condensed(s))
      return rc;
  }
  if (!s->h_status) {
    HIP_TRY(gar_host_malloc((void **)&s->h_status, sizeof(int) * (size_t)s->batch, hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_status, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_sol, hipEventDisableTiming));
  }
  HIP_TRY(hipMemcpyAsync(s->h_status, s->d_status, sizeof(int) * (size_t)s->batch, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipEventRecord(s->ev_status, s->stream));
  if (s->num_legs == 1)
    if (int rc = prefetch_impl(s, 0))
      return rc;
  if (int rc = launch_forward(s, nullptr))
    return rc;
  {
    const gar_hip_solver *u = s->ulay ? s->ulay : s;
    const size_t nsol = (size_t)u->sol_doubles, ngain = (size_t)(u->ff_all_doubles + u->fb_all_doubles);
    // by a kernel's own stores into the pinned buffer: the copy engine is busy with the gains (3.7 / 9.4 MB) and a
    // hipMemcpyAsync would queue behind them (measured: profiles/r04_seam_eager_rollout_ab.json)
    double *dst = s->h_results + (s->padded ? nsol + ngain : 0);
    const unsigned nblk = (unsigned)std::min<int64_t>((s->sol_doubles + 255) / 256, 256);
    hipLaunchKernelGGL(gar::gar_store_to_host, dim3(nblk), dim3(256), 0, s->stream, dst, s->d_sol, (long long)s->sol_doubles);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(s->ev_sol, s->stream));
  HIP_TRY(hipEventSynchronize(s->ev_status));
  int nf = 0;
  for (int b = 0; b < s->batch; ++b)
    nf += (s->h_status[b] != 0);
  s->last_failed = nf;
  if (nf > 0) {
    // the roll-out, the solution's copy and the gains' read-back of the FAILED sweep are already enqueued: let them
    // finish and disarm the "already in flight" shortcuts, so that a later fetch does its own work (and nobody is
    // handed these gains as if the sweep had succeeded)
    if (s->ev_pref)
      HIP_TRY(hipEventSynchronize(s->ev_pref));
    s->pref_b = -1;
    s->eager_fwd = false;
    return fail(GAR_HIP_ERR_FACTOR, "Failed stage LDL factorization (" + std::to_string(nf) + " problem(s))");
  }
  s->eager_fwd = true;
  return GAR_HIP_OK;
}

int gar_hip_num_failed(gar_hip_solver *s) {
  GAR_GUARD(s);
  if (!s)
    return 0;
  GAR_MULTI(s, multi_num_failed(s));
  std::vector<int> st((size_t)s->batch);
  if (hipMemcpyAsync(st.data(), s->d_status, sizeof(int) * st.size(), hipMemcpyDeviceToHost,
                     s->stream) != hipSuccess ||
      hipStreamSynchronize(s->stream) != hipSuccess)
    return -1;
  int n = 0;
  for (int v : st)
    n += (v != 0);
  s->last_failed = n;
  return n;
}

int gar_hip_slow_path_stages(gar_hip_solver *s, int64_t out[2]) {
  GAR_GUARD(s);
  GAR_MULTI(s, multi_counters(s, out, gar_hip_slow_path_stages));
  if (!s || !out)
    return fail(GAR_HIP_ERR_ARG, "bad argument");
  int c[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(c, s->d_status + s->batch, sizeof(c), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  out[0] = c[0];
  out[1] = c[1];
  return GAR_HIP_OK;
}

int gar_hip_constrained_bk_stages(gar_hip_solver *s, int64_t out[2]) {
  GAR_GUARD(s);
  GAR_MULTI(s, multi_counters(s, out, gar_hip_constrained_bk_stages));
  if (!s || !out)
    return fail(GAR_HIP_ERR_ARG, "bad argument");
  int c[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(c, s->d_status + s->batch + 2, sizeof(c), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  out[0] = c[0];
  out[1] = c[1];
  return GAR_HIP_OK;
}

int gar_hip_backward(gar_hip_solver *s, double mueq) {
  GAR_GUARD(s);
  if (int rc = gar_hip_backward_async(s, mueq))
    return rc;
  const int nf = gar_hip_num_failed(s);
  if (nf < 0)
    return fail(GAR_HIP_ERR_DEVICE, std::string("backward: ") +
                                        hipGetErrorString(hipGetLastError()));
  if (nf > 0)
    return fail(GAR_HIP_ERR_FACTOR, "Failed stage LDL factorization (" + std::to_string(nf) +
                                        " problem(s))");
  return GAR_HIP_OK;
}

int gar_hip_forward_async(gar_hip_solver *s, const double *theta_device) {
  GAR_GUARD_NOJOIN(s);
  if (!s)
    return fail(GAR_HIP_ERR_ARG, "null solver");
  if (pipe_on(s))
    return pipe_forward(s);
  pipe_autojoin(s);
  GAR_MULTI(s, multi_forward(s));
  return launch_forward(s, theta_device);
}

int gar_hip_forward(gar_hip_solver *s, const double *theta) {
  GAR_GUARD(s);
  if (!s)
    return fail(GAR_HIP_ERR_ARG, "null solver");
  if (s->multi) {
    if (int rc = multi_forward(s))
      return rc;
    return multi_sync(s);
  }
  if (s->eager_fwd) { // enqueued behind the sweep by gar_hip_backward_blocks (no parameter: theta has no say)
    HIP_TRY(hipEventSynchronize(s->ev_sol));
    return GAR_HIP_OK;
  }
  const double *th = nullptr;
  if (theta && s->nth0 > 0 && s->num_legs == 1) {
    HIP_TRY(hipMemcpyAsync(s->d_theta, theta, sizeof(double) * (size_t)s->nth0 * s->batch,
                           hipMemcpyHostToDevice, s->stream));
    th = s->d_theta;
  }
  if (int rc = launch_forward(s, th))
    return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  return GAR_HIP_OK;
}

int64_t gar_hip_boundary_doubles(const gar_hip_solver *s) { return s ? s->tuple_doubles : 0; }
double *gar_hip_device_boundary_local(gar_hip_solver *s) { return s ? s->d_bound_local : nullptr; }
double *gar_hip_device_boundary_all(gar_hip_solver *s) { return s ? s->d_bound_all : nullptr; }

int gar_hip_set_refinement(gar_hip_solver *s, double thr, int max_steps) {
  if (!s || max_steps < 0)
    return fail(GAR_HIP_ERR_ARG, "bad refinement settings");
  s->cond_threshold = thr;
  s->max_refinement = max_steps;
  GAR_MULTI(s, multi_all(s, [&](gar_hip_solver *q) { return gar_hip_set_refinement(q, thr, max_steps); }));
  return GAR_HIP_OK;
}

int gar_hip_condensed_info(gar_hip_solver *s, int b, double out[2]) {
  GAR_GUARD(s);
  if (int rc = check_bt(s, b, 0))
    return rc;
  GAR_MULTI(s, gar_hip_condensed_info(s->multi->subs[0], b, out)); // (solved redundantly on every device)
  if (s->num_legs < 2 || !out)
    return fail(GAR_HIP_ERR_ARG, "condensed info needs leg mode");
  const int64_t nblk = 2 * s->num_legs, bs = (int64_t)s->nxb * s->nxb;
  const double *info = s->d_cscratch + (int64_t)b * s->cscratch_doubles + 4 * nblk * bs +
                       4 * nblk * s->nxb;
  if (int rc = d2h(s, out, info, 2))
    return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  return GAR_HIP_OK;
}

static int get_solution_dev(gar_hip_solver *s, int b, double *xs, double *us, double *vs,
                         double *lbdas) {
  const double *base = s->d_sol + (int64_t)b * s->sol_doubles;
  int rc = d2h(s, xs, base + s->sol_x, s->sol_u - s->sol_x);
  rc |= d2h(s, us, base + s->sol_u, s->sol_v - s->sol_u);
  rc |= d2h(s, vs, base + s->sol_v, s->sol_l - s->sol_v);
  int64_t nl = s->nc0;
  for (int t = 0; t < s->horizon; ++t)
    nl += s->meta[t].nx2;
  rc |= d2h(s, lbdas, base + s->sol_l, nl);
  if (rc)
    return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  return GAR_HIP_OK;
}

static int get_gains_dev(gar_hip_solver *s, int b, int t, double *ff, double *fb, double *fth) {
  if (int rc = ensure_expanded(s))
    return rc;
  const gar_stage_meta &m = s->meta[t];
  const int nx2r = s->dense ? 2 * m.nx2 : m.nx2; // stage-dense solver: rows [K; Z; L; Y]
  const gar_factor_offsets o = gar_factor_layout(m.nx, m.nu, m.nc, nx2r, m.nth);
  const double *rec = s->d_fac + (int64_t)b * s->fac_doubles + m.fac_off;
  const int64_t nr = (int64_t)m.nu + m.nc + nx2r;
  int rc = d2h(s, ff, rec + o.ff, nr);
  std::vector<double> tmp;
  // the specialised kernel families keep fb (and fth) in the fbT2 device order
  const bool t2 = records_t2(s, b) && t < s->horizon;
  const bool fbt2 = t2 && fb;
  const bool ftht2 = t2 && fth && m.nth > 0;
  std::vector<double> tmpth;
  if (fbt2) {
    tmp.resize((size_t)(nr * m.nx));
    rc |= d2h(s, tmp.data(), rec + o.fb, nr * m.nx);
  } else {
    rc |= d2h(s, fb, rec + o.fb, nr * m.nx);
  }
  if (ftht2) {
    tmpth.resize((size_t)(nr * m.nth));
    rc |= d2h(s, tmpth.data(), rec + o.fth, nr * m.nth);
  } else {
    rc |= d2h(s, fth, rec + o.fth, nr * m.nth);
  }
  if (rc)
    return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (ftht2) {
    const int NW = (int)nr, NT = m.nth;
    for (int r = 0; r < NW; ++r)
      for (int j = 0; j < NT; ++j)
        fth[(size_t)r * NT + j] = tmpth[(size_t)(j >> 1) * (2 * NW) + 2 * r + (j & 1)];
  }
  if (fbt2) { // back to StageFactor's row-major [K; Z; Aff] (riccati-kernel.hpp:96-97)
    const int NW = (int)nr, NX = m.nx;
    for (int r = 0; r < NW; ++r)
      for (int j = 0; j < NX; ++j)
        fb[(size_t)r * NX + j] = tmp[(size_t)(j >> 1) * (2 * NW) + 2 * r + (j & 1)];
  }
  return GAR_HIP_OK;
}

int gar_hip_gains_doubles(const gar_hip_solver *s, int64_t out[2]) {
  if (!s || !out)
    return fail(GAR_HIP_ERR_ARG, "bad argument");
  out[0] = s->ulay ? s->ulay->ff_all_doubles : s->ff_all_doubles;
  out[1] = s->ulay ? s->ulay->fb_all_doubles : s->fb_all_doubles;
  return GAR_HIP_OK;
}

int gar_hip_gains_offsets(const gar_hip_solver *s, int t, int64_t out[2]) {
  if (int rc = check_bt(s, 0, t))
    return rc;
  const std::vector<long long> &go = s->ulay ? s->ulay->gain_off : s->gain_off;
  out[0] = go[2 * (size_t)t];
  out[1] = go[2 * (size_t)t + 1];
  return GAR_HIP_OK;
}

// [t_lo, t_hi): the stages whose gains are gathered and copied (the whole horizon for a one-device solver; its own
// stages for each device of a multi-device solver, gar_multi.hpp); gains_base: the host buffer [.. | ff_all | fb_all]
// the gains land in (null: the solver's own); sync = false leaves the copies in flight on the solver's stream
static int fetch_results_impl(gar_hip_solver *s, int b, int what, int t_lo, int t_hi, double *gains_base, bool sync) {
  if (int rc = check_bt(s, b, 0))
    return rc;
  // the caller's records (under padding: the real rows / columns only); the device solution record is staged
  // behind them when it has to be stripped on the host
  const gar_hip_solver *u = s->ulay ? s->ulay : s;
  const size_t nsol = (size_t)u->sol_doubles, ngain = (size_t)(u->ff_all_doubles + u->fb_all_doubles);
  const size_t nscratch = s->padded ? (size_t)s->sol_doubles : 0;
  if (!s->h_results) { // first use: the buffers live as long as the solver's layout
    // all three or none: a partial failure must not leave h_results set with the device buffers missing
    double *h = nullptr, *dg = nullptr;
    long long *dgo = nullptr;
    hipError_t e = gar_host_malloc((void **)&h, sizeof(double) * (nsol + ngain + nscratch), hipHostMallocDefault);
    if (e == hipSuccess)
      e = gar_dev_malloc((void **)&dg, sizeof(double) * std::max<size_t>(ngain, 1));
    if (e == hipSuccess)
      e = gar_dev_malloc((void **)&dgo, sizeof(long long) * u->gain_off.size());
    if (e == hipSuccess)
      e = hipMemcpyAsync(dgo, u->gain_off.data(), sizeof(long long) * u->gain_off.size(), hipMemcpyHostToDevice,
                         s->stream);
    if (e != hipSuccess) {
      if (h)
        (void)hipHostFree(h);
      (void)hipFree(dg);
      (void)hipFree(dgo);
      return fail(GAR_HIP_ERR_DEVICE, std::string("gar_hip_fetch_results: ") + hipGetErrorString(e));
    }
    std::memset(h, 0, sizeof(double) * (nsol + ngain + nscratch)); // (alignment padding inside the records reads as zero)
    s->h_results = h;
    s->d_gains = dg;
    s->d_gain_off = dgo;
  }
  if ((what & 2) && s->pref_b == b && !gains_base && t_lo == 0 && t_hi == s->horizon + 1) {
    // the gains are already on their way (gar_hip_prefetch_gains): wait for them; what collapseFeedback changed
    // since -- stage 0 -- comes again
    HIP_TRY(hipEventSynchronize(s->ev_pref));
    s->pref_b = -1;
    if (!s->pref_collapsed)
      what &= ~2;
    else
      t_hi = 1;
  }
  if ((what & 2) && t_hi > t_lo) { // device-side gather (fbT2 -> row-major, dummy rows / columns dropped), then ONE device-to-host copy
    if (s->pref_b >= 0 && s->ev_pref) {
      // a read-back started by gar_hip_prefetch_gains (of another problem, or of this one over another range) may
      // still be writing d_gains / h_results on the second stream: this gather and copy reuse both
      HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_pref, 0));
      HIP_TRY(hipEventSynchronize(s->ev_pref)); // (h_results is host memory: the caller may read it right after)
      s->pref_b = -1;
    }
    if (int rc = ensure_expanded(s))
      return rc;
    const bool t2 = records_t2(s, b);
    hipLaunchKernelGGL(gar::gar_gather_gains, dim3((unsigned)(t_hi - t_lo)), dim3(256), 0, s->stream,
                       s->d_meta, s->d_fac + (int64_t)b * s->fac_doubles, s->d_gains,
                       s->d_gains + u->ff_all_doubles, s->d_gain_off, s->horizon, t2 ? 1 : 0,
                       s->dense ? 1 : 0, s->padded ? s->unx : 0, s->padded ? s->unu : 0, t_lo);
    HIP_TRY(hipGetLastError());
    double *hg = (gains_base ? gains_base : s->h_results) + nsol;
    if (t_lo == 0 && t_hi == s->horizon + 1) {
      HIP_TRY(hipMemcpyAsync(hg, s->d_gains, sizeof(double) * ngain, hipMemcpyDeviceToHost, s->stream));
    } else { // the two slices of this stage range
      const std::vector<long long> &go = u->gain_off;
      const long long f0 = go[2 * (size_t)t_lo], f1 = t_hi <= s->horizon ? go[2 * (size_t)t_hi] : u->ff_all_doubles;
      const long long b0 = go[2 * (size_t)t_lo + 1], b1 = t_hi <= s->horizon ? go[2 * (size_t)t_hi + 1] : u->fb_all_doubles;
      if (f1 > f0)
        HIP_TRY(hipMemcpyAsync(hg + f0, s->d_gains + f0, sizeof(double) * (size_t)(f1 - f0), hipMemcpyDeviceToHost, s->stream));
      if (b1 > b0)
        HIP_TRY(hipMemcpyAsync(hg + u->ff_all_doubles + b0, s->d_gains + u->ff_all_doubles + b0,
                               sizeof(double) * (size_t)(b1 - b0), hipMemcpyDeviceToHost, s->stream));
    }
  }
  const bool sol_there = (what & 1) && s->eager_fwd && b == 0; // copied behind the eager roll-out (gar_hip_backward_blocks)
  if ((what & 1) && !sol_there)
    HIP_TRY(hipMemcpyAsync(s->h_results + (s->padded ? nsol + ngain : 0), s->d_sol + (int64_t)b * s->sol_doubles,
                           sizeof(double) * (size_t)s->sol_doubles, hipMemcpyDeviceToHost, s->stream));
  if (!sync)
    return GAR_HIP_OK;
  if (sol_there && what == 1)
    HIP_TRY(hipEventSynchronize(s->ev_sol));
  else
    HIP_TRY(hipStreamSynchronize(s->stream));
  if ((what & 1) && s->padded)
    strip_solution_rec(s, s->h_results + nsol + ngain, s->h_results);
  return GAR_HIP_OK;
}

// every stage's gains of problem b: gather + ONE device-to-host copy on the second stream, behind what the main
// stream holds now
static int prefetch_impl(gar_hip_solver *s, int b) {
  if (int rc = fetch_results_impl(s, b, 0, 0, 0, nullptr, false)) // the buffers, on first use
    return rc;
  if (!s->aux_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&s->aux_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_main, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_pref, hipEventDisableTiming));
  }
  const gar_hip_solver *u = s->ulay ? s->ulay : s;
  const size_t nsol = (size_t)u->sol_doubles, ngain = (size_t)(u->ff_all_doubles + u->fb_all_doubles);
  HIP_TRY(hipEventRecord(s->ev_main, s->stream));
  HIP_TRY(hipStreamWaitEvent(s->aux_stream, s->ev_main, 0));
  hipLaunchKernelGGL(gar::gar_gather_gains, dim3((unsigned)(s->horizon + 1)), dim3(256), 0, s->aux_stream, s->d_meta,
                     s->d_fac + (int64_t)b * s->fac_doubles, s->d_gains, s->d_gains + u->ff_all_doubles, s->d_gain_off,
                     s->horizon, records_t2(s, b) ? 1 : 0, s->dense ? 1 : 0, s->padded ? s->unx : 0,
                     s->padded ? s->unu : 0, 0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(s->h_results + nsol, s->d_gains, sizeof(double) * ngain, hipMemcpyDeviceToHost, s->aux_stream));
  HIP_TRY(hipEventRecord(s->ev_pref, s->aux_stream));
  s->pref_b = b;
  s->pref_collapsed = false;
  return GAR_HIP_OK;
}

int gar_hip_fetch_results(gar_hip_solver *s, int b, int what) {
  GAR_GUARD(s);
  if (int rc = check_bt(s, b, 0))
    return rc;
  GAR_MULTI(s, multi_fetch_results(s, b, what));
  return fetch_results_impl(s, b, what, 0, s->horizon + 1, nullptr, true);
}

int gar_hip_prefetch_gains(gar_hip_solver *s, int b) {
  GAR_GUARD(s);
  if (int rc = check_bt(s, b, 0))
    return rc;
  // (multi-device and folded solvers: the ordinary fetch does the work -- their read-back has host-side steps)
  if (s->multi || s->fold)
    return GAR_HIP_OK;
  if (s->eager_fwd && s->pref_b == b && !s->pref_collapsed)
    return GAR_HIP_OK; // gar_hip_backward_blocks started it already
  return prefetch_impl(s, b);
}

const double *gar_hip_host_results(gar_hip_solver *s, int64_t offs[3]) {
  if (!s)
    return nullptr;
  const gar_hip_solver *u = s->ulay ? s->ulay : s;
  if (offs) {
    offs[0] = 0;
    offs[1] = u->sol_doubles;
    offs[2] = u->sol_doubles + u->ff_all_doubles;
  }
  if (s->multi)
    return s->multi->h_results;
  return s->h_results;
}

int gar_hip_get_gains_all(gar_hip_solver *s, int b, double *ff_all, double *fb_all) {
  if (int rc = gar_hip_fetch_results(s, b, 2))
    return rc;
  const gar_hip_solver *u = s->ulay ? s->ulay : s;
  const double *h = gar_hip_host_results(s, nullptr);
  if (ff_all)
    std::memcpy(ff_all, h + u->sol_doubles, sizeof(double) * (size_t)u->ff_all_doubles);
  if (fb_all)
    std::memcpy(fb_all, h + u->sol_doubles + u->ff_all_doubles,
                sizeof(double) * (size_t)u->fb_all_doubles);
  return GAR_HIP_OK;
}

static int get_value_dev(gar_hip_solver *s, int b, int t, double *Vxx, double *vx, double *Vxt,
                      double *Vtt, double *vt) {
  if (int rc = ensure_expanded(s))
    return rc;
  const gar_stage_meta &m = s->meta[t];
  const gar_factor_offsets o = gar_factor_layout(m.nx, m.nu, m.nc, s->dense ? 2 * m.nx2 : m.nx2, m.nth);
  const double *rec = s->d_fac + (int64_t)b * s->fac_doubles + m.fac_off;
  std::vector<double> packed; // the serial one-wave family keeps the lower triangle, packed (gar_layout.h)
  int rc = 0;
  if (s->vxx_packed && Vxx) {
    packed.resize((size_t)gar_sym_packed_doubles(m.nx));
    rc = d2h(s, packed.data(), rec + o.Vxx, (int64_t)packed.size());
  } else {
    rc = d2h(s, Vxx, rec + o.Vxx, (int64_t)m.nx * m.nx);
  }
  rc |= d2h(s, vx, rec + o.vx, m.nx);
  rc |= d2h(s, Vxt, rec + o.Vxt, (int64_t)m.nx * m.nth);
  rc |= d2h(s, Vtt, rec + o.Vtt, (int64_t)m.nth * m.nth);
  rc |= d2h(s, vt, rec + o.vt, m.nth);
  if (rc)
    return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (!packed.empty())
    for (int j = 0; j < m.nx; ++j)
      for (int i = 0; i < m.nx; ++i)
        Vxx[(size_t)j * m.nx + i] = packed[(size_t)gar_sym_index(1, m.nx, i, j)];
  return GAR_HIP_OK;
}

static int get_kkt_dev(gar_hip_solver *s, int b, int t, double mueq, double *out) {
  if (s->dense)
    return fail(GAR_HIP_ERR_UNSUPPORTED, "kktMat of the stage-dense solver is not kept (the whole stage matrix: gar_dense.hpp)");
  if (!out)
    return fail(GAR_HIP_ERR_ARG, "null output");
  if (int rc = ensure_expanded(s))
    return rc;
  const gar_stage_meta &m = s->meta[t];
  const int nk = m.nu + m.nc;
  if (nk == 0)
    return GAR_HIP_OK;
  if ((int64_t)nk * nk > s->kkt_doubles) {
    if (s->d_kkt)
      (void)hipFree(s->d_kkt);
    s->d_kkt = nullptr;
    HIP_TRY(gar_dev_malloc((void **)&s->d_kkt, sizeof(double) * (size_t)nk * nk));
    s->kkt_doubles = (int64_t)nk * nk;
  }
  if (commit(s) != GAR_HIP_OK)
    return GAR_HIP_ERR_DEVICE;
  const int nth_st = (m.flags & GAR_KNOT_HAS_PARAM) ? m.nth : 0;
  const gar_knot_offsets ko = gar_knot_layout(m.nx, m.nu, m.nc, m.nx2, nth_st);
  const double *knot = s->d_prob + (int64_t)b * s->prob_doubles + m.in_off;
  // the stage's value-function term: none at the terminal knot and at the last knot of a leg (each leg ends
  // with terminalSolve, parallel-solver.hxx:150-160)
  const double *Vn = nullptr;
  if (t < s->horizon && !(m.flags & GAR_KNOT_LEG_END) && m.nu > 0) {
    const gar_stage_meta &mn = s->meta[t + 1];
    const gar_factor_offsets fn = gar_factor_layout(mn.nx, mn.nu, mn.nc, mn.nx2, mn.nth);
    Vn = s->d_fac + (int64_t)b * s->fac_doubles + mn.fac_off + fn.Vxx;
  }
  hipLaunchKernelGGL(gar::gar_kkt_matrix, dim3(1), dim3(256), 0, s->stream, knot, ko, Vn, m.nx2, m.nu, m.nc, mueq,
                     s->d_kkt, s->vxx_packed ? 1 : 0, (s->qr_packed && t < s->horizon) ? 1 : 0);
  HIP_TRY(hipGetLastError());
  if (int rc = d2h(s, out, s->d_kkt, (int64_t)nk * nk))
    return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  return GAR_HIP_OK;
}

static int get_initial_dev(gar_hip_solver *s, int b, double *kkt0_ff, double *kkt0_fth,
                        double *thGrad, double *thHess) {
  const double *io = s->d_init + (int64_t)b * s->init_doubles;
  const int64_t n0 = s->n0, nth = s->nth0;
  int rc = d2h(s, kkt0_ff, io, n0);
  rc |= d2h(s, kkt0_fth, io + n0, n0 * nth);
  rc |= d2h(s, thGrad, io + n0 + n0 * nth, nth);
  rc |= d2h(s, thHess, io + n0 + n0 * nth + nth, nth * nth);
  if (rc)
    return rc;
  HIP_TRY(hipStreamSynchronize(s->stream));
  return GAR_HIP_OK;
}

int gar_hip_debug_trace(gar_hip_solver *s, int enable, long long out[64]) {
  GAR_GUARD(s);
  if (!s)
    return fail(GAR_HIP_ERR_ARG, "null solver");
  GAR_MULTI(s, gar_hip_debug_trace(s->multi->subs[0], enable, out));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (enable && !s->d_trace) {
    HIP_TRY(gar_dev_malloc((void **)&s->d_trace, sizeof(long long) * 64));
    HIP_TRY(hipMemset(s->d_trace, 0, sizeof(long long) * 64));
  }
  if (out && s->d_trace)
    HIP_TRY(hipMemcpy(out, s->d_trace, sizeof(long long) * 64, hipMemcpyDeviceToHost));
  if (!enable && s->d_trace) {
    (void)hipFree(s->d_trace);
    s->d_trace = nullptr;
  }
  return GAR_HIP_OK;
}

// (the derivative records speak the CALLER's dimensions: under padding the layout is the caller-facing one, `ulay`)
int64_t gar_hip_deriv_doubles(const gar_hip_solver *s) { return s ? (s->ulay ? s->ulay->deriv_doubles : s->deriv_doubles) : 0; }

int gar_hip_deriv_offsets(const gar_hip_solver *s, int t, int64_t out[4]) {
  if (int rc = check_bt(s, 0, t))
    return rc;
  const gar_hip_solver *u = s->ulay ? s->ulay : s;
  out[0] = u->deriv_off[t];
  out[1] = u->d_G0;
  out[2] = u->d_g0;
  out[3] = u->d_iH;
  return GAR_HIP_OK;
}

int gar_hip_update_lq_subproblem_device(gar_hip_solver *s, const double *deriv_dev, double preg,
                                        int hess_exact) {
  GAR_GUARD(s);
  if (!s || !deriv_dev)
    return fail(GAR_HIP_ERR_ARG, "gar_hip_update_lq_subproblem_device: bad argument");
  GAR_MULTI(s, fail(GAR_HIP_ERR_UNSUPPORTED, "device-resident LQ assembly on a multi-device solver: one derivative buffer per device would be needed"));
  if (int rc = commit(s)) // pending host staging first; later host writes flush only their own ranges
    return rc;
  const gar_hip_solver *u = s->ulay ? s->ulay : s; // the layout of the derivative buffer: the caller's dimensions
  if (!s->d_deriv_off) {
    HIP_TRY(gar_dev_malloc((void **)&s->d_deriv_off, sizeof(long long) * u->deriv_off.size()));
    HIP_TRY(hipMemcpy(s->d_deriv_off, u->deriv_off.data(), sizeof(long long) * u->deriv_off.size(),
                      hipMemcpyHostToDevice));
  }
  gar::UpdateParams U{};
  U.meta = s->d_meta;
  U.deriv = deriv_dev;
  U.prob = s->d_prob;
  U.deriv_stride = u->deriv_doubles;
  U.prob_stride = s->prob_doubles;
  U.G0_off = s->G0_off;
  U.g0_off = s->g0_off;
  U.deriv_off = s->d_deriv_off;
  U.d_G0 = u->d_G0;
  U.d_g0 = u->d_g0;
  U.d_iH = u->d_iH;
  U.horizon = s->horizon;
  U.nc0 = s->nc0;
  U.nx0 = s->nx0;
  U.hess_exact = hess_exact;
  U.preg = preg;
  U.qr_packed = s->qr_packed ? 1 : 0;
  const dim3 grid((unsigned)(s->horizon + 1), (unsigned)s->batch);
  if (s->padded) // the caller's records scattered into the padded knots, dummy rows / columns written (gar_generic.hpp)
    hipLaunchKernelGGL(gar::gar_update_lq_padded, grid, dim3(256), 0, s->stream, U, s->unx, s->unu, s->user_nc0);
  else
    hipLaunchKernelGGL(gar::gar_update_lq, grid, dim3(256), 0, s->stream, U);
  HIP_TRY(hipGetLastError());
  return GAR_HIP_OK;
}

int gar_hip_download_packed(gar_hip_solver *s, int b0, int nb, double *packed) {
  GAR_GUARD(s);
  GAR_MULTI(s, multi_download_packed(s, b0, nb, packed));
  if (!s || !packed || b0 < 0 || nb < 0 || b0 + nb > s->batch)
    return fail(GAR_HIP_ERR_ARG, "gar_hip_download_packed: bad argument");
  if (int rc = commit(s))
    return rc;
  if (s->padded) { // device records -> the caller's: the real rows / columns of every block
    const gar_hip_solver *u = s->ulay;
    std::vector<double> dev((size_t)s->prob_doubles);
    for (int b = b0; b < b0 + nb; ++b) {
      HIP_TRY(hipMemcpyAsync(dev.data(), s->d_prob + (int64_t)b * s->prob_doubles, sizeof(double) * dev.size(),
                             hipMemcpyDeviceToHost, s->stream));
      HIP_TRY(hipStreamSynchronize(s->stream));
      double *rec = packed + (int64_t)(b - b0) * u->prob_doubles;
      std::memset(rec, 0, sizeof(double) * (size_t)u->prob_doubles);
      auto take = [](double *dst, const double *src, int r, int c, int R) {
        for (int j = 0; j < c; ++j)
          std::memcpy(dst + (size_t)j * r, src + (size_t)j * R, sizeof(double) * (size_t)r);
      };
      take(rec + u->G0_off, dev.data() + s->G0_off, s->user_nc0, s->unx, s->nc0);
      take(rec + u->g0_off, dev.data() + s->g0_off, s->user_nc0, 1, s->nc0);
      for (int t = 0; t <= s->horizon; ++t) {
        const gar_stage_meta &m = u->meta[t], &M = s->meta[t];
        const gar_knot_offsets o = gar_knot_layout(m.nx, m.nu, 0, m.nx2, 0), O = gar_knot_layout(M.nx, M.nu, 0, M.nx2, 0);
        double *k = rec + m.in_off;
        const double *K = dev.data() + M.in_off;
        if (s->qr_packed && t < s->horizon) {
          unpack_lower(k + o.Q, K + O.Q, m.nx, M.nx);
          unpack_lower(k + o.R, K + O.R, m.nu, M.nu);
        } else {
          take(k + o.Q, K + O.Q, m.nx, m.nx, M.nx);
          take(k + o.R, K + O.R, m.nu, m.nu, M.nu);
        }
        take(k + o.S, K + O.S, m.nx, m.nu, M.nx);
        take(k + o.q, K + O.q, m.nx, 1, M.nx);
        take(k + o.r, K + O.r, m.nu, 1, M.nu);
        take(k + o.A, K + O.A, m.nx2, m.nx, M.nx2);
        take(k + o.B, K + O.B, m.nx2, m.nu, M.nx2);
        take(k + o.f, K + O.f, m.nx2, 1, M.nx2);
      }
    }
    return GAR_HIP_OK;
  }
  HIP_TRY(hipMemcpyAsync(packed, s->d_prob + (int64_t)b0 * s->prob_doubles,
                         sizeof(double) * (size_t)s->prob_doubles * nb, hipMemcpyDeviceToHost,
                         s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (s->qr_packed) { // the device keeps the lower triangles of Q and R, packed: the full symmetric blocks back
    std::vector<double> tmp;
    for (int k = 0; k < nb; ++k)
      for (int t = 0; t < s->horizon; ++t) {
        const gar_stage_meta &m = s->meta[t];
        const gar_knot_offsets o = gar_knot_layout(m.nx, m.nu, m.nc, m.nx2, 0);
        double *rec = packed + (int64_t)k * s->prob_doubles + m.in_off;
        for (const auto &blk : {std::make_pair(o.Q, m.nx), std::make_pair(o.R, m.nu)}) {
          tmp.assign(rec + blk.first, rec + blk.first + (size_t)blk.second * (blk.second + 1) / 2);
          unpack_lower(rec + blk.first, tmp.data(), blk.second, blk.second);
        }
      }
  }
  return GAR_HIP_OK;
}

int gar_hip_set_timing(gar_hip_solver *s, int enable) {
  GAR_GUARD(s);
  if (!s)
    return fail(GAR_HIP_ERR_ARG, "null solver");
  GAR_MULTI(s, multi_all(s, [&](gar_hip_solver *q) { return gar_hip_set_timing(q, enable); }));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (enable && !s->ev[0])
    for (auto &e : s->ev)
      HIP_TRY(hipEventCreate(&e));
  s->timing = enable != 0;
  return GAR_HIP_OK;
}

int gar_hip_last_kernel_ms(gar_hip_solver *s, double out[3]) {
  GAR_GUARD(s);
  GAR_MULTI(s, multi_last_kernel_ms(s, out));
  if (!s || !out)
    return fail(GAR_HIP_ERR_ARG, "bad argument");
  if (!s->timing || !(s->mfma_kernel || s->wave_kernel || s->leg_bwd_kernel || s->seg_bwd_kernel))
    return fail(GAR_HIP_ERR_UNSUPPORTED, "per-kernel timing is recorded for the specialised "
                                         "kernel family after gar_hip_set_timing(s, 1)");
  HIP_TRY(hipStreamSynchronize(s->stream));
  float ms = 0.f;
  if (pipe_on(s)) { // per HALF-batch launch, the mean of the two halves; the initial stage rides inside the sweep
    out[0] = out[1] = out[2] = 0.0;
    for (int h = 0; h < 2; ++h) {
      HIP_TRY(hipEventElapsedTime(&ms, s->pipe_evT[h][0], s->pipe_evT[h][1]));
      out[0] += 0.5 * ms;
      HIP_TRY(hipEventElapsedTime(&ms, s->pipe_evT[h][2], s->pipe_evT[h][3]));
      out[2] += 0.5 * ms;
    }
    return GAR_HIP_OK;
  }
  HIP_TRY(hipEventElapsedTime(&ms, s->ev[0], s->ev[1]));
  out[0] = ms;
  HIP_TRY(hipEventElapsedTime(&ms, s->ev[1], s->ev[2]));
  out[1] = ms;
  HIP_TRY(hipEventElapsedTime(&ms, s->ev[3], s->ev[4]));
  out[2] = ms;
  return GAR_HIP_OK;
}

int gar_hip_collapse_feedback(gar_hip_solver *s) {
  GAR_GUARD(s);
  if (!s)
    return fail(GAR_HIP_ERR_ARG, "null solver");
  GAR_MULTI(s, gar_hip_collapse_feedback(s->multi->subs[0])); // stage 0 lives on the first device
  s->pref_collapsed = true;
  s->eager_fwd = false; // a roll-out after this call is a new one
  if (s->num_legs < 2 || s->leg_begin != 0)
    return GAR_HIP_OK; // no-op except Parallel (riccati-base.hpp:33)
  if (s->fold) { // the wave-leg family's own records (then re-expanded on request); flagged problems: generic records
    const int *flags = s->d_status + s->batch + 4;
    hipLaunchKernelGGL(s->leg_collapse_kernel, dim3((unsigned)s->batch), dim3(256), 0, s->stream, s->d_meta2, s->d_fac2,
                       (long long)s->flay->fac_doubles, s->batch, flags, 0);
    hipLaunchKernelGGL(gar::gar_collapse_feedback, dim3((unsigned)s->batch), dim3(256), 0, s->stream, s->d_meta, s->d_fac,
                       (long long)s->fac_doubles, s->batch, flags, 1);
    s->fold_expanded = false;
  } else {
    hipLaunchKernelGGL(s->leg_collapse_kernel ? s->leg_collapse_kernel : gar::gar_collapse_feedback,
                       dim3((unsigned)s->batch), dim3(256), 0, s->stream, s->d_meta, s->d_fac,
                       (long long)s->fac_doubles, s->batch, (const int *)nullptr, 0);
  }
  HIP_TRY(hipGetLastError());
  return GAR_HIP_OK; // (asynchronous: whoever reads the gains next is ordered behind it on the stream)
}

int gar_hip_cycle_append(gar_hip_solver *s, const int32_t d[5]) {
  GAR_GUARD(s);
  if (!s || !d)
    return fail(GAR_HIP_ERR_ARG, "bad argument");
  GAR_MULTI(s, multi_cycle_append(s, d));
  if (s->ev_pref)
    HIP_TRY(hipEventSynchronize(s->ev_pref));
  s->pref_b = -1;
  s->eager_fwd = false;
  const int N = s->horizon;
  if (N < 1)
    return GAR_HIP_OK;
  if (int rc = commit(s)) // host writes addressed the old stage numbering: flush them first
    return rc;
  // new dims sequence (the CALLER's dimensions): old[1..N-1], new knot, old[N]  (rotate_vec_left(datas,0,1) +
  // re-created last-but-one factor, proximal-riccati.hxx:79-86)
  const std::vector<int32_t> &od = s->user_dims5;
  std::vector<int32_t> nd(od.size());
  for (int t = 0; t + 1 < N; ++t)
    std::copy(&od[5 * (t + 1)], &od[5 * (t + 1)] + 5, &nd[5 * t]);
  std::copy(d, d + 5, &nd[5 * (N - 1)]);
  std::copy(&od[5 * N], &od[5 * N] + 5, &nd[5 * N]);
  bool uniform = true;
  for (int t = 0; t < N; ++t)
    for (int k = 0; k < 5; ++k)
      uniform &= (nd[5 * t + k] == d[k]) && (od[5 * t + k] == d[k]);
  if (uniform && s->num_legs == 1) {
    // A RING, not a copy: logical stage t (< N) moves to slot (t + ring0) mod N, i.e. every record
    // stays where it is -- problem knots and factors alike (rotate_vec_left(datas, 0, 1)) -- and what
    // was stage 0 becomes the last-but-one slot the caller overwrites next.  The kernels get ring0
    // (specialised families) or the rotated per-stage offsets (generic / dense kernels read them
    // from the stage descriptors): 14 KB of descriptors, asynchronously; no record is touched, the
    // stream is not synchronised.  (Round 1 copied (N-1) x 54 KB per problem and synchronised twice.)
    // (the caller's and the device's dimension sequences are unchanged; a padded solver's dummy rows stay in place)
    s->ring0 = (s->ring0 + 1) % N;
    for (int t = 0; t < N; ++t) {
      const int64_t p = (t + s->ring0) % N;
      s->meta[t].in_off = s->uni_in0 + p * s->uni_in_rec;
      s->meta[t].fac_off = p * s->uni_fac_rec;
    }
    HIP_TRY(hipMemcpyAsync(s->d_meta, s->meta.data(), sizeof(gar_stage_meta) * s->meta.size(),
                           hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemsetAsync(s->d_init, 0, sizeof(double) * (size_t)s->init_doubles * s->batch, s->stream));
    // the last-but-one factor is re-created (zero) like the reference's StageFactor (:84-85): one
    // strided memset over the batch, asynchronous
    HIP_TRY(hipMemset2DAsync(s->d_fac + s->meta[N - 1].fac_off, sizeof(double) * (size_t)s->fac_doubles, 0,
                             sizeof(double) * (size_t)s->uni_fac_rec, (size_t)s->batch, s->stream));
    if (s->padded) { // the slot's dummy diagonals (Q = I, R = I on the padded part) are part of the knot the caller
      // uploads next through gar_hip_upload_stage, which writes whole padded blocks: nothing to do here
    }
    return GAR_HIP_OK;
  }
  // dimensions changed (or leg mode: "just reinitialise everything", parallel-solver.hxx:246-258): rebuild the
  // layout and the buffers.  The new configuration -- padding, layouts, LDS plan, kernel family -- is validated on
  // a TRIAL object first: on failure this solver is untouched (ring position, stage offsets, device records and
  // descriptors included).  On success every device pointer handed out earlier (gar_hip_device_*) is invalid and
  // must be fetched again; resident problem data is not carried over.
  {
    gar_hip_solver trial;
    trial.device = s->device;
    trial.horizon = s->horizon;
    trial.user_nc0 = s->user_nc0;
    trial.batch = s->batch;
    trial.num_legs = s->num_legs;
    trial.rank = s->rank;
    trial.world = s->world;
    trial.dense = s->dense;
    trial.user_dims5 = nd;
    const int rc = configure(&trial);
    delete trial.ulay;
    delete trial.flay;
    trial.ulay = trial.flay = nullptr;
    if (rc != GAR_HIP_OK)
      return rc; // g_last_error says why
  }
  HIP_TRY(hipStreamSynchronize(s->stream));
  free_device(s);
  s->staged = s->dirty = false;
  s->user_dims5 = nd;
  // the pipelined schedule belongs to the kernel family that was bound: it is re-validated against the new one (its
  // half streams and events are kept), and silently off when the new shape has no such family
  const int wanted = s->pipe_requested; // 2: the caller's explicit wish; -1: the library's own choice; 0: off
  s->pipe_halves = 0;
  s->pipe_forked = false;
  s->pipe_evB_valid[0] = s->pipe_evB_valid[1] = false;
  if (int rc = configure(s))
    return rc;
  if (int rc = allocate(s))
    return rc;
  if (wanted != 0) { // (kept as the caller's wish even where this shape refuses it: a later rebuild may serve it again)
    if (gar_hip_set_pipeline(s, wanted) != GAR_HIP_OK)
      s->pipe_halves = 0;
    s->pipe_requested = wanted;
  }
  return GAR_HIP_OK;
}


#include "gar_entry_io.hpp"

} // extern "C"
