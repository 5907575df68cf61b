// gar_multi.hpp -- ONE process, SEVERAL devices: the horizon of one LQ problem (or of one batch of them) sharded over
// up to GAR_MULTI_MAX_DEVICES GPUs behind the SAME gar_hip_solver handle, so that the one
// `gar::RiccatiSolverBase<double>` object SolverProxDDPTpl holds in `linear_solver_`
// (/root/reference/include/aligator/solvers/proxddp/solver-proxddp.hpp:56,181; calls at solver-proxddp.hxx:608-611)
// can use every GPU of the node without torchrun.  (Included by gar_hip.cpp; not a translation unit of its own.)
//
// In the reference, ParallelRiccatiSolver is ONE object constructed with a thread count
// (gar/parallel-solver.hxx:42-46); its legs are OpenMP threads and the "boundary exchange" is the implicit barrier
// that closes the parallel region (:150-164) before assembleCondensedSystem (:169).  Here:
//
//   device r owns legs [r J / W, (r+1) J / W) of J = num_legs (get_work over devices, :23-28) -- one ranked
//   solver per device (gar_hip_solver_create_ranked), each with its own HIP stream;
//   backward(mu):  every device: leg-parallel backward sweep of its legs            (:150-164), event e_r
//                  every device: waits for the W-1 other events, then GATHERS the boundary tuples
//                  (Vxx | Vxt | Vtt | vx | vt of each leg's first stage, 3 nx^2 + 2 nx doubles = 31.7 KB at nx = 36)
//                  of all devices into its own buffer -- ONE kernel reading the peers' buffers over xGMI
//                  (peer access enabled, "pull"), or W-1 hipMemcpyPeerAsync when a pair has no peer access
//                  ("copy") -- and solves the condensed block-tridiagonal system redundantly (:169-202);
//   forward:       every device: leg-parallel roll-out of its legs                  (:209-243).
//   No host synchronisation inside a sweep; no second exchange (the condensed solution is replicated).
//   xGMI is point-to-point and the payload is tens of KB per leg, so the exchange is latency-bound: one all-to-all
//   read per device and sweep, no O(log J) rounds.
//
// Every other entry point of include/gar_hip.h routes by the stage's owner (uploads, gains, value functions),
// broadcasts (G0 / g0, refinement settings) or merges (solution, bulk read-back: each device gathers and copies
// ONLY its own stages, straight into one pinned host buffer, all devices concurrently).  Entry points that hand
// out or take DEVICE pointers (gar_hip_device_*, gar_hip_upload_packed_device, gar_hip_update_lq_subproblem_device,
// gar_hip_set_stream) have no meaning across devices and answer GAR_HIP_ERR_UNSUPPORTED / null.
#pragma once

#define GAR_MULTI_MAX_DEVICES 16

namespace gar {
struct MultiGatherParams {
  const double *src[GAR_MULTI_MAX_DEVICES]; // every device's local tuples [batch][chunk of legs][tuple]
  double *dst;                              // this device's gathered buffer [device][batch][chunk][tuple]
  long long chunk;                          // doubles per device
};
// grid (blocks, W), 256 threads: dst[r][i] = src[r][i] -- src[r] may live on a peer device (xGMI reads)
__global__ void __launch_bounds__(256) gar_multi_gather(MultiGatherParams P) {
  const double *s = P.src[blockIdx.y];
  double *d = P.dst + (long long)blockIdx.y * P.chunk;
  const long long step = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P.chunk; i += step)
    d[i] = s[i];
}
} // namespace gar

struct gar_multi {
  std::vector<gar_hip_solver *> subs; // subs[r]: the ranked solver of device r (rank r of world W)
  std::vector<hipEvent_t> ev_legs;    // device r's leg sweep of the current backward is done
  std::vector<hipEvent_t> ev_gath;    // device r has read every peer's tuples of the current backward
  std::vector<int> t_lo, t_hi;        // stages [t_lo, t_hi) live on device r
  std::vector<int> owner;             // per stage
  bool pull = false;                  // exchange: one gather kernel per device over peer-mapped buffers
  bool swept = false;
  double *h_results = nullptr;        // pinned, merged [solution | ff_all | fb_all] of one problem
};

namespace {

inline const gar_hip_solver *caller_layout(const gar_hip_solver *s) { return s->ulay ? s->ulay : s; }

void multi_ranges(gar_hip_solver *s) {
  gar_multi *M = s->multi;
  const int W = (int)M->subs.size(), J = s->num_legs, N = s->horizon;
  M->t_lo.assign(W, 0);
  M->t_hi.assign(W, 0);
  M->owner.assign((size_t)N + 1, 0);
  for (int r = 0; r < W; ++r) {
    const int l0 = (int)((long long)r * J / W), l1 = (int)((long long)(r + 1) * J / W);
    int e;
    gar_get_work(N, l0, J, &M->t_lo[r], &e);
    gar_get_work(N, l1 - 1, J, &e, &M->t_hi[r]);
    for (int t = M->t_lo[r]; t < M->t_hi[r]; ++t)
      M->owner[(size_t)t] = r;
  }
}

int multi_sync(gar_hip_solver *s);
void multi_destroy(gar_hip_solver *s) {
  gar_multi *M = s->multi;
  // every device idle FIRST: in pull mode the gather kernels on the other devices' streams read this device's
  // boundary buffer across xGMI, and hipFree on one device is not ordered against a peer's stream
  for (gar_hip_solver *q : M->subs)
    if (q) {
      DeviceGuard g(q->device);
      (void)hipStreamSynchronize(q->stream);
    }
  for (size_t r = 0; r < M->subs.size(); ++r) {
    if (!M->subs[r])
      continue;
    {
      DeviceGuard g(M->subs[r]->device);
      (void)hipStreamSynchronize(M->subs[r]->stream);
      if (r < M->ev_legs.size() && M->ev_legs[r])
        (void)hipEventDestroy(M->ev_legs[r]);
      if (r < M->ev_gath.size() && M->ev_gath[r])
        (void)hipEventDestroy(M->ev_gath[r]);
    }
    gar_hip_solver_destroy(M->subs[r]);
  }
  if (M->h_results)
    (void)hipHostFree(M->h_results);
  delete M;
  s->multi = nullptr;
  delete s->ulay;
  delete s->flay;
  delete s;
}

// ---- problem upload ------------------------------------------------------------------------------------------
int multi_set_init(gar_hip_solver *s, int b, const double *G0, const double *g0) {
  for (gar_hip_solver *q : s->multi->subs) // every device solves the condensed system: all need G0, g0
    if (int rc = gar_hip_set_init(q, b, G0, g0))
      return rc;
  return GAR_HIP_OK;
}

// the doubles [lo, hi) of problems [b0, b0 + nb) of `packed` (device layout == caller layout: not padded)
int upload_packed_range(gar_hip_solver *q, int b0, int nb, const double *packed, int64_t lo, int64_t hi) {
  DeviceGuard g(q->device);
  const int64_t P = q->prob_doubles;
  for (int b = b0; b < b0 + nb; ++b) {
    const double *src = packed + (int64_t)(b - b0) * P + lo;
    if (q->staged) {
      std::memcpy(q->h_prob + (int64_t)b * P + lo, src, sizeof(double) * (size_t)(hi - lo));
      mark_dirty(q, b, lo, hi);
    } else {
      HIP_TRY(hipMemcpyAsync(q->d_prob + (int64_t)b * P + lo, src, sizeof(double) * (size_t)(hi - lo),
                             hipMemcpyHostToDevice, q->stream));
    }
  }
  if (!q->staged)
    HIP_TRY(hipStreamSynchronize(q->stream));
  return GAR_HIP_OK;
}

int multi_upload_packed(gar_hip_solver *s, int b0, int nb, const double *packed) {
  gar_multi *M = s->multi;
  if (!packed || b0 < 0 || nb < 0 || b0 + nb > s->batch)
    return fail(GAR_HIP_ERR_ARG, "gar_hip_upload_packed: bad argument");
  const int N = s->horizon;
  if (s->padded) { // the caller's records, knot by knot through the padding path of the stage's owner
    const gar_hip_solver *u = s->ulay;
    for (int b = b0; b < b0 + nb; ++b) {
      const double *rec = packed + (int64_t)(b - b0) * u->prob_doubles;
      for (int t = 0; t <= N; ++t) {
        const gar_stage_meta &m = u->meta[t];
        const gar_knot_offsets o = gar_knot_layout(m.nx, m.nu, 0, m.nx2, 0);
        const double *k = rec + m.in_off;
        if (int rc = gar_hip_upload_stage(M->subs[(size_t)M->owner[(size_t)t]], b, t, k + o.Q, k + o.S, k + o.R, k + o.q,
                                          k + o.r, k + o.A, k + o.B, k + o.f, nullptr, nullptr, nullptr, nullptr,
                                          nullptr, nullptr, nullptr, nullptr))
          return rc;
      }
      if (int rc = multi_set_init(s, b, rec + u->G0_off, rec + u->g0_off))
        return rc;
    }
    return GAR_HIP_OK;
  }
  const int64_t head = s->meta[0].in_off; // G0 | g0
  for (size_t r = 0; r < M->subs.size(); ++r) {
    const int64_t lo = s->meta[(size_t)M->t_lo[r]].in_off;
    const int64_t hi = M->t_hi[r] <= N ? s->meta[(size_t)M->t_hi[r]].in_off : s->prob_doubles;
    if (r > 0 && head > 0)
      if (int rc = upload_packed_range(M->subs[r], b0, nb, packed, 0, head))
        return rc;
    if (int rc = upload_packed_range(M->subs[r], b0, nb, packed, r == 0 ? 0 : lo, hi))
      return rc;
  }
  return GAR_HIP_OK;
}

int multi_download_packed(gar_hip_solver *s, int b0, int nb, double *packed) {
  gar_multi *M = s->multi;
  if (!packed || b0 < 0 || nb < 0 || b0 + nb > s->batch)
    return fail(GAR_HIP_ERR_ARG, "gar_hip_download_packed: bad argument");
  const gar_hip_solver *L = caller_layout(s);
  const int N = s->horizon;
  std::vector<double> tmp((size_t)L->prob_doubles * (size_t)nb);
  for (size_t r = 0; r < M->subs.size(); ++r) {
    if (int rc = gar_hip_download_packed(M->subs[r], b0, nb, tmp.data()))
      return rc;
    const int64_t lo = r == 0 ? 0 : L->meta[(size_t)M->t_lo[r]].in_off;
    const int64_t hi = M->t_hi[r] <= N ? L->meta[(size_t)M->t_hi[r]].in_off : L->prob_doubles;
    for (int k = 0; k < nb; ++k)
      std::memcpy(packed + (int64_t)k * L->prob_doubles + lo, tmp.data() + (int64_t)k * L->prob_doubles + lo,
                  sizeof(double) * (size_t)(hi - lo));
  }
  return GAR_HIP_OK;
}

// ---- the sweep -------------------------------------------------------------------------------------------------
int multi_backward_legs(gar_hip_solver *s, double mueq) {
  gar_multi *M = s->multi;
  const size_t W = M->subs.size();
  for (size_t r = 0; r < W; ++r) {
    gar_hip_solver *q = M->subs[r];
    DeviceGuard g(q->device);
    // the previous sweep's readers of this device's tuples must be done before they are overwritten
    if (M->swept)
      for (size_t p = 0; p < W; ++p)
        if (p != r)
          HIP_TRY(hipStreamWaitEvent(q->stream, M->ev_gath[p], 0));
    if (int rc = gar_hip_backward_legs_async(q, mueq)) // parallel-solver.hxx:150-164, this device's legs
      return rc;
    HIP_TRY(hipEventRecord(M->ev_legs[r], q->stream));
  }
  return GAR_HIP_OK;
}

int multi_exchange_and_condensed(gar_hip_solver *s) {
  gar_multi *M = s->multi;
  const size_t W = M->subs.size();
  const long long chunk = (long long)s->batch * M->subs[0]->legs_per_rank * s->tuple_doubles;
  RoctxRange range_("gar::multi_device_boundary_exchange+condensed");
  for (size_t r = 0; r < W; ++r) {
    gar_hip_solver *q = M->subs[r];
    DeviceGuard g(q->device);
    for (size_t p = 0; p < W; ++p)
      if (p != r)
        HIP_TRY(hipStreamWaitEvent(q->stream, M->ev_legs[p], 0));
    if (M->pull) { // ONE kernel: W chunks read (W-1 of them over xGMI), written into this device's gathered buffer
      gar::MultiGatherParams P{};
      for (size_t p = 0; p < W; ++p)
        P.src[p] = M->subs[p]->d_bound_local;
      P.dst = q->d_bound_all;
      P.chunk = chunk;
      const unsigned blocks = (unsigned)std::max<long long>(1, std::min<long long>(64, (chunk + 255) / 256));
      hipLaunchKernelGGL(gar::gar_multi_gather, dim3(blocks, (unsigned)W), dim3(256), 0, q->stream, P);
      HIP_TRY(hipGetLastError());
    } else {
      for (size_t p = 0; p < W; ++p) {
        double *dst = q->d_bound_all + (long long)p * chunk;
        const double *src = M->subs[p]->d_bound_local;
        if (M->subs[p]->device == q->device)
          HIP_TRY(hipMemcpyAsync(dst, src, sizeof(double) * (size_t)chunk, hipMemcpyDeviceToDevice, q->stream));
        else
          HIP_TRY(hipMemcpyPeerAsync(dst, q->device, src, M->subs[p]->device, sizeof(double) * (size_t)chunk, q->stream));
      }
    }
    HIP_TRY(hipEventRecord(M->ev_gath[r], q->stream));
    if (int rc = gar_hip_condensed_solve_async(q)) // :169-202, redundantly on every device
      return rc;
  }
  M->swept = true;
  return GAR_HIP_OK;
}

int multi_forward(gar_hip_solver *s) {
  for (gar_hip_solver *q : s->multi->subs)
    if (int rc = gar_hip_forward_legs_async(q)) // :209-243
      return rc;
  return GAR_HIP_OK;
}

int multi_sync(gar_hip_solver *s) {
  for (gar_hip_solver *q : s->multi->subs) {
    DeviceGuard g(q->device);
    HIP_TRY(hipStreamSynchronize(q->stream));
  }
  return GAR_HIP_OK;
}

// problems whose backward failed on ANY device
int multi_num_failed(gar_hip_solver *s) {
  std::vector<int> acc((size_t)s->batch, 0), st((size_t)s->batch);
  for (gar_hip_solver *q : s->multi->subs) {
    DeviceGuard g(q->device);
    if (hipMemcpyAsync(st.data(), q->d_status, sizeof(int) * st.size(), hipMemcpyDeviceToHost, q->stream) != hipSuccess ||
        hipStreamSynchronize(q->stream) != hipSuccess)
      return -1;
    for (size_t i = 0; i < st.size(); ++i)
      acc[i] |= st[i];
  }
  int n = 0;
  for (int v : acc)
    n += (v != 0);
  s->last_failed = n;
  return n;
}

int multi_counters(gar_hip_solver *s, int64_t out[2], int (*get)(gar_hip_solver *, int64_t *)) {
  if (!out)
    return fail(GAR_HIP_ERR_ARG, "bad argument");
  out[0] = out[1] = 0;
  for (gar_hip_solver *q : s->multi->subs) {
    int64_t c[2];
    if (int rc = get(q, c))
      return rc;
    out[0] += c[0];
    out[1] += c[1];
  }
  return GAR_HIP_OK;
}

// ---- results ---------------------------------------------------------------------------------------------------
// the four parts of a packed solution record of the caller: [begin, end) of device r's stages in each
struct SolRange { int64_t x0, x1, u0, u1, v0, v1, l0, l1; };
SolRange multi_sol_range(const gar_hip_solver *s, size_t r) {
  const gar_multi *M = s->multi;
  const gar_hip_solver *L = caller_layout(s);
  const int N = s->horizon, lo = M->t_lo[r], hi = M->t_hi[r];
  SolRange R;
  int64_t nl = L->nc0;
  for (int t = 0; t < N; ++t)
    nl += L->meta[(size_t)t].nx2;
  R.x0 = L->meta[(size_t)lo].x_off - L->sol_x;
  R.x1 = (hi <= N ? L->meta[(size_t)hi].x_off : L->sol_u) - L->sol_x;
  R.u0 = L->meta[(size_t)lo].u_off - L->sol_u;
  R.u1 = (hi <= N ? L->meta[(size_t)hi].u_off : L->sol_v) - L->sol_u;
  R.v0 = L->meta[(size_t)lo].v_off - L->sol_v;
  R.v1 = (hi <= N ? L->meta[(size_t)hi].v_off : L->sol_l) - L->sol_v;
  // lbdas[t] travels with stage t: at a leg start it comes from the condensed solution every device holds
  // (parallel-solver.hxx:215-220), otherwise from the leg that holds stage t - 1 -- the same leg
  R.l0 = lo == 0 ? 0 : L->meta[(size_t)lo].l_off - L->sol_l;
  R.l1 = hi <= N ? L->meta[(size_t)hi].l_off - L->sol_l : nl;
  return R;
}

int multi_get_solution(gar_hip_solver *s, int b, double *xs, double *us, double *vs, double *lbdas) {
  gar_multi *M = s->multi;
  const gar_hip_solver *L = caller_layout(s);
  int64_t nl = L->nc0;
  for (int t = 0; t < s->horizon; ++t)
    nl += L->meta[(size_t)t].nx2;
  std::vector<double> tx((size_t)(L->sol_u - L->sol_x)), tu((size_t)(L->sol_v - L->sol_u)), tv((size_t)(L->sol_l - L->sol_v)),
      tl((size_t)nl);
  for (size_t r = 0; r < M->subs.size(); ++r) {
    if (int rc = gar_hip_get_solution(M->subs[r], b, xs ? tx.data() : nullptr, us ? tu.data() : nullptr,
                                      vs ? tv.data() : nullptr, lbdas ? tl.data() : nullptr))
      return rc;
    const SolRange R = multi_sol_range(s, r);
    if (xs)
      std::copy(tx.begin() + R.x0, tx.begin() + R.x1, xs + R.x0);
    if (us)
      std::copy(tu.begin() + R.u0, tu.begin() + R.u1, us + R.u0);
    if (vs)
      std::copy(tv.begin() + R.v0, tv.begin() + R.v1, vs + R.v0);
    if (lbdas)
      std::copy(tl.begin() + R.l0, tl.begin() + R.l1, lbdas + R.l0);
  }
  return GAR_HIP_OK;
}

int multi_fetch_results(gar_hip_solver *s, int b, int what) {
  gar_multi *M = s->multi;
  const gar_hip_solver *L = caller_layout(s);
  const size_t nsol = (size_t)L->sol_doubles, ngain = (size_t)(L->ff_all_doubles + L->fb_all_doubles);
  if (!M->h_results) {
    double *h = nullptr;
    const hipError_t e = gar_host_malloc((void **)&h, sizeof(double) * (nsol + ngain), hipHostMallocPortable);
    if (e != hipSuccess)
      return fail(GAR_HIP_ERR_DEVICE, std::string("gar_hip_fetch_results: ") + hipGetErrorString(e));
    std::memset(h, 0, sizeof(double) * (nsol + ngain));
    M->h_results = h;
  }
  // every device gathers and copies its own stages only, all devices at once; then one wait per device
  for (size_t r = 0; r < M->subs.size(); ++r) {
    DeviceGuard g(M->subs[r]->device);
    if (int rc = fetch_results_impl(M->subs[r], b, what, M->t_lo[r], M->t_hi[r], M->h_results, false))
      return rc;
  }
  if (int rc = multi_sync(s))
    return rc;
  if (what & 1)
    for (size_t r = 0; r < M->subs.size(); ++r) {
      gar_hip_solver *q = M->subs[r];
      if (q->padded)
        strip_solution_rec(q, q->h_results + nsol + ngain, q->h_results);
      const SolRange R = multi_sol_range(s, r);
      const double *src = q->h_results;
      double *dst = M->h_results;
      std::copy(src + L->sol_x + R.x0, src + L->sol_x + R.x1, dst + L->sol_x + R.x0);
      std::copy(src + L->sol_u + R.u0, src + L->sol_u + R.u1, dst + L->sol_u + R.u0);
      std::copy(src + L->sol_v + R.v0, src + L->sol_v + R.v1, dst + L->sol_v + R.v0);
      std::copy(src + L->sol_l + R.l0, src + L->sol_l + R.l1, dst + L->sol_l + R.l0);
    }
  return GAR_HIP_OK;
}

int multi_cycle_append(gar_hip_solver *s, const int32_t d[5]) {
  gar_multi *M = s->multi;
  const int N = s->horizon;
  if (N < 1)
    return GAR_HIP_OK;
  // leg mode: "just reinitialise everything" (parallel-solver.hxx:246-258) -- on every device; the first device's
  // trial configuration rejects a bad knot before anything changes.  Every device idle first: re-initialising a
  // sub-solver frees the boundary buffer its peers' gather kernels may still be reading.
  if (int rc = multi_sync(s))
    return rc;
  for (gar_hip_solver *q : M->subs)
    if (int rc = gar_hip_cycle_append(q, d))
      return rc;
  const std::vector<int32_t> od = s->user_dims5;
  std::vector<int32_t> &nd = s->user_dims5;
  for (int t = 0; t + 1 < N; ++t)
    std::copy(&od[5 * (size_t)(t + 1)], &od[5 * (size_t)(t + 1)] + 5, &nd[5 * (size_t)t]);
  std::copy(d, d + 5, &nd[5 * (size_t)(N - 1)]);
  if (int rc = configure(s))
    return rc;
  multi_ranges(s);
  if (M->h_results) // the merged record's size follows the layout
    (void)hipHostFree(M->h_results);
  M->h_results = nullptr;
  M->swept = false;
  return GAR_HIP_OK;
}

int multi_last_kernel_ms(gar_hip_solver *s, double out[3]) {
  if (!out)
    return fail(GAR_HIP_ERR_ARG, "bad argument");
  out[0] = out[1] = out[2] = 0.0;
  for (gar_hip_solver *q : s->multi->subs) { // the devices run concurrently: the slowest bounds the sweep
    double o[3];
    if (int rc = gar_hip_last_kernel_ms(q, o))
      return rc;
    for (int k = 0; k < 3; ++k)
      out[k] = std::max(out[k], o[k]);
  }
  return GAR_HIP_OK;
}

template <class F> int multi_all(gar_hip_solver *s, F f) {
  for (gar_hip_solver *q : s->multi->subs)
    if (int rc = f(q))
      return rc;
  return GAR_HIP_OK;
}

inline gar_hip_solver *multi_owner(gar_hip_solver *s, int t) { return s->multi->subs[(size_t)s->multi->owner[(size_t)t]]; }

gar_hip_solver *create_impl(int device, int horizon, const int32_t *dims5, int nc0, int batch, int num_legs, int rank,
                            int world, bool dense);

gar_hip_solver *multi_create(int ndev, const int *dev_ids, int horizon, const int32_t *dims5, int nc0, int batch,
                             int num_legs) {
  if (ndev < 1 || ndev > GAR_MULTI_MAX_DEVICES || !dev_ids || horizon < 0 || !dims5 || nc0 < 0 || batch < 1 ||
      num_legs < 1) {
    fail(GAR_HIP_ERR_ARG, "gar_hip_multi_create: bad argument");
    return nullptr;
  }
  if (ndev == 1) // one device: the plain solver (serial in time when num_legs == 1)
    return create_impl(dev_ids[0], horizon, dims5, nc0, batch, num_legs, 0, 1, false);
  if (num_legs < ndev) {
    fail(GAR_HIP_ERR_ARG, "gar_hip_multi_create: every device needs a leg (num_legs >= number of devices)");
    return nullptr;
  }
  const int ndevices = gar_hip_device_count();
  for (int r = 0; r < ndev; ++r)
    if (dev_ids[r] < 0 || dev_ids[r] >= ndevices) {
      fail(GAR_HIP_ERR_DEVICE, "gar_hip_multi_create: no such HIP device (the HIP backend has no CPU fallback)");
      return nullptr;
    }
  gar_hip_solver *s = new gar_hip_solver();
  s->multi = new gar_multi();
  s->device = dev_ids[0];
  s->horizon = horizon;
  s->user_nc0 = nc0;
  s->batch = batch;
  s->num_legs = num_legs;
  s->rank = 0;
  s->world = ndev;
  s->user_dims5.assign(dims5, dims5 + 5 * ((size_t)horizon + 1));
  normalise_terminal(s);
  gar_multi *M = s->multi;
  M->subs.assign((size_t)ndev, nullptr);
  M->ev_legs.assign((size_t)ndev, nullptr);
  M->ev_gath.assign((size_t)ndev, nullptr);
  if (configure(s) != GAR_HIP_OK) { // the layout only: this object owns no device memory
    multi_destroy(s);
    return nullptr;
  }
  for (int r = 0; r < ndev; ++r) {
    M->subs[(size_t)r] = create_impl(dev_ids[r], horizon, dims5, nc0, batch, num_legs, r, ndev, false);
    if (!M->subs[(size_t)r]) {
      const std::string why = g_last_error;
      multi_destroy(s);
      fail(GAR_HIP_ERR_DEVICE, why);
      return nullptr;
    }
    DeviceGuard g(dev_ids[r]);
    if (hipEventCreateWithFlags(&M->ev_legs[(size_t)r], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&M->ev_gath[(size_t)r], hipEventDisableTiming) != hipSuccess) {
      multi_destroy(s);
      fail(GAR_HIP_ERR_DEVICE, "hipEventCreate failed");
      return nullptr;
    }
  }
  multi_ranges(s);
  // peer access between every pair of distinct devices: the gather kernel reads the peers' tuples in place
  bool peers = true;
  for (int r = 0; r < ndev && peers; ++r)
    for (int p = 0; p < ndev && peers; ++p) {
      if (dev_ids[r] == dev_ids[p])
        continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, dev_ids[r], dev_ids[p]) != hipSuccess || !can)
        peers = false;
    }
  const char *ex = gar_option("GAR_HIP_MULTI_EXCHANGE");
  const bool want_copy = ex && std::string(ex) == "copy";
  if (peers && !want_copy)
    for (int r = 0; r < ndev && peers; ++r) {
      DeviceGuard g(dev_ids[r]);
      for (int p = 0; p < ndev && peers; ++p) {
        if (dev_ids[r] == dev_ids[p])
          continue;
        const hipError_t e = hipDeviceEnablePeerAccess(dev_ids[p], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
          peers = false;
        (void)hipGetLastError();
      }
    }
  M->pull = peers && !want_copy;
  return s;
}

} // namespace
