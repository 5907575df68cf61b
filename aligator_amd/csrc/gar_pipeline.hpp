// gar_pipeline.hpp -- the pipelined sweep of gar_hip_set_pipeline (include/gar_hip.h): host side.  Included by
// gar_hip.cpp inside its anonymous namespace, behind launch_backward / launch_forward (it uses make_mfma_params,
// make_mfma_fwd_params, make_params, commit and the solver struct).  The device side: gar_forward_lean.hpp (the roll-out
// that fits beside a backward wave), gar_wave.hpp (gar_backward_wave_half: the sweep under its half-batch launch name,
// MfmaParams::init_small), gar_generic.hpp (gar_initial_wave on the flagged problems).  DESIGN.md 5.1b.
#pragma once

// ---- the pipelined sweep (gar_hip_set_pipeline) -------------------------------------------------------------------
// LDS of a CU, planned: 4 backward waves (one per SIMD, wave_lds_doubles_small each: the launch without the fused
// initial stage's kkt0 overlay) + ONE forward workgroup (gar_forward_lean: four waves, one per SIMD).  The forward
// launch ASKS for more than half of the CU's LDS, so that a second forward workgroup never fits: two forward waves
// on a SIMD would take the registers a backward wave needs (432 + 80 of 512).
// (kCuLdsBytes, lds_round: gar_hip.cpp, ahead of launch_forward)
int pipe_plan(gar_hip_solver *s, size_t lean_used) {
  const size_t bwd = lds_round((size_t)s->wave_lds_doubles_small * sizeof(double));
  size_t ask = lds_round(lean_used);
  if (ask <= kCuLdsBytes / 2)
    ask = lds_round(kCuLdsBytes / 2 + 1);
  if (4 * bwd + ask > kCuLdsBytes)
    return fail(GAR_HIP_ERR_UNSUPPORTED, "pipelined sweep: four backward waves (" + std::to_string(bwd) +
                                             " B of LDS each) and one forward workgroup (" + std::to_string(ask) +
                                             " B) do not share a CU");
  s->lean_fwd_lds_bytes = ask;
  return GAR_HIP_OK;
}
inline bool pipe_on(const gar_hip_solver *s) { return s && s->pipe_halves == 2; }
inline void pipe_range(const gar_hip_solver *s, int h, int *b0, int *nb) {
  const int first = (s->batch + 1) / 2;
  *b0 = h == 0 ? 0 : first;
  *nb = h == 0 ? first : s->batch - first;
}
// the caller's stream behind everything the half streams hold
int pipe_join(gar_hip_solver *s) {
  if (!s->pipe_forked)
    return GAR_HIP_OK;
  for (int h = 0; h < 2; ++h) {
    HIP_TRY(hipEventRecord(s->pipe_evF[h], s->pipe_stream[h])); // (covers the sweeps too: stream order)
    HIP_TRY(hipStreamWaitEvent(s->stream, s->pipe_evF[h], 0));
  }
  s->pipe_forked = false;
  return GAR_HIP_OK;
}
void pipe_autojoin(const gar_hip_solver *s) {
  if (s && s->pipe_forked)
    (void)pipe_join(const_cast<gar_hip_solver *>(s));
}
// ... and the half streams behind what the caller's stream holds (uploads, a previous unpipelined sweep)
// (`again`: a sweep re-records the fork even while the halves are forked -- what the caller enqueued on the solver's
// stream since the first fork, a producer kernel or a copy into the records, is then waited for as gar_hip.h
// promises; an event on a stream that holds nothing new is complete at once, so the halves keep overlapping)
int pipe_fork(gar_hip_solver *s, bool again = false) {
  if (s->pipe_forked && !again)
    return GAR_HIP_OK;
  HIP_TRY(hipEventRecord(s->pipe_evFork, s->stream));
  for (int h = 0; h < 2; ++h)
    HIP_TRY(hipStreamWaitEvent(s->pipe_stream[h], s->pipe_evFork, 0));
  s->pipe_forked = true;
  return GAR_HIP_OK;
}
int pipe_backward(gar_hip_solver *s, double mueq) {
  RoctxRange range_("gar::backwardImpl+factor_initial (pipelined)");
  s->eager_fwd = false;
  if (s->ev_pref) {
    HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_pref, 0));
    s->pref_b = -1;
  }
  if (s->dirty) { // staged host data goes out on the caller's stream: order it, then fork again
    if (int rc = pipe_join(s))
      return rc;
    if (int rc = commit(s))
      return rc;
  }
  if (int rc = pipe_fork(s, /*again=*/true))
    return rc;
  const gar::MfmaParams M0 = make_mfma_params(s, mueq);
  const gar::GenericParams G0 = make_params(s, mueq);
  for (int h = 0; h < 2; ++h) {
    int b0, nb;
    pipe_range(s, h, &b0, &nb);
    hipStream_t st = s->pipe_stream[h];
    // backward sweeps alternate between the halves: this one starts when the other half's last one has ended --
    // which is also when that half's forward sweep starts (its stream's next kernel)
    if (s->pipe_evB_valid[1 - h])
      HIP_TRY(hipStreamWaitEvent(st, s->pipe_evB[1 - h], 0));
    HIP_TRY(hipMemsetAsync(s->d_status + b0, 0, sizeof(int) * (size_t)nb, st));
    if (h == 0) // the slow-path counters of "the last backward" cover both halves (the second half runs behind this one)
      HIP_TRY(hipMemsetAsync(s->d_status + s->batch, 0, sizeof(int) * 4, st));
    gar::MfmaParams M = M0;
    M.prob += (long long)b0 * M.prob_stride;
    M.fac += (long long)b0 * M.fac_stride;
    M.status += b0;
    M.resume += b0;
    M.init = s->d_init + (long long)b0 * M.init_stride;
    M.init_small = 1;
    M.trace = nullptr;
    if (s->timing)
      HIP_TRY(hipEventRecord(s->pipe_evT[h][0], st));
    hipLaunchKernelGGL(s->wave_half_kernel, dim3((unsigned)nb), dim3(64), (size_t)s->wave_lds_doubles_small * sizeof(double),
                       st, M, nb);
    if (s->timing)
      HIP_TRY(hipEventRecord(s->pipe_evT[h][1], st));
    // the problems whose initial condition is not "x0 given" (no closed form): every other wave leaves at once
    gar::GenericParams G = G0;
    G.prob += (long long)b0 * G.prob_stride;
    G.fac += (long long)b0 * G.fac_stride;
    G.init += (long long)b0 * G.init_stride;
    G.status += b0;
    G.only = M.resume;
    hipLaunchKernelGGL(gar::gar_initial_wave, dim3((unsigned)nb), dim3(64),
                       (size_t)gar::gar_initial_wave_lds_doubles(s->n0, s->nth0) * sizeof(double), st, G);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->pipe_evB[h], st));
    s->pipe_evB_valid[h] = true;
  }
  return GAR_HIP_OK;
}
int pipe_forward(gar_hip_solver *s) {
  RoctxRange range_("gar::forwardImpl (pipelined)");
  if (int rc = pipe_fork(s))
    return rc;
  const gar::MfmaFwdParams F0 = make_mfma_fwd_params(s);
  for (int h = 0; h < 2; ++h) {
    int b0, nb;
    pipe_range(s, h, &b0, &nb);
    hipStream_t st = s->pipe_stream[h];
    gar::MfmaFwdParams F = F0;
    F.fac += (long long)b0 * F.fac_stride;
    F.init += (long long)b0 * F.init_stride;
    F.sol += (long long)b0 * F.sol_stride;
    if (s->timing)
      HIP_TRY(hipEventRecord(s->pipe_evT[h][2], st));
    hipLaunchKernelGGL(s->lean_fwd_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), s->lean_fwd_lds_bytes, st, F, nb);
    HIP_TRY(hipGetLastError());
    if (s->timing)
      HIP_TRY(hipEventRecord(s->pipe_evT[h][3], st));
  }
  return GAR_HIP_OK;
}

