// gar_launch.hpp -- kernel parameter blocks and launches of the sweeps: launch_backward, launch_forward, the pipelined schedule (gar_pipeline.hpp), launch_condensed.
// Part of the ONE translation unit gar_hip.cpp (included in place: it uses the solver struct and the helpers defined
// above its include line); split out for readability only.
#pragma once

// parameters of the serial specialised sweeps (gar_backward_mfma, gar_backward_wave and its chain)
gar::MfmaParams make_mfma_params(gar_hip_solver *s, double mueq) {
  gar::MfmaParams M{};
  M.prob = s->d_prob;
  M.fac = s->d_fac;
  M.status = s->d_status;
  M.slow = s->d_status + s->batch;
  M.resume = s->d_status + s->batch + 4;
  M.prob_stride = s->prob_doubles;
  M.fac_stride = s->fac_doubles;
  const int N = s->horizon;
  M.in_off0 = s->uni_in0;
  M.in_rec = s->uni_in_rec;
  M.in_offN = s->meta[N].in_off;
  M.fac_rec = s->uni_fac_rec;
  M.fac_offN = s->meta[N].fac_off;
  M.horizon = N;
  M.trace = s->d_trace;
  const bool fused = s->wave_kernel && s->wave_fused_init;
  M.init = fused ? s->d_init : nullptr;
  M.init_stride = s->init_doubles;
  M.G0_off = s->G0_off;
  M.g0_off = s->g0_off;
  M.nc0 = s->nc0;
  M.mueq = mueq;
  M.init_closed = s->init_closed ? 1 : 0;
  M.ring0 = s->ring0;
  {
    const char *sa = gar_option("GAR_HIP_SPD_ACCEPT");
    M.spd_accept = (sa && sa[0] == '0') ? 0 : 1;
  }
  return M;
}

// [l0, l1): the legs swept by this call (default: every leg of this solver; gar_hip_backward_blocks sweeps them in
// chunks, as their knots arrive).  The kernels index legs as blockIdx.x + leg_begin and the tuples as blockIdx.x:
// a chunk is the same launch with leg_begin = l0 and the tuple buffer advanced to leg l0's slot.
int launch_backward(gar_hip_solver *s, double mueq, int l0 = -1, int l1 = -1) {
  RoctxRange range_(s->num_legs > 1 ? "gar::parallel_backward" : "gar::backwardImpl+factor_initial");
  const bool chunk = l0 >= 0;
  if (!chunk)
    l0 = s->leg_begin, l1 = s->leg_end;
  const bool first = l0 == s->leg_begin, last = l1 == s->leg_end;
  const long long tup_shift = (long long)(l0 - s->leg_begin) * s->tuple_doubles;
  if (s->leg_bwd_kernel) {
    gar::LegParams Q = make_leg_params(s);
    Q.leg_begin = l0;
    Q.boundary += tup_shift;
    const dim3 grid((unsigned)(l1 - l0), (unsigned)s->batch);
    if (s->timing && first)
      HIP_TRY(hipEventRecord(s->ev[0], s->stream));
    if (s->fold) { // knots with nc > 0: fold C, d into Q, q (gar_fold.hpp); problems with D != 0 get flagged
      s->fold_mueq = mueq;
      s->fold_expanded = s->coupled_known = false;
      hipLaunchKernelGGL(gar::gar_fold_constraints, dim3((unsigned)(s->horizon + 1), (unsigned)s->batch), dim3(256),
                         fold_lds_bytes(s), s->stream, make_fold_params(s));
    }
    hipLaunchKernelGGL(s->leg_bwd_kernel, grid, dim3(64 * s->leg_waves),
                       (size_t)s->leg_lds_doubles * sizeof(double), s->stream, Q);
    hipLaunchKernelGGL(s->leg_tuple_kernel, grid, dim3(256), 0, s->stream, Q);
    if (s->fold && s->cseg_on) { // ... and are swept by the constrained segment legs (gar_cstr_seg.hpp; every other problem: an early exit)
      const int *flagged = s->d_status + s->batch + 4;
      const int N = s->horizon;
      gar::MfmaParams M{};
      M.prob = s->d_prob;
      M.fac = s->d_fac2;
      M.status = s->d_status;
      M.slow = s->d_status + s->batch;
      M.resume = s->d_cseg_resume;
      M.prob_stride = s->prob_doubles;
      M.fac_stride = s->flay->fac_doubles;
      M.in_off0 = s->uni_in0;
      M.in_rec = s->uni_in_rec;
      M.in_offN = s->meta[N].in_off;
      M.fac_rec = s->cseg.rec;
      M.fac_offN = (long long)N * s->cseg.rec;
      M.horizon = N;
      M.mueq = mueq;
      {
        const char *sa = gar_option("GAR_HIP_SPD_ACCEPT");
        M.spd_accept = (sa && sa[0] == '0') ? 0 : 1;
      }
      // (gar_cstr_seg.hpp) the leg-end stages -- V' = 0: no MFMA work, and the matrix on which Bunch-Kaufman pivots -- by a
      // workgroup each, then the chain once, leg by leg, from the knot below: decoupled -> coupled -> LDS Bunch-Kaufman;
      // CSTR_SEG_LEG_END = 0: the leg ends through the chain too, which then runs in two rounds -- coupled stage and LDS
      // Bunch-Kaufman for ONE stage each, then the same again from the hand-over knot, to the end
      const char *le = gar_option("GAR_HIP_CSTR_SEG_LEG_END");
      if (!(le && le[0] == '0')) {
        hipLaunchKernelGGL(s->cseg.leg_end, grid, dim3((unsigned)s->cseg.stage_threads),
                           (size_t)s->cseg.leg_end_lds_doubles * sizeof(double), s->stream, M, s->num_legs, l0, flagged);
        for (int ph = 0; ph < 3; ++ph)
          hipLaunchKernelGGL(s->cseg.backward[ph], grid, dim3(64), (size_t)s->cseg.backward_lds_doubles * sizeof(double),
                             s->stream, M, s->num_legs, l0, flagged, ph == 0 ? gar::kCsegReenter : 0);
      } else
      for (int round = 0; round < 2; ++round)
        for (int ph = 0; ph < 3; ++ph)
          hipLaunchKernelGGL(s->cseg.backward[ph], grid, dim3(64), (size_t)s->cseg.backward_lds_doubles * sizeof(double),
                             s->stream, M, s->num_legs, l0, flagged,
                             (round == 1 && ph == 0 ? gar::kCsegReenter : 0) | (round == 0 && ph >= 1 ? gar::kCsegSingle : 0));
      gar::CsegParams Cp{};
      Cp.meta = s->d_meta;
      Cp.prob = s->d_prob;
      Cp.fac2 = s->d_fac2;
      Cp.fac = s->d_fac;
      Cp.status = s->d_status;
      Cp.only = flagged;
      Cp.prob_stride = s->prob_doubles;
      Cp.fac_stride = s->fac_doubles;
      Cp.fac2_stride = s->flay->fac_doubles;
      Cp.in_off0 = s->uni_in0;
      Cp.in_rec = s->uni_in_rec;
      Cp.horizon = N;
      Cp.num_legs = s->num_legs;
      Cp.leg_begin = l0;
      Cp.local_legs = l1 - l0;
      Cp.mueq = mueq;
      hipLaunchKernelGGL(s->cseg.chain, grid, dim3((unsigned)s->cseg.chain_threads),
                         (size_t)s->cseg.chain_lds_doubles * sizeof(double), s->stream, Cp);
      hipLaunchKernelGGL(s->cseg.stage, dim3((unsigned)N + 1, (unsigned)s->batch), dim3((unsigned)s->cseg.stage_threads),
                         (size_t)s->cseg.stage_lds_doubles * sizeof(double), s->stream, Cp);
      gar::LegParamParams Lp{};
      Lp.meta = s->d_meta;
      Lp.meta2 = s->d_meta2;
      Lp.prob = s->d_prob;
      Lp.fac2 = s->d_fac2;
      Lp.fac = s->d_fac;
      Lp.boundary = s->d_bound_local + tup_shift;
      Lp.status = s->d_status;
      Lp.prob_stride = s->prob_doubles;
      Lp.fac_stride = s->fac_doubles;
      Lp.fac2_stride = s->flay->fac_doubles;
      Lp.boundary_stride = (long long)s->legs_per_rank * s->tuple_doubles;
      Lp.horizon = N;
      Lp.num_legs = s->num_legs;
      Lp.leg_begin = l0;
      Lp.tuple_doubles = (int)s->tuple_doubles;
      Lp.nxb = s->nxb;
      Lp.nxM = s->dims5[0];
      Lp.nuM = s->dims5[1];
      Lp.local_legs = l1 - l0;
      Lp.only = flagged;
      hipLaunchKernelGGL(gar::gar_leg_param_finish, grid, dim3(1024), 0, s->stream, Lp);
    } else if (s->fold) { // ... and are swept by the generic leg kernels (every other problem: an early exit)
      gar::GenericParams G = make_params(s, mueq);
      G.only = s->d_status + s->batch + 4;
      G.leg_begin = l0;
      G.local_legs = l1 - l0;
      G.boundary += tup_shift;
      hipLaunchKernelGGL(gar::gar_backward_generic, grid, dim3(GAR_BACKWARD_THREADS), (size_t)s->lds.total * sizeof(double), s->stream, G);
    }
    HIP_TRY(hipGetLastError());
    if (s->timing && last)
      HIP_TRY(hipEventRecord(s->ev[1], s->stream));
    return GAR_HIP_OK;
  }
  if (s->seg_bwd_kernel) { // segment legs (gar_leg_seg.hpp): plain part, then the parameter recursion + tuples
    const gar_hip_solver *f = s->flay;
    const int N = s->horizon;
    gar::MfmaParams M{};
    M.prob = s->d_prob;
    M.fac = s->d_fac2;
    M.status = s->d_status;
    M.slow = s->d_status + s->batch;
    M.resume = s->d_status + s->batch + 4;
    M.prob_stride = s->prob_doubles;
    M.fac_stride = f->fac_doubles;
    M.in_off0 = s->uni_in0;
    M.in_rec = s->uni_in_rec;
    M.in_offN = s->meta[N].in_off;
    M.fac_rec = f->uni_fac_rec;
    M.fac_offN = f->meta[N].fac_off;
    M.horizon = N;
    M.mueq = mueq;
    {
      const char *sa = gar_option("GAR_HIP_SPD_ACCEPT");
      M.spd_accept = (sa && sa[0] == '0') ? 0 : 1;
    }
    const dim3 grid((unsigned)(l1 - l0), (unsigned)s->batch);
    if (s->timing && first)
      HIP_TRY(hipEventRecord(s->ev[0], s->stream));
    hipLaunchKernelGGL(s->seg_bwd_kernel, grid, dim3(128), (size_t)s->seg_lds_doubles * sizeof(double), s->stream, M,
                       s->num_legs, l0);
    gar::LegParamParams Q{};
    Q.meta = s->d_meta;
    Q.meta2 = s->d_meta2;
    Q.prob = s->d_prob;
    Q.fac2 = s->d_fac2;
    Q.fac = s->d_fac;
    Q.boundary = s->d_bound_local + tup_shift;
    Q.status = s->d_status;
    Q.prob_stride = s->prob_doubles;
    Q.fac_stride = s->fac_doubles;
    Q.fac2_stride = f->fac_doubles;
    Q.boundary_stride = (long long)s->legs_per_rank * s->tuple_doubles;
    Q.horizon = N;
    Q.num_legs = s->num_legs;
    Q.leg_begin = l0;
    Q.tuple_doubles = (int)s->tuple_doubles;
    Q.nxb = s->nxb;
    Q.nxM = s->dims5[0];
    Q.nuM = s->dims5[1];
    Q.local_legs = l1 - l0;
    // the chain of Vxt alone per leg; everything else of every stage at once; the running sums and the tuples
    {
      hipLaunchKernelGGL(gar::gar_leg_param_chain, grid, dim3(GAR_LEG_PARAM_THREADS),
                         (size_t)gar::leg_chain_lds_doubles(s->dims5[0]) * sizeof(double), s->stream, Q);
      hipLaunchKernelGGL(gar::gar_leg_param_stage, dim3((unsigned)N + 1, (unsigned)s->batch), dim3(GAR_LEG_STAGE_THREADS),
                         (size_t)gar::leg_stage_lds_doubles(s->dims5[0], s->dims5[1]) * sizeof(double), s->stream, Q);
      hipLaunchKernelGGL(gar::gar_leg_param_finish, grid, dim3(1024), 0, s->stream, Q);
    }
    HIP_TRY(hipGetLastError());
    if (s->timing && last)
      HIP_TRY(hipEventRecord(s->ev[1], s->stream));
    return GAR_HIP_OK;
  }
  gar::GenericParams P = make_params(s, mueq);
  if (s->dense) {
    hipLaunchKernelGGL(gar::gar_backward_dense, dim3((unsigned)s->batch), dim3(GAR_DENSE_THREADS),
                       (size_t)s->dense_lds.total * sizeof(double), s->stream, P);
    HIP_TRY(hipGetLastError());
    return GAR_HIP_OK;
  }
  if (s->mfma_kernel || s->wave_kernel) {
    const gar::MfmaParams M = make_mfma_params(s, mueq);
    const bool fused = s->wave_kernel && s->wave_fused_init;
    if (s->timing)
      HIP_TRY(hipEventRecord(s->ev[0], s->stream));
    if (s->wave_kernel) {
      const int wpb = s->waves_per_block;
      hipLaunchKernelGGL(s->wave_kernel, dim3((unsigned)((s->batch + wpb - 1) / wpb)),
                         dim3(s->wave_block_threads * wpb), (size_t)s->wave_lds_doubles * wpb * sizeof(double),
                         s->stream, M, s->batch);
      // constrained sweeps: the chain decoupled stage -> coupled stage -> LDS Bunch-Kaufman (gar_wave.hpp)
      for (auto k : {s->wave_coupled_kernel, s->wave_bk_kernel})
        if (k)
          hipLaunchKernelGGL(k, dim3((unsigned)((s->batch + wpb - 1) / wpb)), dim3(s->wave_block_threads * wpb),
                             (size_t)s->wave_lds_doubles * wpb * sizeof(double), s->stream, M, s->batch);
    } else {
      hipLaunchKernelGGL(s->mfma_kernel, dim3((unsigned)s->batch), dim3(256),
                         (size_t)s->mfma_lds_doubles * sizeof(double), s->stream, M);
    }
    HIP_TRY(hipGetLastError());
    if (s->timing)
      HIP_TRY(hipEventRecord(s->ev[1], s->stream));
    if (fused) {
      // nothing to launch: gar_backward_wave already produced kkt0.ff
    } else if (s->n0 <= 128) { // one wave per problem (wave-scope Bunch-Kaufman handles n <= 128)
      hipLaunchKernelGGL(gar::gar_initial_wave, dim3((unsigned)s->batch), dim3(64),
                         (size_t)gar::gar_initial_wave_lds_doubles(s->n0, s->nth0) * sizeof(double),
                         s->stream, P);
    } else {
      hipLaunchKernelGGL(gar::gar_initial_generic, dim3((unsigned)s->batch), dim3(256),
                         (size_t)s->lds.total * sizeof(double), s->stream, P);
    }
    HIP_TRY(hipGetLastError());
    if (s->timing)
      HIP_TRY(hipEventRecord(s->ev[2], s->stream));
    return GAR_HIP_OK;
  }
  const dim3 grid((unsigned)(l1 - l0), (unsigned)s->batch);
  P.leg_begin = l0;
  P.local_legs = l1 - l0;
  if (P.boundary)
    P.boundary += tup_shift;
  hipLaunchKernelGGL(gar::gar_backward_generic, grid, dim3(GAR_BACKWARD_THREADS),
                     (size_t)s->lds.total * sizeof(double), s->stream, P);
  HIP_TRY(hipGetLastError());
  return GAR_HIP_OK;
}

constexpr size_t kCuLdsBytes = 160 * 1024, kLdsGranule = 1280; // gfx950: 160 KiB per CU, allocated in 320-dword pieces
inline size_t lds_round(size_t b) { return (b + kLdsGranule - 1) / kLdsGranule * kLdsGranule; }

gar::MfmaFwdParams make_mfma_fwd_params(gar_hip_solver *s) {
  gar::MfmaFwdParams F{};
  const int N = s->horizon;
  F.fac = s->d_fac;
  F.init = s->d_init;
  F.sol = s->d_sol;
  F.fac_stride = s->fac_doubles;
  F.init_stride = s->init_doubles;
  F.sol_stride = s->sol_doubles;
  F.fac_rec = s->uni_fac_rec;
  F.fac_offN = s->meta[N].fac_off;
  F.horizon = N;
  F.nc0 = s->nc0;
  F.sol_u = (int)s->sol_u;
  F.sol_l = (int)s->sol_l;
  F.sol_v = (int)s->sol_v;
  F.ring0 = s->ring0;
  return F;
}

int launch_forward(gar_hip_solver *s, const double *theta_dev) {
  RoctxRange range_(s->num_legs > 1 ? "gar::parallel_forward" : "gar::forwardImpl");
  if (s->leg_fwd_kernel) {
    gar::LegParams Q = make_leg_params(s);
    if (s->timing)
      HIP_TRY(hipEventRecord(s->ev[3], s->stream));
    hipLaunchKernelGGL(s->leg_fwd_kernel, dim3((unsigned)(s->leg_end - s->leg_begin), (unsigned)s->batch),
                       dim3(64), 0, s->stream, Q);
    if (s->fold) { // v_t = zff + Z x_t on this rank's stages; flagged problems: the generic roll-out
      hipLaunchKernelGGL(gar::gar_constraint_multipliers, dim3((unsigned)(s->horizon + 1), (unsigned)s->batch), dim3(64), 0,
                         s->stream, make_fold_params(s));
      const char *cf = gar_option("GAR_HIP_CSTR_SEG_FORWARD");
      if (s->cseg_on && !(cf && std::string(cf) == "generic")) { // the constrained segment legs' own roll-out (gar_cstr_seg.hpp)
        gar::CsegFwdParams F{};
        F.meta = s->d_meta;
        F.fac = s->d_fac;
        F.sol = s->d_sol;
        F.csol = s->d_csol;
        F.only = s->d_status + s->batch + 4;
        F.fac_stride = s->fac_doubles;
        F.sol_stride = s->sol_doubles;
        F.horizon = s->horizon;
        F.num_legs = s->num_legs;
        F.leg_begin = s->leg_begin;
        F.nxb = s->nxb;
        F.nc0 = s->nc0;
        hipLaunchKernelGGL(s->cseg.forward, dim3((unsigned)(s->leg_end - s->leg_begin), (unsigned)s->batch), dim3(64), 0, s->stream, F);
      } else {
      gar::GenericParams G = make_params(s, 0.0);
      G.only = s->d_status + s->batch + 4;
      hipLaunchKernelGGL(gar::gar_forward_generic, dim3((unsigned)(s->leg_end - s->leg_begin), (unsigned)s->batch), dim3(GAR_FORWARD_THREADS),
                         (size_t)s->lds.ftotal * sizeof(double), s->stream, G);
      }
    }
    HIP_TRY(hipGetLastError());
    if (s->timing)
      HIP_TRY(hipEventRecord(s->ev[4], s->stream));
    return GAR_HIP_OK;
  }
  if (s->mfma_fwd_kernel) {
    const gar::MfmaFwdParams F = make_mfma_fwd_params(s);
    if (s->timing)
      HIP_TRY(hipEventRecord(s->ev[3], s->stream));
    // FORWARD = lean (per launch): the LDS-DMA roll-out of the pipelined schedule (gar_forward_lean.hpp, bit for bit the
    // same solution) for the whole batch in the plain schedule too -- one workgroup of four problems per CU at a time
    const char *fw = s->lean_fwd_kernel ? gar_option("GAR_HIP_FORWARD") : nullptr;
    if (fw && std::string(fw) == "lean") {
      if (s->lean_fwd_lds_bytes == 0) {
        s->lean_fwd_lds_bytes = std::max(lds_round(s->lean_fwd_used), lds_round(kCuLdsBytes / 2 + 1));
        HIP_TRY(hipFuncSetAttribute((const void *)s->lean_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)s->lean_fwd_lds_bytes));
      }
      hipLaunchKernelGGL(s->lean_fwd_kernel, dim3((unsigned)((s->batch + 3) / 4)), dim3(256), s->lean_fwd_lds_bytes, s->stream, F,
                         s->batch);
    } else
    hipLaunchKernelGGL(s->mfma_fwd_kernel, dim3((unsigned)s->batch), dim3(64), s->mfma_fwd_lds_bytes, s->stream, F);
    HIP_TRY(hipGetLastError());
    if (s->timing)
      HIP_TRY(hipEventRecord(s->ev[4], s->stream));
    return GAR_HIP_OK;
  }
  gar::GenericParams P = make_params(s, 0.0);
  P.theta = theta_dev;
  if (s->dense) {
    hipLaunchKernelGGL(gar::gar_forward_dense, dim3((unsigned)s->batch), dim3(256),
                       (size_t)s->lds.ftotal * sizeof(double), s->stream, P);
    HIP_TRY(hipGetLastError());
    return GAR_HIP_OK;
  }
  const dim3 grid((unsigned)(s->leg_end - s->leg_begin), (unsigned)s->batch);
  if (s->timing)
    HIP_TRY(hipEventRecord(s->ev[3], s->stream));
  if (s->seg_bwd_kernel && s->seg_fwd_kernel && s->num_legs > 1) // segment legs: a wave per (leg, problem)
    hipLaunchKernelGGL(s->seg_fwd_kernel, grid, dim3(64), 0, s->stream, P);
  else
    hipLaunchKernelGGL(gar::gar_forward_generic, grid, dim3(GAR_FORWARD_THREADS),
                       (size_t)s->lds.ftotal * sizeof(double), s->stream, P);
  HIP_TRY(hipGetLastError());
  if (s->timing)
    HIP_TRY(hipEventRecord(s->ev[4], s->stream));
  return GAR_HIP_OK;
}

#include "gar_pipeline.hpp"

int launch_condensed(gar_hip_solver *s) {
  RoctxRange range_("gar::assembleCondensedSystem+symmetricBlockTridiagSolve");
  gar::CondensedParams C{};
  C.ball = s->d_bound_all;
  C.prob = s->d_prob;
  C.scratch = s->d_cscratch;
  C.csol = s->d_csol;
  C.status = s->d_status;
  C.prob_stride = s->prob_doubles;
  C.scratch_stride = s->cscratch_doubles;
  C.G0_off = s->G0_off;
  C.g0_off = s->g0_off;
  C.batch = s->batch;
  C.num_legs = s->num_legs;
  C.legs_per_rank = s->legs_per_rank;
  C.world = s->world;
  C.tuple_doubles = (int)s->tuple_doubles;
  C.nxb = s->nxb;
  C.nc0 = s->nc0;
  C.nx0 = s->nx0;
  C.max_refinement = s->max_refinement;
  C.threshold = s->cond_threshold;
  C.backward_ok = s->cond_backward_ok;
  C.trace = s->d_trace;
  C.gated = 0;
  if (s->cyc_setup_kernel) {
    gar::CyclicParams Y{};
    Y.C = C;
    Y.h = 0;
    const int J = s->num_legs;
    const size_t lds = (size_t)s->cyc_lds_doubles * sizeof(double);
    // J waves for the legs + two for the initial condition's row (S_0 / r_0 and C_0), see gar_cyclic_setup
    hipLaunchKernelGGL(s->cyc_setup_kernel, dim3((unsigned)J + 2, (unsigned)s->batch), dim3(64),
                       lds + (size_t)s->cyc_block_doubles * sizeof(double), s->stream, Y);
    for (int h = 1; h < J; h *= 2) {
      Y.h = h;
      hipLaunchKernelGGL(s->cyc_reduce_kernel,
                         dim3((unsigned)((J + 2 * h - 1) / (2 * h)), (unsigned)s->batch), dim3(192),
                         2 * lds + (64 + (size_t)s->cyc_block_doubles) * sizeof(double), s->stream, Y);
    }
    // back-substitution: the levels holding at most 4 blocks in one workgroup, the wider ones a
    // launch each; then the states and the residual, a wave per leg
    int hmax = 1;
    while (2 * hmax < J)
      hmax *= 2;
    int htop = hmax;
    while (htop > 1 && (J / (htop / 2) + 1) / 2 <= 4)
      htop /= 2;
    Y.h = htop;
    hipLaunchKernelGGL(s->cyc_top_kernel, dim3((unsigned)s->batch), dim3(256), lds, s->stream, Y);
    for (int h = htop / 2; h >= 1; h /= 2) {
      Y.h = h;
      hipLaunchKernelGGL(s->cyc_backlevel_kernel,
                         dim3((unsigned)((J / h + 1) / 2), (unsigned)s->batch), dim3(64), 0,
                         s->stream, Y);
    }
    hipLaunchKernelGGL(s->cyc_recover_kernel, dim3((unsigned)J, (unsigned)s->batch), dim3(64),
                       (size_t)s->cyc_block_doubles * sizeof(double), s->stream, Y); // LDS: G0, padded
    HIP_TRY(hipGetLastError());
    C.gated = 1; // the chain kernel (with refinement) re-solves only what missed the threshold
  }
  if (s->cond_wave_kernel) {
    hipLaunchKernelGGL(s->cond_wave_kernel, dim3((unsigned)s->batch), dim3(64),
                       (size_t)s->cond_wave_lds_doubles * sizeof(double), s->stream, C);
  } else {
    if (!C.gated && s->cond_reduced) {
      // the leg states eliminated leg-parallel, the chain on the J remaining blocks, the states back leg-parallel
      // (gar_generic.hpp: gar_condensed_leg_eliminate); the full chain then runs gated, like behind cyclic reduction
      const dim3 grid((unsigned)s->num_legs, (unsigned)s->batch);
      hipLaunchKernelGGL(gar::gar_condensed_leg_eliminate, grid, dim3(GAR_CONDENSED_THREADS),
                         (size_t)gar::gar_condensed_leg_lds_doubles(s->nxb) * sizeof(double), s->stream, C);
      if (s->cond_cr) {
        // the J remaining blocks by block cyclic reduction: a workgroup per block and level (gar_condensed_cr.hpp)
        const int J = s->num_legs;
        const size_t blk_bytes = (size_t)s->nxb * s->nxb * sizeof(double);
        // (the products of a level read their operands from LDS when four blocks fit a CU)
        const int staged = (size_t)gar::gar_condensed_cr_update_lds_doubles(s->nxb, 1) * sizeof(double) <= 160 * 1024;
        const size_t upd_bytes = (size_t)gar::gar_condensed_cr_update_lds_doubles(s->nxb, staged) * sizeof(double);
        hipLaunchKernelGGL(gar::gar_condensed_cr_assemble, grid, dim3(GAR_CONDENSED_THREADS), blk_bytes, s->stream, C);
        for (int h = 1; h < J; h *= 2) {
          hipLaunchKernelGGL(gar::gar_condensed_cr_eliminate, dim3((unsigned)((J - 1 + h) / (2 * h)), (unsigned)s->batch),
                             dim3(GAR_CONDENSED_THREADS),
                             (size_t)gar::gar_condensed_leg_lds_doubles(s->nxb) * sizeof(double), s->stream, C, h);
          hipLaunchKernelGGL(gar::gar_condensed_cr_update, dim3((unsigned)((J + 2 * h - 1) / (2 * h)), (unsigned)s->batch),
                             dim3(GAR_CONDENSED_THREADS), upd_bytes, s->stream, C, h, staged);
        }
        hipLaunchKernelGGL(gar::gar_condensed_cr_back, dim3((unsigned)s->batch), dim3(GAR_CONDENSED_THREADS),
                           (size_t)gar::gar_condensed_cr_back_lds_doubles(s->nxb, J) * sizeof(double), s->stream, C);
      } else {
        gar::CondensedParams R = C;
        R.reduced = 1;
        hipLaunchKernelGGL(gar::gar_condensed_generic, dim3((unsigned)s->batch), dim3(GAR_CONDENSED_THREADS),
                           (size_t)s->cond_lds_doubles * sizeof(double), s->stream, R);
      }
      hipLaunchKernelGGL(gar::gar_condensed_leg_states, grid, dim3(256), (size_t)(5 * s->nxb + 2) * sizeof(double),
                         s->stream, C);
      C.gated = 1;
    }
    hipLaunchKernelGGL(gar::gar_condensed_generic, dim3((unsigned)s->batch), dim3(GAR_CONDENSED_THREADS),
                       (size_t)s->cond_lds_doubles * sizeof(double), s->stream, C);
  }
  HIP_TRY(hipGetLastError());
  if (s->timing) // leg mode: the "initial stage" slot of the timing API is the condensed solve
    HIP_TRY(hipEventRecord(s->ev[2], s->stream));
  return GAR_HIP_OK;
}

int check_bt(const gar_hip_solver *s, int b, int t) {
  if (!s)
    return fail(GAR_HIP_ERR_ARG, "null solver");
  if (b < 0 || b >= s->batch)
    return fail(GAR_HIP_ERR_ARG, "problem index out of range");
  if (t < 0 || t > s->horizon)
    return fail(GAR_HIP_ERR_ARG, "stage index out of range");
  return GAR_HIP_OK;
}

int d2h(gar_hip_solver *s, double *dst, const double *src, int64_t n) {
  if (!dst || n <= 0)
    return GAR_HIP_OK;
  HIP_TRY(hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s->stream));
  return GAR_HIP_OK;
}

