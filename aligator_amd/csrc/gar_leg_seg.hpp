// gar_leg_seg.hpp -- ParallelRiccatiSolver (gar/parallel-solver.hxx:131-240) for shapes whose plain stage has a
// specialised kernel but no wave-leg family of its own: the wide (56, 24) shape of the Talos walk
// (bench/talos-walk.cpp:102-127 and bench/lqr.cpp:112-134 run it with LQSolverChoice::PARALLEL).
//
// A leg of the reference is backwardImpl over its knots with the leg-end knot treated by terminalSolve under
// configure_knot (Gx = A^T, Gu = B^T, Gth = 0, gamma = f, :136-141) and every other knot carrying the implicit
// parameterisation (:52-60).  Two observations split that into pieces that already exist:
//   (1) terminalSolve on the leg-end knot (riccati-kernel.hxx:151-183) is stageKernelSolve with a ZERO next value
//       function: Qhat = Q, Rhat = R, Shat = S, so K = -R^-1 S^T, kff = -R^-1 r, Vxx = Q + S K, vx = q + S kff --
//       and its parameter outputs are that stage's closed loop: Vxt = Gx + K^T Gu = (A + B K)^T = Aff^T,
//       vt = gamma + Gu^T kff = f + B kff = yff, Kth = -R^-1 Gu = -R^-1 B^T, Vtt = Gu^T Kth = B Kth.
//       So the PLAIN part of a leg is the serial stage kernel started from V' = 0, v' = 0 behind the leg end
//       (gar_backward_pair_leg: gar_wave_pair.hpp's two-wave stage over the stage range of one leg);
//   (2) the parameter part (:278-311 with Gx = Gu = Gth = gamma = 0) needs from the plain part only what its
//       records hold -- Aff, yff, and Vxx' for Rhat = R + B^T V' B -- and is a matrix recursion
//         Ghat_u = B^T Vxt',  Kth = -Rhat^-1 Ghat_u,  Yth = B Kth,
//         Vxt = Aff^T Vxt',   Vtt = Vtt' + Ghat_u^T Kth,  vt = vt' + Vxt'^T yff
//       started from Vxt' = I, Vtt' = 0, vt' = 0 -- which reproduces (1)'s parameter outputs at the leg end.
//       The gar_leg_param_* kernels below run it for ANY dimensions (blocks in LDS, f64 MFMA through wg_gemm, the
//       workgroup LDL^T / Bunch-Kaufman of gar_device.hpp on Rhat -- the reference's own factorisation), after the
//       plain kernel of the same launch sequence, and write the caller-visible records (fth = [Kth; Yth], Vxt,
//       Vtt, vt beside the copied ff, fb, Vxx, vx) and the leg's boundary tuple.
// Like terminalSolve, the leg-end record keeps yff, Aff and Yth at zero (:130-193 never writes them).
#pragma once
#include "gar_device.hpp"
#include "gar_generic.hpp"
#include "gar_wave_pair.hpp"

namespace gar {

// ---- (1) the plain part: the two-wave stage of gar_wave_pair.hpp over ONE leg's stage range -------------------
// grid (local legs, batch) x 128; factor records go to the uniform (nth = 0) scratch layout P.fac / P.fac_rec
template <int NX, int NU>
__global__ void __launch_bounds__(128, (NX + NU > 64) ? 1 : 2) gar_backward_pair_leg(MfmaParams P, int num_legs, int leg_begin) {
  using C = WaveCfg<NX, NU, 0>;
  using M = MfmaCfg<NX, NU, 0>;
  constexpr int PK = C::PK;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int leg = (int)blockIdx.x + leg_begin, b = (int)blockIdx.y;
  double *sm = gar_smem;
  const double *prob = P.prob + (long long)b * P.prob_stride;
  double *fac = P.fac + (long long)b * P.fac_stride;
  const int N = P.horizon;
  int t_beg, t_end;
  gar_get_work(N, leg, num_legs, &t_beg, &t_end);
  const bool last_leg = (leg == num_legs - 1);
  double *V = sm + C::oV, *vn = sm + C::oVn;
  WaveLane<NX, NU, 0> L;
  wave_lane_init<NX, NU>(L, lane);
  int t_first;
  if (last_leg) { // the true terminal knot (terminalSolve, nu = 0, :146-149, :175-178): Vxx = Q, vx = q
    const double *rec = prob + P.in_offN;
    double *out = fac + P.fac_offN;
    for (int e = tid; e < NX * NX; e += 128) {
      const int j = e / NX, i = e - j * NX;
      const double v = (i >= j) ? rec[M::tQ + e] : rec[M::tQ + i * NX + j];
      V[i * PK + j] = v;
      out[M::tVxx + e] = v;
    }
    if (tid < NX) {
      const double v = rec[M::tq + tid];
      vn[tid] = v;
      out[M::tvx + tid] = v;
    }
    t_first = N - 1;
  } else { // behind a leg end: no value function (see (1) above)
    for (int e = tid; e < NX * PK; e += 128)
      V[e] = 0.0;
    if (tid < NX)
      vn[tid] = 0.0;
    t_first = t_end - 1;
  }
  __syncthreads();
  if (t_first < t_beg)
    return; // a leg made of the terminal knot alone
  int failed = 0;
  const double *rec1 = prob + P.in_off0 + P.slot(t_first) * P.in_rec;
#if GAR_PAIR_REFRESH_LANE
  // (the lane offsets re-derived per stage, as in gar_backward_pair: the uneven first half of pair_stage needs the room)
  if (wave == 0) {
    WaveStage<NX, NU> S;
    pair_load<NX, NU, 0>(rec1, L, S, lane);
    for (int t = t_first; t >= t_beg; --t) {
      const int lane_t = lane + fence0(S.Fo[0][0][0]);
      WaveLane<NX, NU, 0> Lt;
      wave_lane_init<NX, NU>(Lt, lane_t);
      pair_stage<NX, NU, 0>(P, sm, prob, fac, t, lane_t, Lt, S, failed);
    }
  } else {
    WaveStage<NX, NU> S;
    pair_load<NX, NU, 1>(rec1, L, S, lane);
    for (int t = t_first; t >= t_beg; --t) {
      const int lane_t = lane + fence0(S.Fo[PairCfg<NX, NU>::SPLIT][0][0]);
      WaveLane<NX, NU, 0> Lt;
      wave_lane_init<NX, NU>(Lt, lane_t);
      pair_stage<NX, NU, 1>(P, sm, prob, fac, t, lane_t, Lt, S, failed);
    }
  }
#else
  if (wave == 0) {
    WaveStage<NX, NU> S;
    pair_load<NX, NU, 0>(rec1, L, S, lane);
    for (int t = t_first; t >= t_beg; --t)
      pair_stage<NX, NU, 0>(P, sm, prob, fac, t, lane, L, S, failed);
  } else {
    WaveStage<NX, NU> S;
    pair_load<NX, NU, 1>(rec1, L, S, lane);
    for (int t = t_first; t >= t_beg; --t)
      pair_stage<NX, NU, 1>(P, sm, prob, fac, t, lane, L, S, failed);
  }
#endif
  if (failed && lane == 0)
    atomicOr(&P.status[b], failed);
}

// ---- (2) the parameter part, any dimensions ------------------------------------------------------------------
struct LegParamParams {
  const gar_stage_meta *meta;  // the solver's layout: caller-visible records (nth = nx on non-final legs)
  const gar_stage_meta *meta2; // the plain kernels' scratch layout (same knots, nth = 0 everywhere)
  const double *prob;
  const double *fac2; // scratch records written by the plain kernel
  double *fac;        // caller-visible records
  double *boundary;   // [problem][local leg][tuple]
  int *status;
  long long prob_stride, fac_stride, fac2_stride, boundary_stride;
  int horizon, num_legs, leg_begin, tuple_doubles, nxb;
  int nxM, nuM; // largest nx, nu (LDS carve)
  int local_legs;
  const int *only; // gar_leg_param_finish behind the constrained segment legs (gar_cstr_seg.hpp): problems with only[b] == 1 (null: all)
};

__host__ __device__ inline int leg_prepare_lds_doubles(int nx, int nu) { // V' | Bm VB Tm | Rh Rc | wk | sub piv ctrl (the front of gar_leg_param_stage's carve)
  auto a2 = [](int x) { return (x + 1) & ~1; };
  return a2(nx * nx) + 3 * a2(nx * nu) + 2 * a2(nu * nu) + a2(GAR_LDL_PANEL * nu) + 2 * a2(nu) + 64;
}

#ifndef GAR_LEG_PARAM_THREADS
#define GAR_LEG_PARAM_THREADS 1024
#endif
// ---- (2) the parameter part, split by what is sequential ---------------------------------------------------------
// Of the recursion above only  Vxt_t = Aff_t^T Vxt_{t+1}  chains the stages of a leg; everything else of stage t is a
// function of Vxt_{t+1} (and of that stage's own operands), and Vtt / vt are running sums of per-stage increments:
//   (2a) gar_leg_param_chain   one workgroup per (leg, problem): the chain of products alone, Vxt of every stage
//                              written to its record (a stage: one 56 x 56 x 56 product, its operand prefetched;
//                              traced, scripts/ctrace_leg_chain.py: 15-16 k cycles per stage, 7 k of them the product on
//                              wave 0 + 2.7 k waiting for the other 15 -- 14 MFMAs per wave, four waves per SIMD: the
//                              masking and address arithmetic of the any-dimension wg_gemm shares the issue port with the
//                              MFMAs; an odd LDS pitch for Vxt', fewer waves, operands fetched a group ahead: no effect);
//   (2b) gar_leg_param_stage   one workgroup per (STAGE, problem), all stages at once: Rhat_t = R + B^T V' B, its
//                              factorisation, T = Rhat^{-1} B^T, then
//                              Ghat_u = B^T Vxt', Kth = -T Vxt', Yth = B Kth, the increments Ghat_u^T Kth and
//                              Vxt'^T yff, and the caller-visible record;
//   (2c) gar_leg_param_finish  one workgroup per (leg, problem): Vtt, vt summed from the leg end down (elementwise,
//                              a thread per element), the boundary tuple.
// The sequential path of a leg of 8 stages drops from 8 x (4 products + record traffic) to 8 products.
__host__ __device__ inline int leg_chain_lds_doubles(int nx) {
  auto a2 = [](int x) { return (x + 1) & ~1; };
  return 3 * a2(nx * nx) + 64;
}
__host__ __device__ inline int leg_stage_lds_doubles(int nx, int nu) { // leg_prepare's carve + Gh Kt | yf
  auto a2 = [](int x) { return (x + 1) & ~1; };
  return leg_prepare_lds_doubles(nx, nu) + 2 * a2(nx * nu) + a2(nx);
}

#ifdef GAR_CTRACE
__device__ long long g_ptrace[16];
#define CHT(id)                                                                                                        \
  {                                                                                                                    \
    const long long now_ = clock64();                                                                                  \
    if (w.tid == 0 && blockIdx.x == 1 && blockIdx.y == 0)                                                              \
      g_ptrace[id] += now_ - tprev;                                                                                    \
    tprev = now_;                                                                                                      \
  }
#else
#define CHT(id)
#endif
__global__ void __launch_bounds__(GAR_LEG_PARAM_THREADS) gar_leg_param_chain(LegParamParams P) {
#ifdef GAR_CTRACE
  long long tprev = clock64();
#endif
  const WG w = wg_self();
  double *sm = gar_smem;
  const int leg = (int)blockIdx.x + P.leg_begin, b = (int)blockIdx.y;
  if (leg == P.num_legs - 1)
    return;
  int t_beg, t_end;
  gar_get_work(P.horizon, leg, P.num_legs, &t_beg, &t_end);
  const double *fac2 = P.fac2 + (long long)b * P.fac2_stride;
  double *fac = P.fac + (long long)b * P.fac_stride;
  auto a2 = [](int x) { return (x + 1) & ~1; };
  const int nxM = P.nxM;
  double *Xt = sm, *Xn = Xt + a2(nxM * nxM), *Af = Xn + a2(nxM * nxM);
  const int nth = P.meta[t_end - 1].nx2;
  constexpr int PRE = 4; // the next stage's Aff in flight under the product (registers), up to PRE x nthr doubles
  double pre[PRE];
  // the stage descriptors run TWO stages ahead of the product (a descriptor read, then the operand reads it addresses,
  // are two dependent round trips to memory: neither may sit on the chain)
  struct Desc {
    int nx, nu, nx2;
    long long aff, vxt; // offsets of Aff (scratch records) and of Vxt (caller-visible records)
  };
  auto desc = [&](int t) {
    const gar_stage_meta m = P.meta[t < t_beg ? t_beg : t];
    Desc d;
    d.nx = m.nx, d.nu = m.nu, d.nx2 = m.nx2;
    d.aff = P.meta2[t < t_beg ? t_beg : t].fac_off + gar_factor_layout(m.nx, m.nu, 0, m.nx2, 0).fb + m.nu * m.nx;
    d.vxt = m.fac_off + gar_factor_layout(m.nx, m.nu, 0, m.nx2, nth).Vxt;
    return d;
  };
  Desc cur = desc(t_end - 1), nxt = desc(t_end - 2);
  {
    const double *a = fac2 + cur.aff;
    wg_move8(w, cur.nx2 * cur.nx, [&](int e) { return a[e]; }, [&](int e, double v) { Af[e] = v; });
    for (int e = w.tid; e < nth * nth; e += w.nthr)
      Xt[e] = ((e / nth) == (e % nth)) ? 1.0 : 0.0;
  }
  // (the barriers order LDS only: the record stores of one stage and the operand loads of the next stay in flight)
  for (int t = t_end - 1; t >= t_beg; --t) {
    const int nx = cur.nx, nx2 = cur.nx2;
    CHT(0)
    wg_lds_bar(); // Af, Xt complete
    CHT(1)
    const Desc nnv = desc(t - 2);
    const int ncnt = nxt.nx2 * nxt.nx;
    const double *na = fac2 + nxt.aff;
    if (t > t_beg) {
#pragma unroll
      for (int q = 0; q < PRE; ++q) {
        const int e = w.tid + q * w.nthr;
        pre[q] = na[e < ncnt ? e : ncnt - 1];
      }
    }
    CHT(2)
    // Vxt = Aff^T Vxt'  (:305-306; at the leg end Aff^T I = A^T + K^T B^T, :186)
    wg_gemm(w, nx, nth, nx2, rowmajor(Af, nx).T(), colmajor(Xt, nx2), MatV{nullptr, 0, 0}, colmajor(Xn, nx), 1.0);
    CHT(3)
    wg_lds_bar();
    CHT(4)
    // first everything that WAITS for loads -- the next operand into LDS, the descriptor two stages ahead into scalar
    // registers -- then the record stores: gfx9 counts loads and stores on one in-order counter, so a wait for a load
    // issued behind a store is a wait for that store's acknowledgement too
    if (t > t_beg) {
#pragma unroll
      for (int q = 0; q < PRE; ++q) {
        const int e = w.tid + q * w.nthr;
        if (e < ncnt)
          Af[e] = pre[q];
      }
      for (int e = w.tid + PRE * w.nthr; e < ncnt; e += w.nthr)
        Af[e] = na[e];
    }
    CHT(5)
    Desc nn;
    nn.nx = __builtin_amdgcn_readfirstlane(nnv.nx), nn.nu = __builtin_amdgcn_readfirstlane(nnv.nu);
    nn.nx2 = __builtin_amdgcn_readfirstlane(nnv.nx2);
    nn.aff = ((long long)__builtin_amdgcn_readfirstlane((int)(nnv.aff >> 32)) << 32) |
             (unsigned)__builtin_amdgcn_readfirstlane((int)nnv.aff);
    nn.vxt = ((long long)__builtin_amdgcn_readfirstlane((int)(nnv.vxt >> 32)) << 32) |
             (unsigned)__builtin_amdgcn_readfirstlane((int)nnv.vxt);
    CHT(6)
    double *dst = fac + cur.vxt;
    wg_move8(w, nx * nth, [&](int e) { return Xn[e]; }, [&](int e, double v) { dst[e] = v; });
    CHT(7)
    double *tmp = Xt;
    Xt = Xn;
    Xn = tmp;
    cur = nxt;
    nxt = nn;
  }
}

#ifndef GAR_LEG_STAGE_THREADS
#define GAR_LEG_STAGE_THREADS 512
#endif
// grid (N + 1, batch) x GAR_LEG_STAGE_THREADS
__global__ void __launch_bounds__(GAR_LEG_STAGE_THREADS) gar_leg_param_stage(LegParamParams P) {
  const WG w = wg_self();
  double *sm = gar_smem;
  const int t = (int)blockIdx.x, b = (int)blockIdx.y;
  int leg = P.leg_begin, t_beg = 0, t_end = 0;
  for (; leg < P.leg_begin + P.local_legs; ++leg) { // the leg of stage t (among this rank's)
    gar_get_work(P.horizon, leg, P.num_legs, &t_beg, &t_end);
    if (t >= t_beg && t < t_end)
      break;
  }
  if (leg >= P.leg_begin + P.local_legs)
    return;
  const double *prob = P.prob + (long long)b * P.prob_stride;
  const double *fac2 = P.fac2 + (long long)b * P.fac2_stride;
  double *fac = P.fac + (long long)b * P.fac_stride;
  const gar_stage_meta m = P.meta[t];
  const int nx = m.nx, nu = m.nu, nx2 = m.nx2, nr = nu + nx2;
  const double *src = fac2 + P.meta2[t].fac_off;
  double *dst = fac + m.fac_off;
  if (leg == P.num_legs - 1) { // unparameterised records: the scratch layout's, at the caller-visible offsets
    const int n = (int)gar_factor_doubles(m.nx, m.nu, m.nc, m.nx2, 0);
    wg_move8(w, n, [&](int e) { return src[e]; }, [&](int e, double v) { dst[e] = v; });
    return;
  }
  const bool leg_end = (t == t_end - 1);
  const int nth = P.meta[t_end - 1].nx2; // the parameter: the next leg's first costate
  auto a2 = [](int x) { return (x + 1) & ~1; };
  const int nxM = P.nxM, nuM = P.nuM;
  double *p = sm;
  auto take = [&](int n) { double *o = p; p += a2(n); return o; };
  double *Vn = take(nxM * nxM), *Bm = take(nxM * nuM), *VB = take(nxM * nuM), *Tm = take(nxM * nuM);
  double *Rh = take(nuM * nuM), *Rc = take(nuM * nuM), *wk = take(GAR_LDL_PANEL * nuM), *sub = take(nuM);
  int *piv = (int *)take(nuM), *ctrl = (int *)take(16);
  double *Gh = take(nuM * nxM), *Kt = take(nuM * nxM), *yf = take(nxM);
  const gar_knot_offsets ko = gar_knot_layout(m.nx, nu, 0, nx2, 0);
  const gar_factor_offsets f2 = gar_factor_layout(nx, nu, 0, nx2, 0), fo = gar_factor_layout(nx, nu, 0, nx2, nth);
  const double *knot = prob + m.in_off;
  wg_move8(w, nx2 * nu, [&](int e) { return knot[ko.B + e]; }, [&](int e, double v) {
    Bm[e] = v;
    Tm[e] = v; // B^T (nu x nx2, row-major) = B column-major, as it is
  });
  wg_move8(w, nu * nu, [&](int e) { return knot[ko.R + e]; }, [&](int e, double v) { Rh[e] = v; });
  for (int e = w.tid; e < nx2; e += w.nthr)
    yf[e] = src[f2.ff + nu + e];
  if (!leg_end) { // V' symmetrised from its lower triangle as the consuming stage does (:216)
    const gar_stage_meta mn = P.meta[t + 1];
    const double *Vg = fac2 + P.meta2[t + 1].fac_off + gar_factor_layout(mn.nx, mn.nu, 0, mn.nx2, 0).Vxx;
    wg_move8(w, nx2 * nx2, [&](int e) {
      const int j = e / nx2, i = e - j * nx2;
      return (i >= j) ? Vg[e] : Vg[i * nx2 + j]; }, [&](int e, double v) { Vn[e] = v; });
  }
  // the plain part of the caller-visible record, ff | fb | Vxx | vx, while the operands arrive (leg end: yff, Aff zero)
  for (int e = w.tid; e < nr; e += w.nthr)
    dst[fo.ff + e] = (leg_end && e >= nu) ? 0.0 : src[f2.ff + e];
  wg_move8(w, nr * nx, [&](int e) { return src[f2.fb + e]; },
           [&](int e, double v) { dst[fo.fb + e] = (leg_end && e >= nu * nx) ? 0.0 : v; });
  wg_move8(w, nx * nx + nx, [&](int e) { return src[f2.Vxx + e]; }, [&](int e, double v) { dst[fo.Vxx + e] = v; }); // Vxx | vx
  __syncthreads();
  const MatV B = colmajor(Bm, nx2);
  if (!leg_end) { // Rhat = R + B^T (V' B)  (:221, :225)
    wg_gemm(w, nx2, nu, nx2, colmajor(Vn, nx2), B, MatV{nullptr, 0, 0}, colmajor(VB, nx2), 1.0);
    __syncthreads();
    wg_gemm(w, nu, nu, nx2, B.T(), colmajor(VB, nx2), colmajor(Rh, nu), colmajor(Rh, nu), 1.0);
    __syncthreads();
  }
  // Vxt' of this stage (the chain kernel's record of stage t + 1; the identity behind the leg end) takes V's place
  if (leg_end) {
    for (int e = w.tid; e < nx2 * nth; e += w.nthr)
      Vn[e] = ((e / nx2) == (e % nx2)) ? 1.0 : 0.0;
  } else {
    const gar_stage_meta mn = P.meta[t + 1];
    const double *Xg = fac + mn.fac_off + gar_factor_layout(mn.nx, mn.nu, 0, mn.nx2, nth).Vxt;
    wg_move8(w, nx2 * nth, [&](int e) { return Xg[e]; }, [&](int e, double v) { Vn[e] = v; });
  }
  // Rhat > 0 on a well-posed stage: blocked elimination without pivoting; otherwise Bunch-Kaufman as in the reference
  int failed = 0, indefinite = 1;
  if (nu >= 8 && nu <= 64) {
    for (int e = w.tid; e < nu * nu; e += w.nthr)
      Rc[e] = Rh[e];
    __syncthreads();
    indefinite = wg_ldl_definite_factor(w, nu, Rh, nu, sub, piv, wk, ctrl);
    if (indefinite) {
      for (int e = w.tid; e < nu * nu; e += w.nthr)
        Rh[e] = Rc[e];
      __syncthreads();
    }
  }
  if (indefinite)
    failed |= wg_bk_factor(w, nu, Rh, nu, sub, piv, ctrl);
  __syncthreads();
  wg_bk_solve(w, nu, Rh, nu, sub, piv, Tm, nx2, 1, nx2); // T = Rhat^{-1} B^T, rows of nx2
  __syncthreads();
  const MatV X = colmajor(Vn, nx2), G = rowmajor(Gh, nth), K = rowmajor(Kt, nth);
  // Ghat_u = B^T Vxt' (:286-287),  Kth = -Rhat^{-1} Ghat_u = -T Vxt' (:288-292)
  wg_gemm(w, nu, nth, nx2, B.T(), X, MatV{nullptr, 0, 0}, G, 1.0);
  wg_gemm(w, nu, nth, nx2, rowmajor(Tm, nx2), X, MatV{nullptr, 0, 0}, K, -1.0);
  __syncthreads();
  // the increments of the running sums (gar_leg_param_finish adds them up): Ghat_u^T Kth (:308-310), Vxt'^T yff (:301);
  // Yth = B Kth (:295) -- straight into the record
  wg_gemm(w, nth, nth, nu, G.T(), K, MatV{nullptr, 0, 0}, colmajor(dst + fo.Vtt, nth), 1.0);
  wg_gemm(w, nth, 1, nx2, X.T(), colmajor(yf, nx2), MatV{nullptr, 0, 0}, colmajor(dst + fo.vt, nth), 1.0);
  if (!leg_end)
    wg_gemm(w, nx2, nth, nu, B, K, MatV{nullptr, 0, 0}, rowmajor(dst + fo.fth + nu * nth, nth), 1.0);
  // fth = [Kth; Yth]  (Vxt: the chain kernel's; Vtt, vt: the increments, summed by gar_leg_param_finish)
  for (int e = w.tid; e < nu * nth; e += w.nthr)
    dst[fo.fth + e] = Kt[e];
  if (leg_end)
    for (int e = w.tid; e < nx2 * nth; e += w.nthr)
      dst[fo.fth + nu * nth + e] = 0.0;
  if (failed && w.tid == 0)
    atomicOr(&P.status[b], failed);
}

// grid (local legs, batch) x 1024
__global__ void __launch_bounds__(1024) gar_leg_param_finish(LegParamParams P) {
  const WG w = wg_self();
  const int leg = (int)blockIdx.x + P.leg_begin, b = (int)blockIdx.y;
  if (P.only != nullptr && P.only[b] != 1)
    return;
  int t_beg, t_end;
  gar_get_work(P.horizon, leg, P.num_legs, &t_beg, &t_end);
  const bool last_leg = (leg == P.num_legs - 1);
  double *fac = P.fac + (long long)b * P.fac_stride;
  const gar_stage_meta m0 = P.meta[t_beg];
  const int nxb = P.nxb, bs = nxb * nxb, nx = m0.nx, nth = last_leg ? 0 : P.meta[t_end - 1].nx2;
  double *tup = P.boundary + (long long)b * P.boundary_stride + (long long)blockIdx.x * P.tuple_doubles;
  const gar_factor_offsets fo0 = gar_factor_layout(m0.nx, m0.nu, m0.nc, m0.nx2, nth);
  const double *rec = fac + m0.fac_off;
  // the boundary tuple of this leg: (Vxx | Vxt | Vtt | vx | vt) of its first stage, blocks of nxb;
  // every element of the tuple is written exactly once
  wg_move8(w, 2 * bs, [&](int e) {
    const bool second = e >= bs;
    const int ee = second ? e - bs : e, j = ee / nxb, i = ee - j * nxb;
    const bool real = i < nx && j < (second ? nth : nx);
    return real ? rec[(second ? fo0.Vxt : fo0.Vxx) + j * nx + i] : 0.0; }, [&](int e, double v) { tup[e] = v; });
  for (int e = w.tid; e < nxb; e += w.nthr)
    tup[3 * bs + e] = e < nx ? rec[fo0.vx + e] : 0.0;
  for (int e = w.tid + 3 * bs + 2 * nxb; e < P.tuple_doubles; e += w.nthr)
    tup[e] = 0.0;
  // Vtt_t = Vtt_{t+1} + Ghat_u^T Kth, vt_t = vt_{t+1} + Vxt'^T yff from the leg end down: elementwise, in place, the
  // increments of up to CH stages loaded before the first sum is stored
  constexpr int CH = 8;
  for (int e = w.tid; e < bs + nxb; e += w.nthr) {
    const bool mat = e < bs;
    const int j = mat ? e / nxb : 0, i = mat ? e - j * nxb : e - bs;
    const bool real = i < nth && (!mat || j < nth);
    double acc = 0.0;
    if (real)
      for (int t1 = t_end - 1; t1 >= t_beg; t1 -= CH) {
        double v[CH];
        double *q[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int t = (t1 - c >= t_beg) ? t1 - c : t_beg;
          const gar_stage_meta m = P.meta[t];
          const gar_factor_offsets fo = gar_factor_layout(m.nx, m.nu, m.nc, m.nx2, nth);
          q[c] = fac + m.fac_off + (mat ? fo.Vtt + j * nth + i : fo.vt + i);
          v[c] = *q[c];
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
          if (t1 - c >= t_beg) {
            acc = acc + v[c];
            *q[c] = acc;
          }
      }
    tup[(mat ? 2 * bs : 3 * bs + nxb - bs) + e] = acc;
  }
}

// ---- (3) the roll-out of the segment legs: one wave per (leg, problem), lane = row ------------------------------
// forwardImpl over one leg (riccati-kernel.hxx:314-377 under parallel-solver.hxx:209-243) in gar_forward_wide's
// scheme -- lane r < NU owns row r of K and Kth, lane r < NX row r of Aff, Yth, Vxx' and Vxt' (fb, fth ROW-major: a
// row is 448 contiguous bytes; Vxx' symmetric: its row is its column; Vxt' column-major: one coalesced load per
// column), the state and the parameter broadcast from the lanes that hold them:
//   u = kff + K x + Kth th,   x' = yff + Aff x + Yth th,   lbd' = vx' + Vxx' x' + Vxt' th
// with x, lbd at the leg's first stage and th (the next leg's first costate) from the condensed solution.  Replaces
// the any-dimension roll-out on this family (74 -> 54 us per sweep at N = 256, 32 legs: a lone wave draws its 230 KB
// per stage at about 35 GB/s).
// one stage of the roll-out: (MORE) x_{t+1}, lbd_{t+1}, then u_t.  A lane's six rows are 336 doubles -- more than
// the register file holds -- so the order is pinned: the rows of the state chain (Aff, Yth) and of the costate
// (Vxx', Vxt') are requested first, the rows of the controls (K, Kth) into the registers the chain releases
template <int NX, int NU, bool LAST, bool MORE>
__device__ __forceinline__ void wide_leg_stage(const GenericParams &P, const double *fac, double *sol, int t, int lane,
                                               double &xs, double th) {
  constexpr int NTH = LAST ? 0 : NX;
  const int iA = lane < NX ? lane : NX - 1, iK = lane < NU ? lane : NU - 1;
  const gar_stage_meta m = P.meta[t];
  const int nu = m.nu; // (0 at the terminal knot)
  const gar_factor_offsets fo = gar_factor_layout(NX, nu, 0, m.nx2, NTH);
  const double *rec = fac + m.fac_off;
  const gar_stage_meta mn = P.meta[MORE ? t + 1 : t];
  const gar_factor_offsets fn = gar_factor_layout(NX, mn.nu, 0, mn.nx2, NTH);
  const double *recn = fac + mn.fac_off;
  const double x_in = xs;
  const bool has_u = nu > 0;
  double2_t kro[NX / 2], kth[LAST ? 1 : NX / 2];
  auto load_k = [&] {
    const double *kp = rec + fo.fb + (long long)iK * NX, *ktp = rec + fo.fth + (long long)iK * NX;
#pragma unroll
    for (int q = 0; q < NX / 2; ++q) {
      kro[q] = *reinterpret_cast<const double2_t *>(kp + 2 * q);
      if (!LAST)
        kth[q] = *reinterpret_cast<const double2_t *>(ktp + 2 * q);
    }
  };
  if (MORE) {
    double2_t aff[NX / 2], yth[LAST ? 1 : NX / 2], vrow[NX / 2];
    double vxt[LAST ? 1 : NX];
    const double *ap = rec + fo.fb + (long long)(nu + iA) * NX, *ytp = rec + fo.fth + (long long)(nu + iA) * NX;
#pragma unroll
    for (int q = 0; q < NX / 2; ++q) {
      aff[q] = *reinterpret_cast<const double2_t *>(ap + 2 * q);
      if (!LAST)
        yth[q] = *reinterpret_cast<const double2_t *>(ytp + 2 * q);
    }
#pragma unroll
    for (int q = 0; q < NX / 2; ++q)
      vrow[q] = *reinterpret_cast<const double2_t *>(recn + fn.Vxx + (long long)iA * NX + 2 * q);
    if (!LAST) {
#pragma unroll
      for (int j = 0; j < NX; ++j)
        vxt[j] = recn[fn.Vxt + (long long)j * NX + iA];
    }
    const double yff = rec[fo.ff + nu + iA], vxn = recn[fn.vx + iA];
    __builtin_amdgcn_sched_barrier(0);
    double x0 = yff, x1 = 0.0;
#pragma unroll
    for (int q = 0; q < NX / 2; ++q) {
      x0 = __builtin_fma(aff[q].x, lane_bcast(x_in, 2 * q), x0);
      x1 = __builtin_fma(aff[q].y, lane_bcast(x_in, 2 * q + 1), x1);
      if (!LAST) {
        x0 = __builtin_fma(yth[q].x, lane_bcast(th, 2 * q), x0);
        x1 = __builtin_fma(yth[q].y, lane_bcast(th, 2 * q + 1), x1);
      }
    }
    const double xn = x0 + x1;
    if (lane < NX)
      sol[mn.x_off + lane] = xn;
    double l0 = vxn, l1 = 0.0; // lbd' = vx' + Vxx' x' + Vxt' th  (:369-374)
#pragma unroll
    for (int q = 0; q < NX / 2; ++q) {
      l0 = __builtin_fma(vrow[q].x, lane_bcast(xn, 2 * q), l0);
      l1 = __builtin_fma(vrow[q].y, lane_bcast(xn, 2 * q + 1), l1);
      if (!LAST) {
        l0 = __builtin_fma(vxt[2 * q], lane_bcast(th, 2 * q), l0);
        l1 = __builtin_fma(vxt[2 * q + 1], lane_bcast(th, 2 * q + 1), l1);
      }
    }
    if (lane < NX)
      sol[mn.l_off + lane] = l0 + l1;
    xs = xn;
    __builtin_amdgcn_sched_barrier(0);
  }
  if (has_u) { // u = kff + K x + Kth th: off the chain, its rows into the registers the chain released (requesting
    load_k();  //  them under the costate's products overflows the register file: 348 B of scratch per lane)
    double u0 = rec[fo.ff + iK], u1 = 0.0;
#pragma unroll
    for (int q = 0; q < NX / 2; ++q) {
      u0 = __builtin_fma(kro[q].x, lane_bcast(x_in, 2 * q), u0);
      u1 = __builtin_fma(kro[q].y, lane_bcast(x_in, 2 * q + 1), u1);
      if (!LAST) {
        u0 = __builtin_fma(kth[q].x, lane_bcast(th, 2 * q), u0);
        u1 = __builtin_fma(kth[q].y, lane_bcast(th, 2 * q + 1), u1);
      }
    }
    if (lane < NU)
      sol[m.u_off + lane] = u0 + u1;
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NX, int NU>
__global__ void __launch_bounds__(64) gar_forward_wide_leg(GenericParams P) {
  static_assert(NX <= 64 && NX % 2 == 0 && NU <= 64, "state and controls live in the first lanes");
  const int lane = (int)threadIdx.x;
  const int leg = (int)blockIdx.x + P.leg_begin, b = (int)blockIdx.y;
  if (P.only != nullptr && P.only[b] == 0)
    return;
  const double *fac = P.fac + (long long)b * P.fac_stride;
  double *sol = P.sol + (long long)b * P.sol_stride;
  int t_beg, t_end;
  gar_get_work(P.horizon, leg, P.num_legs, &t_beg, &t_end);
  const bool last = (leg == P.num_legs - 1);
  const int nxb = P.nxb;
  const double *cs = P.csol + (long long)b * (2 * P.num_legs) * nxb;
  const gar_stage_meta m0 = P.meta[t_beg];
  const int iA = lane < NX ? lane : NX - 1;
  for (int e = lane; e < (leg == 0 ? P.nc0 : NX); e += 64) // scatter of the condensed solution (:215-220)
    sol[m0.l_off + e] = cs[(2 * leg) * nxb + e];
  double xs = cs[(2 * leg + 1) * nxb + iA];
  if (lane < NX)
    sol[m0.x_off + lane] = xs;
  if (last) {
    for (int t = t_beg; t + 1 < t_end; ++t)
      wide_leg_stage<NX, NU, true, true>(P, fac, sol, t, lane, xs, 0.0);
    wide_leg_stage<NX, NU, true, false>(P, fac, sol, t_end - 1, lane, xs, 0.0);
  } else {
    const double th = cs[(2 * (leg + 1)) * nxb + iA]; // theta = lbdas[end] (:234-236)
    for (int t = t_beg; t + 1 < t_end; ++t)
      wide_leg_stage<NX, NU, false, true>(P, fac, sol, t, lane, xs, th);
    wide_leg_stage<NX, NU, false, false>(P, fac, sol, t_end - 1, lane, xs, th);
  }
}

} // namespace gar
