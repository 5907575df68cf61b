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
//       gar_leg_param_generic runs it for ANY dimensions (one workgroup per (leg, problem), blocks in LDS, f64
//       MFMA through wg_gemm, the workgroup Bunch-Kaufman of gar_device.hpp on Rhat -- the reference's own
//       factorisation), after the plain kernel of the same launch sequence, and writes the caller-visible
//       records (fth = [Kth; Yth], Vxt, Vtt, vt beside the copied ff, fb, Vxx, vx) and the leg's boundary tuple.
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
  if (wave == 0) {
    WaveStage<NX, NU> S;
    pair_load<NX, NU, 0>(rec1, L, S);
    for (int t = t_first; t >= t_beg; --t)
      pair_stage<NX, NU, 0>(P, sm, prob, fac, t, lane, L, S, failed);
  } else {
    WaveStage<NX, NU> S;
    pair_load<NX, NU, 1>(rec1, L, S);
    for (int t = t_first; t >= t_beg; --t)
      pair_stage<NX, NU, 1>(P, sm, prob, fac, t, lane, L, S, failed);
  }
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
  double *tgain;          // [problem][stage][nuM x nxM]: T_t = Rhat_t^{-1} B_t^T (gar_leg_param_prepare)
  long long tgain_stride; // doubles per problem
  int local_legs;
};

__host__ __device__ inline int leg_param_lds_doubles(int nx, int nu) { // the recursion: Xa Xb Tt Af | Bm Tm Gh Kt | vt vtn yf
  auto a2 = [](int x) { return (x + 1) & ~1; };
  return 4 * a2(nx * nx) + 4 * a2(nx * nu) + 3 * a2(nx) + 64;
}
__host__ __device__ inline int leg_prepare_lds_doubles(int nx, int nu) { // V' | Bm VB Tm | Rh | wk | sub piv ctrl
  auto a2 = [](int x) { return (x + 1) & ~1; };
  return a2(nx * nx) + 3 * a2(nx * nu) + 2 * a2(nu * nu) + a2(GAR_LDL_PANEL * nu) + 2 * a2(nu) + 64;
}

// What the parameter recursion needs of Rhat_t = R_t + B_t^T V'_{t+1} B_t is the operator T_t = Rhat_t^{-1} B_t^T
// (Kth = -T_t Vxt', a product) -- and T_t does not depend on the recursion's state: every stage of every non-final
// leg at once, one workgroup each (grid (stages, batch) x 256).  The factorisation and the substitution leave the
// sequential path this way.  V'_{t+1} is the plain kernel's Vxx of stage t + 1 (zero behind a leg end).
__global__ void __launch_bounds__(256) gar_leg_param_prepare(LegParamParams P) {
  const WG w = wg_self();
  double *sm = gar_smem;
  const int t = (int)blockIdx.x, b = (int)blockIdx.y;
  int leg = P.leg_begin, t_beg = 0, t_end = 0;
  for (; leg < P.leg_begin + P.local_legs; ++leg) { // the leg of stage t (among this rank's)
    gar_get_work(P.horizon, leg, P.num_legs, &t_beg, &t_end);
    if (t >= t_beg && t < t_end)
      break;
  }
  if (leg >= P.leg_begin + P.local_legs || leg == P.num_legs - 1)
    return;
  const bool leg_end = (t == t_end - 1);
  const double *prob = P.prob + (long long)b * P.prob_stride;
  const double *fac2 = P.fac2 + (long long)b * P.fac2_stride;
  const gar_stage_meta m = P.meta[t];
  const int nx = m.nx, nu = m.nu, nx2 = m.nx2;
  (void)nx;
  auto a2 = [](int x) { return (x + 1) & ~1; };
  const int nxM = P.nxM, nuM = P.nuM;
  double *p = sm;
  auto take = [&](int n) { double *o = p; p += a2(n); return o; };
  double *Vn = take(nxM * nxM), *Bm = take(nxM * nuM), *VB = take(nxM * nuM), *Tm = take(nxM * nuM);
  double *Rh = take(nuM * nuM), *Rc = take(nuM * nuM), *wk = take(GAR_LDL_PANEL * nuM), *sub = take(nuM);
  int *piv = (int *)take(nuM), *ctrl = (int *)take(16);
  const gar_knot_offsets ko = gar_knot_layout(m.nx, nu, 0, nx2, 0);
  const double *knot = prob + m.in_off;
  for (int e = w.tid; e < nx2 * nu; e += w.nthr) {
    const double v = knot[ko.B + e];
    Bm[e] = v;
    Tm[(e / nx2) * nx2 + (e % nx2)] = v; // B^T (nu x nx2, row-major) = B column-major, as it is
  }
  for (int e = w.tid; e < nu * nu; e += w.nthr)
    Rh[e] = knot[ko.R + e];
  if (!leg_end) { // V' symmetrised from its lower triangle as the consuming stage does (:216)
    const gar_stage_meta mn = P.meta[t + 1];
    const double *Vg = fac2 + P.meta2[t + 1].fac_off + gar_factor_layout(mn.nx, mn.nu, 0, mn.nx2, 0).Vxx;
    for (int e = w.tid; e < nx2 * nx2; e += w.nthr) {
      const int j = e / nx2, i = e - j * nx2;
      Vn[e] = (i >= j) ? Vg[e] : Vg[i * nx2 + j];
    }
  }
  __syncthreads();
  const MatV B = colmajor(Bm, nx2);
  if (!leg_end) { // Rhat = R + B^T (V' B)  (:221, :225)
    wg_gemm(w, nx2, nu, nx2, colmajor(Vn, nx2), B, MatV{nullptr, 0, 0}, colmajor(VB, nx2), 1.0);
    __syncthreads();
    wg_gemm(w, nu, nu, nx2, B.T(), colmajor(VB, nx2), colmajor(Rh, nu), colmajor(Rh, nu), 1.0);
    __syncthreads();
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
  double *Tg = P.tgain + (long long)b * P.tgain_stride + (long long)t * nuM * nxM;
  for (int e = w.tid; e < nu * nx2; e += w.nthr)
    Tg[e] = Tm[e];
  if (failed && w.tid == 0)
    atomicOr(&P.status[b], failed);
}

// (debug build -DGAR_CTRACE: cycles per phase of workgroup (0, 0), read with gar_hip_debug_ptrace)
#ifdef GAR_CTRACE
__device__ long long g_ptrace[16];
#define PT(id)                                                                                                         \
  {                                                                                                                    \
    __syncthreads();                                                                                                   \
    const long long now_ = clock64();                                                                                  \
    if (w.tid == 0 && blockIdx.x == 0 && blockIdx.y == 0)                                                              \
      g_ptrace[id] += now_ - tprev;                                                                                    \
    tprev = now_;                                                                                                      \
  }
#else
#define PT(id)
#endif
// grid (local legs, batch) x GAR_LEG_PARAM_THREADS
// (GAR_LEG_PARAM_THREADS threads: the copies between HBM and LDS and the tiles of the products spread over 8 waves;
// the panel factorisation and the in-block substitutions stay one wave's work)
#ifndef GAR_LEG_PARAM_THREADS
#define GAR_LEG_PARAM_THREADS 1024
#endif
__global__ void __launch_bounds__(GAR_LEG_PARAM_THREADS) gar_leg_param_generic(LegParamParams P) {
#ifdef GAR_CTRACE
  long long tprev = clock64();
#endif
  const WG w = wg_self();
  double *sm = gar_smem;
  const int leg = (int)blockIdx.x + P.leg_begin, b = (int)blockIdx.y;
  int t_beg, t_end;
  gar_get_work(P.horizon, leg, P.num_legs, &t_beg, &t_end);
  const bool last_leg = (leg == P.num_legs - 1);
  const double *prob = P.prob + (long long)b * P.prob_stride;
  const double *fac2 = P.fac2 + (long long)b * P.fac2_stride;
  double *fac = P.fac + (long long)b * P.fac_stride;
  auto a2 = [](int x) { return (x + 1) & ~1; };
  const int nxM = P.nxM, nuM = P.nuM;
  double *p = sm;
  auto take = [&](int n) { double *o = p; p += a2(n); return o; };
  double *Xa = take(nxM * nxM), *Xb = take(nxM * nxM), *Tt = take(nxM * nxM), *Af = take(nxM * nxM);
  double *Bm = take(nxM * nuM), *Tm = take(nxM * nuM), *Gh = take(nuM * nxM), *Kt = take(nuM * nxM);
  double *vt = take(nxM), *vtn = take(nxM), *yf = take(nxM);
  double *Xt = Xa, *Xn = Xb; // Vxt' (current) and the buffer the new Vxt goes to
  const int failed = 0;      // (the factorisations are gar_leg_param_prepare's)

  if (last_leg) { // unparameterised records: the scratch layout's, at the caller-visible offsets
    for (int t = t_beg; t < t_end; ++t) {
      const gar_stage_meta m = P.meta[t];
      const int n = (int)gar_factor_doubles(m.nx, m.nu, m.nc, m.nx2, 0);
      const double *src = fac2 + P.meta2[t].fac_off;
      double *dst = fac + m.fac_off;
      for (int e = w.tid; e < n; e += w.nthr)
        dst[e] = src[e];
    }
  } else {
    const int nth = P.meta[t_end - 1].nx2; // the parameter: the next leg's first costate
    for (int e = w.tid; e < nth * nth; e += w.nthr) {
      Xt[e] = ((e / nth) == (e % nth)) ? 1.0 : 0.0;
      Tt[e] = 0.0;
    }
    for (int e = w.tid; e < nth; e += w.nthr)
      vt[e] = 0.0;
    __syncthreads();
    for (int t = t_end - 1; t >= t_beg; --t) {
      const gar_stage_meta m = P.meta[t];
      const int nx = m.nx, nu = m.nu, nx2 = m.nx2, nr = nu + nx2;
      const bool leg_end = (t == t_end - 1);
      const gar_knot_offsets ko = gar_knot_layout(nx, nu, 0, nx2, 0);
      const gar_factor_offsets f2 = gar_factor_layout(nx, nu, 0, nx2, 0), fo = gar_factor_layout(nx, nu, 0, nx2, nth);
      const double *knot = prob + m.in_off;
      const double *src = fac2 + P.meta2[t].fac_off;
      double *dst = fac + m.fac_off;
      PT(0)
      // operands: B (nx2 x nu, column-major), Aff (rows nu.. of the row-major fb), yff, T = Rhat^{-1} B^T (prepared)
      const double *Tg = P.tgain + (long long)b * P.tgain_stride + (long long)t * nuM * nxM;
      for (int e = w.tid; e < nx2 * nu; e += w.nthr) {
        Bm[e] = knot[ko.B + e];
        Tm[e] = Tg[e];
      }
      for (int e = w.tid; e < nx2 * nx; e += w.nthr)
        Af[e] = src[f2.fb + nu * nx + e];
      for (int e = w.tid; e < nx2; e += w.nthr)
        yf[e] = src[f2.ff + nu + e];
      __syncthreads();
      PT(1)
      const MatV B = colmajor(Bm, nx2), X = colmajor(Xt, nx2), G = rowmajor(Gh, nth), K = rowmajor(Kt, nth);
      // everything that is a function of Vxt' alone, in one phase:
      //   Ghat_u = B^T Vxt' (:286-287),  Kth = -Rhat^{-1} Ghat_u = -T Vxt' (:288-292),  Vxt = Aff^T Vxt' (:305-306; at the
      //   leg end Aff^T I = A^T + K^T B^T, :186),  vt += Vxt'^T yff (:301)
      wg_gemm(w, nu, nth, nx2, B.T(), X, MatV{nullptr, 0, 0}, G, 1.0);
      wg_gemm(w, nu, nth, nx2, rowmajor(Tm, nx2), X, MatV{nullptr, 0, 0}, K, -1.0);
      wg_gemm(w, nx, nth, nx2, rowmajor(Af, nx).T(), X, MatV{nullptr, 0, 0}, colmajor(Xn, nx), 1.0);
      wg_gemm(w, nth, 1, nx2, X.T(), colmajor(yf, nx2), colmajor(vt, nth), colmajor(vtn, nth), 1.0);
      __syncthreads();
      PT(4)
      // Vtt += Ghat_u^T Kth  (:308-310),  Yth = B Kth  (:295) straight into the record
      wg_gemm(w, nth, nth, nu, G.T(), K, colmajor(Tt, nth), colmajor(Tt, nth), 1.0);
      if (!leg_end)
        wg_gemm(w, nx2, nth, nu, B, K, MatV{nullptr, 0, 0}, rowmajor(dst + fo.fth + nu * nth, nth), 1.0);
      __syncthreads();
      PT(5)
      // the caller-visible record: ff | fb | fth | Vxx | vx | Vxt | Vtt | vt  (gar_layout.h)
      for (int e = w.tid; e < nr; e += w.nthr)
        dst[fo.ff + e] = (leg_end && e >= nu) ? 0.0 : src[f2.ff + e];
      for (int e = w.tid; e < nr * nx; e += w.nthr)
        dst[fo.fb + e] = (leg_end && e >= nu * nx) ? 0.0 : src[f2.fb + e];
      for (int e = w.tid; e < nu * nth; e += w.nthr)
        dst[fo.fth + e] = Kt[e];
      if (leg_end)
        for (int e = w.tid; e < nx2 * nth; e += w.nthr)
          dst[fo.fth + nu * nth + e] = 0.0;
      for (int e = w.tid; e < nx * nx + nx; e += w.nthr)
        dst[fo.Vxx + e] = src[f2.Vxx + e]; // Vxx | vx, contiguous in both layouts
      for (int e = w.tid; e < nx * nth; e += w.nthr)
        dst[fo.Vxt + e] = Xn[e];
      for (int e = w.tid; e < nth * nth; e += w.nthr)
        dst[fo.Vtt + e] = Tt[e];
      for (int e = w.tid; e < nth; e += w.nthr) {
        dst[fo.vt + e] = vtn[e];
        vt[e] = vtn[e];
      }
      __syncthreads();
      PT(6)
      double *tmp = Xt;
      Xt = Xn;
      Xn = tmp;
    }
  }
  __syncthreads();
  // the boundary tuple of this leg: (Vxx | Vxt | Vtt | vx | vt) of its first stage, blocks of nxb
  {
    const gar_stage_meta m = P.meta[t_beg];
    const int nxb = P.nxb, bs = nxb * nxb, nx = m.nx, nth = last_leg ? 0 : m.nth;
    const gar_factor_offsets fo = gar_factor_layout(m.nx, m.nu, m.nc, m.nx2, nth);
    const double *rec = fac + m.fac_off;
    double *tup = P.boundary + (long long)b * P.boundary_stride + (long long)blockIdx.x * P.tuple_doubles;
    for (int e = w.tid; e < P.tuple_doubles; e += w.nthr)
      tup[e] = 0.0;
    __syncthreads();
    for (int e = w.tid; e < nx * nx; e += w.nthr)
      tup[(e / nx) * nxb + (e % nx)] = rec[fo.Vxx + e];
    for (int e = w.tid; e < nx * nth; e += w.nthr)
      tup[bs + (e / nx) * nxb + (e % nx)] = rec[fo.Vxt + e];
    for (int e = w.tid; e < nth * nth; e += w.nthr)
      tup[2 * bs + (e / nth) * nxb + (e % nth)] = rec[fo.Vtt + e];
    for (int e = w.tid; e < nx; e += w.nthr)
      tup[3 * bs + e] = rec[fo.vx + e];
    for (int e = w.tid; e < nth; e += w.nthr)
      tup[3 * bs + nxb + e] = rec[fo.vt + e];
  }
  if (failed && w.tid == 0)
    atomicOr(&P.status[b], failed);
}

} // namespace gar
