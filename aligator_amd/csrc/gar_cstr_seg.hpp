// gar_cstr_seg.hpp -- ParallelRiccatiSolver (gar/parallel-solver.hxx:131-240) on problems whose knots carry COUPLED
// equality constraints (C x + D u + d = mu v with D != 0 somewhere), on the specialised constrained kernels (round 6).
//
// Leg mode folds constrained knots with D = 0 into the unconstrained wave-leg family (gar_fold.hpp: the reduced KKT
// matrix [Rhat D^T; D -mu I] of riccati-kernel.hxx:232-241 is block diagonal then).  With D != 0 it is not, and until
// round 6 such a problem was left to the any-dimension leg kernels (one 256-thread workgroup per leg, every block in
// LDS: 5.85 / 1.34 ms per sweep with 6 / 32 legs at (36, 12, 32), N = 256, against 0.78 / 0.42 for the fold).  Here the
// SEGMENT-LEG split of gar_leg_seg.hpp is applied to the constrained chain of gar_wave.hpp:
//   (1) the plain part of a leg = the serial constrained sweep (decoupled stage -> coupled stage -> LDS Bunch-Kaufman,
//       the three kernels of gar_backward_wave_body) over the leg's stage range, started from V' = 0, v' = 0 behind a
//       leg end: terminalSolve on the leg-end knot under configure_knot (parallel-solver.hxx:136-141;
//       riccati-kernel.hxx:151-183) IS stageKernelSolve with a zero next value function -- Rhat = R, Shat = S, the
//       reduced KKT matrix [R D^T; D -mu I] -- and its parameter outputs are that stage's closed loop
//       (Vxt = Gx + K^T Gu = Aff^T, vt = gamma + Gu^T kff = yff, [Kth; Zth] = -M^-1 [B^T; 0], Vtt = Gu^T Kth = B Kth);
//   (2) the parameter part (:278-311 with Gx = Gu = Gth = gamma = 0 and Gv = 0, :52-60) needs from the plain part what
//       its records hold -- Aff, yff, Vxx' -- and the knot's own R, B, D:
//         Ghat_u = B^T Vxt',  [Kth; Zth] = -M^-1 [Ghat_u; 0],  Yth = B Kth,
//         Vxt = Aff^T Vxt',   Vtt = Vtt' + Ghat_u^T Kth,        vt = vt' + Vxt'^T yff
//       started from Vxt' = I, Vtt' = 0, vt' = 0; M = [Rhat D^T; D -mu I] is re-formed per stage from R, B, D and the
//       stored Vxx' and factorised by the workgroup Bunch-Kaufman of gar_device.hpp -- the reference's own
//       factorisation of kktMat (:232-241).  Only Vxt chains the stages of a leg (gar_cseg_param_chain); the rest is
//       one workgroup per (stage, problem), all stages at once (gar_cseg_param_stage), which also writes the
//       caller-visible record (row-major fb, full Vxx: the layout the any-dimension roll-out, the getters and the
//       collapse read for such a problem, as before); gar_leg_param_finish sums Vtt, vt and builds the boundary tuple.
// The kernels run on the problems the fold flagged (only[b] == 1) and leave every other problem to the fold's family:
// same launch sequence for every problem, no host decision.  The plain kernel's records are scratch: they go to the
// flagged problem's slice of the wave-leg family's factor buffer (d_fac2), which that family does not touch for it.
// This header is compiled in a translation unit of its own (gar_cstr_seg.cpp): the stage functions it instantiates are
// register-critical, and a translation unit of their own keeps their code generation -- and that of everybody else's
// kernels -- independent of each other (see gar_wave_wide.cpp).
#pragma once
#include "gar_wave.hpp"
#include "gar_cstr_seg_api.hpp"

namespace gar {

// scratch records of the plain part (per problem): stage t < N at t * rec, the terminal knot at N * rec, then one
// dump slot of NX * NX doubles per leg (where the first stage of a leg "flushes" the value function it started from)
template <int NX, int NU, int NC> struct CsegCfg {
  using M = MfmaCfg<NX, NU, NC>;
  static constexpr int rec = M::fvx + NX;     // = gar_factor_doubles(NX, NU, NC, NX, 0)
  static constexpr int trec = M::tvx + NX;    // terminal record
  __host__ __device__ static constexpr long long dump(int horizon, int leg) {
    return (long long)horizon * rec + ((trec + 1) & ~1) + (long long)leg * NX * NX;
  }
  __host__ __device__ static constexpr long long doubles(int horizon, int num_legs) { return dump(horizon, num_legs); }
};

// ---- (1) the plain part: gar_backward_wave_body's chain over ONE leg's stage range ------------------------------
// grid (local legs, batch) x 64.  P.resume: one entry per (problem, local leg); P.fac / P.fac_rec: the scratch records.
// The serial chain is one-way (a problem that reached the LDS Bunch-Kaufman kernel stays there to the end): fine where
// pivoting is rare.  A LEG END is where it is not -- behind a leg end V' = 0, so the reduced KKT matrix is
// [R D^T; D -mu I] with the knot's bare R against D, and Bunch-Kaufman's first test (:61) fails on it whenever D's
// entries outweigh R's diagonal: with the one-way chain every leg of such a problem spent ALL its stages on the LDS
// Bunch-Kaufman (traced, 32 legs: 860 us of a 1.2 ms leg sweep).  The chain therefore runs in TWO rounds:
//   round 1: decoupled -> coupled for ONE stage -> LDS Bunch-Kaufman for ONE stage (flags & CSEG_SINGLE: the kernel
//            hands the leg back at the next knot);  round 2: decoupled (CSEG_REENTER: from the hand-over knot) ->
//            coupled -> LDS Bunch-Kaufman, each to the end of the leg.
// (The coupled kernel too takes one stage only in round 1: the launches of a round are serialised, so a leg that never
// pivots -- the last one, which starts from the true terminal knot -- would otherwise sweep all its stages in round 1
// while the others wait to sweep theirs in round 2: 186 + 122 + 252 us measured where 31 + 122 + 217 are needed.)
constexpr int CSEG_REENTER = 1, CSEG_SINGLE = 2;
template <int NX, int NU, int NC, int PHASE>
__global__ void __launch_bounds__(64, 1) gar_cseg_backward(MfmaParams P, int num_legs, int leg_begin, const int *only,
                                                            int flags) {
  using C = WaveCfg<NX, NU, NC>;
  using M = MfmaCfg<NX, NU, NC>;
  static_assert(NC > 0 && !M::WIDE, "the constrained one-wave family");
  constexpr int PK = C::PK;
  constexpr bool QP = GAR_QR_PACKED != 0;
  const int lane = (int)threadIdx.x & 63;
  const int leg = (int)blockIdx.x + leg_begin, b = (int)blockIdx.y;
  if (only[b] != 1)
    return;
  const int unit = b * (int)gridDim.x + (int)blockIdx.x;
  double *sm = gar_smem;
  const double *prob = P.prob + (long long)b * P.prob_stride;
  double *fac = P.fac + (long long)b * P.fac_stride;
  const int N = P.horizon;
  double *V = sm + C::oV, *vn = sm + C::oVn;
  int t_beg, t_end;
  gar_get_work(N, leg, num_legs, &t_beg, &t_end);
  const bool last_leg = (leg == num_legs - 1);
  const int t_first = last_leg ? N - 1 : t_end - 1;
  int tstart = t_first;
  const bool resumed = PHASE > 0 || (flags & CSEG_REENTER) != 0;
  if (resumed) {
    // resume[unit]: t >= 0: the knot the NEXT kernel of the chain takes over at; -1: the leg is done; -2 - t: the coupled
    // kernel of round 1 handed the leg back at knot t without having met a pivot -- round 2's first kernel picks it up,
    // the LDS Bunch-Kaufman kernel in between leaves it alone
    tstart = P.resume[unit];
    if (PHASE == 0 && tstart <= -2)
      tstart = -2 - tstart;
    if (tstart < 0)
      return;
  }
  if constexpr (PHASE == 0) {
    // D != 0 at the first knot: the decoupled stage would find out half-way through (wave_stage2: d_nonzero) and hand
    // over from this very knot -- found here at the price of one load per lane
    if (tstart >= t_beg) {
      const double *Dk = prob + P.in_off0 + (long long)tstart * P.in_rec + M::kD;
      bool nz = false;
      for (int e = lane; e < NC * NU; e += 64)
        nz |= (Dk[e] != 0.0);
      if (wave_ballot(nz) != 0ull) {
        if (lane == 0)
          P.resume[unit] = tstart;
        return;
      }
    }
  }
  WaveLane<NX, NU, NC> L;
  wave_lane_init<NX, NU, NC, QP>(L, lane);
  WaveStage<NX, NU> S;
  if (tstart >= t_beg) {
    wave_load_a<NX, NU>(prob + P.in_off0 + (long long)tstart * P.in_rec, L, S);
    wave_load_b<NX, NU, WaveLane<NX, NU, NC>, QP>(prob + P.in_off0 + (long long)tstart * P.in_rec, L, S);
  }
  double *vflush;
  if (resumed && tstart < t_first) { // V' = Vxx, vx' = vx of knot tstart + 1: complete in its record
    const double *rn = fac + (long long)(tstart + 1) * P.fac_rec;
    for (int e = lane; e < NX * NX; e += 64) {
      const int j = e / NX, i = e - j * NX;
      V[i * PK + j] = rn[M::fVxx + gar_sym_index(GAR_VXX_PACKED, NX, i, j)];
    }
    if (lane < NX)
      vn[lane] = rn[M::fvx + lane];
    vflush = fac + (long long)(tstart + 1) * P.fac_rec + M::fVxx;
  } else if (last_leg) {
    // the true terminal knot (terminalSolve, nu = 0, :146-149, :175-178): Z = C / mu, zff = d / mu,
    // Vxx = Q + C^T Z, vx = q + C^T zff
    const double *rec = prob + P.in_offN;
    double *out = fac + P.fac_offN;
    for (int e = lane; e < NX * NX; e += 64) {
      const int j = e / NX, i = e - j * NX;
      double v = (i >= j) ? rec[M::tQ + e] : rec[M::tQ + i * NX + j];
      for (int k = 0; k < NC; ++k)
        v = __builtin_fma(rec[M::tC + i * NC + k], rec[M::tC + j * NC + k] / P.mueq, v);
      V[i * PK + j] = v;
      if (!GAR_VXX_PACKED)
        out[M::tVxx + e] = v;
      else if (i >= j)
        out[M::tVxx + gar_sym_index(1, NX, i, j)] = v;
    }
    if (lane < NX) {
      double v = rec[M::tq + lane];
      for (int k = 0; k < NC; ++k)
        v = __builtin_fma(rec[M::tC + lane * NC + k], rec[M::td + k] / P.mueq, v);
      vn[lane] = v;
      out[M::tvx + lane] = v;
    }
    for (int e = lane; e < NC + NX; e += 64)
      out[e] = e < NC ? rec[M::td + e] / P.mueq : 0.0;
    for (int e = lane; e < (NC + NX) * NX; e += 64) {
      const int i = e / NX, j = e - i * NX;
      out[(NC + NX) + e] = i < NC ? rec[M::tC + j * NC + i] / P.mueq : 0.0;
    }
    vflush = fac + P.fac_offN + M::tVxx;
  } else { // behind a leg end: no value function (see (1) above); the first stage's deferred flush goes to the dump slot
    for (int e = lane; e < NX * PK; e += 64)
      V[e] = 0.0;
    if (lane < NX)
      vn[lane] = 0.0;
    vflush = fac + CsegCfg<NX, NU, NC>::dump(N, leg);
  }
  wave_sync();
  int failed = 0;
  const bool tracing = false;
  for (int t = tstart; t >= t_beg; --t) {
    if constexpr (PHASE == 2) {
      if (lane == 0)
        atomicAdd(&P.slow[3], 1); // (gar_hip_constrained_bk_stages, as the serial chain counts)
      wave_stage<NX, NU, 0, 0, NC, GAR_VXX_PACKED != 0>(P, sm, prob, fac, t, lane, L, S, failed, tracing);
      if (flags & CSEG_SINGLE) { // (its Vxx, vx are complete in the record: this stage flushes its own)
        if (lane == 0) {
          P.resume[unit] = t - 1 >= t_beg ? t - 1 : -1;
          if (failed)
            atomicOr(&P.status[b], failed);
        }
        return;
      }
    } else {
      if (PHASE == 1 && lane == 0)
        atomicAdd(&P.slow[2], 1);
      // (the lane offsets re-derived per stage: gar_backward_wave_body, GAR_COUPLED_REFRESH_LANE / GAR_CSTR_REFRESH_LANE)
      constexpr bool REFRESH = PHASE == 1 ? (GAR_COUPLED_REFRESH_LANE != 0) : (GAR_CSTR_REFRESH_LANE != 0);
      const int lane_t = REFRESH ? lane + fence0(S.fi) : lane;
      WaveLane<NX, NU, NC> Lt;
      if constexpr (REFRESH)
        wave_lane_init<NX, NU, NC, QP>(Lt, lane_t);
      if (!wave_stage2<NX, NU, NC, PHASE == 1>(P, sm, prob, fac, t, lane_t, REFRESH ? Lt : L, S, failed, vflush, tracing)) {
        if (lane == 0) { // over to the next kernel of the chain from this knot on
          P.resume[unit] = t;
          if (PHASE == 1)
            atomicAdd(&P.slow[2], -1);
          if (failed)
            atomicOr(&P.status[b], failed);
        }
        return;
      }
      if (PHASE == 1 && (flags & CSEG_SINGLE)) {
        tstart = t - 1; // (handed back below, behind the flush of this stage's Vxx)
        break;
      }
    }
  }
  if constexpr (PHASE < 2) {
    if (lane == 0)
      P.resume[unit] = (PHASE == 1 && (flags & CSEG_SINGLE) && tstart >= t_beg) ? -2 - tstart : -1;
    wave_flush_vxx<NX, GAR_VXX_PACKED != 0, PK>(V, vflush, lane);
  }
  if (failed && lane == 0)
    atomicOr(&P.status[b], failed);
}

#ifndef GAR_CSEG_BK_BLOCKED
#define GAR_CSEG_BK_BLOCKED 1
#endif
#ifndef GAR_CSEG_STAGE_THREADS
#define GAR_CSEG_STAGE_THREADS 512
#endif
// ---- (1b) the leg-END stage of every non-final leg, one workgroup per (leg, problem) ---------------------------------
// Behind a leg end V' = 0: the stage is [kff K; zff Z] = -M^-1 [r S^T; d C] with M = [R D^T; D -mu I] of the knot's own
// blocks (:151-172), yff = f + B kff, Aff = A + B K, Vxx = Q + S K + C^T Z, vx = q + S kff + C^T zff (:175-183) -- no
// product with a value function, so nothing of it needs the stage kernels' MFMA pipeline, and M is exactly the matrix on
// which Bunch-Kaufman pivots (R bare against D).  Through the chain it cost a failed attempt of the coupled kernel + one
// stage of the one-wave LDS Bunch-Kaufman kernel + a second round of launches (51 + 122 us); here it is one workgroup
// with the panel-blocked Bunch-Kaufman (the reference's own factorisation, as in gar_cseg_param_stage), and the chain
// behind it runs ONCE, from the knot below (flags & CSEG_REENTER on its first kernel).  Writes the stage's scratch
// record in the stage kernels' format (fbT2, packed lower Vxx) and hands the leg on through P.resume.
// grid (local legs, batch) x GAR_CSEG_STAGE_THREADS
template <int NX, int NU, int NC> __host__ __device__ constexpr int cseg_leg_end_lds_doubles() {
  constexpr int NK = NU + NC;
  // Mk | G | Sm Bm | Cm | Am | Qm | q f yff vx | sub | piv | ctrl | wk
  return NK * NK + NK * (NX + 2) + 2 * NX * NU + NC * NX + 2 * NX * NX + 4 * NX + NK + NK + 16 + NK * GAR_BK_PANEL + 16;
}
template <int NX, int NU, int NC>
__global__ void __launch_bounds__(GAR_CSEG_STAGE_THREADS) gar_cseg_leg_end(MfmaParams P, int num_legs, int leg_begin,
                                                                          const int *only) {
  using M = MfmaCfg<NX, NU, NC>;
  constexpr int NK = M::NK, bs = NX * NX, GLD = NX + 2; // G = [rhs | matrix | pad]: rows of NX + 2 doubles
  const WG w = wg_self();
  double *sm = gar_smem;
  const int leg = (int)blockIdx.x + leg_begin, b = (int)blockIdx.y;
  if (only[b] != 1)
    return;
  const int unit = b * (int)gridDim.x + (int)blockIdx.x;
  const int N = P.horizon;
  int t_beg, t_end;
  gar_get_work(N, leg, num_legs, &t_beg, &t_end);
  if (leg == num_legs - 1) { // the last leg starts from the true terminal knot: the chain's first kernel does that
    if (w.tid == 0)
      P.resume[unit] = (N - 1 >= t_beg) ? -2 - (N - 1) : -1;
    return;
  }
  const int t = t_end - 1;
  if (t < t_beg) {
    if (w.tid == 0)
      P.resume[unit] = -1;
    return;
  }
  const double *knot = P.prob + (long long)b * P.prob_stride + P.in_off0 + (long long)t * P.in_rec;
  double *out = P.fac + (long long)b * P.fac_stride + (long long)t * P.fac_rec;
  double *p = sm;
  auto take = [&](int n) { double *o = p; p += n; return o; };
  double *Mk = take(NK * NK), *G = take(NK * GLD), *Sm = take(NX * NU), *Bm = take(NX * NU), *Cm = take(NC * NX);
  double *Am = take(bs), *Qm = take(bs), *qv = take(NX), *fv = take(NX), *yf = take(NX), *vx = take(NX), *sub = take(NK);
  int *piv = (int *)take(NK), *ctrl = (int *)take(16);
  double *wk = take(NK * GAR_BK_PANEL);
  for (int e = w.tid; e < NK * NK; e += w.nthr) { // [R D^T; D -mu I] (:151-154), column-major, both triangles
    const int j = e / NK, i = e - j * NK;
    const int a = i >= j ? i : j, c = i >= j ? j : i;
    double v;
    if (a < NU)
      v = GAR_QR_PACKED ? knot[M::kR + gar_lower_index(NU, a, c)] : knot[M::kR + c * NU + a];
    else if (c < NU)
      v = knot[M::kD + c * NC + (a - NU)];
    else
      v = (a == c) ? -P.mueq : 0.0;
    Mk[e] = v;
  }
  for (int e = w.tid; e < NK * GLD; e += w.nthr) { // -[r S^T; d C] (:156-159), row i: rhs | matrix row
    const int i = e / GLD, j = e - i * GLD;
    double v = 0.0;
    if (j == 0)
      v = i < NU ? -knot[M::kr + i] : -knot[M::kd + (i - NU)];
    else if (j <= NX)
      v = i < NU ? -knot[M::kS + i * NX + (j - 1)] : -knot[M::kC + (j - 1) * NC + (i - NU)];
    G[e] = v;
  }
  for (int e = w.tid; e < NX * NU; e += w.nthr) {
    Sm[e] = knot[M::kS + e];
    Bm[e] = knot[M::kB + e];
  }
  for (int e = w.tid; e < NC * NX; e += w.nthr)
    Cm[e] = knot[M::kC + e];
  for (int e = w.tid; e < bs; e += w.nthr) {
    const int j = e / NX, i = e - j * NX;
    Am[e] = knot[M::kA + e];
    Qm[e] = GAR_QR_PACKED ? knot[M::kQ + (i >= j ? gar_lower_index(NX, i, j) : gar_lower_index(NX, j, i))] : knot[M::kQ + e];
  }
  for (int e = w.tid; e < NX; e += w.nthr) {
    qv[e] = knot[M::kq + e];
    fv[e] = knot[M::kf + e];
  }
  __syncthreads();
  int failed;
  if constexpr (GAR_CSEG_BK_BLOCKED != 0 && NK >= 24)
    failed = wg_bk_factor_blocked(w, NK, Mk, NK, sub, piv, ctrl, wk);
  else
    failed = wg_bk_factor(w, NK, Mk, NK, sub, piv, ctrl);
  __syncthreads();
  wg_bk_solve(w, NK, Mk, NK, sub, piv, G, GLD, 1, NX + 1); // [kff K; zff Z]
  __syncthreads();
  const MatV Km = rowmajor(G + 1, GLD), Zm = rowmajor(G + NU * GLD + 1, GLD), Bv = colmajor(Bm, NX), Sv = colmajor(Sm, NX);
  // yff = f + B kff ; Aff = A + B K (:266-267) ; Vxx = Q + S K (+ C^T Z below) ; vx = q + S kff + C^T zff (:175-183)
  wg_gemm(w, NX, NX, NU, Bv, Km, colmajor(Am, NX), colmajor(Am, NX), 1.0);
  wg_gemm(w, NX, NX, NU, Sv, Km, colmajor(Qm, NX), colmajor(Qm, NX), 1.0);
  for (int i = w.tid; i < NX; i += w.nthr) {
    double y = fv[i], v = qv[i];
    for (int k = 0; k < NU; ++k) {
      y = __builtin_fma(Bm[k * NX + i], G[k * GLD], y);
      v = __builtin_fma(Sm[k * NX + i], G[k * GLD], v);
    }
    for (int k = 0; k < NC; ++k)
      v = __builtin_fma(Cm[i * NC + k], G[(NU + k) * GLD], v);
    yf[i] = y;
    vx[i] = v;
  }
  __syncthreads();
  wg_gemm(w, NX, NX, NC, colmajor(Cm, NC).T(), Zm, colmajor(Qm, NX), colmajor(Qm, NX), 1.0);
  __syncthreads();
  // the scratch record, in the stage kernels' format: ff = [kff; zff; yff], fb = [K; Z; Aff] as fbT2, packed lower Vxx, vx
  for (int e = w.tid; e < M::NR; e += w.nthr)
    out[M::fFF + e] = e < NK ? G[e * GLD] : yf[e - NK];
  for (int e = w.tid; e < M::NR * NX; e += w.nthr) {
    const int r = e / NX, j = e - r * NX;
    out[M::fFB + M::fbT2(r, j)] = r < NK ? G[r * GLD + 1 + j] : Am[j * NX + (r - NK)];
  }
  for (int e = w.tid; e < bs; e += w.nthr) {
    const int j = e / NX, i = e - j * NX;
    if (i >= j)
      out[M::fVxx + gar_sym_index(GAR_VXX_PACKED, NX, i, j)] = Qm[e];
  }
  for (int e = w.tid; e < NX; e += w.nthr)
    out[M::fvx + e] = vx[e];
  if (w.tid == 0) {
    P.resume[unit] = (t - 1 >= t_beg) ? -2 - (t - 1) : -1;
    if (failed)
      atomicOr(&P.status[b], 1);
  }
}

// ---- (2) the parameter part -------------------------------------------------------------------------------------------

#ifndef GAR_CSEG_CHAIN_THREADS
#define GAR_CSEG_CHAIN_THREADS 576
#endif
template <int NX> __host__ __device__ constexpr int cseg_chain_lds_doubles() { return 3 * NX * NX + 16; }

// (2a) the chain Vxt_t = Aff_t^T Vxt_{t+1} (:305-306; at the leg end Aff^T I, :186), Vxt of every stage written to its
// caller-visible record.  grid (local legs, batch) x GAR_CSEG_CHAIN_THREADS
template <int NX, int NU, int NC>
__global__ void __launch_bounds__(GAR_CSEG_CHAIN_THREADS) gar_cseg_param_chain(CsegParams P) {
  using M = MfmaCfg<NX, NU, NC>;
  using CS = CsegCfg<NX, NU, NC>;
  constexpr int NK = M::NK, bs = NX * NX, NPRE = (bs + GAR_CSEG_CHAIN_THREADS - 1) / GAR_CSEG_CHAIN_THREADS;
  const WG w = wg_self();
  double *sm = gar_smem;
  const int leg = (int)blockIdx.x + P.leg_begin, b = (int)blockIdx.y;
  if (P.only[b] != 1 || leg == P.num_legs - 1)
    return;
  int t_beg, t_end;
  gar_get_work(P.horizon, leg, P.num_legs, &t_beg, &t_end);
  const double *fac2 = P.fac2 + (long long)b * P.fac2_stride;
  double *fac = P.fac + (long long)b * P.fac_stride;
  double *Xt = sm, *Xn = Xt + bs, *Af = Xn + bs;
  // Aff(r, j) of stage t: row NK + r of the fbT2 block of the scratch record; kept row-major (nx2 x nx) in LDS
  auto aff_src = [&](int t, int e) {
    const int r = e / NX, j = e - r * NX;
    return fac2[(long long)t * CS::rec + M::fFB + M::fbT2(NK + r, j)];
  };
  for (int e = w.tid; e < bs; e += w.nthr) {
    Af[e] = aff_src(t_end - 1, e);
    Xt[e] = ((e / NX) == (e % NX)) ? 1.0 : 0.0;
  }
  double pre[NPRE];
  for (int t = t_end - 1; t >= t_beg; --t) {
    __syncthreads(); // Af, Xt complete
    if (t > t_beg) { // the next stage's Aff in flight under the product
#pragma unroll
      for (int q = 0; q < NPRE; ++q) {
        const int e = w.tid + q * GAR_CSEG_CHAIN_THREADS;
        pre[q] = aff_src(t - 1, e < bs ? e : bs - 1);
      }
    }
    wg_gemm(w, NX, NX, NX, rowmajor(Af, NX).T(), colmajor(Xt, NX), MatV{nullptr, 0, 0}, colmajor(Xn, NX), 1.0);
    __syncthreads();
    if (t > t_beg) {
#pragma unroll
      for (int q = 0; q < NPRE; ++q) {
        const int e = w.tid + q * GAR_CSEG_CHAIN_THREADS;
        if (e < bs)
          Af[e] = pre[q];
      }
    }
    const gar_stage_meta m = P.meta[t];
    double *dst = fac + m.fac_off + gar_factor_layout(NX, NU, NC, NX, NX).Vxt;
    for (int e = w.tid; e < bs; e += w.nthr)
      dst[e] = Xn[e];
    double *tmp = Xt;
    Xt = Xn;
    Xn = tmp;
  }
}

template <int NX, int NU, int NC> __host__ __device__ constexpr int cseg_stage_lds_doubles() {
  constexpr int NK = NU + NC;
  // Vn | Bm VB | Mk | Tm | Gh | Kt | yf | sub | piv | ctrl | wk (the blocked Bunch-Kaufman's panel)
  return NX * NX + 2 * NX * NU + NK * NK + NK * NX + NU * NX + NK * NX + NX + NK + NK + 16 + NK * GAR_BK_PANEL + 16;
}

// (2b) everything else of stage t, all stages at once; also the caller-visible record (row-major fb, full Vxx: what the
// any-dimension roll-out, the getters and gar_leg_param_finish read).  grid (N + 1, batch) x GAR_CSEG_STAGE_THREADS
template <int NX, int NU, int NC>
__global__ void __launch_bounds__(GAR_CSEG_STAGE_THREADS) gar_cseg_param_stage(CsegParams P) {
  using M = MfmaCfg<NX, NU, NC>;
  using CS = CsegCfg<NX, NU, NC>;
  constexpr int NK = M::NK, NR = M::NR, bs = NX * NX;
  const WG w = wg_self();
  double *sm = gar_smem;
  const int t = (int)blockIdx.x, b = (int)blockIdx.y;
  if (P.only[b] != 1)
    return;
  int leg = P.leg_begin, t_beg = 0, t_end = 0;
  for (; leg < P.leg_begin + P.local_legs; ++leg) { // the leg of stage t (among this rank's)
    gar_get_work(P.horizon, leg, P.num_legs, &t_beg, &t_end);
    if (t >= t_beg && t < t_end)
      break;
  }
  if (leg >= P.leg_begin + P.local_legs)
    return;
  const int N = P.horizon;
  const double *prob = P.prob + (long long)b * P.prob_stride;
  const double *fac2 = P.fac2 + (long long)b * P.fac2_stride;
  double *fac = P.fac + (long long)b * P.fac_stride;
  const gar_stage_meta m = P.meta[t];
  const double *src = fac2 + (long long)t * CS::rec; // (t = N: the terminal record sits at N * rec too)
  double *dst = fac + m.fac_off;
  const bool last_leg = (leg == P.num_legs - 1);
  const bool leg_end = !last_leg && (t == t_end - 1);
  auto vxx_src = [&](const double *rec_vxx, int e) { // full symmetric block from the packed lower triangle
    const int j = e / NX, i = e - j * NX;
    return rec_vxx[gar_sym_index(GAR_VXX_PACKED, NX, i, j)];
  };
  if (t == N) { // the terminal knot (nu = 0): ff, fb row-major already; Vxx packed
    const gar_factor_offsets fo = gar_factor_layout(NX, 0, NC, m.nx2, 0);
    for (int e = w.tid; e < (NC + NX) * (1 + NX); e += w.nthr)
      dst[fo.ff + e] = src[e];
    for (int e = w.tid; e < bs; e += w.nthr)
      dst[fo.Vxx + e] = vxx_src(src + M::tVxx, e);
    for (int e = w.tid; e < NX; e += w.nthr)
      dst[fo.vx + e] = src[M::tvx + e];
    return;
  }
  const int nth = last_leg ? 0 : NX;
  const gar_factor_offsets fo = gar_factor_layout(NX, NU, NC, NX, nth);
  // the plain part of the caller-visible record: ff | fb (fbT2 -> row-major) | Vxx | vx  (leg end: yff, Aff zero, like
  // terminalSolve, :130-193)
  for (int e = w.tid; e < NR; e += w.nthr)
    dst[fo.ff + e] = (leg_end && e >= NK) ? 0.0 : src[M::fFF + e];
  for (int e = w.tid; e < NR * NX; e += w.nthr) {
    const int r = e / NX, j = e - r * NX;
    dst[fo.fb + e] = (leg_end && r >= NK) ? 0.0 : src[M::fFB + M::fbT2(r, j)];
  }
  for (int e = w.tid; e < bs; e += w.nthr)
    dst[fo.Vxx + e] = vxx_src(src + M::fVxx, e);
  for (int e = w.tid; e < NX; e += w.nthr)
    dst[fo.vx + e] = src[M::fvx + e];
  if (last_leg)
    return;
  double *p = sm;
  auto take = [&](int n) { double *o = p; p += n; return o; };
  double *Vn = take(bs), *Bm = take(NX * NU), *VB = take(NX * NU), *Mk = take(NK * NK), *Tm = take(NK * NX);
  double *Gh = take(NU * NX), *Kt = take(NK * NX), *yf = take(NX), *sub = take(NK);
  int *piv = (int *)take(NK), *ctrl = (int *)take(16);
  [[maybe_unused]] double *wk = take(NK * GAR_BK_PANEL);
  const double *knot = prob + P.in_off0 + (long long)t * P.in_rec;
  for (int e = w.tid; e < NK * NX; e += w.nthr) { // [B^T; 0]: B^T (nu x nx2, row-major) = B column-major, as it is
    const double v = e < NU * NX ? knot[M::kB + e] : 0.0;
    Tm[e] = v;
    if (e < NU * NX)
      Bm[e] = v;
  }
  for (int e = w.tid; e < NK * NK; e += w.nthr) { // [R D^T; D -mu I] (:232-236), column-major, both triangles
    const int j = e / NK, i = e - j * NK;
    const int a = i >= j ? i : j, c = i >= j ? j : i; // a >= c
    double v;
    if (a < NU)
      v = GAR_QR_PACKED ? knot[M::kR + gar_lower_index(NU, a, c)] : knot[M::kR + c * NU + a];
    else if (c < NU)
      v = knot[M::kD + c * NC + (a - NU)];
    else
      v = (a == c) ? -P.mueq : 0.0;
    Mk[e] = v;
  }
  for (int e = w.tid; e < NX; e += w.nthr)
    yf[e] = src[M::fFF + NK + e];
  if (!leg_end) { // V' symmetrised from its lower triangle as the consuming stage does (:216)
    const double *Vg = fac2 + (long long)(t + 1) * CS::rec + M::fVxx;
    for (int e = w.tid; e < bs; e += w.nthr)
      Vn[e] = vxx_src(Vg, e);
  }
  __syncthreads();
  const MatV B = colmajor(Bm, NX);
  if (!leg_end) { // Rhat = R + B^T (V' B)  (:221, :225)
    wg_gemm(w, NX, NU, NX, colmajor(Vn, NX), B, MatV{nullptr, 0, 0}, colmajor(VB, NX), 1.0);
    __syncthreads();
    wg_gemm(w, NU, NU, NX, B.T(), colmajor(VB, NX), colmajor(Mk, NK), colmajor(Mk, NK), 1.0);
    __syncthreads();
  }
  // Vxt' of this stage (the chain kernel's record of stage t + 1; the identity behind the leg end) takes V's place
  if (leg_end) {
    for (int e = w.tid; e < bs; e += w.nthr)
      Vn[e] = ((e / NX) == (e % NX)) ? 1.0 : 0.0;
  } else {
    const double *Xg = fac + P.meta[t + 1].fac_off + gar_factor_layout(NX, NU, NC, NX, NX).Vxt;
    for (int e = w.tid; e < bs; e += w.nthr)
      Vn[e] = Xg[e];
  }
  // the reference's own factorisation of kktMat: Bunch-Kaufman (:237-241)
  // (from 24 columns on in its panel-blocked form, as the any-dimension stage kernel does: one wave runs the pivot search,
  // the trailing matrix is updated once per panel on MFMA tiles -- bunchkaufman.hpp:172-344 is the reference's own)
  int failed;
  if constexpr (GAR_CSEG_BK_BLOCKED != 0 && NK >= 24)
    failed = wg_bk_factor_blocked(w, NK, Mk, NK, sub, piv, ctrl, wk);
  else
    failed = wg_bk_factor(w, NK, Mk, NK, sub, piv, ctrl);
  __syncthreads();
  wg_bk_solve(w, NK, Mk, NK, sub, piv, Tm, NX, 1, NX); // T = M^{-1} [B^T; 0], rows of nx2
  __syncthreads();
  const MatV X = colmajor(Vn, NX), G = rowmajor(Gh, NX), K = rowmajor(Kt, NX);
  // Ghat_u = B^T Vxt' (:286-287),  [Kth; Zth] = -M^{-1} [Ghat_u; 0] = -T Vxt' (:288-292)
  wg_gemm(w, NU, NX, NX, B.T(), X, MatV{nullptr, 0, 0}, G, 1.0);
  wg_gemm(w, NK, NX, NX, rowmajor(Tm, NX), X, MatV{nullptr, 0, 0}, K, -1.0);
  __syncthreads();
  // the increments of the running sums (gar_leg_param_finish adds them up): Ghat_u^T Kth (:308-310), Vxt'^T yff (:301);
  // Yth = B Kth (:295) -- straight into the record
  wg_gemm(w, NX, NX, NU, G.T(), K, MatV{nullptr, 0, 0}, colmajor(dst + fo.Vtt, NX), 1.0);
  wg_gemm(w, NX, 1, NX, X.T(), colmajor(yf, NX), MatV{nullptr, 0, 0}, colmajor(dst + fo.vt, NX), 1.0);
  if (!leg_end)
    wg_gemm(w, NX, NX, NU, B, K, MatV{nullptr, 0, 0}, rowmajor(dst + fo.fth + NK * NX, NX), 1.0);
  // fth = [Kth; Zth; Yth]  (Vxt: the chain kernel's; Vtt, vt: the increments, summed by gar_leg_param_finish)
  for (int e = w.tid; e < NK * NX; e += w.nthr)
    dst[fo.fth + e] = Kt[e];
  if (leg_end)
    for (int e = w.tid; e < NX * NX; e += w.nthr)
      dst[fo.fth + NK * NX + e] = 0.0;
  if (failed && w.tid == 0)
    atomicOr(&P.status[b], 1);
}

// ---- (3) the roll-out: one wave per (leg, problem), lane = row --------------------------------------------------------
// forwardImpl over one leg (riccati-kernel.hxx:314-377 under parallel-solver.hxx:209-243) in gar_forward_wide_leg's scheme
// (gar_leg_seg.hpp), with the constraint rows: lane r < nu + nc owns row r of [K; Z] and [Kth; Zth], lane r < NX row r
// of Aff, Yth, Vxx' and Vxt' (fb, fth row-major, Vxt' column-major); the state and the parameter are broadcast from the
// lanes that hold them:
//   [u; v] = [kff; zff] + [K; Z] x + [Kth; Zth] th,   x' = yff + Aff x + Yth th,   lbd' = vx' + Vxx' x' + Vxt' th
// Replaces the any-dimension roll-out on these problems (60 us per sweep at N = 256, 32 legs).
template <int NX, bool LAST, bool MORE>
__device__ __forceinline__ void cseg_fwd_stage(const CsegFwdParams &P, const double *fac, double *sol, int t, int lane,
                                               double &xs, double th) {
  constexpr int NTH = LAST ? 0 : NX;
  const gar_stage_meta m = P.meta[t];
  const int nu = m.nu, nk = m.nu + m.nc; // (nu = 0 at the terminal knot)
  const int iA = lane < NX ? lane : NX - 1, iK = lane < nk ? lane : (nk > 0 ? nk - 1 : 0);
  const gar_factor_offsets fo = gar_factor_layout(NX, nu, m.nc, m.nx2, NTH);
  const double *rec = fac + m.fac_off;
  const double x_in = xs;
  if (MORE) {
    const gar_stage_meta mn = P.meta[t + 1];
    const gar_factor_offsets fn = gar_factor_layout(NX, mn.nu, mn.nc, mn.nx2, NTH);
    const double *recn = fac + mn.fac_off;
    double2_t aff[NX / 2], yth[LAST ? 1 : NX / 2], vrow[NX / 2];
    double vxt[LAST ? 1 : NX];
    const double *ap = rec + fo.fb + (long long)(nk + iA) * NX, *ytp = rec + fo.fth + (long long)(nk + iA) * NX;
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
    const double yff = rec[fo.ff + nk + iA], vxn = recn[fn.vx + iA];
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
  if (nk > 0) { // [u; v] = [kff; zff] + [K; Z] x + [Kth; Zth] th: off the chain
    double2_t kro[NX / 2], kth[LAST ? 1 : NX / 2];
    const double *kp = rec + fo.fb + (long long)iK * NX, *ktp = rec + fo.fth + (long long)iK * NX;
#pragma unroll
    for (int q = 0; q < NX / 2; ++q) {
      kro[q] = *reinterpret_cast<const double2_t *>(kp + 2 * q);
      if (!LAST)
        kth[q] = *reinterpret_cast<const double2_t *>(ktp + 2 * q);
    }
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
    if (lane < nu)
      sol[m.u_off + lane] = u0 + u1;
    else if (lane < nk)
      sol[m.v_off + (lane - nu)] = u0 + u1;
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NX, int NU, int NC>
__global__ void __launch_bounds__(64) gar_cseg_forward(CsegFwdParams P) {
  static_assert(NX <= 64 && NX % 2 == 0 && NU + NC <= 64, "state and [u; v] live in the first lanes");
  const int lane = (int)threadIdx.x;
  const int leg = (int)blockIdx.x + P.leg_begin, b = (int)blockIdx.y;
  if (P.only[b] != 1)
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
      cseg_fwd_stage<NX, true, true>(P, fac, sol, t, lane, xs, 0.0);
    cseg_fwd_stage<NX, true, false>(P, fac, sol, t_end - 1, lane, xs, 0.0);
  } else {
    const double th = cs[(2 * (leg + 1)) * nxb + iA]; // theta = lbdas[end] (:234-236)
    for (int t = t_beg; t + 1 < t_end; ++t)
      cseg_fwd_stage<NX, false, true>(P, fac, sol, t, lane, xs, th);
    cseg_fwd_stage<NX, false, false>(P, fac, sol, t_end - 1, lane, xs, th);
  }
}

} // namespace gar
