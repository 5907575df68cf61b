// gar_wave_leg.hpp -- the parallel-in-time (leg) sweeps of ParallelRiccatiSolver
// (gar/parallel-solver.hxx:131-240) for uniform unconstrained problems, at the speed of
// the one-wave-per-problem kernels of gar_wave.hpp: one workgroup per (problem, leg).
//
//   backward: non-final legs run the parameterised recursion (nth = NX, the costate of the next
//             leg's first state is the parameter): the leg-end knot through wave_stage<2>
//             (terminalSolve under configure_knot), every other stage through
//             stageKernelSolve + :278-311 -- by ONE wave (gar_backward_wave_leg: wave_stage<1>) or,
//             the default, split over TWO (gar_backward_wave_leg2: wave A the plain recursion
//             wave_stage<3>, wave B the parameter part wave_param_stage, one barrier behind);
//             the final leg is the plain recursion (wave_stage<0>) down from the true terminal
//             knot.  Parallelism comes from the legs: a single long-horizon problem occupies
//             2 num_legs SIMDs instead of one.
//   condensed: gar_condensed_wave (the reference's elimination chain, here) and block cyclic
//             reduction (gar_cyclic.hpp, the default).
//   tuples  : (Vxx, Vxt, Vtt, vx, vt) of every leg's first stage -> the boundary buffer that the
//             all-gather / condensed solve consume.
//   forward : the closed-loop roll-out of a leg from the condensed solution; theta (constant
//             along the leg) is held in scalar registers, so its terms (Kth theta, Yth theta,
//             Vxt' theta) are independent FMAs beside the state recursion, not extra latency.
#pragma once
#include "gar_generic.hpp"
#include "gar_wave.hpp"

namespace gar {

struct LegParams {
  MfmaParams M;            // prob, fac, status, strides, in_off0/in_rec/in_offN, horizon
  const gar_stage_meta *meta;
  int num_legs, leg_begin; // this launch covers legs leg_begin + blockIdx.x
  // forward
  const double *csol;      // condensed solution [problem][2*num_legs][NX]
  double *sol;
  long long sol_stride;
  int sol_u, sol_l, nc0;
  // tuples
  double *boundary;        // [problem][local leg][tuple]
  long long boundary_stride;
  int tuple_doubles;
  double *cinfo;           // condensed-solve scratch: info slot of problem 0 (residual, steps)
  long long cinfo_stride;
  const int *skip;         // folded solvers (gar_fold.hpp): problems with skip[b] != 0 belong to the generic kernels
};

template <int NX, int NU>
__global__ void __launch_bounds__(64, 1) gar_backward_wave_leg(LegParams Q) {
  using C = WaveCfg<NX, NU>;
  using M = MfmaCfg<NX, NU>;
  constexpr int PK = C::PK;
  const int lane = (int)threadIdx.x & 63;
  const int leg = (int)blockIdx.x + Q.leg_begin;
  const int b = (int)blockIdx.y;
  if (Q.skip != nullptr && Q.skip[b] != 0)
    return;
  double *sm = gar_smem;
  MfmaParams P = Q.M;
  const double *prob = P.prob + (long long)b * P.prob_stride;
  const int N = P.horizon;
  int t_beg, t_end;
  gar_get_work(N, leg, Q.num_legs, &t_beg, &t_end);
  const bool last_leg = (leg == Q.num_legs - 1);
  // factor records are uniform inside a leg: stage t at fac0 + t * fac_rec
  P.fac_rec = last_leg ? (long long)(M::fvx + NX) : (long long)C::prec;
  P.fac_rec = (P.fac_rec + 1) & ~1ll;
  double *fac = P.fac + (long long)b * P.fac_stride + Q.meta[t_beg].fac_off - (long long)t_beg * P.fac_rec;
  double *V = sm + C::oV, *vn = sm + C::oVn;
  // debug build (make trace, -DGAR_TRACE; scripts/trace_leg.py): cycle stamps of the second stage of leg 0
  // (wave_stage stamps stage horizon/2); compiled out otherwise -- the run-time test alone costs
  // 3 % of the leg sweep
#ifdef GAR_TRACE
  const bool tracing = P.trace != nullptr && b == 0 && leg == 0 && lane == 0;
  if (P.trace != nullptr)
    P.horizon = 2 * (t_beg + 1);
#else
  const bool tracing = false;
#endif
  WaveLane<NX, NU> L;
  wave_lane_init<NX, NU>(L, lane);
  WaveStage<NX, NU> S;
  int failed = 0;
  if (last_leg) {
    // ---- true terminal knot (terminalSolve, nu = 0, nc = 0, :175-178): Vxx = Q, vx = q
    const int t1 = N - 1 >= t_beg ? N - 1 : t_beg;
    if (N - 1 >= t_beg) {
      wave_load_a<NX, NU>(prob + P.in_off0 + (long long)t1 * P.in_rec, L, S);
      wave_load_b<NX, NU>(prob + P.in_off0 + (long long)t1 * P.in_rec, L, S);
    }
    {
      const double *rec = prob + P.in_offN;
      double *out = P.fac + (long long)b * P.fac_stride + Q.meta[N].fac_off;
      for (int e = lane; e < NX * NX; e += 64) {
        const int j = e / NX, i = e - j * NX;
        const double v = (i >= j) ? rec[M::tQ + e] : rec[M::tQ + i * NX + j];
        V[i * PK + j] = v;
        out[M::tVxx + e] = v;
      }
      if (lane < NX) {
        const double v = rec[M::tq + lane];
        vn[lane] = v;
        out[M::tvx + lane] = v;
      }
    }
    wave_sync();
    for (int t = N - 1; t >= t_beg; --t)
      wave_stage<NX, NU, 0>(P, sm, prob, fac, t, lane, L, S, failed, tracing);
  } else {
    const int te = t_end - 1;
    wave_load_a<NX, NU>(prob + P.in_off0 + (long long)te * P.in_rec, L, S);
    wave_load_b<NX, NU>(prob + P.in_off0 + (long long)te * P.in_rec, L, S);
    wave_stage<NX, NU, 2>(P, sm, prob, fac, te, lane, L, S, failed, tracing);
    for (int t = te - 1; t >= t_beg; --t)
      wave_stage<NX, NU, 1>(P, sm, prob, fac, t, lane, L, S, failed, tracing);
  }
  if (failed && lane == 0)
    atomicOr(&P.status[b], failed);
}

// ---- two waves per leg -----------------------------------------------------------------------
// The parameter part of a leg stage (riccati-kernel.hxx:278-311) needs from the plain part only
// K, the factorisation of Rhat and yff.  Wave A runs the plain recursion (wave_stage<.., 3, PAR>)
// and publishes those in LDS; wave B (this function) runs one stage behind the publication:
//   before the barrier of stage t : F operands of knot t, Ghat_u = B^T Vxt'
//   after it                      : Kth = -Rhat^{-1} Ghat_u, Aff = A + B K (recomputed: 27 MFMAs
//                                   are cheaper than moving Aff between waves), Yth, vt, Vxt, Vtt
// The critical path of a stage is wave A's plain stage (the parameter part is shorter), instead
// of the sum of both.  Published blocks alternate with the stage parity, so one workgroup barrier
// per stage orders everything.
template <int NX, int NU, int PAR>
__device__ __forceinline__ void wave_param_stage(const MfmaParams &P, double *sm, const double *prob,
                                                 double *fac, int t, int lane,
                                                 const WaveLane<NX, NU> &L, WaveStage<NX, NU> &S,
                                                 int &failed) {
  using C = WaveCfg<NX, NU>;
  constexpr int NW = C::NW, PG = C::PG, TX = C::TX, KS = C::KS, KU = C::KU;
  const int li = lane & 15, lk = lane >> 4;
  const double *sb = sm + PAR * C::pub_shift; // what wave A publishes for this stage
  const double *Gp = sb + C::oG, *Lr = sb + C::oLr, *ndi = sb + C::oDi;
  const double *yfp = sm + (PAR ? C::oYf1 : C::oYf);
  double *Xt = sm + C::oXt + lane, *Tt = sm + C::oTt + lane;
  double *Gt = sm + C::oGt, *vtl = sm + C::oVt;
  const unsigned lkx = 8u * (unsigned)(lk * NX + li);
  double *out = fac + (long long)t * P.fac_rec;
  const double *rec = prob + P.in_off0 + (long long)t * P.in_rec;
  const double *recn = rec - (t > 0 ? P.in_rec : 0);
  // ---- before the barrier: B of this knot, Ghat_u = B^T Vxt' (:286-287) ----------------------
  double Bop[TX][KU], Bop4[KU];
#pragma unroll
  for (int ti = 0; ti < TX; ++ti)
#pragma unroll
    for (int s = 0; s < KU; ++s)
      Bop[ti][s] = WaveLane<NX, NU>::x_in(ti) ? ldg_b(rec, 4 * s * NX + 16 * ti, L.bop0)
                                              : ldg_b(rec, 4 * s * NX, L.bopX);
  if (C::REM4) {
#pragma unroll
    for (int s = 0; s < KU; ++s)
      Bop4[s] = ldg_b(rec, 4 * s * NX, L.bop4);
  }
  constexpr int shLoT = C::shTile(0), shHiT = C::shTile(KU - 1);
  double4_t Gh[shHiT - shLoT + 1][TX];
#pragma unroll
  for (int tu = shLoT; tu <= shHiT; ++tu)
#pragma unroll
    for (int tj = 0; tj < TX; ++tj) {
      double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < KS; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(S.fo(tu, s), Xt[(s * TX + tj) * 64], acc, 0, 0, 0);
      Gh[tu - shLoT][tj] = acc;
    }
#pragma unroll
  for (int sp = 0; sp < KU; ++sp)
#pragma unroll
    for (int tj = 0; tj < TX; ++tj)
      if (16 * tj + 15 < NX || 16 * tj + li < NX)
        Gt[(4 * sp + lk) * PG + 16 * tj + li] = Gh[C::shTile(sp) - shLoT][tj][C::shReg(sp)];
  __syncthreads(); // wave A has published stage t
  // ---- Kth = -Rhat^{-1} Ghat_u (:288-291) with wave A's factorisation ---------------------------
  if (sb[C::oFlag] == 0.0) {
    const int rowu = lane < NU ? lane : NU - 1;
    double a_row[NU], nd[NU], y[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      a_row[j] = Lr[rowu * NU + j];
      nd[j] = ndi[j];
    }
    const int ct = lane < NX ? lane : NX - 1;
#pragma unroll
    for (int k = 0; k < NU; ++k)
      y[k] = Gt[k * PG + ct];
    ldl_solve_regs_bcast<NU>(a_row, nd, y);
    if (lane < NX) {
#pragma unroll
      for (int k = 0; k < NU; ++k)
        Gt[k * PG + ct] = y[k];
    }
  } else { // Bunch-Kaufman factors (interchanges / 2x2 pivots) in the published block
    for (int e = lane; e < NU * PG; e += 64)
      Gt[e] = -Gt[e];
    const double *sub = sb + C::oBk;
    const int *piv = (const int *)(sub + 16);
    const WG w1 = wave_self();
    wave_sync();
    wg_bk_solve(w1, NU, sb + C::oM, NU, sub, piv, Gt, PG, 1, NX);
  }
  wave_sync();
  double Kb[TX][KU], Kthb[TX][KU];
#pragma unroll
  for (int tj = 0; tj < TX; ++tj) {
    const int cc = (16 * tj + li) < NX ? (16 * tj + li) : NX - 1;
#pragma unroll
    for (int s = 0; s < KU; ++s) {
      Kb[tj][s] = Gp[(4 * s + lk) * PG + 1 + cc];
      Kthb[tj][s] = Gt[(4 * s + lk) * PG + cc];
      if (16 * tj + li < NX)
        stg_b(out, C::pFTH + 8 * tj * 2 * NW + 8 * s, L.fbl, Kthb[tj][s]);
    }
  }
  // ---- vt = vt' + Vxt'^T yff (:298-301), before Vxt' is replaced ---------------------------------
  {
    double ys[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
      ys[s] = yfp[4 * s + lk];
    double pt[TX];
#pragma unroll
    for (int tj = 0; tj < TX; ++tj) {
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s & 1)
          a1 = __builtin_fma(Xt[(s * TX + tj) * 64], ys[s], a1);
        else
          a0 = __builtin_fma(Xt[(s * TX + tj) * 64], ys[s], a0);
      }
      double a = a0 + a1;
      a += __shfl_xor(a, 16);
      a += __shfl_xor(a, 32);
      pt[tj] = a;
    }
    double st = pt[0];
#pragma unroll
    for (int tj = 1; tj < TX; ++tj)
      st = (lk == tj) ? pt[tj] : st;
    if (lane < NX) {
      const double v = vtl[lane] + st;
      vtl[lane] = v;
      out[C::pvt + lane] = v;
    }
  }
  // ---- Aff = A + B K in place on the F operand registers (as wave A does) -------------------------
  double4_t accT[TX];
  if (C::KST > 0 && !C::REM4) {
#pragma unroll
    for (int tj = 0; tj < TX; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        accT[tj][r] = (r < C::KST) ? S.FoT[tj][r] : 0.0;
  }
#pragma unroll
  for (int s = 0; s < KU; ++s)
#pragma unroll
    for (int tj = 0; tj < TX; ++tj)
#pragma unroll
      for (int ti = 0; ti < TX; ++ti) {
        if (ti < C::KSF)
          S.Fo[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bop[ti][s], Kb[tj][s], S.Fo[tj][ti], 0, 0, 0);
        else if (C::REM4)
          S.FoT[tj][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(Bop4[s], Kb[tj][s], S.FoT[tj][0], 0, 0, 0);
        else
          accT[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bop[ti][s], Kb[tj][s], accT[tj], 0, 0, 0);
      }
  // ---- Yth = B Kth (:295), fth rows NU.. -----------------------------------------------------------
#pragma unroll
  for (int tj = 0; tj < TX; ++tj)
#pragma unroll
    for (int ti = 0; ti < TX; ++ti) {
      double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < KU; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Bop[ti][s], Kthb[tj][s], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * ti + lk + 4 * r, j = 16 * tj + li;
        if (16 * ti + 4 * r < NX) {
          if (i < NX && j < NX)
            stg_b(out, C::pFTH + 8 * tj * 2 * NW + 2 * (NU + 16 * ti + 4 * r), L.fbl, acc[r]);
        }
      }
    }
  // ---- Vxt = Aff^T Vxt' (:304-306), one parameter column tile at a time ---------------------------
#pragma unroll
  for (int tj = 0; tj < TX; ++tj) {
    double4_t nv[TX];
#pragma unroll
    for (int ti = 0; ti < TX; ++ti) {
      double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const double aq = (s < 4 * C::KSF || C::REM4) ? S.fo(ti, s) : accT[ti][s - 4 * C::KSF];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, Xt[(s * TX + tj) * 64], acc, 0, 0, 0);
      }
      nv[ti] = acc;
    }
#pragma unroll
    for (int ti = 0; ti < TX; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * ti + 4 * r < NX) {
          Xt[((4 * ti + r) * TX + tj) * 64] = nv[ti][r];
          if (16 * tj + 15 < NX || 16 * tj + li < NX)
            stg_b(out, C::pVxt + 16 * tj * NX + 16 * ti + 4 * r, 8u * (unsigned)(li * NX + lk), nv[ti][r]);
        }
  }
  // ---- Vtt = Vtt' + Ghat_u^T Kth, computed transposed (see wave_stage) ----------------------------
#pragma unroll
  for (int ti = 0; ti < TX; ++ti)
#pragma unroll
    for (int tj = 0; tj < TX; ++tj) {
      double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * ti + 4 * r < NX)
          acc[r] = Tt[((4 * ti + r) * TX + tj) * 64];
#pragma unroll
      for (int s = 0; s < KU; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Kthb[ti][s], Gh[C::shTile(s) - shLoT][tj][C::shReg(s)],
                                                   acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * ti + 4 * r < NX) {
          Tt[((4 * ti + r) * TX + tj) * 64] = acc[r];
          if (16 * tj + 15 < NX || 16 * tj + li < NX)
            stg_b(out, C::pVtt + (16 * ti + 4 * r) * NX + 16 * tj, lkx, acc[r]);
        }
    }
  // ---- knot t-1: its F operands replace Aff
  wave_load_a<NX, NU>(recn, L, S);
  (void)failed;
}

template <int NX, int NU>
__global__ void __launch_bounds__(128, 1) gar_backward_wave_leg2(LegParams Q) {
  using C = WaveCfg<NX, NU>;
  using M = MfmaCfg<NX, NU>;
  constexpr int PK = C::PK;
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  const int leg = (int)blockIdx.x + Q.leg_begin;
  const int b = (int)blockIdx.y;
  if (Q.skip != nullptr && Q.skip[b] != 0)
    return;
  double *sm = gar_smem;
  MfmaParams P = Q.M;
  const double *prob = P.prob + (long long)b * P.prob_stride;
  const int N = P.horizon;
  int t_beg, t_end;
  gar_get_work(N, leg, Q.num_legs, &t_beg, &t_end);
  const bool last_leg = (leg == Q.num_legs - 1);
  P.fac_rec = last_leg ? (long long)(M::fvx + NX) : (long long)C::prec;
  P.fac_rec = (P.fac_rec + 1) & ~1ll;
  double *fac = P.fac + (long long)b * P.fac_stride + Q.meta[t_beg].fac_off - (long long)t_beg * P.fac_rec;
  double *V = sm + C::oV, *vn = sm + C::oVn;
  const bool tracing = false;
  WaveLane<NX, NU> L;
  wave_lane_init<NX, NU>(L, lane);
  WaveStage<NX, NU> S;
  int failed = 0;
  if (last_leg) { // the plain recursion: one wave
    if (wave != 0)
      return;
    const int t1 = N - 1 >= t_beg ? N - 1 : t_beg;
    if (N - 1 >= t_beg) {
      wave_load_a<NX, NU>(prob + P.in_off0 + (long long)t1 * P.in_rec, L, S);
      wave_load_b<NX, NU>(prob + P.in_off0 + (long long)t1 * P.in_rec, L, S);
    }
    {
      const double *rec = prob + P.in_offN;
      double *out = P.fac + (long long)b * P.fac_stride + Q.meta[N].fac_off;
      for (int e = lane; e < NX * NX; e += 64) {
        const int j = e / NX, i = e - j * NX;
        const double v = (i >= j) ? rec[M::tQ + e] : rec[M::tQ + i * NX + j];
        V[i * PK + j] = v;
        out[M::tVxx + e] = v;
      }
      if (lane < NX) {
        const double v = rec[M::tq + lane];
        vn[lane] = v;
        out[M::tvx + lane] = v;
      }
    }
    wave_sync();
    for (int t = N - 1; t >= t_beg; --t)
      wave_stage<NX, NU, 0>(P, sm, prob, fac, t, lane, L, S, failed, tracing);
  } else {
    const int te = t_end - 1;
    if (wave == 0) {
      wave_load_a<NX, NU>(prob + P.in_off0 + (long long)te * P.in_rec, L, S);
      wave_load_b<NX, NU>(prob + P.in_off0 + (long long)te * P.in_rec, L, S);
      wave_stage<NX, NU, 2>(P, sm, prob, fac, te, lane, L, S, failed, tracing); // writes Xt, Tt, vt
      __syncthreads();
      // the first alternating stage reads vx' where the leg-end stage left it (block 0): PAR = 1
      int t = te - 1;
      for (; t - 1 >= t_beg; t -= 2) {
        wave_stage<NX, NU, 3, 1>(P, sm, prob, fac, t, lane, L, S, failed, tracing);
        wave_stage<NX, NU, 3, 0>(P, sm, prob, fac, t - 1, lane, L, S, failed, tracing);
      }
      if (t >= t_beg)
        wave_stage<NX, NU, 3, 1>(P, sm, prob, fac, t, lane, L, S, failed, tracing);
    } else {
      if (te - 1 >= t_beg)
        wave_load_a<NX, NU>(prob + P.in_off0 + (long long)(te - 1) * P.in_rec, L, S);
      __syncthreads();
      int t = te - 1;
      for (; t - 1 >= t_beg; t -= 2) {
        wave_param_stage<NX, NU, 1>(P, sm, prob, fac, t, lane, L, S, failed);
        wave_param_stage<NX, NU, 0>(P, sm, prob, fac, t - 1, lane, L, S, failed);
      }
      if (t >= t_beg)
        wave_param_stage<NX, NU, 1>(P, sm, prob, fac, t, lane, L, S, failed);
    }
  }
  if (failed && lane == 0)
    atomicOr(&P.status[b], failed);
}

// (Vxx | Vxt | Vtt | vx | vt) of each local leg's first stage, blocks of NX
template <int NX, int NU>
__global__ void __launch_bounds__(256) gar_leg_tuples(LegParams Q) {
  using C = WaveCfg<NX, NU>;
  using M = MfmaCfg<NX, NU>;
  const int leg = (int)blockIdx.x + Q.leg_begin;
  const int b = (int)blockIdx.y;
  int t_beg, t_end;
  gar_get_work(Q.M.horizon, leg, Q.num_legs, &t_beg, &t_end);
  const bool last_leg = (leg == Q.num_legs - 1);
  const double *rec = Q.M.fac + (long long)b * Q.M.fac_stride + Q.meta[t_beg].fac_off;
  double *tup = Q.boundary + (long long)b * Q.boundary_stride + (long long)blockIdx.x * Q.tuple_doubles;
  if (blockIdx.x == 0 && threadIdx.x == 0 && Q.cinfo != nullptr) {
    // the cyclic-reduction kernels accumulate the residual norm with atomicMax
    Q.cinfo[(long long)b * Q.cinfo_stride] = 0.0;
    Q.cinfo[(long long)b * Q.cinfo_stride + 1] = 0.0;
    Q.cinfo[(long long)b * Q.cinfo_stride + 2] = 0.0; // max_i (|rhs| + |A| |sol|)_i (gar_cyclic_recover)
  }
  if (Q.skip != nullptr && Q.skip[b] != 0)
    return; // (the generic backward kernel writes this problem's tuples itself)
  const bool term = (t_beg == Q.M.horizon); // a leg made of the terminal knot alone
  const int oVxx = last_leg ? (term ? M::tVxx : M::fVxx) : C::pVxx;
  const int ovx = last_leg ? (term ? M::tvx : M::fvx) : C::pvx;
  constexpr int bs = NX * NX;
  for (int e = (int)threadIdx.x; e < bs; e += (int)blockDim.x) {
    tup[e] = rec[oVxx + e];
    tup[bs + e] = last_leg ? 0.0 : rec[C::pVxt + e];
    tup[2 * bs + e] = last_leg ? 0.0 : rec[C::pVtt + e];
  }
  for (int e = (int)threadIdx.x; e < NX; e += (int)blockDim.x) {
    tup[3 * bs + e] = rec[ovx + e];
    tup[3 * bs + NX + e] = last_leg ? 0.0 : rec[C::pvt + e];
  }
}

// collapseFeedback (parallel-solver.hpp:41-51) on the device order of this kernel family:
// K0 -= Kth0 * Vxt(0)^T on the factor record of stage 0
template <int NX, int NU>
__global__ void gar_collapse_feedback_t2(const gar_stage_meta *meta, double *fac,
                                         long long fac_stride, int batch, const int *flags, int want) {
  using C = WaveCfg<NX, NU>;
  using M = MfmaCfg<NX, NU>;
  const int b = (int)blockIdx.x;
  if (b >= batch || meta[0].nth == 0 || (flags != nullptr && (flags[b] != 0) != (want != 0)))
    return;
  double *rec = fac + (long long)b * fac_stride + meta[0].fac_off;
  for (int e = (int)threadIdx.x; e < NU * NX; e += (int)blockDim.x) {
    const int i = e / NX, j = e - i * NX;
    double s = 0.0;
    for (int k = 0; k < NX; ++k)
      s += rec[C::pFTH + M::fbT2(i, k)] * rec[C::pVxt + k * NX + j];
    rec[M::fFB + M::fbT2(i, j)] -= s;
  }
}


// ---------------------------------------------------------------------------------------------
// Condensed (leg-boundary) system, one WAVE per problem, blocks of NX resident in LDS.
// Same elimination as gar_condensed_generic / the reference (assembleCondensedSystem,
// parallel-solver.hxx:85-129; symmetricBlockTridiagSolve, block-tridiagonal.hpp:82-138):
// up-looking from the last block, D_i <- D_i - B_i D_{i+1}^{-1} B_i^T, then the down sweep.
// What changes is how a block step runs:
//   * D is factorised in registers (lane = row; unpivoted LDL^T while the first Bunch-Kaufman
//     test holds at every column -- then it IS Bunch-Kaufman's factorisation; otherwise the
//     generic device Bunch-Kaufman takes over for that block) and W = D^{-1} is formed once
//     (lane = column of the identity); every later use of the factorisation -- the right-hand
//     side, U = D^{-1} B^T, the refinement sweeps -- is a product with W;
//   * the coupling blocks that are -I by construction (super[2k+2], parallel-solver.hxx:105)
//     are not multiplied: U = -W and D_i <- D_i - W, which is what the products evaluate to;
//   * U = W B^T and D_i -= B U run on v_mfma_f64_16x16x4 tiles straight from LDS.
// The chain is serial in the block index (2*num_legs steps); everything inside a step is
// wave-parallel.  LDS: 4 blocks + the solution vector.
// ---------------------------------------------------------------------------------------------
template <int NX> struct CondCfg {
  // diagonal blocks of the blocked triangular inverse
  static constexpr int BS = (NX % 12 == 0) ? 12 : ((NX % 16 == 0) ? 16 : NX);
  static constexpr int NB = NX / BS, KB = BS / 4;
  static constexpr int TX = (NX + 15) / 16, KS = NX / 4;
  static_assert(NX % 4 == 0 && BS <= 16 && NB >= 1 && NB <= 3, "unsupported block size");
};

// one 16x16 accumulator tile += A[r0 + li][c0 + 4s + lk] * B[c0b + 4s + lk][j0 + li], s < KB:
// a BS x BS x BS product of sub-blocks of two column-major pitch-NX matrices in LDS (rows and
// columns past BS are clamped: they only feed accumulator entries that are never stored)
template <int NX>
__device__ __forceinline__ double4_t cond_blk_gemm(const double *A, int ar0, int ac0, const double *B,
                                                   int br0, int bc0, double4_t acc, int li, int lk) {
  using K = CondCfg<NX>;
  const int lc = li < K::BS ? li : K::BS - 1;
#pragma unroll
  for (int s = 0; s < K::KB; ++s) {
    const double aq = A[(ac0 + 4 * s + lk) * NX + ar0 + lc];
    const double bq = B[(bc0 + lc) * NX + br0 + 4 * s + lk];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, bq, acc, 0, 0, 0);
  }
  return acc;
}
// store the valid BS x BS part of a tile into a column-major pitch-NX matrix at (r0, c0)
template <int NX>
__device__ __forceinline__ void cond_blk_store(double *Mx, int r0, int c0, double4_t acc, double sgn,
                                               int li, int lk) {
  using K = CondCfg<NX>;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (4 * r < K::BS) { // compile-time
      if (lk + 4 * r < K::BS && li < K::BS)
        Mx[(c0 + li) * NX + r0 + lk + 4 * r] = sgn * acc[r];
    }
}

// W = D^{-1} from the unpivoted factorisation D = L Dd L^T held lane = row (a_row[j] = L(row, j),
// nd[k] = -1/d_k): L goes to LDS, its inverse M is formed block-wise (diagonal blocks by lane-
// parallel substitution -- lane = (block, column) --, off-diagonal blocks as small MFMA products),
// then W = M^T Dd^{-1} M on MFMA tiles, skipping the k-steps where the triangular M is zero.
// Lm may alias Wm.
// (cond_inverse_from_lds: the second half alone -- unit-lower L column-major in Lm, 1/d_k in Dl already in LDS, as the
// blocked factorisation of gar_ldl_blocked.hpp leaves them)
template <int NX>
__device__ __forceinline__ void cond_inverse_from_lds(double *Lm, double *Mm, double *Wm, double *Tm, double *Dl, int lane);
template <int NX>
__device__ __forceinline__ void cond_inverse(const double (&a_row)[NX], const double (&nd)[NX],
                                             double *Lm, double *Mm, double *Wm, double *Tm,
                                             double *Dl, int lane) {
  // clean unit-lower L, column-major; 1/d_k to lane k
  if (lane < NX) {
#pragma unroll
    for (int j = 0; j < NX; ++j)
      Lm[j * NX + lane] = (j < lane) ? a_row[j] : (j == lane ? 1.0 : 0.0);
  }
  {
    double dl = 0.0;
#pragma unroll
    for (int k = 0; k < NX; ++k)
      dl = (lane == k) ? -nd[k] : dl;
    if (lane < NX)
      Dl[lane] = dl;
  }
  cond_inverse_from_lds<NX>(Lm, Mm, Wm, Tm, Dl, lane);
}
template <int NX>
__device__ __forceinline__ void cond_inverse_from_lds(double *Lm, double *Mm, double *Wm, double *Tm, double *Dl, int lane) {
  using K = CondCfg<NX>;
  constexpr int BS = K::BS, NB = K::NB, TX = K::TX, KS = K::KS;
  const int li = lane & 15, lk = lane >> 4;
  for (int e = lane; e < NX * NX; e += 64)
    Mm[e] = 0.0;
  wave_sync();
  { // diagonal blocks: M_bb = L_bb^{-1}, lane = (block, column)
    const int lc = lane < NX ? lane : NX - 1;
    const int bb = lc / BS, c = lc - bb * BS;
    const double *Lb = Lm + bb * BS * (NX + 1);
    double y[BS];
#pragma unroll
    for (int i = 0; i < BS; ++i)
      y[i] = (i == c) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < BS - 1; ++j)
#pragma unroll
      for (int i = j + 1; i < BS; ++i)
        y[i] = __builtin_fma(-Lb[j * NX + i], y[j], y[i]);
    if (lane < NX) {
#pragma unroll
      for (int i = 0; i < BS; ++i)
        Mm[lc * NX + bb * BS + i] = y[i];
    }
  }
  wave_sync();
  const double4_t z4 = double4_t{0.0, 0.0, 0.0, 0.0};
  if (NB >= 2) {
    // T1 = L10 M00 ; T2 = L21 M11
    double4_t t1 = cond_blk_gemm<NX>(Lm, BS, 0, Mm, 0, 0, z4, li, lk);
    cond_blk_store<NX>(Tm, 0, 0, t1, 1.0, li, lk); // T1 at rows 0.., columns 0..
    if (NB >= 3) {
      double4_t t2 = cond_blk_gemm<NX>(Lm, 2 * BS, BS, Mm, BS, BS, z4, li, lk);
      cond_blk_store<NX>(Tm, BS, 0, t2, 1.0, li, lk); // T2 below T1
    }
    wave_sync();
    // M10 = -M11 T1 ; M21 = -M22 T2
    double4_t m10 = cond_blk_gemm<NX>(Mm, BS, BS, Tm, 0, 0, z4, li, lk);
    cond_blk_store<NX>(Mm, BS, 0, m10, -1.0, li, lk);
    if (NB >= 3) {
      double4_t m21 = cond_blk_gemm<NX>(Mm, 2 * BS, 2 * BS, Tm, BS, 0, z4, li, lk);
      cond_blk_store<NX>(Mm, 2 * BS, BS, m21, -1.0, li, lk);
    }
    wave_sync();
    if (NB >= 3) {
      // T3 = L20 M00 + L21 M10 ; M20 = -M22 T3
      double4_t t3 = cond_blk_gemm<NX>(Lm, 2 * BS, 0, Mm, 0, 0, z4, li, lk);
      t3 = cond_blk_gemm<NX>(Lm, 2 * BS, BS, Mm, BS, 0, t3, li, lk);
      cond_blk_store<NX>(Tm, 0, 0, t3, 1.0, li, lk);
      wave_sync();
      double4_t m20 = cond_blk_gemm<NX>(Mm, 2 * BS, 2 * BS, Tm, 0, 0, z4, li, lk);
      cond_blk_store<NX>(Mm, 2 * BS, 0, m20, -1.0, li, lk);
      wave_sync();
    }
  }
  // W = M^T Dd^{-1} M, lower tiles, mirrored; M(k, a) = 0 for k < a
  double op[TX][KS], dsc[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s)
    dsc[s] = Dl[4 * s + lk];
#pragma unroll
  for (int t = 0; t < TX; ++t) {
    const int c = (16 * t + li) < NX ? (16 * t + li) : NX - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s)
      op[t][s] = (4 * s + 3 >= 16 * t) ? Mm[c * NX + 4 * s + lk] : 0.0;
  }
  wave_sync(); // Lm (which may alias Wm) is dead from here
#pragma unroll
  for (int ta = 0; ta < TX; ++ta)
#pragma unroll
    for (int tb = 0; tb <= ta; ++tb) {
      double4_t acc = z4;
#pragma unroll
      for (int s = 0; s < KS; ++s)
        if (4 * s + 3 >= 16 * ta) // compile-time: rows k >= 16 ta of M's column tile ta
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(op[ta][s] * dsc[s], op[tb][s], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * ta + 4 * r < NX) {
          const int i = 16 * ta + lk + 4 * r, j = 16 * tb + li;
          if (i < NX && j < NX && (ta != tb || i >= j)) { // the lower entry, to both places
            Wm[j * NX + i] = acc[r];
            Wm[i * NX + j] = acc[r];
          }
        }
    }
}

// a pitch-NX block HBM -> LDS: every load is issued before the first LDS write
template <int NX>
__device__ __forceinline__ void cond_copy_block(double *dst, const double *src, int lane) {
  constexpr int bs = NX * NX, NCH = (bs + 63) / 64;
  double tmp[NCH];
#pragma unroll
  for (int q = 0; q < NCH; ++q) {
    const int e = 64 * q + lane;
    tmp[q] = src[(64 * q + 63 < bs || e < bs) ? e : bs - 1];
  }
#pragma unroll
  for (int q = 0; q < NCH; ++q) {
    const int e = 64 * q + lane;
    if (64 * q + 63 < bs || e < bs)
      dst[e] = tmp[q];
  }
}

template <int NX>
__global__ void __launch_bounds__(64, 1) gar_condensed_wave(CondensedParams P) {
  constexpr int bs = NX * NX;
  const int lane = (int)threadIdx.x & 63;
  const int b = (int)blockIdx.x;
  const int nblk = 2 * P.num_legs, N = nblk - 1;
  if (P.gated) { // the cyclic-reduction solve of this problem met the residual threshold?
    double *inf = P.scratch + (long long)b * P.scratch_stride + 4ll * nblk * NX * NX + 4ll * nblk * NX;
    // (refinement disabled: the cyclic-reduction result stands -- unless a block inverse failed
    // outright, which poisons the residual with +inf)
    if (inf[0] <= P.threshold || (P.max_refinement == 0 && inf[0] <= 1.79e308) ||
        (inf[0] <= 1.79e308 && inf[0] <= P.backward_ok * inf[2])) { // (see gar_cyclic_recover)
      if (lane == 0)
        inf[3] = 0.0; // the cyclic-reduction result stands (gar_hip_condensed_resolved)
      return;
    }
    if (lane == 0)
      inf[3] = 1.0; // re-solved here, in the reference's order
  } else if (lane == 0) {
    (P.scratch + (long long)b * P.scratch_stride + 4ll * nblk * NX * NX + 4ll * nblk * NX)[3] = 0.0;
  }
  const WG w1 = wave_self();
  double *sm = gar_smem;
  using K = CondCfg<NX>;
  constexpr int TX = K::TX, KS = K::KS;
  const int li = lane & 15, lk = lane >> 4;
  double *Dm = sm, *Wm = sm + bs, *Bm = sm + 2 * bs, *Mm = sm + 3 * bs;
  double *Tm = sm + 4 * bs;                   // NX x 16 scratch of the blocked inverse
  double *Dl = Tm + 16 * NX;                  // 1/d_k
  double *bsub = Dl + NX;                     // Bunch-Kaufman fallback: sub | piv, ctrl
  int *bpiv = (int *)(bsub + NX + (NX & 1));
  double *solv = bsub + NX + (NX & 1) + (NX + 16) / 2 + 2; // [nblk][NX]
  double *errv = solv + nblk * NX;                         // [nblk][NX]
  double *S = P.scratch + (long long)b * P.scratch_stride;
  double *Wall = S + 2ll * nblk * bs, *Uall = S + 3ll * nblk * bs; // the facD / U slots
  double *info = S + 4ll * nblk * bs + 4ll * nblk * NX;
  double *sol = P.csol + (long long)b * nblk * NX;
  const double *prob = P.prob + (long long)b * P.prob_stride;
  const int nc0 = P.nc0;
  const int row = lane < NX ? lane : NX - 1;
  int failed = 0;

  // right-hand side (assembleCondensedSystem, :118-128); block 0 padded to NX with zeros
  for (int e = lane; e < NX; e += 64)
    solv[e] = e < nc0 ? -prob[P.g0_off + e] : 0.0;
  for (int leg = 0; leg < P.num_legs; ++leg) {
    const double *tup = cond_tuple(P, b, leg);
    if (lane < NX) {
      solv[(2 * leg + 1) * NX + lane] = -tup[3 * bs + lane];
      if (leg + 1 < P.num_legs)
        solv[(2 * leg + 2) * NX + lane] = -tup[3 * bs + NX + lane];
    }
  }
  // D of the last block: Vxx of the last leg
  {
    const double *tup = cond_tuple(P, b, P.num_legs - 1);
    for (int e = lane; e < bs; e += 64)
      Dm[e] = tup[e];
  }
  wave_sync();

  // one pass of "x_ib <- W_ib x_ib ; x_i -= B_i x_ib" uses these two helpers (lane = row)
  auto matvec = [&](const double *Mcol, double x) { // sum_k M(row, k) x_k, M column-major
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < NX; k += 2) {
      s0 = __builtin_fma(Mcol[k * NX + row], lane_bcast(x, k), s0);
      s1 = __builtin_fma(Mcol[(k + 1) * NX + row], lane_bcast(x, k + 1), s1);
    }
    return s0 + s1;
  };
  auto matvecT = [&](const double *Mcol, double x) { // sum_k M(k, row) x_k
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < NX; k += 2) {
      s0 = __builtin_fma(Mcol[row * NX + k], lane_bcast(x, k), s0);
      s1 = __builtin_fma(Mcol[row * NX + k + 1], lane_bcast(x, k + 1), s1);
    }
    return s0 + s1;
  };
  // the coupling block B_i = super[i] (rows: block i, columns: block i+1) into Bm, column-major,
  // zero-padded; returns 1 if it is -I (nothing loaded)
  auto load_coupling = [&](int i) -> int {
    if (i >= 2 && (i & 1) == 0)
      return 1;
    if (i == 0) { // G0: nc0 x NX
      for (int e = lane; e < bs; e += 64) {
        const int j = e / NX, r = e - j * NX;
        Bm[e] = r < nc0 ? prob[P.G0_off + j * nc0 + r] : 0.0;
      }
    } else { // Vxt of leg (i-1)/2
      cond_copy_block<NX>(Bm, cond_tuple(P, b, (i - 1) >> 1) + bs, lane);
    }
    return 0;
  };

  // ---- up-looking elimination -------------------------------------------------------------
#ifdef GAR_TRACE
  const bool tracing = P.trace != nullptr && b == 0 && lane == 0;
#define GAR_CMARK(id)                                                          \
  __builtin_amdgcn_sched_barrier(0);                                           \
  if (tracing && (ib == N - 2 || ib == N - 3))                                 \
    P.trace[(id) + 16 * (N - 2 - ib)] = (long long)clock64();                  \
  __builtin_amdgcn_sched_barrier(0);
#else
#define GAR_CMARK(id)
#endif
  for (int ib = N; ib >= 0; --ib) {
    const int n = ib == 0 ? nc0 : NX;
    GAR_CMARK(0)
    if (ib == 0) { // pad the nc0 x nc0 block with an identity
      for (int e = lane; e < bs; e += 64) {
        const int j = e / NX, r = e - j * NX;
        if (r >= n || j >= n)
          Dm[e] = (r == j) ? 1.0 : 0.0;
      }
      wave_sync();
    }
    {
      double a_row[NX], nd[NX];
      const int verdict = wave_ldl_fast<NX>(Dm, lane, a_row, nd);
      GAR_CMARK(1)
      if (verdict == 0) {
        cond_inverse<NX>(a_row, nd, Wm, Mm, Wm, Tm, Dl, lane);
      } else {
        for (int e = lane; e < bs; e += 64) {
          const int j = e / NX, r = e - j * NX;
          Wm[e] = (r == j) ? 1.0 : 0.0;
        }
        wave_sync();
        failed |= wg_bk_factor(w1, NX, Dm, NX, bsub, bpiv, bpiv + NX + 8);
        wg_bk_solve(w1, NX, Dm, NX, bsub, bpiv, Wm, 1, NX, NX);
      }
    }
    wave_sync();
    GAR_CMARK(2)
    for (int e = lane; e < bs; e += 64)
      Wall[(long long)ib * bs + e] = Wm[e];
    // x_ib <- D^{-1} x_ib
    const double xb = matvec(Wm, solv[ib * NX + row]);
    if (lane < NX)
      solv[ib * NX + lane] = xb;
    if (ib == 0)
      break;
    GAR_CMARK(3)
    const int i = ib - 1;
    const int negI = load_coupling(i);
    // D_i of the next step: the original diagonal block ...
    if (i == 0) {
      for (int e = lane; e < bs; e += 64)
        Dm[e] = 0.0; // -mudyn I with mudyn = 0 (:93-94, :165)
    } else {
      const double *tup = cond_tuple(P, b, (i - 1) >> 1);
      const int off = (i & 1) ? 0 : 2 * bs; // odd: Vxx(leg) ; even: Vtt(leg)
      cond_copy_block<NX>(Dm, tup + off, lane);
    }
    wave_sync();
    GAR_CMARK(4)
    if (negI) {
      // x_i -= (-I) x_ib ; U_i = -W ; D_i -= (-I)(-W)
      if (lane < NX)
        solv[i * NX + lane] += xb;
      for (int e = lane; e < bs; e += 64) {
        const double wv = Wm[e];
        Uall[(long long)i * bs + e] = -wv;
        Dm[e] -= wv;
      }
    } else {
      const double xi = solv[i * NX + row] - matvec(Bm, xb);
      if (lane < NX)
        solv[i * NX + lane] = xi;
      // U_i = W B^T (NX x NX; columns >= dim(i) are zero) ; D_i -= B U_i (lower tiles).
      // W(16t+li, 4s+lk) and B(16t+li, 4s+lk) share one operand pattern; U's accumulator
      // registers ARE the B operand of the second product (D -> B identity of the f64 MFMA)
      double opW[TX][KS], opB[TX][KS];
#pragma unroll
      for (int t = 0; t < TX; ++t) {
        const int c = (16 * t + li) < NX ? (16 * t + li) : NX - 1;
#pragma unroll
        for (int sk = 0; sk < KS; ++sk) {
          opW[t][sk] = Wm[(4 * sk + lk) * NX + c];
          opB[t][sk] = Bm[(4 * sk + lk) * NX + c];
        }
      }
      double4_t Ut[TX][TX];
#pragma unroll
      for (int ta = 0; ta < TX; ++ta)
#pragma unroll
        for (int tb = 0; tb < TX; ++tb) {
          double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int sk = 0; sk < KS; ++sk)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(opW[ta][sk], opB[tb][sk], acc, 0, 0, 0);
          Ut[ta][tb] = acc;
        }
      GAR_CMARK(5)
      double *Ug = Uall + (long long)i * bs;
#pragma unroll
      for (int ta = 0; ta < TX; ++ta)
#pragma unroll
        for (int tb = 0; tb < TX; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * ta + 4 * r < NX) {
              if (16 * tb + 15 < NX || 16 * tb + li < NX)
                Ug[(16 * tb + li) * NX + 16 * ta + 4 * r + lk] = Ut[ta][tb][r];
            }
#pragma unroll
      for (int ta = 0; ta < TX; ++ta)
#pragma unroll
        for (int tb = 0; tb <= ta; ++tb) {
          double4_t acc;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ii = 16 * ta + lk + 4 * r, jj = 16 * tb + li;
            acc[r] = (16 * ta + 4 * r < NX) ? Dm[(jj < NX ? jj : NX - 1) * NX + (ii < NX ? ii : NX - 1)] : 0.0;
          }
#pragma unroll
          for (int sk = 0; sk < KS; ++sk)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-opB[ta][sk], Ut[sk >> 2][tb][sk & 3], acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * ta + 4 * r < NX) {
              const int ii = 16 * ta + lk + 4 * r, jj = 16 * tb + li;
              if (ii < NX && jj < NX)
                Dm[jj * NX + ii] = acc[r];
            }
        }
    }
    wave_sync();
    GAR_CMARK(6)
  }
#undef GAR_CMARK
  // ---- down sweep: x_{i+1} -= U_i x_i  (:131-134) -----------------------------------------------
  __threadfence_block();
  for (int i = 0; i < N; ++i) {
    const double xi = solv[i * NX + row];
    double s0 = 0.0, s1 = 0.0;
    const double *Ug = Uall + (long long)i * bs;
#pragma unroll
    for (int k = 0; k < NX; k += 2) {
      s0 = __builtin_fma(Ug[k * NX + row], lane_bcast(xi, k), s0);
      s1 = __builtin_fma(Ug[(k + 1) * NX + row], lane_bcast(xi, k + 1), s1);
    }
    if (lane < NX)
      solv[(i + 1) * NX + lane] -= s0 + s1;
    wave_sync();
  }

  // ---- iterative refinement (parallel-solver.hxx:184-202; residual against the true
  // right-hand side, as gar_condensed_generic) ----------------------------------------------------
  int steps = 0;
  double resdl = 0.0;
  for (int it = 0; it < P.max_refinement; ++it) {
    // err = rhs - A sol, block row by block row (blockTridiagMatMul, :52-75)
    double mx = 0.0;
    for (int i = 0; i <= N; ++i) {
      const double xi = solv[i * NX + row];
      double r;
      if (i == 0) {
        r = row < nc0 ? -prob[P.g0_off + row] : 0.0; // D_0 = 0
      } else {
        const double *tup = cond_tuple(P, b, (i - 1) >> 1);
        const int off = (i & 1) ? 0 : 2 * bs;
        r = -tup[3 * bs + ((i & 1) ? 0 : NX) + row] - matvec(tup + off, xi);
      }
      if (i > 0) { // sub[i-1] = super[i-1]^T applied to x_{i-1}
        const double xp = solv[(i - 1) * NX + row];
        if (i - 1 == 0) {
          double s = 0.0;
          for (int k = 0; k < nc0; ++k)
            s += prob[P.G0_off + row * nc0 + k] * lane_bcast(xp, k);
          r -= s;
        } else if (((i - 1) & 1) == 0) {
          r += xp;
        } else {
          r -= matvecT(cond_tuple(P, b, (i - 2) >> 1) + bs, xp);
        }
      }
      if (i < N) { // super[i] applied to x_{i+1}
        const double xn = solv[(i + 1) * NX + row];
        if (i == 0) {
          double s = 0.0;
          for (int k = 0; k < NX; ++k)
            s += (row < nc0 ? prob[P.G0_off + k * nc0 + row] : 0.0) * lane_bcast(xn, k);
          r -= s;
        } else if ((i & 1) == 0) {
          r += xn;
        } else {
          r -= matvec(cond_tuple(P, b, (i - 1) >> 1) + bs, xn);
        }
      }
      if (lane >= NX || (i == 0 && lane >= nc0))
        r = 0.0;
      if (lane < NX)
        errv[i * NX + lane] = r;
      const double av = fabs(r);
      mx = (av > mx || av != av) ? av : mx;
    }
    // infinity norm over the wave
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const double other = __shfl_xor(mx, o);
      mx = (other > mx || other != other) ? other : mx;
    }
    resdl = mx;
    wave_sync();
    if (resdl <= P.threshold)
      break;
    // blockTridiagRefinementStep (:147-182) with the stored W and U
    for (int ib = N; ib >= 0; --ib) {
      const double eb = matvec(Wall + (long long)ib * bs, errv[ib * NX + row]);
      if (lane < NX)
        errv[ib * NX + lane] = eb;
      if (ib == 0)
        break;
      const int i = ib - 1;
      double ei = errv[i * NX + row];
      if (i >= 2 && (i & 1) == 0) {
        ei += eb;
      } else if (i == 0) {
        double s = 0.0;
        for (int k = 0; k < NX; ++k)
          s += (row < nc0 ? prob[P.G0_off + k * nc0 + row] : 0.0) * lane_bcast(eb, k);
        ei -= s;
      } else {
        ei -= matvec(cond_tuple(P, b, (i - 1) >> 1) + bs, eb);
      }
      if (lane < NX)
        errv[i * NX + lane] = ei;
      wave_sync();
    }
    for (int i = 0; i < N; ++i) {
      const double s = matvec(Uall + (long long)i * bs, errv[i * NX + row]);
      if (lane < NX)
        errv[(i + 1) * NX + lane] -= s;
      wave_sync();
    }
    for (int e = lane; e < nblk * NX; e += 64)
      solv[e] += errv[e];
    steps = it + 1;
    wave_sync();
  }
  for (int e = lane; e < nblk * NX; e += 64)
    sol[e] = solv[e];
  if (lane == 0) {
    info[0] = resdl;
    info[1] = (double)steps;
    if (failed)
      atomicOr(&P.status[b], 4);
  }
}

// ---- forward roll-out of one leg (parallel-solver.hxx:215-240 around forwardImpl,
// riccati-kernel.hxx:314-377) ---------------------------------------------------------------
template <int NX> struct LegFwdStage {
  double2_t g[NX / 2];  // [K; Aff] row r
  double2_t gt[NX / 2]; // [Kth; Yth] row r
  double vrow[NX];      // Vxx' row iv
  double trow[NX];      // Vxt' row iv
  double ff, vxn;
};

template <int NX, int NU, bool PARAM>
__device__ __forceinline__ void leg_fwd_load(const double *rec, const double *recn, int oVn, int ovn,
                                             int r, int iv, bool with_next, LegFwdStage<NX> &S) {
  using C = WaveCfg<NX, NU>;
  using M = MfmaCfg<NX, NU>;
  constexpr int NW = C::NW;
#pragma unroll
  for (int m = 0; m < NX / 2; ++m) {
    S.g[m] = *reinterpret_cast<const double2_t *>(rec + M::fFB + m * 2 * NW + 2 * r);
    if (PARAM)
      S.gt[m] = *reinterpret_cast<const double2_t *>(rec + C::pFTH + m * 2 * NW + 2 * r);
  }
  S.ff = rec[M::fFF + r];
  if (with_next) {
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      S.vrow[j] = recn[oVn + (iv >= j ? j * NX + iv : iv * NX + j)]; // symmetric: the lower triangle only (as gar_mfma.hpp::fwd_load)
      if (PARAM)
        S.trow[j] = recn[C::pVxt + j * NX + iv];
    }
    S.vxn = recn[ovn + iv];
  }
}

template <int NX, int NU, bool PARAM>
__device__ __forceinline__ void leg_forward_body(const LegParams &Q, int leg, int b, int lane) {
  using C = WaveCfg<NX, NU>;
  using M = MfmaCfg<NX, NU>;
  constexpr int NW = C::NW;
  const int N = Q.M.horizon;
  int t_beg, t_end;
  gar_get_work(N, leg, Q.num_legs, &t_beg, &t_end);
  const double *facb = Q.M.fac + (long long)b * Q.M.fac_stride;
  double *sol = Q.sol + (long long)b * Q.sol_stride;
  const double *cs = Q.csol + (long long)b * (2 * Q.num_legs) * NX;
  const long long frec = ((PARAM ? (long long)C::prec : (long long)(M::fvx + NX)) + 1) & ~1ll;
  const double *fac = facb + Q.meta[t_beg].fac_off - (long long)t_beg * frec;
  const int r = lane < NW ? lane : NW - 1;
  const int iv = lane < NX ? lane : NX - 1;
  const int ix = (lane >= NU && lane < NW) ? lane - NU : 0;
  // scatter of the condensed solution (:215-220): lbdas[t_beg], xs[t_beg]
  double xs = cs[(2 * leg + 1) * NX + ix];
  if (lane >= NU && lane < NW)
    sol[(long long)t_beg * NX + ix] = xs;
  {
    const int nl0 = (leg == 0) ? Q.nc0 : NX;
    const long long lo = (leg == 0) ? Q.sol_l : Q.sol_l + Q.nc0 + (long long)(t_beg - 1) * NX;
    for (int e = lane; e < nl0; e += 64)
      sol[lo + e] = cs[(2 * leg) * NX + e];
  }
  // theta = lbdas[t_end] = the next leg's first costate (:234-236), kept wave-uniform
  double th[NX];
  if (PARAM) {
    const double tv = cs[(2 * (leg + 1)) * NX + iv];
#pragma unroll
    for (int k = 0; k < NX; ++k)
      th[k] = lane_bcast(tv, k);
  }
  const int oVxx = PARAM ? C::pVxx : M::fVxx, ovx = PARAM ? C::pvx : M::fvx;
  for (int t = t_beg; t < t_end; ++t) {
    const bool last = (t == t_end - 1);
    if (!PARAM && t == N)
      break; // the terminal knot has no controls
    const double *rec = fac + (long long)t * frec;
    // the knot after t inside this leg (the final leg ends on the terminal knot's record)
    const bool next_is_term = (!PARAM && t + 1 == N);
    const double *recn = next_is_term ? facb + Q.meta[N].fac_off : rec + frec;
    LegFwdStage<NX> S;
    leg_fwd_load<NX, NU, PARAM>(rec, recn, next_is_term ? M::tVxx : oVxx, next_is_term ? M::tvx : ovx,
                                r, iv, !last, S);
    // u = kff + K x + Kth theta ; x' = yff + Aff x + Yth theta  (:334-336, :360-361)
    double acc = S.ff, acc1 = 0.0;
#pragma unroll
    for (int m = 0; m < NX / 2; ++m) {
      acc = __builtin_fma(S.g[m].x, lane_bcast(xs, NU + 2 * m), acc);
      acc1 = __builtin_fma(S.g[m].y, lane_bcast(xs, NU + 2 * m + 1), acc1);
    }
    if (PARAM) {
      double a2 = 0.0, a3 = 0.0;
#pragma unroll
      for (int m = 0; m < NX / 2; ++m) {
        a2 = __builtin_fma(S.gt[m].x, th[2 * m], a2);
        a3 = __builtin_fma(S.gt[m].y, th[2 * m + 1], a3);
      }
      acc1 += a2 + a3;
    }
    acc += acc1;
    if (lane < NU)
      sol[Q.sol_u + (long long)t * NU + lane] = acc;
    if (last)
      break;
    if (lane >= NU && lane < NW)
      sol[(long long)(t + 1) * NX + (lane - NU)] = acc;
    // lbd' = vx' + Vxx' x' + Vxt' theta  (:369-374)
    double lam = S.vxn, lam1 = 0.0;
#pragma unroll
    for (int j = 0; j < NX; j += 2) {
      lam = __builtin_fma(S.vrow[j], lane_bcast(acc, NU + j), lam);
      lam1 = __builtin_fma(S.vrow[j + 1], lane_bcast(acc, NU + j + 1), lam1);
    }
    if (PARAM) {
      double l2 = 0.0, l3 = 0.0;
#pragma unroll
      for (int j = 0; j < NX; j += 2) {
        l2 = __builtin_fma(S.trow[j], th[j], l2);
        l3 = __builtin_fma(S.trow[j + 1], th[j + 1], l3);
      }
      lam1 += l2 + l3;
    }
    lam += lam1;
    if (lane < NX)
      sol[Q.sol_l + Q.nc0 + (long long)t * NX + lane] = lam;
    xs = acc;
  }
}

// one launch for all local legs: the parameterised roll-out for the non-final legs, the plain one
// for the final leg
template <int NX, int NU>
__global__ void __launch_bounds__(64) gar_forward_wave_leg(LegParams Q) {
  const int lane = (int)threadIdx.x;
  const int leg = (int)blockIdx.x + Q.leg_begin;
  const int b = (int)blockIdx.y;
  if (Q.skip != nullptr && Q.skip[b] != 0)
    return;
  if (leg < Q.num_legs - 1)
    leg_forward_body<NX, NU, true>(Q, leg, b, lane);
  else
    leg_forward_body<NX, NU, false>(Q, leg, b, lane);
}

} // namespace gar
