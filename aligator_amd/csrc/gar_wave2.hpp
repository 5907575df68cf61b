// gar_wave2.hpp -- the plain stage (MODE 0, nc = 0) of the one-wave-per-problem backward sweep,
// second generation.  Same arithmetic as ProximalRiccatiKernel::stageKernelSolve
// (gar/riccati-kernel.hxx:209-277) and the same register / LDS / record layouts as wave_stage
// (gar_wave.hpp); what changed is WHERE the non-MFMA work of the serial chain
//     Rhat -> factorisation -> [kff | K] -> Aff, Vxx
// runs (round-1 profile: 13 k of a stage's 25 k cycles were VALU issue + exposed latency, 530 of
// the ~1 400 VALU instructions v_readlane broadcasts):
//   * vplus = vx' + V'f (:217-218) is never formed.  [qhat; rhat] = [q; r] + F^T vplus is
//     evaluated as [q; r] + F^T vx' + P^T f with P = V'F, which the stage computes anyway: 27 more
//     FMAs on registers that are already there instead of 36 LDS row reads + 72 v_readlane + 36
//     FMAs + an LDS round trip;
//   * the column tiles are swept from the control columns down: Rhat is complete after the first
//     tile column(s), goes through LDS into the lane = row layout, and its register LDL^T (the
//     latency-bound pivot chain) is issued while the MFMAs of the remaining tile columns keep the
//     matrix pipe busy -- the two are independent until the solve;
//   * the triangular solves run ON THE MFMA LAYOUT: Shat^T already sits in H's D registers as
//     X[4s'+lk][16tj+li], which is the B operand and the D result of v_mfma_f64_4x4x4 (four
//     independent 4x4x4 blocks, 16 cycles).  With L = (L_pq) in 4x4 blocks, forward substitution is
//       X_p += (-L_pq) X_q (q < p),   X_p <- (I + N_p)^{-1} X_p  by  y <- x - N_p y, three times
//     (N_p strictly lower, nilpotent: row i is final after i sweeps -- the plain substitution, in
//     block form), then X_p *= -1/d, then the same with the transposed blocks, bottom up.  The
//     A operands (-L_pq and its transpose, lane (li,lk) -> element (li&3, lk)) are twelve ds_read_b64
//     from the row-major copy of L the factorisation leaves in LDS.  No v_readlane substitution
//     (264 v_readlane + 144 FMA per stage before), no export of Shat^T to LDS, and K comes out in
//     the B-operand layout Aff and Vxx consume: no LDS round trip after the solve either.
//     rhat rides along as one more column (a spare lane column of the last state tile when
//     NX % 16 != 0, else a tile of its own whose result is kff replicated over li).
// The rare stage whose Rhat fails the first Bunch-Kaufman test takes wave_slow_factor_solve
// exactly as before (complete rule; generic device Bunch-Kaufman when it really pivots).
#pragma once
#include "gar_wave.hpp"

namespace gar {

// LDS hand-off between lanes of ONE wave.  The LDS executes a wave's instructions in order, so a
// ds_write followed by a ds_read of the same wave needs no barrier in hardware; the compiler only
// has to keep the two in order -- and, unlike wave_sync(), this does not fence the MFMAs around it.
__device__ __forceinline__ void wave_lds_order() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" ::: "memory");
#else
  wave_sync(); // the CPU emulator runs one thread per lane
#endif
}

// v from lane N of the caller's 16-lane row, in one VALU instruction (v_mov_b64_dpp row_newbcast:N:
// the only DPP control CDNA's double-precision ALU accepts) -- no SGPR round trip, no
// VALU-writes-SGPR wait states: a v_readlane broadcast of a double is two instructions plus an
// s_nop before its first use.  N must be a constant after unrolling.
template <int N> __device__ __forceinline__ double row_bcast_c(double v, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
  (void)lane;
  return __builtin_amdgcn_mov_dpp(v, 0x150 + N, 0xF, 0xF, true);
#else
  return __shfl(v, (lane & 48) | N);
#endif
}
__device__ __forceinline__ double row_bcast(double v, int n, int lane) {
  switch (n) {
  case 0: return row_bcast_c<0>(v, lane);
  case 1: return row_bcast_c<1>(v, lane);
  case 2: return row_bcast_c<2>(v, lane);
  case 3: return row_bcast_c<3>(v, lane);
  case 4: return row_bcast_c<4>(v, lane);
  case 5: return row_bcast_c<5>(v, lane);
  case 6: return row_bcast_c<6>(v, lane);
  case 7: return row_bcast_c<7>(v, lane);
  case 8: return row_bcast_c<8>(v, lane);
  case 9: return row_bcast_c<9>(v, lane);
  case 10: return row_bcast_c<10>(v, lane);
  case 11: return row_bcast_c<11>(v, lane);
  case 12: return row_bcast_c<12>(v, lane);
  case 13: return row_bcast_c<13>(v, lane);
  case 14: return row_bcast_c<14>(v, lane);
  default: return row_bcast_c<15>(v, lane);
  }
}

// wave_ldl_fast (gar_wave.hpp) leaving -L(i,j) in a[j] (j < i): the sign the block updates
// X_p += (-L_pq) X_q want.  Product for product the same factorisation (a * (-d) == -(a * d)).
// The rows live in the first 16-lane row of the wave (NU <= 16) and every broadcast is a DPP row
// broadcast; the other three rows of lanes compute on copies of row NU-1 (never read: nd is taken
// from lane 0, L from lanes < NU, the verdict from lanes k..NU-1).
template <int NU>
__device__ __forceinline__ int wave_ldl_fast_neg(const double *M, int lane, double (&a)[NU],
                                                 double (&nd)[NU]) {
  static_assert(NU <= 16, "lane = row inside one 16-lane DPP row");
  const double alpha = (1.0 + 4.123105625617661) / 8.0;
  const int row = lane < NU ? lane : NU - 1;
#pragma unroll
  for (int j = 0; j < NU; ++j)
    a[j] = M[j * NU + row];
  unsigned long long bad = 0ull;
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const double akk = row_bcast(a[k], k, lane);
    const unsigned long long nok = __ballot(!(fabs(akk) >= alpha * fabs(a[k])) || akk == 0.0);
    const unsigned long long from_k = ((1ull << NU) - 1ull) & ~((1ull << k) - 1ull);
    bad |= nok & from_k;
    const double nd_k = -fast_rcp(akk);
    const double nlik = a[k] * nd_k; // -L(i,k)
#pragma unroll
    for (int j = k + 1; j < NU; ++j)
      a[j] = __builtin_fma(row_bcast(nlik, j, lane), a[k], a[j]); // a(i,j) -= L(j,k) a(i,k)
    a[k] = nlik;
    nd[k] = nd_k;
  }
  return bad != 0ull;
}

// X <- -(L D L^T)^{-1} X on the MFMA layout: X[p] holds rows 4p+lk of one 16-column tile.
// An[p][q] / At[p][q] (p >= q): A operands of -L_pq and of its transpose (diagonal blocks:
// strictly lower / strictly upper part only); ndv[p] = -1/d[4p+lk].
template <int KU>
__device__ __forceinline__ void ldl_solve_mfma4(const double (&An)[KU][KU], const double (&At)[KU][KU],
                                                const double (&ndv)[KU], double (&X)[KU]) {
#pragma unroll
  for (int p = 0; p < KU; ++p) {
#pragma unroll
    for (int q = 0; q < p; ++q)
      X[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(An[p][q], X[q], X[p], 0, 0, 0);
    const double x0 = X[p];
    double y = x0;
#pragma unroll
    for (int it = 0; it < 3; ++it)
      y = __builtin_amdgcn_mfma_f64_4x4x4f64(An[p][p], y, x0, 0, 0, 0);
    X[p] = y;
  }
#pragma unroll
  for (int p = 0; p < KU; ++p)
    X[p] *= ndv[p];
#pragma unroll
  for (int p = KU - 1; p >= 0; --p) {
#pragma unroll
    for (int q = KU - 1; q > p; --q)
      X[p] = __builtin_amdgcn_mfma_f64_4x4x4f64(At[q][p], X[q], X[p], 0, 0, 0);
    const double x0 = X[p];
    double y = x0;
#pragma unroll
    for (int it = 0; it < 3; ++it)
      y = __builtin_amdgcn_mfma_f64_4x4x4f64(At[p][p], y, x0, 0, 0, 0);
    X[p] = y;
  }
}

template <int NX, int NU>
__device__ __forceinline__ void wave_stage2(const MfmaParams &P, double *sm, const double *prob,
                                            double *fac, int t, int lane,
                                            const WaveLane<NX, NU, 0> &L, WaveStage<NX, NU> &S,
                                            int &failed, const bool tracing) {
  using C = WaveCfg<NX, NU, 0>;
  using M = MfmaCfg<NX, NU, 0>;
  constexpr int NK = C::NK, NR = C::NR;
  constexpr int NW = C::NW, PK = C::PK, PG = C::PG, TX = C::TX, TW = C::TW, KS = C::KS, KU = C::KU;
  static_assert(NW <= 64, "the vector recursions keep [qhat; rhat] one entry per lane");
  const int li = lane & 15, lk = lane >> 4;
  double *V = sm + C::oV, *G = sm + C::oG, *Mm = sm + C::oM, *vn = sm + C::oVn;
  double *Lr = sm + C::oLr, *ndi = sm + C::oDi;
  constexpr int oVxx = M::fVxx, ovx = M::fvx;
#ifdef GAR_DIAG_SAMEREC // timing diagnostic (wrong results): every stage reads / writes record 0
  double *out = fac;
  const double *rec = prob + P.in_off0;
  const double *recn = rec;
#else
  double *out = fac + (long long)t * P.fac_rec;
  const double *rec = prob + P.in_off0 + (long long)t * P.in_rec;
  const double *recn = rec - (t > 0 ? P.in_rec : 0); // knot t-1 (t = 0: harmless re-read)
#endif
// cycle stamps of scripts/trace_wave2.py: only in the debug build (make trace), where every mark also
// pins the schedule (sched_barrier) so that a phase's instructions stay inside its stamps
#ifdef GAR_TRACE
#define GAR_WMARK(id)                                                          \
  __builtin_amdgcn_sched_barrier(0);                                           \
  if (tracing && t == (P.horizon >> 1))                                        \
    P.trace[(id)] = (long long)clock64();                                      \
  __builtin_amdgcn_sched_barrier(0);
#else
#define GAR_WMARK(id)
#endif
  GAR_WMARK(0)
  // ---- operands of the vector recursion: vx'[4s+lk] (LDS), f[4s+lk] (this knot: L2 hit) ------
  double vxs[KS], fs[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s)
    vxs[s] = vn[4 * s + lk];
  {
    const unsigned lkb = 8u * (unsigned)lk;
#pragma unroll
    for (int s = 0; s < KS; ++s)
      fs[s] = ldg_b(rec, M::kf + 4 * s, lkb);
  }
  // ---- P = V' F, H = W + F^T P (:216-228), tile columns from the control columns down ---------
  constexpr int TXF = C::REM4 ? TX - 1 : TX;
  const int i4 = lane & 3, k4 = lane >> 4;
  constexpr int cR = NX >> 4; // first tile column holding control columns: Rhat needs tj >= cR
  double part[TW];            // (F^T vx' + P^T f)[16 tj + li], summed over this lane's rows
  double a_row[NU], nd[NU];
  int verdict = 0;
#pragma unroll
  for (int tj = TW - 1; tj >= 0; --tj) {
    double4_t Pt[TX];
    double p4 = 0.0;
#pragma unroll
    for (int tm = 0; tm < TX; ++tm)
      Pt[tm] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const double bq = S.fo(tj, s);
#pragma unroll
      for (int tm = 0; tm < TXF; ++tm) {
        const int ic = (16 * tm + li) < NX ? (16 * tm + li) : NX - 1;
        const double aq = V[ic * PK + 4 * s + lk];
        Pt[tm] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, bq, Pt[tm], 0, 0, 0);
      }
      if (C::REM4)
        p4 = __builtin_amdgcn_mfma_f64_4x4x4f64(V[(NX - 4 + i4) * PK + 4 * s + k4], bq, p4, 0, 0, 0);
    }
    {
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const double pq = (C::REM4 && (s >> 2) == TX - 1) ? p4 : Pt[s >> 2][s & 3];
        a0 = __builtin_fma(S.fo(tj, s), vxs[s], a0);
        a1 = __builtin_fma(pq, fs[s], a1);
      }
      part[tj] = a0 + a1;
    }
#pragma unroll
    for (int ti = tj; ti < TW; ++ti) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const double pq = (C::REM4 && (s >> 2) == TX - 1) ? p4 : Pt[s >> 2][s & 3];
        S.Hc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(S.fo(ti, s), pq, S.Hc[ti][tj], 0, 0, 0);
      }
    }
    if (tj == cR) {
      GAR_WMARK(1)
      // ---- Rhat (lower) is complete: LDS -> lane = row -> register LDL^T under the first
      // Bunch-Kaufman test, issued while the remaining tile columns run on the matrix pipe
#pragma unroll
      for (int ti = cR; ti < TW; ++ti)
#pragma unroll
        for (int tc = cR; tc <= ti; ++tc)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * ti + lk + 4 * r, c = 16 * tc + li;
            if (16 * ti + 4 * r >= NX && 16 * ti + 4 * r < NW) { // compile-time: control rows
              if (c >= NX && c <= row)
                Mm[(c - NX) * NK + (row - NX)] = S.Hc[ti][tc][r];
            }
          }
      wave_lds_order();
      verdict = wave_ldl_fast_neg<NU>(Mm, lane, a_row, nd);
      if (lane < NU) {
#pragma unroll
        for (int j = 0; j < NU; ++j)
          Lr[lane * NU + j] = a_row[j]; // -L row-major (entries j >= i: not L, masked at the reads)
      }
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NU; ++j)
          ndi[j] = nd[j];
      }
      GAR_WMARK(2)
    }
  }
  GAR_WMARK(3)
  // ---- [qhat; rhat] = [q; r] + F^T vx' + P^T f (:217-218, :227-228) ---------------------------
  double hq;
  const double fi = S.fi;
  {
#pragma unroll
    for (int tt = 0; tt < TW; ++tt) {
      double a = part[tt];
      a += __shfl_xor(a, 16);
      a += __shfl_xor(a, 32);
      part[tt] = a; // column 16 tt + li, replicated over lk
    }
    double sel = part[0];
#pragma unroll
    for (int tt = 1; tt < TW; ++tt)
      sel = (lk == tt) ? part[tt] : sel;
    hq = S.qri + sel;
    if (lane >= NX && lane < NW)
      G[(lane - NX) * PG] = hq; // rhat
  }
  // B of this knot as the A operand of Aff = A + B K
  double Bop[TX][KU];
  double Bop4[KU];
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
  wave_lds_order();
  GAR_WMARK(4)
  // ---- [kff | K] = -Rhat^{-1} [rhat | Shat^T] (:248-262) ---------------------------------------
  double Kb[TX][KU]; // K[4s'+lk][16tj+li]: the B operand of Aff and Vxx
  constexpr bool SPARE = (NX % 16) != 0; // a free lane column in the last state tile for rhat
  constexpr int lc = NX % 16;
  if (verdict == 0) {
    double An[KU][KU], At[KU][KU], ndv[KU], rh[KU];
    {
      const int i3 = li & 3;
#pragma unroll
      for (int p = 0; p < KU; ++p)
#pragma unroll
        for (int q = 0; q <= p; ++q) {
          const double vn_ = Lr[(4 * p + i3) * NU + 4 * q + lk];
          const double vt_ = Lr[(4 * p + lk) * NU + 4 * q + i3];
          An[p][q] = (p == q && !(i3 > lk)) ? 0.0 : vn_;
          At[p][q] = (p == q && !(lk > i3)) ? 0.0 : vt_;
        }
#pragma unroll
      for (int p = 0; p < KU; ++p) {
        ndv[p] = ndi[4 * p + lk];
        rh[p] = G[(4 * p + lk) * PG];
      }
    }
#pragma unroll
    for (int tj = 0; tj < TX; ++tj) {
#pragma unroll
      for (int sp = 0; sp < KU; ++sp) {
        const double sv = S.Hc[C::shTile(sp)][tj][C::shReg(sp)]; // Shat^T(4sp+lk, 16tj+li)
        Kb[tj][sp] = (SPARE && tj == TX - 1 && li == lc) ? rh[sp] : sv;
      }
    }
    GAR_WMARK(5)
    double kfx[KU]; // !SPARE: kff[4sp+lk] in every lane
#pragma unroll
    for (int sp = 0; sp < KU; ++sp)
      kfx[sp] = rh[sp];
#pragma unroll
    for (int tj = 0; tj < TX; ++tj)
      ldl_solve_mfma4<KU>(An, At, ndv, Kb[tj]);
    if (!SPARE)
      ldl_solve_mfma4<KU>(An, At, ndv, kfx);
#pragma unroll
    for (int sp = 0; sp < KU; ++sp) {
      if (SPARE) {
        if (li == lc)
          G[(4 * sp + lk) * PG] = Kb[TX - 1][sp];
      } else {
        if (li == 0)
          G[(4 * sp + lk) * PG] = kfx[sp];
      }
    }
  } else {
    // the first Bunch-Kaufman test failed somewhere: Shat^T joins rhat in LDS and the stage goes
    // through the complete rule / the generic device Bunch-Kaufman (wave_slow_factor_solve)
#pragma unroll
    for (int ti = 0; ti < TW; ++ti)
#pragma unroll
      for (int tj = 0; tj <= ti; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * ti + lk + 4 * r, c = 16 * tj + li;
          if (16 * ti + 4 * r >= NX && 16 * ti + 4 * r < NW) {
            if (c < NX)
              G[(row - NX) * PG + 1 + c] = S.Hc[ti][tj][r];
          }
        }
    wave_sync();
    failed |= wave_slow_factor_solve<NX, NU, false>(sm, lane, P.slow);
#pragma unroll
    for (int tj = 0; tj < TX; ++tj) {
      const int cc = (16 * tj + li) < NX ? (16 * tj + li) : NX - 1;
#pragma unroll
      for (int s = 0; s < KU; ++s)
        Kb[tj][s] = G[(4 * s + lk) * PG + 1 + cc];
    }
  }
  wave_lds_order();
  GAR_WMARK(6)
  // ---- K -> fb rows 0..NU-1 (fbT2) ---------------------------------------------------------------
#pragma unroll
  for (int tj = 0; tj < TX; ++tj)
#pragma unroll
    for (int s = 0; s < KU; ++s)
      if (16 * tj + 15 < NX || 16 * tj + li < NX)
        stg_b(out, M::fFB + 8 * tj * 2 * NR + 8 * s, L.fbl, Kb[tj][s]);
  GAR_WMARK(11)
  // ---- kff; yff = f + B kff (:266); vx = qhat + Shat kff (:275-276) ----------------
  {
    double kf[KU]; // kff[4s'+lk]
#pragma unroll
    for (int s = 0; s < KU; ++s)
      kf[s] = G[(4 * s + lk) * PG];
    double py[TX], pv[TX];
#pragma unroll
    for (int ti = 0; ti < TX; ++ti) {
      double a = 0.0, c = 0.0;
#pragma unroll
      for (int s = 0; s < KU; ++s) {
        a = __builtin_fma(Bop[ti][s], kf[s], a);
        c = __builtin_fma(S.Hc[C::shTile(s)][ti][C::shReg(s)], kf[s], c); // Shat(16ti+li, 4s+lk)
      }
      a += __shfl_xor(a, 16);
      c += __shfl_xor(c, 16);
      a += __shfl_xor(a, 32);
      c += __shfl_xor(c, 32);
      py[ti] = a;
      pv[ti] = c;
    }
    double sy = py[0], sv = pv[0];
#pragma unroll
    for (int ti = 1; ti < TX; ++ti) {
      sy = (lk == ti) ? py[ti] : sy;
      sv = (lk == ti) ? pv[ti] : sv;
    }
    const double yf = fi + sy, vxv = hq + sv;
    if (lane < NU)
      out[M::fFF + lane] = G[lane * PG];
    if (lane < NX) {
      out[M::fFF + NK + lane] = yf;
      out[ovx + lane] = vxv;
      vn[lane] = vxv;
    }
  }
  GAR_WMARK(7)
  // ---- Aff = A + B K (:267), in place on the F operand registers ------------------
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
  GAR_WMARK(12)
#pragma unroll
  for (int tj = 0; tj < TX; ++tj)
#pragma unroll
    for (int ti = 0; ti < TX; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * ti + lk + 4 * r, j = 16 * tj + li;
        if (16 * ti + 4 * r < NX) { // compile-time
          if (i < NX && j < NX)     // fbT2(NU+i, j), i = 16ti+4r+lk
            stg_b(out, M::fFB + 8 * tj * 2 * NR + 2 * (NK + 16 * ti + 4 * r), L.fbl,
                  ti < C::KSF ? S.Fo[tj][ti][r] : (C::REM4 ? S.FoT[tj][0] : accT[tj][r]));
        }
      }
  // ---- knot t-1: the F operands and vectors go into the registers Aff just released
  wave_load_a<NX, NU>(recn, L, S);
  GAR_WMARK(8)
  // ---- Vxx = Qhat + Shat K (:272-273), lower tiles, mirrored into LDS --------------
  constexpr int shLo = C::shTile(0);
  double4_t accS[TX][TX];
  double acc4[TX];
  double sh4[KU];
#pragma unroll
  for (int tj = 0; tj < TX; ++tj)
#pragma unroll
    for (int ti = tj; ti < TX; ++ti)
      if (ti >= shLo) {
        if (C::REM4 && ti == TX - 1)
          acc4[tj] = S.Hc[ti][tj][0];
        else
          accS[ti][tj] = S.Hc[ti][tj];
      }
  if (C::REM4) {
#pragma unroll
    for (int s = 0; s < KU; ++s)
      sh4[s] = __shfl(S.Hc[C::shTile(s)][TX - 1][C::shReg(s)], (lane & 48) | ((NX - 4) & 15) | (lane & 3));
  }
#pragma unroll
  for (int s = 0; s < KU; ++s)
#pragma unroll
    for (int tj = 0; tj < TX; ++tj)
#pragma unroll
      for (int ti = tj; ti < TX; ++ti) {
        const double aq = S.Hc[C::shTile(s)][ti][C::shReg(s)];
        if (C::REM4 && ti == TX - 1)
          acc4[tj] = __builtin_amdgcn_mfma_f64_4x4x4f64(sh4[s], Kb[tj][s], acc4[tj], 0, 0, 0);
        else if (ti >= shLo)
          accS[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, Kb[tj][s], accS[ti][tj], 0, 0, 0);
        else
          S.Hc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, Kb[tj][s], S.Hc[ti][tj], 0, 0, 0);
      }
  GAR_WMARK(13)
#pragma unroll
  for (int tj = 0; tj < TX; ++tj)
#pragma unroll
    for (int ti = tj; ti < TX; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * ti + lk + 4 * r, c = 16 * tj + li;
        if (16 * ti + 4 * r < NX) { // compile-time
          const bool ok = (i < NX && c < NX && i >= c);
          const double v = (C::REM4 && ti == TX - 1) ? acc4[tj]
                                                     : (ti >= shLo ? accS[ti][tj][r] : S.Hc[ti][tj][r]);
          if (ti > tj && 16 * ti + 4 * r + 3 < NX && 16 * tj + 15 < NX) { // compile-time: all lanes valid
            V[i * PK + c] = v;
            V[c * PK + i] = v;
          } else {
            V[ok ? i * PK + c : C::oDump] = v;
            V[ok ? c * PK + i : C::oDump + 1] = v;
          }
        }
      }
  wave_sync();
  GAR_WMARK(14)
  // ---- knot t-1: its Hessian tiles replace H
  wave_load_b<NX, NU>(recn, L, S);
  GAR_WMARK(9)
  // ---- Vxx -> HBM (column-major, symmetric), 16 B per lane -----------------------
  {
    constexpr int NCH = (NX * NX / 2 + 63) / 64;
    double2_t vbuf[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int e = 64 * q + lane;
      const int ec = (64 * q + 63 < NX * NX / 2) ? e : (e < NX * NX / 2 ? e : NX * NX / 2 - 1);
      vbuf[q] = *reinterpret_cast<const double2_t *>(&V[2 * ec]);
    }
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int e = 64 * q + lane;
      if (64 * q + 63 < NX * NX / 2 || e < NX * NX / 2)
        *reinterpret_cast<double2_t *>(&out[oVxx + 2 * e]) = vbuf[q];
    }
  }
  GAR_WMARK(10)
#undef GAR_WMARK
}

} // namespace gar
