// gar_wave_pair.hpp -- TWO WAVES PER PROBLEM: the backward sweep for the wide shapes (nx + nu > 64;
// the Talos-walk LQ shape nx = 56, nu = 22 -> 24, bench/talos-walk.cpp:20-28, bench/lqr.cpp:25-26).
//
// Same arithmetic as ProximalRiccatiKernel::stageKernelSolve (gar/riccati-kernel.hxx:209-277) and the
// same building blocks as the one-wave stage (gar_wave2.hpp: F in MFMA operand layout, P = V'F feeding
// H = W + F^T P from its D registers, [qhat; rhat] = [q; r] + F^T vx' + P^T f, register LDL^T of Rhat
// under the complete Bunch-Kaufman rule, triangular solves on v_mfma_f64_4x4x4, Aff in place on the F
// registers).  What a single wave cannot do at this size is HOLD the stage: F is 70 doubles per lane,
// the 15 lower tiles of H another 60, and the register file's split into 256 VALU-addressable and 256
// accumulator registers overflows (wave<56,24>: 1.9 KB of scratch per lane, 4.7x the useful HBM
// traffic in spills, 230 k cycles per stage for 46 k of MFMA).  So the tile COLUMNS are split:
//   wave 0: tile columns 0 .. SPLIT-1      wave 1: tile columns SPLIT .. TW-1   (SPLIT = TW/2)
// For (56, 24): wave 0 holds all of F and 9 tiles of H (238 MFMAs in the first half), wave 1 holds F's
// tile columns 2..4 and 6 tiles (252 MFMAs) -- and Rhat, whose tiles all lie in wave 1's columns, so
// wave 1 factorises.  Every later step is column-local: a wave solves K = -Rhat^{-1} Shat^T for ITS
// state columns (the solve is column-independent; rhat rides in wave 1's spare lane column), forms Aff
// and the Vxx tiles of its columns, stores its part of the record.  What crosses between the waves
// goes through LDS: V' (shared operand), L and 1/d, Shat^T (the A operand of the other wave's Vxx
// tiles), [qhat; rhat], kff, vx.  Four workgroup barriers per stage.
// fb is written ROW-major (the generic record layout): the initial stage and the forward sweep of
// these shapes are the generic kernels.
#pragma once
#include "gar_wave2.hpp"
#include "gar_ldl_blocked.hpp"

namespace gar {

// Wide F loads (round 6).  The contraction index of P = V'F and H = W + F^T P -- the NEXT state's index, the rows of
// F = [A | B] -- is free to be relabelled: MFMA k-step s, lane group lk may stand for ANY row as long as both operands
// of a product agree.  With row(s, lk) = KS lk + s instead of 4 s + lk a lane's KS operand values of one tile column
// are KS CONTIGUOUS doubles of a column of the column-major knot block: KS / 2 16-byte loads (global_load_dwordx4)
// instead of KS 8-byte ones -- 35 instead of 70 load instructions for wave 0 at (56, 24), 21 instead of 42 for wave 1,
// on a wave whose end-of-stage burst of ~100 loads behind ~40 record stores is what it waits for (gfx9: ONE in-order
// counter of 63 outstanding memory instructions).  Everything indexed by the next state follows the same relabelling
// pos -> row: pos = 4 s + lk (the position the hardware's C/D layout gives the accumulators) holds row
// pi(pos) = KS (pos & 3) + (pos >> 2): the rows of V' read as the A operand of P (so P comes out with its rows in
// position order and feeds H unchanged), vx' and f in the vector part, the rows of B as the A operand of Aff, the
// rows of Aff / yff on their way out.  Results: the same sums in the same order per entry (the k-steps visit the
// rows in another order: last-digit differences against the 8-byte build, none against the record layout).
// MEASURED AND NOT ADOPTED (round 6; same box, alternating launches, 1 024 distinct device-generated problems, N = 275;
// profiles/r06_ab_pair_wide_f_loads_and_load_order_not_kept.log): backward 14.45 ms with the 8-byte loads, 14.72 ms
// with the 16-byte ones, 14.70 ms with those issued in consumption order (GAR_PAIR_ORDER) -- solutions equal to
// 5e-15.  The number of load instructions is not what the production schedule waits for (the tracing build, whose
// marks pin the schedule, shows the burst shrink from 23.7 k to 7.9 k cycles on wave 1: the compiler's own placement
// of the loads had hidden it already).  Both stay available: make variant NAME=pairwide DEFS="-DGAR_PAIR_WIDE_F=1".
#ifndef GAR_PAIR_WIDE_F
#define GAR_PAIR_WIDE_F 0
#endif
#ifndef GAR_PAIR_ORDER
#define GAR_PAIR_ORDER 0
#endif
// Packed records for the SERIAL wide family (round 6; what the headline family has had since rounds 3 / 4): the knot
// records keep Q and R as packed lower triangles (gar_layout.h: gar_lower_index; 1 540 + 276 of 9 672 doubles less read
// per knot at (56, 24) -- the upper triangles never reach a result), the factor records the lower triangle of Vxx,
// rectangular packed (gar_sym_index: 1 596 instead of 3 136 doubles written per stage and read by the roll-out,
// gar_forward_wide<.., true>).  The segment-leg kernels (gar_leg_seg.hpp) keep full blocks: PKD = false there.
#ifndef GAR_PAIR_PACKED
#define GAR_PAIR_PACKED 1
#endif
// The 24 x 24 register L D L^T of Rhat (276 broadcast-FMA pairs through v_readlane: 10.6 - 11.8 k cycles of the stage
// with the other wave idle) as TWO 12-column panels on DPP broadcasts + one MFMA trailing update
// (gar_ldl_blocked.hpp).  Shapes whose Rhat fits one DPP row (NU <= 16) keep the register version.
// MEASURED AND NOT ADOPTED (profiles/r06_ab_pair_packed_records_and_blocked_ldl.log, r06_trace_pair_packed_blocked_build.log):
// backward 14.03 ms with it, 14.03 without; the traced factorisation phase 12.3 k cycles against 11.8 k -- a 12-column
// DPP panel costs ~6 k cycles by itself (the pivot-to-pivot chain: broadcast, test, reciprocal, scale), two of them
// plus the LDS round trips of the trailing update are what the 276 v_readlane pairs were.
#ifndef GAR_PAIR_BLOCKED_LDL
#define GAR_PAIR_BLOCKED_LDL 0
#endif
// (measured and NOT adopted here: 402 instead of 501 registers, a third fewer accumulator-register copies -- and the
// backward sweep 4 % SLOWER, 14.60 against 14.04 ms; the same on wave<36,12,32> with D = 0 (+5.5 %) and on the headline
// sweep (+5 %): profiles/r06_ab5_*_lane_offsets_rederived_not_kept.log.  It pays where the parked values had gone to
// scratch -- the coupled stage -- and nowhere else.)
#ifndef GAR_PAIR_REFRESH_LANE
#define GAR_PAIR_REFRESH_LANE 1   // (by itself a loss, see above; the uneven first half below needs the room it makes)
#endif
// Uneven FIRST half (round 6): wave 1 keeps only the tile columns that hold Rhat ({3, 4} at (56, 24): 154 MFMAs) and
// factorises Rhat right behind them, while wave 0 computes the three state tile columns {0, 1, 2} (336 MFMAs): the two
// sides are balanced at ~22-23 k cycles and ONE workgroup barrier replaces two (the 24 x 24 factorisation, ~12 k cycles,
// used to run with wave 0 idle behind an even 238 / 252 split).  The second half keeps its even split by STATE columns
// ({0, 1} / {2, 3}): what wave 1 then needs of tile column 2 comes through LDS -- Shat^T from G, where it is exported
// anyway, the two Qhat tiles through a 4 KB hand-off buffer.
#ifndef GAR_PAIR_UNEVEN
#define GAR_PAIR_UNEVEN 1
#endif
#ifndef GAR_PAIR_EARLY_LOADS
#define GAR_PAIR_EARLY_LOADS 1
#endif
#ifndef GAR_WIDE_FWD_PIPELINED
#define GAR_WIDE_FWD_PIPELINED 1
#endif

template <int NX, int NU> struct PairCfg {
  using C = WaveCfg<NX, NU, 0>;
  static constexpr int SPLIT = C::TW / 2;
  // (the wide shapes only: their records carry fb row-major; an even number of k-steps keeps every piece 16-byte aligned)
  static constexpr bool PX = GAR_PAIR_WIDE_F && MfmaCfg<NX, NU, 0>::WIDE && (C::KS % 2 == 0) && (NX % 4 == 0);
  // next-state row that k-step s, lane group lk stands for / that accumulator position pos holds
  __host__ __device__ static constexpr int krow(int s, int lk) { return PX ? C::KS * lk + s : 4 * s + lk; }
  __host__ __device__ static constexpr int prow(int pos) { return PX ? C::KS * (pos & 3) + (pos >> 2) : pos; }
  static constexpr int oHq = (C::total + 1) & ~1;          // [qhat; rhat], one entry per index
  static constexpr int oFlag2 = oHq + ((C::NW + 1) & ~1);  // verdict of the factorisation (int)
  static constexpr bool BLK = GAR_PAIR_BLOCKED_LDL && NU > 16 && NU % 4 == 0;
  static constexpr int oLdl = oFlag2 + 2;                  // blocked factorisation: working copy of Rhat | -d_k
  // first-half split (tile columns >= SPLIT1 belong to wave 1): uneven for the wide shape, where Rhat's tiles all lie in
  // the columns >= NX / 16
  static constexpr bool UNEVEN = GAR_PAIR_UNEVEN && MfmaCfg<NX, NU, 0>::WIDE && (NX >> 4) > SPLIT;
  static constexpr int SPLIT1 = UNEVEN ? (NX >> 4) : SPLIT;
  static constexpr bool EARLY = UNEVEN && (GAR_PAIR_EARLY_LOADS != 0) && !GAR_PAIR_ORDER;
  static constexpr int nXq = UNEVEN ? (SPLIT1 - SPLIT) * C::TX * 4 * 64 : 0; // hand-off of the Qhat tiles of the columns that change hands
  static constexpr int oXq = oLdl + (BLK ? LdlBlockedLds<NU>::total : 0);
  static constexpr int total = oXq + nXq;
  __host__ __device__ static constexpr int owner(int tj) { return tj >= SPLIT ? 1 : 0; }   // second half: state columns
  __host__ __device__ static constexpr int owner1(int tj) { return tj >= SPLIT1 ? 1 : 0; } // first half: tile columns
};

// knot t's operands for wave W: F tile columns it multiplies with (all of them for wave 0: H(ti, tj)
// needs F(:, ti) for every ti >= tj), the Hessian tiles of its own tile columns
template <int NX, int NU, class LANE>
__device__ __forceinline__ void pair_load_F(const double *rec, const LANE &L, WaveStage<NX, NU> &S, int t, int lane) {
  using C = WaveCfg<NX, NU>;
  using PC = PairCfg<NX, NU>;
  if constexpr (PC::PX) {
    // rows KS lk .. KS lk + KS - 1 of column 16 t + li: KS / 2 pieces of 16 bytes
    using M = MfmaCfg<NX, NU, 0>;
    const int li = lane & 15, lk = lane >> 4;
    const int col = (16 * t + li) < C::NW ? (16 * t + li) : C::NW - 1;
    const unsigned lb = 8u * (unsigned)(M::kA + (WaveLane<NX, NU>::fo_in(t) ? li : col) * NX + C::KS * lk);
    const double *base = rec + (WaveLane<NX, NU>::fo_in(t) ? 16 * t * NX : 0);
#pragma unroll
    for (int s = 0; s < C::KS; s += 2) {
      const double2_t v = *reinterpret_cast<const double2_t *>(reinterpret_cast<const char *>(base + s) + lb);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (s + e < 4 * C::KSF)
          S.Fo[t][(s + e) >> 2][(s + e) & 3] = v[e];
        else
          S.FoT[t][s + e - 4 * C::KSF] = v[e];
      }
    }
    return;
  }
#pragma unroll
  for (int s = 0; s < C::KS; ++s) {
    const double v = WaveLane<NX, NU>::fo_in(t) ? ldg_b(rec, 16 * t * NX + 4 * s, L.fo0) : ldg_b(rec, 4 * s, L.foX);
    if (s < 4 * C::KSF)
      S.Fo[t][s >> 2][s & 3] = v;
    else
      S.FoT[t][s - 4 * C::KSF] = v;
  }
}
template <int NX, int NU, bool QP, class LANE>
__device__ __forceinline__ void pair_load_H(const double *rec, const LANE &L, WaveStage<NX, NU> &S, int ti, int tj) {
  using C = WaveCfg<NX, NU>;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row0 = 16 * ti + 4 * r; // + lk
    if (row0 + 3 < NX)
      S.Hc[ti][tj][r] = QP ? ldg_b(rec, row0, L.hcq[tj])
                           : (WaveLane<NX, NU>::x_in(tj) ? ldg_b(rec, 16 * tj * NX + row0, L.hcx0) : ldg_b(rec, row0, L.hcxX[tj]));
    else if (row0 >= NX && row0 + 3 < C::NW)
      S.Hc[ti][tj][r] = WaveLane<NX, NU>::x_in(tj) ? ldg_b(rec, (row0 - NX) * NX + 16 * tj, L.hcu0)
                                                   : ldg_b(rec, 0, L.hcuX[tj][(row0 - NX) >> 2]);
    else
      S.Hc[ti][tj][r] = 0.0;
  }
}
// PART (uneven first half, GAR_PAIR_EARLY_LOADS): 0 = everything; 1 = only what this wave's registers can take right
// behind its FIRST half -- wave 0: F's tile columns >= SPLIT (its Aff needs the first SPLIT only) and the Hessian tiles of
// the state columns it computed for wave 1; wave 1: F's pure control column(s) and the Rhat-only tile(s) --; 2 = the rest.
// The workgroup barriers of the stage are fences the compiler does not move loads across: without the split every one of
// these requests waits for the end of the stage although its destination has been dead since the first half.
template <int NX, int NU, int W, bool QP = false, int PART = 0, class LANE>
__device__ __forceinline__ void pair_load(const double *rec, const LANE &L, WaveStage<NX, NU> &S, int lane) {
  using C = WaveCfg<NX, NU>;
  using PC = PairCfg<NX, NU>;
  [[maybe_unused]] auto early_F = [](int t) { return PC::UNEVEN && (W == 0 ? t >= PC::SPLIT : 16 * t >= NX); };
  [[maybe_unused]] auto early_H = [](int ti, int tj) {
    return PC::UNEVEN && (W == 0 ? PC::owner(tj) == 1 : (16 * tj >= NX && 16 * ti >= NX));
  };
#if GAR_PAIR_ORDER
  // in the order the next stage consumes them (loads return in order: its first products wait for the first pieces
  // only): tile columns from the last one down -- F(:, tj) feeds P(:, tj), then H(ti, tj), ti = tj .., wants F(:, ti)
  // and its Hessian tile
  bool have[C::TW] = {};
#pragma unroll
  for (int tj = C::TW - 1; tj >= 0; --tj) {
    if (PC::owner1(tj) != W)
      continue;
#pragma unroll
    for (int ti = tj; ti < C::TW; ++ti) {
      if (!have[ti]) {
        pair_load_F<NX, NU>(rec, L, S, ti, lane);
        have[ti] = true;
      }
      pair_load_H<NX, NU, QP>(rec, L, S, ti, tj);
    }
  }
#pragma unroll
  for (int t = 0; t < C::TW; ++t) // (F columns a wave needs for Aff only)
    if (!have[t] && !(W == 1 && t < PC::SPLIT))
      pair_load_F<NX, NU>(rec, L, S, t, lane);
#else
#pragma unroll
  for (int t = 0; t < C::TW; ++t)
    if (!(W == 1 && t < PC::SPLIT) && (PART == 0 || (PART == 1) == early_F(t)))
      pair_load_F<NX, NU>(rec, L, S, t, lane);
#pragma unroll
  for (int ti = 0; ti < C::TW; ++ti)
#pragma unroll
    for (int tj = 0; tj <= ti; ++tj)
      if (PC::owner1(tj) == W && (PART == 0 || (PART == 1) == early_H(ti, tj)))
        pair_load_H<NX, NU, QP>(rec, L, S, ti, tj);
#endif
}

template <int NX, int NU, int W, bool PKD = false>
__device__ __forceinline__ void pair_stage(const MfmaParams &P, double *sm, const double *prob, double *fac,
                                           int t, int lane, const WaveLane<NX, NU, 0> &L,
                                           WaveStage<NX, NU> &S, int &failed) {
  using C = WaveCfg<NX, NU, 0>;
  using M = MfmaCfg<NX, NU, 0>;
  using PC = PairCfg<NX, NU>;
  constexpr int NK = C::NK, NW = C::NW, PK = C::PK, PG = C::PG, TX = C::TX, TW = C::TW, KS = C::KS, KU = C::KU;
  constexpr int SPLIT = PC::SPLIT, cR = NX >> 4;
  // (Rhat and the spare lane column for rhat belong to wave 1; a 4-row remainder tile (NX % 16 == 4)
  // runs on the 16x16x4 instruction here)
  static_assert(cR >= SPLIT && (NX % 16) != 0 && SPLIT >= 1, "Rhat and the spare column belong to wave 1");
  constexpr bool WIDE = M::WIDE; // fb row-major (generic forward sweep) / fbT2 (gar_forward_mfma)
  constexpr int NR = C::NR;
  const int li = lane & 15, lk = lane >> 4;
  const unsigned fbrm = 8u * (unsigned)(lk * NX + li); // row-major fb: element (lk, li)
  const unsigned lib = 8u * (unsigned)li;
  double *V = sm + C::oV, *G = sm + C::oG, *Mm = sm + C::oM, *vn = sm + C::oVn;
  double *Lr = sm + C::oLr, *ndi = sm + C::oDi, *hqv = sm + PC::oHq;
  int *flag = reinterpret_cast<int *>(sm + PC::oFlag2);
  constexpr int oVxx = M::fVxx, ovx = M::fvx;
  double *out = fac + P.slot(t) * P.fac_rec;
  const double *rec = prob + P.in_off0 + P.slot(t) * P.in_rec;
  const double *recn = prob + P.in_off0 + P.slot(t > 0 ? t - 1 : 0) * P.in_rec;
#ifdef GAR_TRACE
#define GAR_PMARK(id)                                                          \
  __builtin_amdgcn_sched_barrier(0);                                           \
  if (P.trace != nullptr && blockIdx.x == 0 && lane == 0 && t == (P.horizon >> 1)) \
    P.trace[W * 16 + (id)] = (long long)clock64();                             \
  __builtin_amdgcn_sched_barrier(0);
#else
#define GAR_PMARK(id)
#endif
  GAR_PMARK(0)
  // ---- first half: P = V'F and H = W + F^T P for this wave's tile columns -----------------------
  double vxs[KS], fs[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s)
    vxs[s] = PC::PX ? vn[KS * lk + s] : vn[4 * s + lk];
  {
    const unsigned lkb = 8u * (unsigned)(PC::PX ? KS * lk : lk);
#pragma unroll
    for (int s = 0; s < KS; ++s)
      fs[s] = ldg_b(rec, M::kf + (PC::PX ? s : 4 * s), lkb);
  }
  double qrv[TW]; // [q; r] entries of this wave's columns: issued here, used after the products
#pragma unroll
  for (int tj = 0; tj < TW; ++tj)
    if (PC::owner1(tj) == W) {
      const int j = 16 * tj + li;
      qrv[tj] = ldg_b(rec, M::kq + (16 * tj + 15 < NW ? 16 * tj : 0), 16 * tj + 15 < NW ? lib : 8u * (unsigned)(j < NW ? j : NW - 1));
    }
  double part[TW];
#pragma unroll
  for (int tj = TW - 1; tj >= 0; --tj) {
    if (PC::owner1(tj) != W)
      continue;
    double4_t Pt[TX];
#pragma unroll
    for (int tm = 0; tm < TX; ++tm)
      Pt[tm] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const double bq = S.fo(tj, s);
#pragma unroll
      for (int tm = 0; tm < TX; ++tm) {
        // (PX: the row of V' whose product lands on accumulator position 16 tm + li, the column of k-step s)
        const int ic = (16 * tm + li) < NX ? PC::prow(16 * tm + li) : NX - 1;
        Pt[tm] = __builtin_amdgcn_mfma_f64_16x16x4f64(V[ic * PK + PC::krow(s, lk)], bq, Pt[tm], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ti = tj; ti < TW; ++ti)
#pragma unroll
      for (int s = 0; s < KS; ++s)
        S.Hc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(S.fo(ti, s), Pt[s >> 2][s & 3], S.Hc[ti][tj], 0, 0, 0);
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      a0 = __builtin_fma(S.fo(tj, s), vxs[s], a0);
      a1 = __builtin_fma(Pt[s >> 2][s & 3], fs[s], a1);
    }
    part[tj] = a0 + a1;
  }
  // B (the A operand of Aff) and f of this knot: in flight during the factorisation instead of at their
  // first use.  (Measured and NOT kept: spreading knot t-1's operand loads over the stage as registers
  // are released -- live ranges grow, 0.32 -> 0.26 of the roofline; warming L2 with one load per line
  // ahead of the burst at the end of the stage -- 0.33 -> 0.31.)
  double Bop[TX][KU]; // B[16ti+li][4s'+lk]: the A operand of Aff
#pragma unroll
  for (int ti = 0; ti < TX; ++ti)
#pragma unroll
    for (int s = 0; s < KU; ++s)
      if constexpr (PC::PX) { // B(pi(16 ti + li), 4 s + lk): the row whose Aff lands on accumulator position 16 ti + li
        const int p = 16 * ti + li, row = p < NX ? PC::prow(p) : NX - 1;
        Bop[ti][s] = ldg_b(rec, M::kB + 4 * s * NX, 8u * (unsigned)(lk * NX + row));
      } else {
        Bop[ti][s] = WaveLane<NX, NU>::x_in(ti) ? ldg_b(rec, 4 * s * NX + 16 * ti, L.bop0) : ldg_b(rec, 4 * s * NX, L.bopX);
      }
  double fyf[TX];
#pragma unroll
  for (int ti = 0; ti < TX; ++ti)
    if (PC::owner(ti) == W) {
      const int i = 16 * ti + li, ic = i < NX ? PC::prow(i) : NX - 1;
      fyf[ti] = ldg_b(rec, M::kf, 8u * (unsigned)ic);
    }
  GAR_PMARK(1)
  // ---- [qhat; rhat] entries of this wave's columns (:217-218, :227-228), Shat^T, Rhat -> LDS -------
#pragma unroll
  for (int tj = 0; tj < TW; ++tj) {
    if (PC::owner1(tj) != W)
      continue;
    const int j = 16 * tj + li;
    const double e = qrv[tj] + rows_sum(part[tj], lane);
    if (lk == 0 && j < NW) {
      hqv[j] = e;
      if (j >= NX)
        G[(j - NX) * PG] = e; // rhat: right-hand side of kff (:248), sign applied in the solve
    }
  }
#pragma unroll
  for (int ti = 0; ti < TW; ++ti)
#pragma unroll
    for (int tj = 0; tj <= ti; ++tj) {
      if (PC::owner1(tj) != W)
        continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + lk + 4 * r, c = 16 * tj + li;
        if (16 * ti + 4 * r >= NX && 16 * ti + 4 * r < NW) { // compile-time: control rows
          if (c < NX)
            G[(row - NX) * PG + 1 + c] = S.Hc[ti][tj][r]; // Shat^T(u, c)
          else if (c <= row)
            Mm[(c - NX) * NK + (row - NX)] = S.Hc[ti][tj][r]; // Rhat (lower), wave 1 only
        }
      }
    }
  double *Xq = sm + PC::oXq;
  if constexpr (PC::UNEVEN) { // the Qhat tiles of the state columns that change hands (computed here, consumed by the other wave)
#pragma unroll
    for (int tj = 0; tj < TX; ++tj) {
      if (!(PC::owner1(tj) == W && PC::owner(tj) != W))
        continue;
#pragma unroll
      for (int ti = tj; ti < TX; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Xq[(((tj - PC::SPLIT) * TX + ti) * 4 + r) * 64 + lane] = S.Hc[ti][tj][r];
    }
  }
  GAR_PMARK(2)
  if constexpr (!PC::UNEVEN)
    __syncthreads(); // (1) [qhat; rhat], Shat^T, Rhat in LDS
  else
    wave_lds_order(); // (uneven split: Rhat is this wave's own export; everything else is read behind barrier (2))
  if constexpr (PC::EARLY)
    pair_load<NX, NU, W, PKD, 1>(recn, L, S, lane); // knot t-1 into the registers the first half released
  GAR_PMARK(3)
  // ---- wave 1: register LDL^T of Rhat under the complete Bunch-Kaufman rule; -L, -1/d -> LDS -------
  if (W == 1) {
    bool first_failed = false;
    int verdict = 2;
    if constexpr (PC::BLK) { // two DPP panels + an MFMA trailing update on a working copy (Mm stays pristine for the
      // complete rule / the device Bunch-Kaufman below)
      double *Wc = sm + PC::oLdl + LdlBlockedLds<NU>::oW, *npv = sm + PC::oLdl + LdlBlockedLds<NU>::oNp;
      for (int e = lane; e < NU * NU; e += 64)
        Wc[e] = Mm[e];
      wave_lds_order();
      verdict = wave_ldl_blocked<NU, 12, false>(Wc, npv, Lr, ndi, lane, first_failed, P.spd_accept != 0, Wc[lane]);
      wave_lds_order();
    }
    if (verdict == 2) { // (the register version: every shape with NU <= 16; a column that needs the complete rule)
      double a_row[NU], nd[NU];
      const int frow = lane < NU ? lane : NU - 1;
#pragma unroll
      for (int j = 0; j < NU; ++j)
        a_row[j] = Mm[j * NU + frow];
      verdict = wave_ldl_fast_neg_pre<NU>(lane, a_row, nd, first_failed, nullptr, P.spd_accept != 0);
      if (lane < NU) {
#pragma unroll
        for (int j = 0; j < NU; ++j)
          Lr[lane * NU + j] = a_row[j];
      }
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NU; ++j)
          ndi[j] = nd[j];
      }
    }
    if (lane == 0) {
      flag[0] = verdict;
      if (first_failed) {
        atomicAdd(&P.slow[0], 1);
        if (verdict != 0)
          atomicAdd(&P.slow[1], 1);
      }
    }
    if (!PC::UNEVEN && verdict != 0) {
      // Bunch-Kaufman interchanges or takes a 2x2 pivot here: the generic device Bunch-Kaufman, exactly
      // what the reference does, by this wave alone; [kff | K] goes to the V buffer (V' is dead until
      // this stage's Vxx is written), Shat^T stays in G for the Vxx products
      wave_sync();
      double *Kv = V;
      for (int e = lane; e < NU * PG; e += 64)
        Kv[e] = -G[e];
      double *sub = sm + C::oBk;
      int *piv = (int *)(sub + C::BKS);
      const WG w1 = wave_self();
      wave_sync();
      failed |= wg_bk_factor(w1, NU, Mm, NU, sub, piv, piv + C::BKS);
      wg_bk_solve(w1, NU, Mm, NU, sub, piv, Kv, PG, 1, NX + 1);
      wave_sync();
    }
  }
  GAR_PMARK(4)
  __syncthreads(); // (2) the factorisation (or [kff | K] itself) is in LDS  [uneven split: AND everybody's exports]
  GAR_PMARK(5)
  const int verdict = flag[0];
  if constexpr (PC::UNEVEN) {
    if (verdict != 0) { // (wave-uniform over the workgroup) the device Bunch-Kaufman needs wave 0's Shat^T in G: behind the barrier
      if (W == 1) {
        double *Kv = V;
        for (int e = lane; e < NU * PG; e += 64)
          Kv[e] = -G[e];
        double *sub = sm + C::oBk;
        int *piv = (int *)(sub + C::BKS);
        const WG w1 = wave_self();
        wave_sync();
        failed |= wg_bk_factor(w1, NU, Mm, NU, sub, piv, piv + C::BKS);
        wg_bk_solve(w1, NU, Mm, NU, sub, piv, Kv, PG, 1, NX + 1);
        wave_sync();
      }
      __syncthreads();
    }
  }
  // ---- [kff | K] = -Rhat^{-1} [rhat | Shat^T] (:248-262) for this wave's state columns ------------
  double Kb[TX][KU];
  constexpr int lc = NX % 16;
  if (verdict == 0) {
    double An[KU][KU], At[KU][KU], ndv[KU];
    const int i3 = li & 3;
#pragma unroll
    for (int p = 0; p < KU; ++p) {
#pragma unroll
      for (int q = 0; q <= p; ++q) {
        const double vn_ = Lr[(4 * p + i3) * NU + 4 * q + lk];
        const double vt_ = Lr[(4 * p + lk) * NU + 4 * q + i3];
        An[p][q] = (p == q && !(i3 > lk)) ? 0.0 : vn_;
        At[p][q] = (p == q && !(lk > i3)) ? 0.0 : vt_;
      }
      ndv[p] = ndi[4 * p + lk];
    }
#pragma unroll
    for (int tj = 0; tj < TX; ++tj) {
      if (PC::owner(tj) != W)
        continue;
#pragma unroll
      for (int sp = 0; sp < KU; ++sp) {
        // Shat^T(4sp+lk, 16tj+li): this wave's own tile registers, or (a column the other wave computed) its export in G
        const int cs = (16 * tj + li) < NX ? (16 * tj + li) : NX - 1;
        const double sv = PC::owner1(tj) == W ? S.Hc[C::shTile(sp)][tj][C::shReg(sp)] : G[(4 * sp + lk) * PG + 1 + cs];
        const double g0 = tj == TX - 1 ? G[(4 * sp + lk) * PG] : 0.0; // (read by every lane, then selected: a
        Kb[tj][sp] = (tj == TX - 1 && li == lc) ? g0 : sv;            //  conditional load becomes a branch)
      }
      ldl_solve_mfma4<KU>(An, At, ndv, Kb[tj]);
    }
    if (W == 1) {
#pragma unroll
      for (int sp = 0; sp < KU; ++sp)
        if (li == lc)
          G[(4 * sp + lk) * PG] = Kb[TX - 1][sp]; // kff
    }
  } else {
    const double *Kv = V;
#pragma unroll
    for (int tj = 0; tj < TX; ++tj) {
      if (PC::owner(tj) != W)
        continue;
      const int cc = (16 * tj + li) < NX ? (16 * tj + li) : NX - 1;
#pragma unroll
      for (int s = 0; s < KU; ++s)
        Kb[tj][s] = Kv[(4 * s + lk) * PG + 1 + cc];
    }
    if (W == 1 && lane < NU)
      G[lane * PG] = Kv[lane * PG]; // kff
  }
  GAR_PMARK(6)
  __syncthreads(); // (3) kff in G(:, 0); every read of [kff | K] from the V buffer is done
  GAR_PMARK(7)
  // ---- K -> fb rows 0..NU-1 (row-major); kff; yff = f + B kff (:266); vx = qhat + Shat kff (:275-276)
#pragma unroll
  for (int tj = 0; tj < TX; ++tj) {
    if (PC::owner(tj) != W)
      continue;
#pragma unroll
    for (int s = 0; s < KU; ++s)
      if (16 * tj + 15 < NX || 16 * tj + li < NX) {
        if (WIDE)
          stg_b(out, M::fFB + 4 * s * NX + 16 * tj, fbrm, Kb[tj][s]);
        else
          stg_b(out, M::fFB + 8 * tj * 2 * NR + 8 * s, L.fbl, Kb[tj][s]);
      }
  }
  {
    double kf[KU];
#pragma unroll
    for (int s = 0; s < KU; ++s)
      kf[s] = G[(4 * s + lk) * PG];
    if (W == 1 && lane < NU)
      out[M::fFF + lane] = G[lane * PG];
#pragma unroll
    for (int ti = 0; ti < TX; ++ti) {
      if (PC::owner(ti) != W)
        continue;
      const int i = 16 * ti + li, ic = i < NX ? i : NX - 1;
      double a = 0.0, c = 0.0;
#pragma unroll
      for (int s = 0; s < KU; ++s) {
        a = __builtin_fma(Bop[ti][s], kf[s], a);
        c = __builtin_fma(G[(4 * s + lk) * PG + 1 + ic], kf[s], c); // Shat(i, 4s+lk)
      }
      const double yf = fyf[ti] + rows_sum(a, lane);
      const double vxv = hqv[ic] + rows_sum(c, lane);
      if (lk == 0 && i < NX) {
        out[M::fFF + NK + PC::prow(i)] = yf; // (PX: position i holds next-state row pi(i); vx is indexed by THIS state)
        out[ovx + i] = vxv;
        vn[i] = vxv;
      }
    }
  }
  GAR_PMARK(8)
  // ---- Aff = A + B K (:267), in place on the F operand registers, this wave's state columns ---------
#pragma unroll
  for (int tj = 0; tj < TX; ++tj) {
    if (PC::owner(tj) != W)
      continue;
    double4_t accT = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r)
      accT[r] = (r < C::KST) ? S.FoT[tj][r < C::KST ? r : 0] : 0.0;
#pragma unroll
    for (int s = 0; s < KU; ++s)
#pragma unroll
      for (int ti = 0; ti < TX; ++ti) {
        if (ti < C::KSF)
          S.Fo[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bop[ti][s], Kb[tj][s], S.Fo[tj][ti], 0, 0, 0);
        else
          accT = __builtin_amdgcn_mfma_f64_16x16x4f64(Bop[ti][s], Kb[tj][s], accT, 0, 0, 0);
      }
#pragma unroll
    for (int ti = 0; ti < TX; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * ti + lk + 4 * r, j = 16 * tj + li;
        if (16 * ti + 4 * r < NX) {
          if (i < NX && j < NX) {
            const double v = ti < C::KSF ? S.Fo[tj][ti < C::KSF ? ti : 0][r] : accT[r];
            if (PC::PX) // position 16 ti + 4 r + lk holds row KS lk + 4 ti + r
              stg_b(out, M::fFB + (NK + 4 * ti + r) * NX + 16 * tj, 8u * (unsigned)(KS * lk * NX + li), v);
            else if (WIDE)
              stg_b(out, M::fFB + (NK + 16 * ti + 4 * r) * NX + 16 * tj, fbrm, v);
            else
              stg_b(out, M::fFB + 8 * tj * 2 * NR + 2 * (NK + 16 * ti + 4 * r), L.fbl, v);
          }
        }
      }
  }
  GAR_PMARK(9)
  // ---- Vxx = Qhat + Shat K (:272-273), lower tiles of this wave's columns, mirrored into LDS ----------
  // (V' was last read before barrier (1); Shat comes from G: the other wave's columns too)
#pragma unroll
  for (int tj = 0; tj < TX; ++tj) {
    if (PC::owner(tj) != W)
      continue;
#pragma unroll
    for (int ti = tj; ti < TX; ++ti) {
      double4_t acc;
      if constexpr (PC::UNEVEN) {
        if (PC::owner1(tj) == W) {
          acc = S.Hc[ti][tj];
        } else { // Qhat of a column the other wave computed: through the hand-off buffer
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc[r] = Xq[(((tj - PC::SPLIT) * TX + ti) * 4 + r) * 64 + lane];
        }
      } else {
        acc = S.Hc[ti][tj];
      }
      const int ic = (16 * ti + li) < NX ? (16 * ti + li) : NX - 1;
#pragma unroll
      for (int s = 0; s < KU; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(G[(4 * s + lk) * PG + 1 + ic], Kb[tj][s], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * ti + lk + 4 * r, c = 16 * tj + li;
        if (16 * ti + 4 * r < NX) {
          const bool ok = (i < NX && c < NX && i >= c);
          if (ti > tj && 16 * ti + 4 * r + 3 < NX && 16 * tj + 15 < NX) {
            V[i * PK + c] = acc[r];
            V[c * PK + i] = acc[r];
          } else {
            V[ok ? i * PK + c : C::oDump + W] = acc[r];
            V[ok ? c * PK + i : C::oDump + 2 + W] = acc[r];
          }
        }
      }
    }
  }
  GAR_PMARK(10)
  pair_load<NX, NU, W, PKD, PC::EARLY ? 2 : 0>(recn, L, S, lane); // knot t-1 into the registers this stage released
  GAR_PMARK(11)
  __syncthreads(); // (4) V, vx complete
  GAR_PMARK(12)
  // ---- Vxx -> HBM, 16 B per lane, the chunks alternate between the waves: column-major and symmetric for the wide
  // shapes, the packed lower triangle (gar_layout.h) where the roll-out is gar_forward_mfma ----
  {
    using VO = VxxOut<NX, PKD || (GAR_VXX_PACKED && !WIDE), PK>;
#pragma unroll
    for (int q = W; q < VO::NCH; q += 2)
      VO::write(out + oVxx, q, lane, VO::read(V, q, lane));
  }
}

// (the narrow shapes fit 256 registers per wave: two waves share a SIMD, i.e. four problems per CU, and
// one problem's waits hide behind the other's arithmetic; the wide ones need the whole register file)
template <int NX, int NU, bool PKD = false>
__global__ void __launch_bounds__(128, (NX + NU > 64) ? 1 : 2) gar_backward_pair(MfmaParams P, int batch) {
  using C = WaveCfg<NX, NU, 0>;
  using M = MfmaCfg<NX, NU, 0>;
  constexpr int PK = C::PK;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = (int)blockIdx.x;
  if (b >= batch)
    return;
  double *sm = gar_smem;
  const double *prob = P.prob + (long long)b * P.prob_stride;
  double *fac = P.fac + (long long)b * P.fac_stride;
  const int N = P.horizon;
  double *V = sm + C::oV, *vn = sm + C::oVn;
  WaveLane<NX, NU, 0> L;
  wave_lane_init<NX, NU, 0, PKD>(L, lane);
  const double *recN1 = prob + P.in_off0 + P.slot(N - 1) * P.in_rec;
  { // terminal knot (terminalSolve, nu = 0, :146-149, :175-178): Vxx = Q, vx = q (its record keeps the full Q)
    const double *rec = prob + P.in_offN;
    double *out = fac + P.fac_offN;
    for (int e = tid; e < NX * NX; e += 128) {
      const int j = e / NX, i = e - j * NX;
      const double v = (i >= j) ? rec[M::tQ + e] : rec[M::tQ + i * NX + j];
      V[i * PK + j] = v;
      if (!PKD && (M::WIDE || !GAR_VXX_PACKED))
        out[M::tVxx + e] = v;
      else if (i >= j)
        out[M::tVxx + gar_sym_index(1, NX, i, j)] = v; // (packed lower triangle: gar_layout.h)
    }
    if (tid < NX) {
      const double v = rec[M::tq + tid];
      vn[tid] = v;
      out[M::tvx + tid] = v;
    }
  }
  __syncthreads();
  int failed = 0;
  // one loop per wave: each keeps only ITS tiles in registers across the stages (a common loop would
  // carry the union of both waves' state through either path)
#if GAR_PAIR_REFRESH_LANE
  // (as in the coupled constrained stage, gar_wave.hpp GAR_COUPLED_REFRESH_LANE: the lane offsets re-derived per stage
  // from a lane index the compiler cannot prove loop-invariant, instead of ~100 values parked in accumulator registers
  // and copied back at every use)
  if (wave == 0) {
    WaveStage<NX, NU> S;
    pair_load<NX, NU, 0, PKD>(recN1, L, S, lane);
    for (int t = N - 1; t >= 0; --t) {
      const int lane_t = lane + fence0(S.Fo[0][0][0]);
      WaveLane<NX, NU, 0> Lt;
      wave_lane_init<NX, NU, 0, PKD>(Lt, lane_t);
      pair_stage<NX, NU, 0, PKD>(P, sm, prob, fac, t, lane_t, Lt, S, failed);
    }
  } else {
    WaveStage<NX, NU> S;
    pair_load<NX, NU, 1, PKD>(recN1, L, S, lane);
    for (int t = N - 1; t >= 0; --t) {
      const int lane_t = lane + fence0(S.Fo[PairCfg<NX, NU>::SPLIT][0][0]);
      WaveLane<NX, NU, 0> Lt;
      wave_lane_init<NX, NU, 0, PKD>(Lt, lane_t);
      pair_stage<NX, NU, 1, PKD>(P, sm, prob, fac, t, lane_t, Lt, S, failed);
    }
  }
#else
  if (wave == 0) {
    WaveStage<NX, NU> S;
    pair_load<NX, NU, 0, PKD>(recN1, L, S, lane);
    for (int t = N - 1; t >= 0; --t)
      pair_stage<NX, NU, 0, PKD>(P, sm, prob, fac, t, lane, L, S, failed);
  } else {
    WaveStage<NX, NU> S;
    pair_load<NX, NU, 1, PKD>(recN1, L, S, lane);
    for (int t = N - 1; t >= 0; --t)
      pair_stage<NX, NU, 1, PKD>(P, sm, prob, fac, t, lane, L, S, failed);
  }
#endif
  if (failed && lane == 0)
    atomicOr(&P.status[b], failed);
}

// ---------------------------------------------------------------------------------------------------
// Forward sweep of the wide shapes (computeInitial + forwardImpl, gar/riccati-kernel.hxx:195-207,
// 314-377), one wave per problem, lane = row: lane r < NU owns row r of K, lane r < NX row r of Aff
// and row r of Vxx' (the records carry fb ROW-major and Vxx symmetric: a row is 448 contiguous bytes at
// nx = 56, read as 16-byte pieces).  u = kff + K x, x' = yff + Aff x, lbd' = vx' + Vxx' x' with the state
// broadcast from the lanes that hold it (v_readlane).  Replaces the generic 256-thread kernel on these
// shapes (18.4 of 48.6 ms per 2 048 sweeps at (56, 22), N = 275).
// VPACK (round 6): the records keep the lower triangle of Vxx, rectangular packed (gar_layout.h) -- the next stage's
// 12.8 KB at nx = 56 are fetched LINEARLY (13 whole-line requests per lane instead of 28 pieces of a 448-byte row
// each) and become rows through LDS, as in gar_forward_mfma.
template <int NX, int NU, bool VPACK = false>
__global__ void __launch_bounds__(64) gar_forward_wide(MfmaFwdParams P) {
  using M = MfmaCfg<NX, NU, 0>;
  using VO = VxxOut<NX, true>;
  double *vb = gar_smem; // VPACK: nx (nx + 1) / 2 doubles of dynamic LDS
  static_assert(NX <= 64 && NX % 2 == 0, "the state lives in the first NX lanes");
  constexpr int NR = M::NR;
  const int lane = (int)threadIdx.x;
  const int b = (int)blockIdx.x;
  const double *fac = P.fac + (long long)b * P.fac_stride;
  double *sol = P.sol + (long long)b * P.sol_stride;
  const double *io = P.init + (long long)b * P.init_stride;
  const int N = P.horizon;
  const int iA = lane < NX ? lane : NX - 1, iK = lane < NU ? lane : NU - 1;
  double xs = io[iA]; // x0 from the initial-stage solve (kkt0.ff)
  if (lane < NX)
    sol[lane] = xs;
  for (int e = lane; e < P.nc0; e += 64)
    sol[P.sol_l + e] = io[NX + e]; // lbd0
#if GAR_WIDE_FWD_PIPELINED
  if constexpr (VPACK) {
    // The packed roll-out, software-pipelined (round 6): a stage's operands are two groups -- G1 = its [kff | K],
    // [yff | Aff] rows (u, x' need them), G2 = the NEXT stage's packed Vxx', vx' (lbd' needs them) -- and each group is
    // requested again, for the next stage, the moment its registers are free: G1(t + 1) travels under the lbd' phase of
    // stage t, G2(t + 1) under the u / x' phase of stage t + 1.  Same sums in the same order as the plain loop below.
    double2_t aff[NX / 2], kro[NX / 2], vp[VO::NCH];
    double kff, yff, vxn;
    auto load_g1 = [&](int t) {
      const double *rec = fac + P.slot(t) * P.fac_rec;
#pragma unroll
      for (int m = 0; m < NX / 2; ++m) {
        kro[m] = *reinterpret_cast<const double2_t *>(rec + M::fFB + iK * NX + 2 * m);
        aff[m] = *reinterpret_cast<const double2_t *>(rec + M::fFB + (NU + iA) * NX + 2 * m);
      }
      kff = rec[M::fFF + iK];
      yff = rec[M::fFF + NU + iA];
    };
    auto load_g2 = [&](int t) {
      const double *recn = (t + 1 < N) ? fac + P.slot(t + 1) * P.fac_rec : fac + P.fac_offN;
      const int oVn = (t + 1 < N) ? M::fVxx : M::tVxx, ovn = (t + 1 < N) ? M::fvx : M::tvx;
#pragma unroll
      for (int q = 0; q < VO::NCH; ++q) {
        const int e = 64 * q + lane, ec = (64 * q + 63 < VO::NP2 || e < VO::NP2) ? e : VO::NP2 - 1;
        vp[q] = *reinterpret_cast<const double2_t *>(recn + oVn + 2 * ec);
      }
      vxn = recn[ovn + iA];
    };
    if (N > 0) {
      load_g1(0);
      load_g2(0);
    }
    for (int t = 0; t < N; ++t) {
      double u0 = kff, u1 = 0.0, x0 = yff, x1 = 0.0;
#pragma unroll
      for (int m = 0; m < NX / 2; ++m) {
        const double xa = lane_bcast(xs, 2 * m), xb = lane_bcast(xs, 2 * m + 1);
        u0 = __builtin_fma(kro[m].x, xa, u0);
        u1 = __builtin_fma(kro[m].y, xb, u1);
        x0 = __builtin_fma(aff[m].x, xa, x0);
        x1 = __builtin_fma(aff[m].y, xb, x1);
      }
      const double u = u0 + u1, xn = x0 + x1;
      if (lane < NU)
        sol[P.sol_u + t * NU + lane] = u;
      if (lane < NX)
        sol[(t + 1) * NX + lane] = xn;
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < N)
        load_g1(t + 1);
      __builtin_amdgcn_sched_barrier(0);
      double l0 = vxn, l1 = 0.0; // lbd' = vx' + Vxx' x'  (:369-371)
      wave_sync(); // (the previous stage's row reads are done)
#pragma unroll
      for (int q = 0; q < VO::NCH; ++q) {
        const int e = 64 * q + lane;
        if (64 * q + 63 < VO::NP2 || e < VO::NP2)
          *reinterpret_cast<double2_t *>(&vb[2 * e]) = vp[q];
      }
      wave_sync();
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < N)
        load_g2(t + 1);
      __builtin_amdgcn_sched_barrier(0);
      const int lowbase = 2 * iA < NX ? iA * NX : (NX - 1 - iA) * (NX + 1) + 1;
#pragma unroll
      for (int j = 0; j < NX; j += 2) {
        const int c0 = 2 * j < NX ? j * NX : (NX - 1 - j) * (NX + 1) + 1;
        const int c1 = 2 * (j + 1) < NX ? (j + 1) * NX : (NX - 2 - j) * (NX + 1) + 1;
        l0 = __builtin_fma(vb[iA >= j ? c0 + iA : lowbase + j], lane_bcast(xn, j), l0);
        l1 = __builtin_fma(vb[iA >= j + 1 ? c1 + iA : lowbase + j + 1], lane_bcast(xn, j + 1), l1);
      }
      if (lane < NX)
        sol[P.sol_l + P.nc0 + t * NX + lane] = l0 + l1;
      xs = xn;
    }
    return;
  }
#endif
  for (int t = 0; t < N; ++t) {
    const double *rec = fac + P.slot(t) * P.fac_rec;
    const double *recn = (t + 1 < N) ? fac + P.slot(t + 1) * P.fac_rec : fac + P.fac_offN;
    const int oVn = (t + 1 < N) ? M::fVxx : M::tVxx, ovn = (t + 1 < N) ? M::fvx : M::tvx;
    double2_t aff[NX / 2], kro[NX / 2], vrow[VPACK ? 1 : NX / 2], vp[VPACK ? VO::NCH : 1];
#pragma unroll
    for (int m = 0; m < NX / 2; ++m) {
      kro[m] = *reinterpret_cast<const double2_t *>(rec + M::fFB + iK * NX + 2 * m);
      aff[m] = *reinterpret_cast<const double2_t *>(rec + M::fFB + (NU + iA) * NX + 2 * m);
    }
    if constexpr (VPACK) {
#pragma unroll
      for (int q = 0; q < VO::NCH; ++q) {
        const int e = 64 * q + lane, ec = (64 * q + 63 < VO::NP2 || e < VO::NP2) ? e : VO::NP2 - 1;
        vp[q] = *reinterpret_cast<const double2_t *>(recn + oVn + 2 * ec);
      }
    } else {
#pragma unroll
      for (int m = 0; m < NX / 2; ++m)
        vrow[m] = *reinterpret_cast<const double2_t *>(recn + oVn + iA * NX + 2 * m);
    }
    const double kff = rec[M::fFF + iK], yff = rec[M::fFF + NU + iA], vxn = recn[ovn + iA];
    double u0 = kff, u1 = 0.0, x0 = yff, x1 = 0.0;
#pragma unroll
    for (int m = 0; m < NX / 2; ++m) {
      const double xa = lane_bcast(xs, 2 * m), xb = lane_bcast(xs, 2 * m + 1);
      u0 = __builtin_fma(kro[m].x, xa, u0);
      u1 = __builtin_fma(kro[m].y, xb, u1);
      x0 = __builtin_fma(aff[m].x, xa, x0);
      x1 = __builtin_fma(aff[m].y, xb, x1);
    }
    const double u = u0 + u1, xn = x0 + x1;
    if (lane < NU)
      sol[P.sol_u + t * NU + lane] = u;
    if (lane < NX)
      sol[(t + 1) * NX + lane] = xn;
    double l0 = vxn, l1 = 0.0; // lbd' = vx' + Vxx' x'  (:369-371)
    if constexpr (VPACK) {
      wave_sync(); // (the previous stage's row reads are done)
#pragma unroll
      for (int q = 0; q < VO::NCH; ++q) {
        const int e = 64 * q + lane;
        if (64 * q + 63 < VO::NP2 || e < VO::NP2)
          *reinterpret_cast<double2_t *>(&vb[2 * e]) = vp[q];
      }
      wave_sync();
      // row iA of the symmetric matrix from its packed lower triangle (gar_sym_index): elements (iA, j), j <= iA, at
      // cj + iA; (j, iA), j > iA, at lowbase + j
      const int lowbase = 2 * iA < NX ? iA * NX : (NX - 1 - iA) * (NX + 1) + 1;
#pragma unroll
      for (int j = 0; j < NX; j += 2) {
        const int c0 = 2 * j < NX ? j * NX : (NX - 1 - j) * (NX + 1) + 1;
        const int c1 = 2 * (j + 1) < NX ? (j + 1) * NX : (NX - 2 - j) * (NX + 1) + 1;
        l0 = __builtin_fma(vb[iA >= j ? c0 + iA : lowbase + j], lane_bcast(xn, j), l0);
        l1 = __builtin_fma(vb[iA >= j + 1 ? c1 + iA : lowbase + j + 1], lane_bcast(xn, j + 1), l1);
      }
    } else {
#pragma unroll
      for (int m = 0; m < NX / 2; ++m) {
        l0 = __builtin_fma(vrow[m].x, lane_bcast(xn, 2 * m), l0);
        l1 = __builtin_fma(vrow[m].y, lane_bcast(xn, 2 * m + 1), l1);
      }
    }
    if (lane < NX)
      sol[P.sol_l + P.nc0 + t * NX + lane] = l0 + l1;
    xs = xn;
  }
}

} // namespace gar
