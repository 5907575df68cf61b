// gar_ldl_blocked.hpp -- wave-scope BLOCKED L D L^T (round 6): panels of NB columns eliminated in registers, the
// trailing matrix updated on v_mfma_f64_16x16x4.
//
// The register factorisations of gar_wave.hpp / gar_wave2.hpp (wave_ldl_fast, wave_ldl_fast_neg_pre: lane = row, one
// broadcast + FMA per (column, later column) pair) cost n (n - 1) / 2 such pairs: 66 at n = 12, where every broadcast
// is one v_mov_b64_dpp inside a 16-lane row -- and 276 / 630 / 946 at n = 24 / 36 / 44 (the Talos shape's Rhat, the
// 36 x 36 blocks of the condensed system, the coupled reduced KKT matrix of bench/gar-riccati.cpp's shape), where the
// rows span several DPP rows and every broadcast is a v_readlane pair through an SGPR: 10 - 35 k cycles, the longest
// single phase of those stages.  Here:
//   * a panel of NB <= 16 columns is eliminated with DPP broadcasts only: every 16-lane row keeps, beside the panel
//     entries of its lanes' OWN matrix rows, a copy of the panel's diagonal block (lane k of each row = block row k)
//     and eliminates it redundantly -- the multipliers -L(c0 + j, c0 + k) every lane needs are then in lane j of its
//     own DPP row;
//   * the rest of the matrix lives in LDS between panels and receives the panel's contribution
//     A22 -= L21 D L21^T as 16 x 16 x 4 MFMA tiles (operands = the panel as it was just written for the solve).
// Same arithmetic as the reference's elimination (core/bunchkaufman.hpp:104-121) with the sums of the trailing update
// in another order (the reference's own factorisation is the BLOCKED one for n > 32, :172-344); the pivot rule is
// checked exactly as the register versions check it: first test of Bunch-Kaufman (:61) at every column on the
// up-to-date column, lane-parallel.
#pragma once
#include "gar_wave2.hpp"

namespace gar {

// LDS the routine needs beside its outputs: the working copy (N x N) and -d_k (N)
template <int N> struct LdlBlockedLds {
  static constexpr int oW = 0, oNp = N * N, total = (N * N + N + 1) & ~1;
};

// Index policies: where element (i, j) of the working matrix / of the factor lives in its LDS array.
template <int N> struct LdlColMajor {  // full column-major block, pitch N (upper part: whatever the array holds)
  __device__ static __forceinline__ int at(int i, int j) { return j * N + i; }
};
template <int N> struct LdlRowMajor {  // full row-major block, pitch N
  __device__ static __forceinline__ int at(int i, int j) { return i * N + j; }
};
struct LdlRowPacked {                  // lower triangle packed by rows: (i, j), j <= i, at i (i + 1) / 2 + j; (i, j) = (j, i)
  __device__ static __forceinline__ int at(int i, int j) {
    const int a = i >= j ? i : j, b = i >= j ? j : i;
    return ((a * (a + 1)) >> 1) + b;
  }
};

// W   : LDS, the N x N matrix, element (i, j) at WI::at(i, j), lower triangle valid; DESTROYED.
// npv : LDS, N doubles of scratch (-d_k).
// Lr  : LDS out, -L: (i, j), j < i, at LI::at(i, j) (entries j >= i are not written); POSL: +L instead of -L.
//       Lr may BE W with the same policy: the factorisation is then in place.
// ndi : LDS out, -1 / d_k
// DEFINITE: the caller knows the matrix is definite: no pivot test, only a zero / non-finite pivot is reported (1).
// Otherwise: returns 0 when Bunch-Kaufman's first test held at every column (its elimination is then this one);
// a column that fails it: with spd_accept the elimination goes on and stands iff every pivot is positive (else 1);
// without, 2 is returned at once -- the caller runs the complete rule (the unblocked register version, from the
// pristine matrix).  first_failed: some column failed the first test (diagnostics).
// NPANELS > 0: only the first NPANELS panels are eliminated (with their trailing updates): the caller goes on from the
// updated matrix in W (the hybrid of the coupled constrained stage: one DPP panel for the Rhat columns, whose trailing
// update forms the 32 x 32 Schur complement on MFMA tiles, the rest in registers).
template <int N, int NB, bool DEFINITE, class WI = LdlColMajor<N>, class LI = LdlRowMajor<N>, bool POSL = false, int NPANELS = 0>
__device__ __forceinline__ int wave_ldl_blocked(double *W, double *npv, double *Lr, double *ndi, int lane,
                                                bool &first_failed, const bool spd_accept, const double dep = 0.0) {
  static_assert(NB % 4 == 0 && NB <= 16 && N <= 64, "panels of k-steps of four inside one DPP row; lane = row");
  // `dep`: any value the calling stage has just produced.  Every LDS address below is a function of the lane alone:
  // inside a stage loop the compiler hoists ALL of them out of the loop (~100 loop-invariant registers), finds no room
  // in a kernel that is full already and SPILLS them -- measured in the coupled stage: 95 scratch reloads, each waited
  // for with vmcnt(0), on the factorisation's critical path (profiles/r06_ab_coupled_blocked_ldl_not_kept.log).  A lane
  // index the compiler believes to depend on `dep` keeps the address arithmetic where it is used.
  lane += fence0(dep);
  const double alpha = (1.0 + 4.123105625617661) / 8.0;
  const int li = lane & 15, lk = lane >> 4;
  const int row = lane < N ? lane : N - 1;
  constexpr int NP = (N + NB - 1) / NB;
  int bad = 0;
  first_failed = false;
  double minpiv = 1.0;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int c0 = p * NB;
    const int nb = (N - c0) < NB ? (N - c0) : NB; // (constant after unrolling)
    // ---- the panel: own rows and, per 16-lane row, the copy of the diagonal block ----------------------
    double a[NB], dg[NB];
    const int drow = c0 + (li < nb ? li : nb - 1);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (j < nb) {
        a[j] = W[WI::at(row, c0 + j)];
        dg[j] = W[WI::at(drow, c0 + j)];
      }
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      if (k >= nb)
        continue;
      const double akk = row_bcast(dg[k], k, lane);
      if (DEFINITE) {
        bad |= (akk == 0.0 || !(fabs(akk) <= 1.79e308)) ? 1 : 0; // wave-uniform
      } else {
        minpiv = !(akk > 0.0) ? -1.0 : minpiv;
        // rows below the pivot (own rows c0 + k < row): |a_kk| >= alpha |a(i, k)|; the pivot's own lane: non-zero
        const unsigned long long nok = wave_ballot(!(fabs(akk) >= alpha * fabs(a[k])) || akk == 0.0);
        const unsigned long long upto = (N >= 64) ? ~0ull : ((1ull << N) - 1ull);
        const unsigned long long from_k = upto & ~((1ull << (c0 + k)) - 1ull);
        if (nok & from_k) { // wave-uniform, rare
          first_failed = true;
          if (!spd_accept)
            return 2;
        }
      }
      const double nd_k = -fast_rcp(akk);
      const double nld = dg[k] * nd_k, nla = a[k] * nd_k; // -L(., c0 + k): block copy / own row
#pragma unroll
      for (int j = k + 1; j < NB; ++j) {
        if (j < nb) {
          const double t = row_bcast(nld, j, lane); // -L(c0 + j, c0 + k)
          dg[j] = __builtin_fma(t, dg[k], dg[j]);
          a[j] = __builtin_fma(t, a[k], a[j]);      // a(i, j) -= L(j, k) a(i, k)
        }
      }
      dg[k] = nld;
      a[k] = nla;
      if (lane == 0) {
        ndi[c0 + k] = nd_k;
        npv[c0 + k] = -akk;
      }
    }
    if (lane < N) {
#pragma unroll
      for (int j = 0; j < NB; ++j)
        if (j < nb && c0 + j < lane)
          Lr[LI::at(lane, c0 + j)] = POSL ? -a[j] : a[j];
    }
    const int c1 = c0 + nb;
    if (c1 >= N)
      break;
    wave_lds_order();
    // ---- trailing update A22 -= L21 D L21^T = A22 + L21 (-D) L21^T, 16 x 16 tiles anchored at c1 ------
    const int nt = (N - c1 + 15) / 16, ks = nb / 4;
#pragma unroll
    for (int ta = 0; ta < (N + 15) / 16; ++ta) {
      if (ta >= nt)
        continue;
      const int ia = c1 + 16 * ta + li, iac = ia < N ? ia : N - 1;
      double opA[NB / 4];
#pragma unroll
      for (int s = 0; s < NB / 4; ++s)
        if (s < ks)
          opA[s] = Lr[LI::at(iac, c0 + 4 * s + lk)];
#pragma unroll
      for (int tb = 0; tb < (N + 15) / 16; ++tb) {
        if (tb > ta)
          continue;
        const int jb = c1 + 16 * tb + li, jbc = jb < N ? jb : N - 1;
        double4_t acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ii = c1 + 16 * ta + 4 * r + lk;
          acc[r] = W[WI::at(ii < N ? ii : N - 1, jbc)];
        }
#pragma unroll
        for (int s = 0; s < NB / 4; ++s)
          if (s < ks) {
            // (the product of the two stored factors is L(i, k) L(j, k) whatever their common sign: times -d_k)
            const double opB = Lr[LI::at(jbc, c0 + 4 * s + lk)] * npv[c0 + 4 * s + lk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[s], opB, acc, 0, 0, 0);
          }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ii = c1 + 16 * ta + 4 * r + lk;
          if (ii < N && jb < N && ii >= jb)
            W[WI::at(ii, jb)] = acc[r];
        }
      }
    }
    wave_lds_order();
    if (NPANELS > 0 && p + 1 >= NPANELS)
      break;
  }
  if (!DEFINITE && spd_accept && first_failed) // the unpivoted factorisation stands only if the matrix proved positive definite
    bad |= __builtin_amdgcn_readfirstlane(minpiv > 0.0 ? 0 : 1);
  return bad;
}

} // namespace gar
