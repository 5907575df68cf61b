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
// (the CPU emulator of tests/emu implements the same instruction: ONE source path)
template <int N> __device__ __forceinline__ double row_bcast_c(double v, int lane) {
  (void)lane;
  return __builtin_amdgcn_mov_dpp(v, 0x150 + N, 0xF, 0xF, true);
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

// The same solve for a factor too large to hold as register operands (the 44 x 44 reduced KKT matrix
// of a constrained stage: 66 + 66 blocks): -L in LDS, PACKED by rows (row r at r(r+1)/2), every 4x4
// block read where it is used and applied to all T right-hand-side tiles; ndp[r] = -1/d[r].
template <int KK, int T>
__device__ __forceinline__ void ldl_solve_mfma4_packed(const double *Lp, const double *ndp, double (&X)[T][KK], int lane) {
  const int i3 = lane & 3, lk = lane >> 4;
  // start of row 4p+x in the packed factor: (4p+x)(4p+x+1)/2 = 2p(4p+1) + 4p x + x(x+1)/2
  const int ti0 = (i3 * (i3 + 1)) >> 1, tk0 = (lk * (lk + 1)) >> 1;
  auto tr_i = [&](int p) { return 2 * p * (4 * p + 1) + 4 * p * i3 + ti0; };
  auto tr_k = [&](int p) { return 2 * p * (4 * p + 1) + 4 * p * lk + tk0; };
  // (the blocks of one block row are fetched together, one block row AHEAD of the products that use them:
  // read where they are used, every ds_read's latency -- 132 of them -- would sit on the critical path)
  double ar[2][KK], adg[2];
  auto load_fwd = [&](int p, int buf) {
#pragma unroll
    for (int q = 0; q < KK; ++q)
      if (q < p)
        ar[buf][q] = Lp[tr_i(p) + 4 * q + lk]; // -L(4p+i3, 4q+lk)
    const double dgv = Lp[tr_i(p) + 4 * p + lk]; // (read by every lane -- the address is valid -- then masked:
    adg[buf] = (i3 > lk) ? dgv : 0.0;            //  a conditional load becomes a branch)
  };
  auto load_bwd = [&](int p, int buf) {
#pragma unroll
    for (int q = 0; q < KK; ++q)
      if (q > p)
        ar[buf][q] = Lp[tr_k(q) + 4 * p + i3]; // -L(4q+lk, 4p+i3): the transposed block
    const double dgv = Lp[tr_k(p) + 4 * p + i3];
    adg[buf] = (lk > i3) ? dgv : 0.0;
  };
  load_fwd(0, 0);
#pragma unroll
  for (int p = 0; p < KK; ++p) {
    const int cb = p & 1;
    if (p + 1 < KK)
      load_fwd(p + 1, cb ^ 1);
#pragma unroll
    for (int q = 0; q < p; ++q) {
#pragma unroll
      for (int t = 0; t < T; ++t)
        X[t][p] = __builtin_amdgcn_mfma_f64_4x4x4f64(ar[cb][q], X[t][q], X[t][p], 0, 0, 0);
    }
    double x0[T], y[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
      x0[t] = y[t] = X[t][p];
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int t = 0; t < T; ++t)
        y[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(adg[cb], y[t], x0[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < T; ++t)
      X[t][p] = y[t];
  }
  load_bwd(KK - 1, 0);
#pragma unroll
  for (int p = 0; p < KK; ++p) {
    const double nd = ndp[4 * p + lk];
#pragma unroll
    for (int t = 0; t < T; ++t)
      X[t][p] *= nd;
  }
#pragma unroll
  for (int p = KK - 1; p >= 0; --p) {
    const int cb = (KK - 1 - p) & 1;
    if (p > 0)
      load_bwd(p - 1, cb ^ 1);
#pragma unroll
    for (int q = KK - 1; q > p; --q) {
#pragma unroll
      for (int t = 0; t < T; ++t)
        X[t][p] = __builtin_amdgcn_mfma_f64_4x4x4f64(ar[cb][q], X[t][q], X[t][p], 0, 0, 0);
    }
    double x0[T], y[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
      x0[t] = y[t] = X[t][p];
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int t = 0; t < T; ++t)
        y[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(adg[cb], y[t], x0[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < T; ++t)
      X[t][p] = y[t];
  }
}

// sum over the four 16-lane rows of the wave, in every lane: two v_permlane{16,32}_swap exchanges on
// the VALU (gfx950) instead of two ds_bpermute round trips through the LDS crossbar
__device__ __forceinline__ double rows_sum(double a, int lane) {
  typedef unsigned uint2v __attribute__((ext_vector_type(2)));
  (void)lane;
  unsigned lo = (unsigned)__double2loint(a), hi = (unsigned)__double2hiint(a);
  uint2v l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  uint2v h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double s = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  lo = (unsigned)__double2loint(s);
  hi = (unsigned)__double2hiint(s);
  l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}

// Reduce-scatter over the four 16-lane rows: every lane holds partial sums p0..p3 (one per
// destination row), row g ends up with the sum of p_g over the four rows.  Recursive halving on
// v_permlane32_swap / v_permlane16_swap: a swap of (keep, send) register pairs IS the exchange --
// three swaps + three adds per double instead of one all-reduce (two swaps, two copies, two adds)
// per destination and a select.
__device__ __forceinline__ double rows_reduce_scatter(double p0, double p1, double p2, double p3, int lane) {
  typedef unsigned uint2v __attribute__((ext_vector_type(2)));
  (void)lane;
  // halves: lower rows {0,1} keep p0 / p1 and send p2 / p3, upper rows the other way round
  auto swap32_add = [](double keep_lo, double keep_hi) {
    const uint2v l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(keep_lo), (unsigned)__double2loint(keep_hi), false, false);
    const uint2v h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(keep_lo), (unsigned)__double2hiint(keep_hi), false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  };
  const double s02 = swap32_add(p0, p2); // rows 0,1: p0 summed over {g, g^2}; rows 2,3: p2
  const double s13 = swap32_add(p1, p3); // rows 0,1: p1 ...                 ; rows 2,3: p3
  const uint2v l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(s02), (unsigned)__double2loint(s13), false, false);
  const uint2v h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(s02), (unsigned)__double2hiint(s13), false, false);
  return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}

// the register LDL^T of wave_ldl_fast_neg on rows already in registers (a[j] = Rhat(row, j))
// Rare path, out of line: column k failed the FIRST Bunch-Kaufman test (|a_kk| >= alpha colmax,
// bunchkaufman.hpp:61).  Evaluate the second one (:63-75) on the rows the factorisation holds in
// registers (lane = row i: R.a[j] = trailing A(i, j), j <= i): imax = first row attaining colmax,
// rowmax = largest off-diagonal entry of row / column imax of the trailing matrix; Bunch-Kaufman
// keeps kp = k iff |a_kk| >= (alpha colmax) (colmax / rowmax).  Returns 0 in that case (the
// unpivoted elimination the caller is doing IS Bunch-Kaufman's), 1 when it interchanges or takes a
// 2x2 pivot (or the column is zero): the stage then goes to wave_slow_factor_solve.
template <int NU> struct LdlRow { double a[NU]; };
template <int NU>
__device__ __attribute__((noinline)) int wave_bk_second_test(LdlRow<NU> R, double akk, int k, int lane) {
  const double alpha = (1.0 + 4.123105625617661) / 8.0;
  double ck = 0.0;
#pragma unroll
  for (int c = 0; c < NU; ++c)
    ck = (c == k) ? R.a[c] : ck;
  const bool below = lane > k && lane < NU;
  const double absc = below ? fabs(ck) : 0.0;
  const double colmax = wave_max_f64(absc);
  const unsigned long long hit = __ballot(below && absc == colmax);
  const int imax = hit ? (int)__builtin_ctzll(hit) : k + 1; // maxCoeff keeps the first maximum
  double rp = 0.0; // own row, columns k .. lane-1
#pragma unroll
  for (int j = 0; j < NU; ++j)
    rp = (j >= k && j < lane) ? fmax(rp, fabs(R.a[j])) : rp;
  const double rowpart = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rp), imax),
                                          __builtin_amdgcn_readlane(__double2loint(rp), imax));
  double ti = 0.0;
#pragma unroll
  for (int c = 0; c < NU; ++c)
    ti = (c == imax) ? R.a[c] : ti;
  const double colpart = wave_max_f64((lane > imax && lane < NU) ? fabs(ti) : 0.0);
  const double rowmax = fmax(rowpart, colpart);
  // lane 0's verdict for the whole wave: akk is the pivot only in the first 16-lane row (the other
  // rows hold copies of row NU-1, see wave_ldl_fast_neg)
  return __builtin_amdgcn_readfirstlane((fabs(akk) >= (alpha * colmax) * (colmax / rowmax)) ? 0 : 1);
}

__device__ __forceinline__ unsigned long long wave_ballot(bool p) {
  return __builtin_amdgcn_ballot_w64(p); // the v_cmp result itself (HIP's __ballot goes through an int)
}
} // namespace gar
// (the blocked wave-scope L D L^T: built on row_bcast / wave_ballot / wave_lds_order above, used by the coupled stage below)
#include "gar_ldl_blocked.hpp"
// MEASURED AND NOT ADOPTED for the coupled stage (same box, alternating launches, 1 024 distinct problems):
//   * first form (profiles/r06_ab_coupled_blocked_ldl_not_kept.log, r06_trace_cstr_coupled.log): backward 15.59 ms
//     against 12.71 with the register version -- every LDS address of the routine is a function of the lane alone, the
//     compiler hoisted them out of the stage loop into a kernel that was full and reloaded them from SCRATCH inside
//     the factorisation (95 reloads, each behind a vmcnt(0)); 84.6 k cycles for the phase;
//   * with the lane index made opaque per stage (fence0: the addresses are computed where they are used):
//     11.98 ms against 12.70 (r06_ab_coupled_blocked_ldl_without_address_spills.log);
//   * and once the SAME cure was applied to the whole coupled stage (GAR_COUPLED_REFRESH_LANE, gar_wave.hpp: no scratch
//     at all, 432 registers) the register version is the faster one: 7.98 ms against 8.54 blocked
//     (r06_ab_coupled_lane_offsets_rederived_per_stage.log).  Results equal to 2e-15 throughout.
#ifndef GAR_COUPLED_BLOCKED_LDL
#define GAR_COUPLED_BLOCKED_LDL 0
#endif
#ifndef GAR_COUPLED_RELOAD_C   // 1: the C operands are loaded again behind the KKT solve (as until round 6); 0: they
#define GAR_COUPLED_RELOAD_C 0 //    stay in registers (the kernel has room since GAR_COUPLED_REFRESH_LANE): 7.82 -> 7.54 ms
#endif
// (the hybrid -- Rhat's columns as one DPP panel with the Schur complement on MFMA tiles, the rest in registers --
// measured and NOT adopted: 7.95 ms against 7.52, profiles/r06_ab_coupled_hybrid_ldl_not_kept.log)
// GAR_CSTR_EARLY_C (round 6; with GAR_CSTR_REFRESH_LANE, gar_wave.hpp): the decoupled constrained stage requests its
// constraint operands C, d at the START of the stage instead of behind the factorisation (where the leave-early branch
// keeps the compiler from moving them up): backward 4.54 -> 4.13 ms at batch 1 024 (0.551 -> 0.606 of the roofline),
// bitwise.  It needs the 80 registers the re-derived lane offsets free: without them the operands spill (5.99 ms).
// profiles/r06_ab_constrained_stage_c_operands_requested_early.log
#ifndef GAR_CSTR_EARLY_C
#define GAR_CSTR_EARLY_C 1
#endif
#ifndef GAR_CSTR_SLOT_C
#define GAR_CSTR_SLOT_C 1
#endif
#ifndef GAR_COUPLED_HYBRID_LDL
#define GAR_COUPLED_HYBRID_LDL 0
#endif
namespace gar {
// The register LDL^T of wave_ldl_fast_neg on rows already in registers (a[j] = Rhat(row, j)), under
// the COMPLETE pivot rule: a column that fails the first test is checked out of line against the
// second one and the elimination goes on when Bunch-Kaufman keeps kp = k.  Returns 0 when it kept
// kp = k at every column; `first_failed` tells whether any column needed the second test.
// (NU <= 16: the rows sit in one 16-lane DPP row and every broadcast is a v_mov_b64_dpp; wider
// Rhat -- the (56, 24) shape -- broadcasts through v_readlane)
template <int NU> __device__ __forceinline__ double ldl_bcast(double v, int n, int lane) {
  if (NU <= 16)
    return row_bcast(v, n & 15, lane);
  return lane_bcast(v, n);
}
// (nd_lds != nullptr: -1/d_k goes to LDS as it is produced -- it is wave-uniform -- instead of into nd[]:
// 2 NU registers less while a wide factorisation runs)
// spd_accept (the plain stage's Rhat only): a column that fails the first test is NOT examined further as long as
// the pivot is positive; if every pivot of the factorisation turns out positive -- Rhat = R + B^T V' B is positive
// definite on every convex stage -- the unpivoted LDL^T stands although Bunch-Kaufman would have interchanged:
// Cholesky-type elimination of a positive definite matrix is backward stable without pivoting, so the gains agree
// with the reference's pivoted factorisation to cond * eps.  A non-positive pivot anywhere returns 1: the stage then
// runs the device Bunch-Kaufman, the reference's own rule (interchanges, 2x2 pivots).  On the reference's generator
// at the north star 3.7 % of the stages used to take that 85 k-cycle path for interchanges that stability does not
// need.  GAR_HIP_SPD_ACCEPT=0 follows the reference's pivot rule literally.
// (K0 > 0: the columns before K0 have been eliminated already -- the trailing matrix is in a[K0 ..])
template <int NU, int NDN = NU, int K0 = 0>
__device__ __forceinline__ int wave_ldl_fast_neg_pre(int lane, double (&a)[NU], double (&nd)[NDN], bool &first_failed,
                                                     double *nd_lds = nullptr, const bool spd_accept = false) {
  const double alpha = (1.0 + 4.123105625617661) / 8.0;
  int bad = 0;
  first_failed = false;
  double minpiv = 1.0; // smallest pivot so far (wave-uniform); NaN-safe: the comparison below is !(x > 0)
#pragma unroll
  for (int k = K0; k < NU; ++k) {
    const double akk = ldl_bcast<NU>(a[k], k, lane);
    minpiv = !(akk > 0.0) ? -1.0 : minpiv;
    const unsigned long long nok = wave_ballot(!(fabs(akk) >= alpha * fabs(a[k])) || akk == 0.0);
    const unsigned long long from_k = ((1ull << NU) - 1ull) & ~((1ull << k) - 1ull);
    if (nok & from_k) { // wave-uniform, rare
      first_failed = true;
      if (!spd_accept) {
        LdlRow<NU> R;
#pragma unroll
        for (int j = 0; j < NU; ++j)
          R.a[j] = a[j];
        bad |= wave_bk_second_test<NU>(R, akk, k, lane);
      }
    }
    const double nd_k = -fast_rcp(akk);
    const double nlik = a[k] * nd_k; // -L(i,k)
#pragma unroll
    for (int j = k + 1; j < NU; ++j)
      a[j] = __builtin_fma(ldl_bcast<NU>(nlik, j, lane), a[k], a[j]); // a(i,j) -= L(j,k) a(i,k)
    a[k] = nlik;
    if (NDN == NU)
      nd[NDN == NU ? k : 0] = nd_k;
    else if (lane == 0)
      nd_lds[k] = nd_k;
  }
  if (spd_accept && first_failed) // the unpivoted factorisation stands only if Rhat proved positive definite
    bad |= __builtin_amdgcn_readfirstlane(minpiv > 0.0 ? 0 : 1);
  return bad;
}

// A memory instruction issued right behind a v_mfma_f64_16x16x4 costs ~8 cycles of the wave's time
// instead of ~20 (the MFMA holds the VALU for 64 cycles; LDS and memory instructions of the same
// wave keep issuing: scripts/ubench/overlap.cpp).  The compiler's scheduler clusters memory
// instructions instead, so the phases below pin their order (sched_barrier) and SLOT the stage's
// independent memory work -- the previous stage's Vxx -> HBM, the rows of Rhat for the
// factorisation, B for Aff, the gains' stores, the next knot's loads, the V tiles -> LDS -- behind
// the MFMAs of the tile columns, of Aff and of Vxx.
#define GAR_SB __builtin_amdgcn_sched_barrier(0)
#ifndef GAR_F_DMA_YOUNGER
#define GAR_F_DMA_YOUNGER 24
#endif

// NC > 0 (equality constraints C x + D u + d = 0 on the knot, riccati-kernel.hxx:232-262): this stage
// serves the DECOUPLED case D = 0 -- what the reference's own generator and benchmark produce
// (tests/gar/test_util.cpp:42-43, bench/gar-riccati.cpp:19-22).  The reduced KKT matrix
// [Rhat D^T; D -mu I] is then block diagonal: Bunch-Kaufman on it is Bunch-Kaufman on Rhat (a column of
// the -mu I block has no off-diagonal entry: it never pivots), [zff | Z] = [d | C] / mu, and the stage
// is the unconstrained one plus Vxx += C^T Z (KC more k-steps per tile), vx += C^T zff and the rows
// [zff | Z] of the record.  Returns 0 WITHOUT having changed anything the caller cannot restore (S.Hc:
// reload with wave_load_b) when D != 0 or Rhat needs a pivot.
// COUPLED (NC > 0): any D.  The register LDL^T under the complete Bunch-Kaufman rule runs on the
// NK = NU + NC rows of [Rhat D^T; D -mu I] (lane = row: rows NU.. are [D(i,:) | 0 .. -mu]); -L goes to
// LDS packed, and the 1 + NX right-hand sides [rhat Shat^T; d C] -- the constraint rows are the C
// operand registers themselves -- are solved on v_mfma_f64_4x4x4 in NK/4 block rows
// (ldl_solve_mfma4_packed).  Returns 0 only when Bunch-Kaufman interchanges or takes a 2x2 pivot
// somewhere: the caller then runs the stage with the LDS Bunch-Kaufman (wave_stage).
// NC = 0: always returns 1.
template <int NX, int NU, int NC = 0, bool COUPLED = false, bool FDMA = false>
__device__ __forceinline__ int wave_stage2(const MfmaParams &P, double *sm, const double *prob,
                                           double *fac, int t, int lane,
                                           const WaveLane<NX, NU, NC> &L, WaveStage<NX, NU> &S,
                                           int &failed, double *&vflush, const bool tracing) {
  using C = WaveCfg<NX, NU, NC>;
  using M = MfmaCfg<NX, NU, NC>;
  constexpr int NK = C::NK, NR = C::NR, KC = C::KC; // NK = NU + NC rows [K; Z] ahead of Aff in the record
  static_assert(NC % 4 == 0 && (NC == 0 || !M::WIDE) && (NC > 0 || !COUPLED), "constraints in k-steps of four");
  constexpr int KUK = NK / 4; // block rows of the reduced KKT matrix
  constexpr int NW = C::NW, PK = C::PK, PG = C::PG, TX = C::TX, TW = C::TW, KS = C::KS, KU = C::KU;
  // [qhat; rhat] one entry per lane; the wide shapes (NW > 64: (56, 24)) keep entries 64.. in a
  // second register (they are control entries: NX <= 64), and write fb ROW-major (the generic
  // record layout: their forward sweep is the generic kernel) instead of the fbT2 device order
  constexpr bool WIDE = M::WIDE;
  static_assert(NX <= 64 && NW <= 128 && TW <= 5, "state entries in the first register");
  const int li = lane & 15, lk = lane >> 4;
  const unsigned fbrm = 8u * (unsigned)(lk * NX + li); // row-major fb: element (lk, li)
  double *V = sm + C::oV, *G = sm + C::oG, *Mm = sm + C::oM, *vn = sm + C::oVn;
  double *Lr = sm + C::oLr, *ndi = sm + C::oDi;
  constexpr int oVxx = M::fVxx, ovx = M::fvx;
  double *out = fac + P.slot(t) * P.fac_rec;
  const double *rec = prob + P.in_off0 + P.slot(t) * P.in_rec;
  const double *recn = prob + P.in_off0 + P.slot(t > 0 ? t - 1 : 0) * P.in_rec; // knot t-1 (t = 0: harmless re-read)
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
  static_assert(!FDMA || (NC == 0 && !WIDE && !COUPLED), "F-DMA: the plain serial stage");
  // F-DMA: [A | B] of knot t-1 on its way into LDS now (the buffer's previous content -- this knot's -- went into
  // the F operand registers during the previous stage); read back in the Aff phase below, a stage's worth of
  // MFMAs later.  Fb: the buffer addressed like the knot record (the lane offsets of F include kA).
  [[maybe_unused]] const double *Fb = sm + C::oF - M::kA;
  if constexpr (FDMA) {
    if (t > 0)
      wave_dma<8 * C::f_doubles>(recn + M::kA, reinterpret_cast<char *>(sm + C::oF), lane);
  }
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
  double td[NC > 0 ? NU : 1]; // NC > 0: D(lane - NU, :) in lanes NU .. NK-1 (row lane of the reduced KKT matrix)
  if (NC > 0) {
    const int crow = lane < NU ? 0 : (lane < NK ? lane - NU : NC - 1);
#pragma unroll
    for (int j = 0; j < NU; ++j)
      td[j] = ldg_b(rec, M::kD + j * NC, 8u * (unsigned)crow);
  }
  constexpr int KC1 = KC > 0 ? KC : 1;
  double Cop[TX][KC1]; // C[4s+lk][16t+li]: A operand (C^T) and, times 1/mu, B operand (Z) of Vxx += C^T Z
  double Cop4[KC1];    // REM4: C[4s+k4][NX-4+i4]
  double dz[KC1];      // zff[4s+lk] = d[4s+lk] / mu
  auto load_cop = [&]() {
    const int i4c = lane & 3, k4c = lane >> 4;
#pragma unroll
    for (int tc = 0; tc < TX; ++tc) {
      const int x = (16 * tc + li) < NX ? (16 * tc + li) : NX - 1;
#pragma unroll
      for (int sc = 0; sc < KC; ++sc)
        Cop[tc][sc] = ldg_b(rec, M::kC + 4 * sc, 8u * (unsigned)(x * NC + lk));
    }
#pragma unroll
    for (int sc = 0; sc < KC; ++sc)
      if (C::REM4)
        Cop4[sc] = ldg_b(rec, M::kC + 4 * sc, 8u * (unsigned)((NX - 4 + i4c) * NC + k4c));
#pragma unroll
    for (int sc = 0; sc < KC; ++sc)
      dz[sc] = ldg_b(rec, M::kd + 4 * sc, 8u * (unsigned)lk);
  };
  // (GAR_CSTR_EARLY_C: the constraint operands are requested at the START of the stage, like D -- they used to be
  // requested behind the factorisation, where the stage's leave-early branch sits in front of them)
  constexpr bool EARLY_C = NC > 0 && !COUPLED && (GAR_CSTR_EARLY_C != 0); // (the coupled stage has no room for them across its 44-row factorisation)
  if constexpr (EARLY_C && !((GAR_CSTR_SLOT_C != 0) && !M::WIDE))
    load_cop();
  double qri1 = 0.0; // WIDE: [q; r][64 + lane]
  if (WIDE)
    qri1 = ldg_b(rec, M::kq + 64, 8u * (unsigned)(lane < NW - 64 ? lane : NW - 65));
  // ---- P = V' F, H = W + F^T P (:216-228), tile columns from the control columns down ---------
  constexpr int TXF = C::REM4 ? TX - 1 : TX;
  const int i4 = lane & 3, k4 = lane >> 4;
  constexpr int cR = NX >> 4; // first tile column holding control columns: Rhat needs tj >= cR
  // V' as the A operand of P = V'F: the same registers serve every tile column (the wide shapes
  // have no registers to spare for that: they read the operands from LDS at every use)
#ifndef GAR_PRELOAD_V
#define GAR_PRELOAD_V 1
#endif
  constexpr bool PRELOAD_V = !WIDE && GAR_PRELOAD_V;
  double Vop[PRELOAD_V ? (TXF > 0 ? TXF : 1) : 1][PRELOAD_V ? KS : 1], Vop4[PRELOAD_V ? KS : 1];
  if (PRELOAD_V) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int tm = 0; tm < TXF; ++tm) {
        const int ic = (16 * tm + li) < NX ? (16 * tm + li) : NX - 1;
        Vop[PRELOAD_V ? tm : 0][PRELOAD_V ? s : 0] = V[ic * PK + 4 * s + lk];
      }
      Vop4[PRELOAD_V ? s : 0] = C::REM4 ? V[(NX - 4 + i4) * PK + 4 * s + k4] : 0.0;
    }
  }
  double part[TW]; // (F^T vx' + P^T f)[16 tj + li], summed over this lane's rows
  double a_row[NU], nd[NU];
  double Bop[TX][KU]; // B of this knot as the A operand of Aff = A + B K: B[16ti+li][4s'+lk]
  double Bop4[KU];    // REM4: B[NX-4+i4][4s'+k4], the A operand of the 4x4x4 blocks
  // ---- memory work slotted behind the MFMAs of the tile columns tj < cR (list A) ----------------
  using VO = VxxOut<NX, GAR_VXX_PACKED && !WIDE, PK>;
  constexpr int NCH = VO::NCH;
  constexpr int nA_flush = NCH + 2, nA_rows = NU, nA_bop = TX * KU + (C::REM4 ? KU : 0);
  // (EARLY_C && GAR_CSTR_SLOT_C: the constraint operands C, d join the list instead of being issued in one burst at the
  // start of the stage -- 40 more loads behind the MFMAs of the state tile columns)
  constexpr bool SLOT_C = EARLY_C && (GAR_CSTR_SLOT_C != 0) && !WIDE;
  constexpr int nA_cop = SLOT_C ? TX * KC + (C::REM4 ? KC : 0) + KC : 0;
  constexpr int nA_base = nA_flush + nA_rows + nA_bop;
  constexpr int nA = nA_base + nA_cop;
  double2_t vbuf[NCH];
  const int frow = lane < NU ? lane : NU - 1;
  auto slotA = [&](int i) { // i: compile-time after unrolling
    if (i < nA_flush) {     // the previous stage's Vxx -> HBM: LDS read of chunk i, store of chunk i-2
      if (i < NCH)
        vbuf[i < NCH ? i : 0] = VO::read(V, i, lane);
      if (i >= 2)
        VO::write(vflush, i - 2, lane, vbuf[i - 2 < NCH ? i - 2 : 0]);
#ifdef GAR_PROBE_AFF_LINEAR // (timing probe: the bytes of Aff as 16-byte-per-lane linear stores, garbage data)
      if (i >= 2 && !WIDE) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
          *reinterpret_cast<double2_t *>(&(vflush - M::NR * NX)[2 * (64 * (2 * (i - 2) + u) + lane)]) = vbuf[i - 2 < NCH ? i - 2 : 0];
      }
#endif
    } else if (i < nA_flush + nA_rows) { // Rhat, lane = row
      const int j = i - nA_flush;
      a_row[j < NU ? j : 0] = Mm[j * NU + frow];
    } else if (i >= nA_base) { // (SLOT_C) C[4s+lk][16t+li], the REM4 operands, d
      const int q = i - nA_base;
      if (q < TX * KC) {
        const int tc = q / KC1, sc = q % KC1;
        const int x = (16 * tc + li) < NX ? (16 * tc + li) : NX - 1;
        Cop[tc < TX ? tc : 0][sc] = ldg_b(rec, M::kC + 4 * sc, 8u * (unsigned)(x * NC + lk));
      } else if (C::REM4 && q < TX * KC + KC) {
        const int sc = q - TX * KC;
        Cop4[sc < KC1 ? sc : 0] = ldg_b(rec, M::kC + 4 * sc, 8u * (unsigned)((NX - 4 + (lane & 3)) * NC + (lane >> 4)));
      } else {
        const int sc = q - TX * KC - (C::REM4 ? KC : 0);
        dz[sc < KC1 ? sc : 0] = ldg_b(rec, M::kd + 4 * sc, 8u * (unsigned)lk);
      }
    } else if (i < nA_base) {
      const int q = i - nA_flush - nA_rows;
      if (q < TX * KU) {
        const int ti = q / KU, sq = q % KU;
        Bop[ti < TX ? ti : 0][sq] = WaveLane<NX, NU>::x_in(ti) ? ldg_b(rec, 4 * sq * NX + 16 * ti, L.bop0)
                                                            : ldg_b(rec, 4 * sq * NX, L.bopX);
      } else {
        const int sq = q - TX * KU;
        Bop4[sq < KU ? sq : 0] = ldg_b(rec, 4 * sq * NX, L.bop4);
      }
    }
  };
  int sA = 0; // next op of list A
  if (WIDE) { // no registers to spare for operands fetched ahead of their use: the flush goes first,
              // the rows of Rhat and B are loaded right where they are consumed
#pragma unroll
    for (int i = 0; i < nA_flush; ++i)
      slotA(i);
    sA = nA_flush;
  }
#pragma unroll
  for (int tj = TW - 1; tj >= 0; --tj) {
    const bool pinned = !WIDE && (tj < cR); // compile-time
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
        const int icv = (16 * tm + li) < NX ? (16 * tm + li) : NX - 1;
        const double aq = PRELOAD_V ? Vop[PRELOAD_V ? tm : 0][PRELOAD_V ? s : 0] : V[icv * PK + 4 * s + lk];
        Pt[tm] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, bq, Pt[tm], 0, 0, 0);
        if (pinned) {
          GAR_SB;
          if (sA < nA)
            slotA(sA++);
          GAR_SB;
        }
      }
      if (C::REM4)
        p4 = __builtin_amdgcn_mfma_f64_4x4x4f64(Vop4[PRELOAD_V ? s : 0], bq, p4, 0, 0, 0);
    }
#pragma unroll
    for (int ti = tj; ti < TW; ++ti) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const double pq = (C::REM4 && (s >> 2) == TX - 1) ? p4 : Pt[s >> 2][s & 3];
        S.Hc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(S.fo(ti, s), pq, S.Hc[ti][tj], 0, 0, 0);
        if (pinned) {
          GAR_SB;
          if (sA < nA)
            slotA(sA++);
          GAR_SB;
        }
      }
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
    if (tj == cR) {
      GAR_WMARK(1)
      // ---- Rhat (lower) is complete: to LDS (column-major), read back lane = row by list A
#pragma unroll
      for (int ti = cR; ti < TW; ++ti)
#pragma unroll
        for (int tc = cR; tc <= ti; ++tc)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * ti + lk + 4 * r, c = 16 * tc + li;
            if (16 * ti + 4 * r >= NX && 16 * ti + 4 * r < NW) { // compile-time: control rows
              if (c >= NX && c <= row)
                Mm[(c - NX) * NU + (row - NX)] = S.Hc[ti][tc][r];
            }
          }
      wave_lds_order();
      if (cR > 0)
        GAR_SB;
    }
  }
  if (cR > 0)
    GAR_SB;
#pragma unroll
  for (int i = 0; i < nA; ++i) // what did not find a shadow (all of it for the shapes with cR = 0)
    if (i >= sA && !(WIDE && i >= nA_flush + nA_rows))
      slotA(i);
  GAR_WMARK(2)
  // ---- register LDL^T of Rhat under the first Bunch-Kaufman test; -L and -1/d to LDS -------------
  bool first_failed;
  int verdict;
  bool d_nonzero = false;
  if (NC > 0) {
    bool nz = false;
#pragma unroll
    for (int j = 0; j < NU; ++j)
      nz |= (td[j] != 0.0);
    d_nonzero = wave_ballot(nz && lane >= NU && lane < NK) != 0ull;
  }
  constexpr int NKC = COUPLED ? NK : 1;
  double a44[NKC], nd44[1];
  double *Lpk = Mm, *nd44p = sm + C::oBk; // COUPLED: -L packed by rows (Rhat in Mm is consumed), -1/d
  if constexpr (COUPLED) {
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      if (j < NU)
        a44[j] = lane < NU ? a_row[j] : td[j];
      else
        a44[j] = (lane == j || (lane >= NK && j == NK - 1)) ? -P.mueq : 0.0;
    }
#if GAR_COUPLED_HYBRID_LDL
    // round 6, hybrid: the NU columns of Rhat as ONE DPP panel of the blocked routine (gar_ldl_blocked.hpp) -- its
    // trailing update forms the NC x NC Schur complement -mu I - D Rhat^{-1} D^T on MFMA tiles instead of NU x NC
    // broadcast-FMA pairs per lane -- and the Schur complement's own factorisation in registers as before.  In place on
    // the row-packed matrix in Mm, where the solve reads -L.  A column of the panel that fails the first test: verdict 2.
    static_assert(NU % 4 == 0 && NU <= 16, "the Rhat columns are one DPP panel");
    if (lane < NK) {
      const int base0 = (lane * (lane + 1)) >> 1;
#pragma unroll
      for (int j = 0; j < NK; ++j)
        if (j <= lane)
          Lpk[base0 + j] = a44[j];
    }
    wave_lds_order();
    verdict = wave_ldl_blocked<NK, NU, false, LdlRowPacked, LdlRowPacked, false, 1>(Lpk, nd44p + ((NK + 3) & ~3), Lpk, nd44p, lane,
                                                                                    first_failed, false, a44[0]);
    wave_lds_order();
    if (verdict == 0) {
      const int rowh = lane < NK ? lane : NK - 1, baseh = (rowh * (rowh + 1)) >> 1;
#pragma unroll
      for (int j = NU; j < NK; ++j)
        a44[j] = Lpk[j <= rowh ? baseh + j : ((j * (j + 1)) >> 1) + rowh]; // (above the diagonal: the mirror entry, unused)
      bool ff2;
      verdict = wave_ldl_fast_neg_pre<NK, 1, NU>(lane, a44, nd44, ff2, nd44p);
      first_failed |= ff2;
      if (lane < NK) {
        const int base1 = (lane * (lane + 1)) >> 1;
#pragma unroll
        for (int j = NU; j < NK - 1; ++j)
          if (j < lane)
            Lpk[base1 + j] = a44[j];
      }
    }
#elif GAR_COUPLED_BLOCKED_LDL
    // round 6: the 44-row factorisation (946 v_readlane broadcast-FMA pairs in registers) in panels of 12 columns on
    // DPP broadcasts, the trailing matrix updated on MFMA tiles (gar_ldl_blocked.hpp) -- IN PLACE on the matrix packed
    // by rows in Mm, which is exactly where and how the solve below reads -L.  A column that fails Bunch-Kaufman's
    // first test leaves with verdict 2: the caller's chain hands the knot to the LDS Bunch-Kaufman kernel (the
    // reference's complete rule), as for a knot where it really pivots.
    if (lane < NK) {
      const int base0 = (lane * (lane + 1)) >> 1;
#pragma unroll
      for (int j = 0; j < NK; ++j)
        if (j <= lane)
          Lpk[base0 + j] = a44[j];
    }
    wave_lds_order();
    verdict = wave_ldl_blocked<NK, 12, false, LdlRowPacked, LdlRowPacked>(Lpk, nd44p + ((NK + 3) & ~3), Lpk, nd44p, lane, first_failed, false, a44[0]);
    wave_lds_order();
#else
    verdict = wave_ldl_fast_neg_pre<NK, 1>(lane, a44, nd44, first_failed, nd44p);
#endif
  } else {
    verdict = wave_ldl_fast_neg_pre<NU>(lane, a_row, nd, first_failed, nullptr, P.spd_accept != 0);
  }
  if (first_failed && lane == 0) { // diagnostics: stages that needed the second test / that really pivot
    atomicAdd(&P.slow[0], 1);
    if (verdict != 0)
      atomicAdd(&P.slow[1], 1);
  }
  if constexpr (COUPLED) {
    if (verdict != 0)
      return 0;
#if !GAR_COUPLED_BLOCKED_LDL && !GAR_COUPLED_HYBRID_LDL
    const int base = (lane * (lane + 1)) >> 1;
#pragma unroll
    for (int j = 0; j < NK - 1; ++j)
      if (j < lane && lane < NK)
        Lpk[base + j] = a44[j];
#endif
  } else {
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
  }
  const double imu = 1.0 / P.mueq;
  if (NC > 0) {
    if (verdict != 0 || (!COUPLED && d_nonzero))
      return 0;
    if constexpr (!EARLY_C)
      load_cop();
  }
  GAR_WMARK(3)
  // ---- [qhat; rhat] = [q; r] + F^T vx' + P^T f (:217-218, :227-228) ---------------------------
  double hq;
  const double fi = S.fi;
  {
    const double sel = rows_reduce_scatter(part[0], TW > 1 ? part[TW > 1 ? 1 : 0] : 0.0,
                                           TW > 2 ? part[TW > 2 ? 2 : 0] : 0.0, TW > 3 ? part[TW > 3 ? 3 : 0] : 0.0, lane);
    hq = S.qri + sel;
    if (WIDE) { // entries 64 + li (tile column 4): summed over the four rows of lanes, in every lane
      const double hq1 = qri1 + rows_sum(part[TW > 4 ? 4 : 0], lane);
      if (lane < NW - 64)
        G[(64 + lane - NX) * PG] = hq1;
    }
    if (lane >= NX && lane < (NW < 64 ? NW : 64))
      G[(lane - NX) * PG] = hq; // rhat
  }
  wave_lds_order();
  GAR_WMARK(4)
  // ---- [kff | K] = -Rhat^{-1} [rhat | Shat^T] (:248-262) ---------------------------------------
  double Kb[TX][KU]; // K[4s'+lk][16tj+li]: the B operand of Aff and Vxx
  constexpr bool SPARE = (NX % 16) != 0; // a free lane column in the last state tile for rhat
  constexpr int lc = NX % 16;
  constexpr int KC2 = COUPLED ? KC1 : 1;
  double Zb[TX][KC2]; // COUPLED: Z[4sc+lk][16tj+li] out of the solve (decoupled: Cop * (1/mu))
  if constexpr (COUPLED) {
    // right-hand sides in B-operand layout: block rows 0..KU-1 = [rhat | Shat^T] (rhat in the spare lane
    // column of the last state tile, or a tile of its own), block rows KU.. = [d | C]
    constexpr bool SPARE_C = (NX % 16) != 0;
    constexpr int lcc = NX % 16, TXN = SPARE_C ? TX : TX + 1;
    double X[TXN][KUK];
#pragma unroll
    for (int tj = 0; tj < TX; ++tj) {
#pragma unroll
      for (int sp = 0; sp < KU; ++sp) {
        const double sv = S.Hc[C::shTile(sp)][tj][C::shReg(sp)];
        const double g0 = (SPARE_C && tj == TX - 1) ? G[(4 * sp + lk) * PG] : 0.0; // (unconditional read, then select)
        X[tj][sp] = (SPARE_C && tj == TX - 1 && li == lcc) ? g0 : sv;
      }
#pragma unroll
      for (int sc = 0; sc < KC; ++sc)
        X[tj][KU + sc] = (SPARE_C && tj == TX - 1 && li == lcc) ? dz[sc] : Cop[tj][sc];
    }
    if (!SPARE_C) {
#pragma unroll
      for (int sp = 0; sp < KU; ++sp)
        X[TXN - 1][sp] = G[(4 * sp + lk) * PG];
#pragma unroll
      for (int sc = 0; sc < KC; ++sc)
        X[TXN - 1][KU + sc] = dz[sc];
    }
    GAR_WMARK(5)
    GAR_WMARK(20)
    ldl_solve_mfma4_packed<KUK, TXN>(Lpk, nd44p, X, lane);
    GAR_WMARK(21)
#pragma unroll
    for (int tj = 0; tj < TX; ++tj) {
#pragma unroll
      for (int sp = 0; sp < KU; ++sp)
        Kb[tj][sp] = X[tj][sp];
#pragma unroll
      for (int sc = 0; sc < KC; ++sc)
        Zb[tj][sc] = X[tj][KU + sc];
    }
#if GAR_COUPLED_RELOAD_C
    load_cop(); // (again, L2 hits: the operands do not stay in registers across the solve)
#endif
    GAR_WMARK(22)
    // [kff; zff] -> column 0 of G (rows 0..NK-1): read back one entry per lane row below
    if (li == (SPARE_C ? lcc : 0)) {
#pragma unroll
      for (int sp = 0; sp < KUK; ++sp)
        G[(4 * sp + lk) * PG] = X[TXN - 1][sp];
    }
  } else if (verdict == 0) {
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
    if constexpr (NC == 0) // (NC > 0 left above: verdict != 0 returns 0)
      failed |= wave_slow_factor_solve<NX, NU, false, true>(sm, lane, nullptr); // (counted above)
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
  if (WIDE) {
#pragma unroll
    for (int i = nA_flush + nA_rows; i < nA_base; ++i)
      slotA(i); // B as the A operand of Aff, and of yff = f + B kff
  }
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
      if (NC > 0) { // vx += C^T zff (:275-276 with the constraint rows)
#pragma unroll
        for (int sc = 0; sc < KC; ++sc)
          c = __builtin_fma(Cop[ti][sc], COUPLED ? G[(NU + 4 * sc + lk) * PG] : dz[sc] * imu, c);
      }
      py[ti] = a;
      pv[ti] = c;
    }
    if (NC > 0 && lane < NC)
      out[M::fFF + NU + lane] = COUPLED ? G[(NU + lane) * PG] : ldg_b(rec, M::kd, 8u * (unsigned)(lane < NC ? lane : 0)) * imu; // zff
    const double sy = rows_reduce_scatter(py[0], TX > 1 ? py[TX > 1 ? 1 : 0] : 0.0, TX > 2 ? py[TX > 2 ? 2 : 0] : 0.0,
                                          TX > 3 ? py[TX > 3 ? 3 : 0] : 0.0, lane);
    const double sv = rows_reduce_scatter(pv[0], TX > 1 ? pv[TX > 1 ? 1 : 0] : 0.0, TX > 2 ? pv[TX > 2 ? 2 : 0] : 0.0,
                                          TX > 3 ? pv[TX > 3 ? 3 : 0] : 0.0, lane);
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
  if constexpr (COUPLED) {
    // the coupled stage is short of registers while the 44-row factorisation and the solve run: the F
    // operands of the state columns are not kept across them but fetched again here (L2 hits)
#pragma unroll
    for (int tcol = 0; tcol < TX; ++tcol)
#pragma unroll
      for (int sq = 0; sq < KS; ++sq) {
        const double v = WaveLane<NX, NU>::fo_in(tcol) ? ldg_b(rec, 16 * tcol * NX + 4 * sq, L.fo0)
                                                       : ldg_b(rec, 4 * sq, L.foX);
        if (sq < 4 * C::KSF)
          S.Fo[tcol][sq >> 2][sq & 3] = v;
        else
          S.FoT[tcol][sq - 4 * C::KSF] = v;
      }
  }
  // ---- Aff = A + B K (:267), in place on the F operand registers, one tile column after the
  // other; behind the MFMAs of column tj: the stores of K (tj = 0) / of Aff's column tj-1 and the
  // loads of the next knot's F operands into the registers that column just released ------------
  double4_t accT[TX];
  if (C::KST > 0 && !C::REM4) {
#pragma unroll
    for (int tj = 0; tj < TX; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        accT[tj][r] = (r < C::KST) ? S.FoT[tj][r] : 0.0;
  }
  constexpr int nKst = TX * KU;
  auto store_K = [&](int q) {
    const int tj = q / KU, sq = q % KU;
    if (16 * tj + 15 < NX || 16 * tj + li < NX) {
      if (WIDE)
        stg_b(out, M::fFB + 4 * sq * NX + 16 * tj, fbrm, Kb[tj < TX ? tj : 0][sq]);
      else
        stg_b(out, M::fFB + 8 * tj * 2 * NR + 8 * sq, L.fbl, Kb[tj < TX ? tj : 0][sq]);
    }
  };
  constexpr int nZst = TX * KC; // Z = C / mu -> fb rows NU .. NU+NC-1
  auto store_Z = [&](int q) {
    const int tj = q / KC1, sq = q % KC1;
    if (16 * tj + 15 < NX || 16 * tj + li < NX)
      stg_b(out, M::fFB + 8 * tj * 2 * NR + 2 * (NU + 4 * sq), L.fbl,
            COUPLED ? Zb[tj < TX ? tj : 0][COUPLED ? sq : 0] : Cop[tj < TX ? tj : 0][sq] * imu);
  };
  int sZ = 0;
  // column tj of Aff -> fb rows NK.. (fbT2), then F's column tile tj of knot t-1 into the same registers
  constexpr int nRow4 = (NX + 3) / 4; // (ti, r) pairs with 16 ti + 4 r < NX
  constexpr int nCol = nRow4 + KS;    // stores, then loads
  auto col_op = [&](int tj, int q) {
    if (q < nRow4) {
      const int ti = q >> 2, r = q & 3;
      const int i = 16 * ti + lk + 4 * r, j = 16 * tj + li;
      if (i < NX && j < NX) {
        const double v = ti < C::KSF ? S.Fo[tj][ti < C::KSF ? ti : 0][r] : (C::REM4 ? S.FoT[tj][0] : accT[tj][r]);
#ifndef GAR_PROBE_NO_AFF_STORE // (timing probe: what the 10.4 KB of Aff per stage cost the sweep; results are then wrong)
        if (WIDE)
          stg_b(out, M::fFB + (NK + 16 * ti + 4 * r) * NX + 16 * tj, fbrm, v);
        else
          stg_b(out, M::fFB + 8 * tj * 2 * NR + 2 * (NK + 16 * ti + 4 * r), L.fbl, v);
#else
        asm volatile("" ::"v"(v));
#endif
      }
    } else if (q < nCol) {
      const int sq = q - nRow4; // k-step of F's column tile tj
      const double *fsrc = FDMA ? Fb : recn; // (F-DMA: the same addressing, of the LDS image)
      const double v = WaveLane<NX, NU>::fo_in(tj) ? ldg_b(fsrc, 16 * tj * NX + 4 * sq, L.fo0)
                                                   : ldg_b(fsrc, 4 * sq, L.foX);
      if (sq < 4 * C::KSF)
        S.Fo[tj][(sq >> 2) < C::KSF ? (sq >> 2) : 0][sq & 3] = v;
      else
        S.FoT[tj][(sq - 4 * C::KSF) < C::KST ? (sq - 4 * C::KSF) : 0] = v;
    }
  };
  if constexpr (FDMA) {
    // the DMA pieces were the first vector-memory instructions of this stage; gfx9 retires loads and stores in order
    // on one counter, so "at most GAR_F_DMA_YOUNGER instructions outstanding" implies "the pieces have landed" as long
    // as at least that many were issued behind them -- the stage issues > 60 (counted in the ISA, Makefile: fdma_check)
    GAR_WAIT_VMCNT(GAR_F_DMA_YOUNGER);
    wave_lds_order();
  }
  GAR_SB;
  int pend_col = -1, pend_q = 0; // column whose stores/loads are being slotted, next op of it
  int sK = 0;
#pragma unroll
  for (int tj = 0; tj < TX; ++tj) {
#pragma unroll
    for (int s = 0; s < KU; ++s)
#pragma unroll
      for (int ti = 0; ti < TX; ++ti) {
#ifdef GAR_PROBE_NO_AFF // (timing probe: ... and its 27 + 3 MFMAs)
        if (false) {
        } else if (ti < C::KSF || C::REM4) {
        } else {
#else
        if (ti < C::KSF) {
          S.Fo[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bop[ti][s], Kb[tj][s], S.Fo[tj][ti], 0, 0, 0);
        } else if (C::REM4) {
          S.FoT[tj][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(Bop4[s], Kb[tj][s], S.FoT[tj][0], 0, 0, 0);
        } else {
#endif
          accT[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bop[ti][s], Kb[tj][s], accT[tj], 0, 0, 0);
        }
        if (ti < C::KSF || !C::REM4) { // behind a 16x16x4
          GAR_SB;
#pragma unroll
          for (int rep = 0; rep < 3; ++rep) {
            if (sK < nKst)
              store_K(sK++);
            else if (pend_col >= 0 && pend_q < nCol)
              col_op(pend_col, pend_q++);
          }
          GAR_SB;
        }
      }
    // finish the previous column's list before this column becomes the pending one
#pragma unroll
    for (int q = 0; q < nCol; ++q)
      if (pend_col >= 0 && q >= pend_q)
        col_op(pend_col, q);
    pend_col = tj;
    pend_q = 0;
    if (tj == 0) {
      GAR_WMARK(11)
    } else if (tj == 1) {
      GAR_WMARK(12)
    }
  }
  GAR_WMARK(8)
  // ---- Vxx = Qhat + Shat K (:272-273), lower tiles, tile after tile; behind the MFMAs: what is left
  // of list B (K, the last Aff column, its F loads) and the finished tiles -> V in LDS (mirrored) ----
  constexpr int shLo = C::shTile(0);
  double sh4[KU];
  if (C::REM4) {
#pragma unroll
    for (int s = 0; s < KU; ++s)
      sh4[s] = __shfl(S.Hc[C::shTile(s)][TX - 1][C::shReg(s)], (lane & 48) | ((NX - 4) & 15) | (lane & 3));
  }
  auto v_write = [&](int ti, int tj, int r, double v) { // element (16ti+lk+4r, 16tj+li) and its mirror
    const int i = 16 * ti + lk + 4 * r, c = 16 * tj + li;
    const bool ok = (i < NX && c < NX && i >= c);
    if (ti > tj && 16 * ti + 4 * r + 3 < NX && 16 * tj + 15 < NX) { // compile-time: all lanes valid
      V[i * PK + c] = v;
      V[c * PK + i] = v;
    } else {
      V[ok ? i * PK + c : C::oDump] = v;
      V[ok ? c * PK + i : C::oDump + 1] = v;
    }
  };
  // the 4-row remainder tiles first (16-cycle MFMAs: no shadow worth slotting into)
  double acc4[TX];
  if (C::REM4) {
#pragma unroll
    for (int tj = 0; tj < TX; ++tj) {
      acc4[tj] = S.Hc[TX - 1][tj][0];
#pragma unroll
      for (int s = 0; s < KU; ++s)
        acc4[tj] = __builtin_amdgcn_mfma_f64_4x4x4f64(sh4[s], Kb[tj][s], acc4[tj], 0, 0, 0);
      if (NC > 0) {
#pragma unroll
        for (int sc = 0; sc < KC; ++sc)
          acc4[tj] = __builtin_amdgcn_mfma_f64_4x4x4f64(Cop4[sc], COUPLED ? Zb[tj][COUPLED ? sc : 0] : Cop[tj][sc] * imu, acc4[tj], 0, 0, 0);
      }
    }
  }
  GAR_WMARK(13)
  GAR_SB;
  // tiles on the 16x16x4 instruction, one after the other; the tile finished last is written while
  // the next one accumulates
  int wr_ti = -1, wr_tj = -1, wr_q = 0; // tile being written (its accumulator is wr_acc), next register
  double4_t wr_acc = double4_t{0.0, 0.0, 0.0, 0.0};
  bool r4_written = !C::REM4;
#pragma unroll
  for (int tj = 0; tj < TX; ++tj)
#pragma unroll
    for (int ti = tj; ti < (C::REM4 ? TX - 1 : TX); ++ti) {
      double4_t acc = S.Hc[ti][tj];
#pragma unroll
      for (int s = 0; s < KU + KC; ++s) {
        // Shat K, then (NC > 0) C^T Z: A = C^T(16ti+li, 4sc+lk), B = Z(4sc+lk, 16tj+li)
        const double aq = s < KU ? S.Hc[C::shTile(s < KU ? s : 0)][ti][C::shReg(s < KU ? s : 0)] : Cop[ti][s >= KU ? s - KU : 0];
        const double bq = s < KU ? Kb[tj][s < KU ? s : 0]
                                 : (COUPLED ? Zb[tj][COUPLED && s >= KU ? s - KU : 0] : Cop[tj][s >= KU ? s - KU : 0] * imu);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, bq, acc, 0, 0, 0);
        GAR_SB;
#pragma unroll
        for (int rep = 0; rep < 3; ++rep) {
          if (sK < nKst) {
            store_K(sK++);
          } else if (pend_col >= 0 && pend_q < nCol) {
            col_op(pend_col, pend_q++);
          } else if (sZ < nZst) {
            store_Z(sZ++);
          } else if (!r4_written) {
#pragma unroll
            for (int c4 = 0; c4 < TX; ++c4)
              v_write(TX - 1, c4, 0, acc4[c4]);
            r4_written = true;
          } else if (wr_ti >= 0 && wr_q < 4) {
            if (16 * wr_ti + 4 * wr_q < NX)
              v_write(wr_ti, wr_tj, wr_q, wr_acc[wr_q]);
            ++wr_q;
          }
        }
        GAR_SB;
      }
      // the previous tile must be fully written before its slot is reused
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (wr_ti >= 0 && q >= wr_q && 16 * wr_ti + 4 * q < NX)
          v_write(wr_ti, wr_tj, q, wr_acc[q]);
      wr_acc = acc;
      wr_ti = ti;
      wr_tj = tj;
      wr_q = 0;
    }
  GAR_WMARK(15)
  GAR_SB;
  // drain: list B, the remainder tiles, the last tile
#pragma unroll
  for (int q = 0; q < nKst; ++q)
    if (q >= sK)
      store_K(q);
#pragma unroll
  for (int q = 0; q < nCol; ++q)
    if (pend_col >= 0 && q >= pend_q)
      col_op(pend_col, q);
#pragma unroll
  for (int q = 0; q < nZst; ++q)
    if (q >= sZ)
      store_Z(q);
  if (!r4_written) {
#pragma unroll
    for (int c4 = 0; c4 < TX; ++c4)
      v_write(TX - 1, c4, 0, acc4[c4]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (wr_ti >= 0 && q >= wr_q && 16 * wr_ti + 4 * q < NX)
      v_write(wr_ti, wr_tj, q, wr_acc[q]);
  // F column tiles Aff does not touch (pure B columns), and the vectors of knot t-1
#pragma unroll
  for (int tcol = TX; tcol < TW; ++tcol)
#pragma unroll
    for (int sq = 0; sq < KS; ++sq) {
      const double v = WaveLane<NX, NU>::fo_in(tcol) ? ldg_b(recn, 16 * tcol * NX + 4 * sq, L.fo0)
                                                     : ldg_b(recn, 4 * sq, L.foX);
      if (sq < 4 * C::KSF)
        S.Fo[tcol][sq >> 2][sq & 3] = v;
      else
        S.FoT[tcol][sq - 4 * C::KSF] = v;
    }
  S.fi = ldg_b(recn, 0, L.fi);
  S.qri = ldg_b(recn, 0, L.qri);
  wave_sync();
  GAR_WMARK(14)
  // ---- knot t-1: its Hessian tiles replace H
  wave_load_b<NX, NU, WaveLane<NX, NU, NC>, (GAR_QR_PACKED && !WIDE)>(recn, L, S);
  GAR_WMARK(9)
  // ---- Vxx -> HBM is left to the next stage (list A) / to the caller after the last one -----------
  vflush = out + oVxx;
  GAR_WMARK(10)
#undef GAR_WMARK
  return 1;
}

} // namespace gar
