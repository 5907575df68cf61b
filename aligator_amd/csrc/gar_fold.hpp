// gar_fold.hpp -- equality-constrained knots (nc > 0) in leg mode on the UNCONSTRAINED wave-leg kernels.
//
// The reference's own benchmark runs ParallelRiccatiSolver on nx = 36, nu = 12 with nc = 32 constraints on every
// knot (bench/gar-riccati.cpp:64-90, BM_parallel) generated with D = 0 (tests/gar/test_util.cpp:42-43).  With
// D = 0 the reduced KKT matrix [Rhat D^T; D -mu I] (riccati-kernel.hxx:232-241) is block diagonal: Bunch-Kaufman
// on it IS Bunch-Kaufman on Rhat (a column of the -mu I block has no off-diagonal entry, it never pivots), and
//   [zff | Z | Zth] = [d | C | Gv] / mu          (:248-262, :288-292; Gv = 0 under the leg parameterisation)
//   Vxx = Qhat + Shat K + C^T Z,  vx = qhat + Shat kff + C^T zff              (:272-277)
// Q and q enter stageKernelSolve / terminalSolve ONLY through those two sums (:224, :227, :175-183), so the
// constrained stage is the unconstrained stage of the knot with
//   Q <- Q + C^T (C / mu),   q <- q + C^T (d / mu)
// -- every other output (K, kff, Aff, yff, Kth, Yth, Vxt, Vtt, vt, the factorisation of Rhat) is unchanged, and
// Z, zff, Zth follow from C, d alone.  Three elementwise kernels around the unchanged wave-leg family:
//   gar_fold_constraints       knot records -> folded knot records (nc = 0 layout), and the D = 0 check: a problem
//                              with any D != 0 is flagged and left to the generic leg kernels (same launch
//                              sequence, no host decision: each family skips the problems of the other)
//   gar_constraint_multipliers v_t = zff + Z x_t after the roll-out (:349-351)
//   gar_expand_constrained     on request (gains / value read-back): the caller-visible factor records with
//                              ff = [kff; zff; yff], fb = [K; Z; Aff], fth = [Kth; 0; Yth]
#pragma once
#include "gar_layout.h"

namespace gar {

struct FoldParams {
  const gar_stage_meta *meta;  // the solver's layout (knots with nc)
  const gar_stage_meta *meta2; // folded layout (same knots, nc = 0)
  const double *prob;          // the caller's knots
  double *prob2;               // folded knots
  double *fac;                 // caller-visible factor records (nc layout)
  const double *fac2;          // factor records of the wave-leg kernels (nc = 0 layout)
  double *sol;
  long long prob_stride, prob2_stride, fac_stride, fac2_stride, sol_stride;
  int *coupled; // per problem: != 0 if some knot has D != 0 (generic kernels take the problem)
  int horizon, t2, t_lo, t_hi;
  double mueq;
  int qr_packed; // the caller's knots t < horizon keep Q, R as packed lower triangles (gar_layout.h); the folded knots: full
  int lds;       // gar_fold_constraints: the launch carries fold_lds_doubles(max nx, max nc) doubles of LDS
};
__host__ __device__ inline int fold_lds_doubles(int nx, int nc) { return nx * (nc + 1) + nx * nc + nc + 2; }

// grid (horizon + 1, batch) x 256
__global__ void __launch_bounds__(256) gar_fold_constraints(FoldParams P) {
  const int t = (int)blockIdx.x, b = (int)blockIdx.y, tid = (int)threadIdx.x;
  const gar_stage_meta m = P.meta[t];
  const int nx = m.nx, nu = m.nu, nc = m.nc, nx2 = m.nx2;
  const gar_knot_offsets ko = gar_knot_layout(nx, nu, nc, nx2, 0);
  const double *src = P.prob + (long long)b * P.prob_stride + m.in_off;
  double *dst = P.prob2 + (long long)b * P.prob2_stride + P.meta2[t].in_off;
  // Q S R q r A B f sit at the same offsets in both layouts (C D d come last)
  const double *Cm = src + ko.C, *dv = src + ko.d;
  const double mu = P.mueq;
  const bool pk = P.qr_packed && t < P.horizon;
  // C and Z = C / mu, zff = d / mu staged in LDS once per knot (round 6: every one of the nx^2 nc products used to divide
  // -- 41 k f64 divisions per (36, 12, 32) knot, 60 us per sweep on one problem; same operations on the same operands in
  // the same order: bitwise the same folded knot).  C's copy has the odd pitch nc + 1 (lanes = consecutive i).
  if (P.lds && nc > 0) {
    double *Cs = gar_smem, *Zs = Cs + nx * (nc + 1), *zd = Zs + nx * nc;
    for (int e = tid; e < nx * nc; e += 256) {
      const int i = e / nc, k = e - i * nc;
      const double c = Cm[e];
      Cs[i * (nc + 1) + k] = c;
      Zs[e] = c / mu;
    }
    for (int k = tid; k < nc; k += 256)
      zd[k] = dv[k] / mu;
    __syncthreads();
    for (int e = tid; e < ko.C; e += 256) {
      double v = src[e];
      if (pk && e >= ko.R && e < ko.R + nu * nu) {
        const int j = (e - ko.R) / nu, i = (e - ko.R) - j * nu;
        v = src[ko.R + (i >= j ? gar_lower_index(nu, i, j) : gar_lower_index(nu, j, i))];
      }
      if (e < nx * nx) { // Q(i, j) += sum_k C(k, i) * (C(k, j) / mu)
        const int j = e / nx, i = e - j * nx;
        if (pk)
          v = src[ko.Q + (i >= j ? gar_lower_index(nx, i, j) : gar_lower_index(nx, j, i))];
        double acc = 0.0;
        for (int k = 0; k < nc; ++k)
          acc = __builtin_fma(Cs[i * (nc + 1) + k], Zs[j * nc + k], acc);
        v += acc;
      } else if (e >= ko.q && e < ko.q + nx) { // q(i) += sum_k C(k, i) * (d(k) / mu)
        const int i = e - ko.q;
        double acc = 0.0;
        for (int k = 0; k < nc; ++k)
          acc = __builtin_fma(Cs[i * (nc + 1) + k], zd[k], acc);
        v += acc;
      }
      dst[e] = v;
    }
  } else
  for (int e = tid; e < ko.C; e += 256) {
    double v = src[e];
    if (pk && e >= ko.R && e < ko.R + nu * nu) { // (the folded knot keeps full blocks: the wave-leg family's format)
      const int j = (e - ko.R) / nu, i = (e - ko.R) - j * nu;
      v = src[ko.R + (i >= j ? gar_lower_index(nu, i, j) : gar_lower_index(nu, j, i))];
    }
    if (e < nx * nx) { // Q(i, j) += sum_k C(k, i) * (C(k, j) / mu)
      const int j = e / nx, i = e - j * nx;
      if (pk)
        v = src[ko.Q + (i >= j ? gar_lower_index(nx, i, j) : gar_lower_index(nx, j, i))];
      double acc = 0.0;
      for (int k = 0; k < nc; ++k)
        acc = __builtin_fma(Cm[i * nc + k], Cm[j * nc + k] / mu, acc);
      v += acc;
    } else if (e >= ko.q && e < ko.q + nx) { // q(i) += sum_k C(k, i) * (d(k) / mu)
      const int i = e - ko.q;
      double acc = 0.0;
      for (int k = 0; k < nc; ++k)
        acc = __builtin_fma(Cm[i * nc + k], dv[k] / mu, acc);
      v += acc;
    }
    dst[e] = v;
  }
  // mu not strictly positive (or so small that C / mu overflows): the fold divides by it -- such a problem goes
  // to the generic leg kernels like one with D != 0, whose Bunch-Kaufman meets the singular [Rhat 0; 0 -mu I] and
  // reports the failed stage as the reference does (riccati-kernel.hxx:239-241)
  int bad = (nc > 0 && !(mu >= 1e-290)) ? 1 : 0;
  for (int e = tid; e < nc * nu; e += 256)
    bad |= (src[ko.D + e] != 0.0);
  if (bad)
    atomicOr(&P.coupled[b], 1);
}

// grid (horizon + 1, batch) x 64: vs[t] = zff + Z xs[t]  (riccati-kernel.hxx:349-351), Z = C / mu, zff = d / mu
__global__ void __launch_bounds__(64) gar_constraint_multipliers(FoldParams P) {
  const int t = (int)blockIdx.x, b = (int)blockIdx.y;
  if (P.coupled[b] || t < P.t_lo || t >= P.t_hi)
    return;
  const gar_stage_meta m = P.meta[t];
  const int nx = m.nx, nc = m.nc;
  if (nc == 0)
    return;
  const gar_knot_offsets ko = gar_knot_layout(nx, m.nu, nc, m.nx2, 0);
  const double *src = P.prob + (long long)b * P.prob_stride + m.in_off;
  double *sol = P.sol + (long long)b * P.sol_stride;
  const double *x = sol + m.x_off;
  const double mu = P.mueq;
  for (int k = (int)threadIdx.x; k < nc; k += 64) {
    double acc = src[ko.d + k] / mu;
    for (int j = 0; j < nx; ++j)
      acc = __builtin_fma(src[ko.C + j * nc + k] / mu, x[j], acc);
    sol[m.v_off + k] = acc;
  }
}

// grid (horizon + 1, batch) x 256: the caller-visible factor record of stage t from the wave-leg family's
__global__ void __launch_bounds__(256) gar_expand_constrained(FoldParams P) {
  const int t = (int)blockIdx.x, b = (int)blockIdx.y, tid = (int)threadIdx.x;
  if (P.coupled[b])
    return;
  const gar_stage_meta m = P.meta[t];
  const int nx = m.nx, nu = m.nu, nc = m.nc, nx2 = m.nx2, nth = m.nth;
  const gar_factor_offsets fo = gar_factor_layout(nx, nu, nc, nx2, nth), f2 = gar_factor_layout(nx, nu, 0, nx2, nth);
  const gar_knot_offsets ko = gar_knot_layout(nx, nu, nc, nx2, 0);
  const double *knot = P.prob + (long long)b * P.prob_stride + m.in_off;
  const double *src = P.fac2 + (long long)b * P.fac2_stride + P.meta2[t].fac_off;
  double *dst = P.fac + (long long)b * P.fac_stride + m.fac_off;
  const int nr = nu + nc + nx2, nr2 = nu + nx2;
  const bool tr = P.t2 && t < P.horizon; // fbT2: element (r, j) at (j / 2) 2 nr + 2 r + (j & 1)
  const double mu = P.mueq;
  for (int r = tid; r < nr; r += 256)
    dst[fo.ff + r] = r < nu ? src[f2.ff + r] : (r < nu + nc ? knot[ko.d + (r - nu)] / mu : src[f2.ff + r - nc]);
  for (int e = tid; e < nr * nx; e += 256) {
    const int r = e / nx, j = e - r * nx;
    double v;
    if (r >= nu && r < nu + nc) {
      v = knot[ko.C + j * nc + (r - nu)] / mu;
    } else {
      const int r2 = r < nu ? r : r - nc;
      v = src[f2.fb + (tr ? (j >> 1) * (2 * nr2) + 2 * r2 + (j & 1) : r2 * nx + j)];
    }
    dst[fo.fb + (tr ? (j >> 1) * (2 * nr) + 2 * r + (j & 1) : e)] = v;
  }
  for (int e = tid; e < nr * nth; e += 256) {
    const int r = e / nth, j = e - r * nth;
    double v = 0.0; // Zth = -Gv / (-mu) = 0 under the leg parameterisation
    if (!(r >= nu && r < nu + nc)) {
      const int r2 = r < nu ? r : r - nc;
      v = src[f2.fth + (tr ? (j >> 1) * (2 * nr2) + 2 * r2 + (j & 1) : r2 * nth + j)];
    }
    dst[fo.fth + (tr ? (j >> 1) * (2 * nr) + 2 * r + (j & 1) : e)] = v;
  }
  const int nval = nx * nx + nx + nx * nth + nth * nth + nth; // Vxx vx Vxt Vtt vt, contiguous in both layouts
  for (int e = tid; e < nval; e += 256)
    dst[fo.Vxx + e] = src[f2.Vxx + e];
}

} // namespace gar
