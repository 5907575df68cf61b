// gar_dense.hpp -- the reference's stage-dense solver on the device: RiccatiSolverDense
// (include/aligator/gar/dense-riccati.hpp:19-56, dense-riccati.hxx:48-116) over DenseKernel
// (include/aligator/gar/dense-kernel.hpp:55-209).
//
// Instead of condensing a stage through R-hat = R + B^T V' B, this solver factorises, per stage,
// the whole symmetric indefinite matrix
//       [ R    D^T   B^T   .  ]   rows/cols: u (nu), v (nc), lambda' (nx2), x' (nx2)
//       [ D   -mu I   .    .  ]
//       [ B     .     .   -I  ]
//       [ .     .    -I   Pxx']
// (Bunch-Kaufman, n = nu + nc + 2 nx2) and solves for [ff | fb | ft] at once: it never forms
// A^T V' A, which is what makes it the reference's choice for badly conditioned stages.  Any
// per-stage dimensions, constraints and user parameters; serial in time (as in the reference).
//
// One 256-thread workgroup per problem.  LDS: the KKT matrix (lower triangle packed by columns) and the
// right-hand sides (row-major, n x (1 + nx + nth), exactly the reference's row-major fb/ft with ff
// as column 0).  The value function (Pxx, px, Pxt, Ptt, pt) is an output anyway: it is written to
// the stage's factor record and the next stage reads it back from there (same workgroup, after a
// barrier; it sits in L2).  Factor record: gar_factor_layout(nx, nu, nc, 2 nx2, nth), i.e. the
// Riccati layout with nr = nu + nc + 2 nx2 rows [K; Z; L; Y], Vxx.. holding Pxx...
//
// Terminal knot: the reference factorises the full matrix with its last 2 nx2 rows zero
// (dense-kernel.hpp:57-74); its Bunch-Kaufman stops at the first zero column (NumericalIssue,
// unchecked).  With nu = nc = 0 nothing is read from it.  Here the leading (nu + nc) block -- the
// system that is meant -- is factorised; rows L, Y of the terminal record are zero.
#pragma once
#include "gar_generic.hpp"

namespace gar {

// (GAR_DENSE_THREADS threads: assembly, trailing updates, substitution strips and the value-function products spread
// over 16 waves; the panel of the factorisation stays one wave's work)
#ifndef GAR_DENSE_THREADS
#define GAR_DENSE_THREADS 1024
#endif
__global__ void __launch_bounds__(GAR_DENSE_THREADS) gar_backward_dense(GenericParams P) {
  const WG w = wg_self();
  double *sm = gar_smem;
  const int b = (int)blockIdx.x;
  const double *prob = P.prob + (long long)b * P.prob_stride;
  double *fac = P.fac + (long long)b * P.fac_stride;
  double *Kmat = sm + P.dense.K, *Rm = sm + P.dense.R, *sub = sm + P.dense.sub;
  int *piv = (int *)(sm + P.dense.piv), *ctrl = piv + 512;
  const double mu = P.mueq;
  int failed = 0;
  for (int t = P.horizon; t >= 0; --t) {
    const gar_stage_meta m = P.meta[t];
    const int nx = m.nx, nu = m.nu, nc = m.nc, nx2 = m.nx2, nth = m.nth;
    const bool term = (t == P.horizon);
    const int o1 = nu, o2 = nu + nc, o3 = o2 + nx2, nfull = o3 + nx2;
    const int n = term ? o2 : nfull; // the block that is factorised
    const int rld = 1 + nx + nth;
    const gar_knot_offsets ko = gar_knot_layout(nx, nu, nc, nx2, (m.flags & GAR_KNOT_HAS_PARAM) ? nth : 0);
    const gar_factor_offsets fo = gar_factor_layout(nx, nu, nc, 2 * nx2, nth);
    const double *k = prob + m.in_off;
    double *fr = fac + m.fac_off;
    // next stage's value function (dense-riccati.hxx:57-61)
    const double *Pn = nullptr, *pn = nullptr, *Pxtn = nullptr;
    if (!term) {
      const gar_stage_meta mn = P.meta[t + 1];
      const gar_factor_offsets fn = gar_factor_layout(mn.nx, mn.nu, mn.nc, 2 * mn.nx2, mn.nth);
      const double *frn = fac + mn.fac_off;
      Pn = frn + fn.Vxx;
      pn = frn + fn.vx;
      Pxtn = (mn.nth == nth) ? frn + fn.Vxt : nullptr;
    }
    // ---- KKT matrix (dense-kernel.hpp:102-113 ; terminal :59-63) ----
    // Only the lower triangle is ever read by the factorisation: it is stored packed by columns
    // (half the LDS, two problems per CU at the north-star shape).  Zero fill, then one
    // branch-free loop per block, so that the global loads of a loop are in flight together.
    const int npk = n * (n + 1) / 2;
    for (int e = w.tid; e < npk; e += w.nthr)
      Kmat[e] = 0.0;
    for (int e = w.tid; e < n * rld; e += w.nthr)
      Rm[e] = 0.0;
    wg_bar(w);
#define KP(i, j) Kmat[bk_idx<GAR_PACKED_LOWER>((i), (j), n)]
#define RR(i, c) Rm[(i) * rld + (c)]
#pragma unroll 4
    for (int e = w.tid; e < nu * nu; e += w.nthr) { // R
      const int j = e / nu, i = e - j * nu;
      const double v = k[ko.R + e];
      if (i >= j)
        KP(i, j) = v;
    }
#pragma unroll 4
    for (int e = w.tid; e < nc * nu; e += w.nthr) { // D
      const int j = e / nc, i = e - j * nc;
      KP(o1 + i, j) = k[ko.D + e];
    }
    for (int i = w.tid; i < nc; i += w.nthr) // -mu I
      KP(o1 + i, o1 + i) = -mu;
    if (!term) {
#pragma unroll 4
      for (int e = w.tid; e < nx2 * nu; e += w.nthr) { // B
        const int j = e / nx2, i = e - j * nx2;
        KP(o2 + i, j) = k[ko.B + e];
      }
      for (int i = w.tid; i < nx2; i += w.nthr) // -I
        KP(o3 + i, o2 + i) = -1.0;
#pragma unroll 4
      for (int e = w.tid; e < nx2 * nx2; e += w.nthr) { // Pxx'
        const int j = e / nx2, i = e - j * nx2;
        const double v = Pn[e];
        if (i >= j)
          KP(o3 + i, o3 + j) = v;
      }
    }
    // ---- right-hand sides [ff | fb | ft], row-major (:119-140 ; terminal :65-76) ----
    for (int i = w.tid; i < nu; i += w.nthr)
      RR(i, 0) = -k[ko.r + i];
    for (int i = w.tid; i < nc; i += w.nthr)
      RR(o1 + i, 0) = -k[ko.d + i];
#pragma unroll 4
    for (int e = w.tid; e < nu * nx; e += w.nthr) { // -S^T
      const int u = e / nx, x = e - u * nx;
      RR(u, 1 + x) = -k[ko.S + e];
    }
#pragma unroll 4
    for (int e = w.tid; e < nc * nx; e += w.nthr) { // -C
      const int x = e / nc, i = e - x * nc;
      RR(o1 + i, 1 + x) = -k[ko.C + e];
    }
    if (!term) {
      for (int i = w.tid; i < nx2; i += w.nthr) {
        RR(o2 + i, 0) = -k[ko.f + i];
        RR(o3 + i, 0) = -pn[i];
      }
#pragma unroll 4
      for (int e = w.tid; e < nx2 * nx; e += w.nthr) { // -A
        const int x = e / nx2, i = e - x * nx2;
        RR(o2 + i, 1 + x) = -k[ko.A + e];
      }
    }
    if (nth > 0) {
      if (m.flags & GAR_KNOT_HAS_PARAM) {
        for (int e = w.tid; e < nu * nth; e += w.nthr) { // -Gu
          const int c = e / nu, i = e - c * nu;
          RR(i, 1 + nx + c) = -k[ko.Gu + e];
        }
        for (int e = w.tid; e < nc * nth; e += w.nthr) { // -Gv
          const int c = e / nc, i = e - c * nc;
          RR(o1 + i, 1 + nx + c) = -k[ko.Gv + e];
        }
      }
      if (!term && Pxtn)
        for (int e = w.tid; e < nx2 * nth; e += w.nthr) { // -Pxt'
          const int c = e / nx2, i = e - c * nx2;
          RR(o3 + i, 1 + nx + c) = -Pxtn[e];
        }
    }
    wg_bar(w);
    // 1. factorise  2. solve (:115, :142-144)
    if (n > 0) {
      if (P.dense.wk >= 0 && n >= 24 && n <= 128) // panel-blocked: pivot search by one wave, MFMA trailing updates
        failed |= wg_bk_factor_blocked<GAR_PACKED_LOWER>(w, n, Kmat, n, sub, piv, ctrl, sm + P.dense.wk);
      else
        failed |= wg_bk_factor<GAR_PACKED_LOWER>(w, n, Kmat, n, sub, piv, ctrl);
      wg_bk_solve<GAR_PACKED_LOWER>(w, n, Kmat, n, sub, piv, Rm, rld, 1, rld);
    }
    // gains to the factor record (the reference's storage orders: fb, ft row-major)
    for (int e = w.tid; e < nfull * rld; e += w.nthr) {
      const int i = e / rld, c = e - i * rld;
      const double v = (i < n) ? Rm[e] : 0.0;
      if (c == 0)
        fr[fo.ff + i] = v;
      else if (c <= nx)
        fr[fo.fb + i * nx + (c - 1)] = v;
      else
        fr[fo.fth + i * nth + (c - 1 - nx)] = v;
    }
    // 3. value function (:150-171 ; terminal :81-97).  R(i, c) = Rm[i * rld + c]:
    //   [px | Pxx] = [q | Q] + S [kff | K] + C^T [zff | Z] (+ A^T [lff | L])      on f64 MFMA tiles (wg_gemm), the
    // left factors read from the knot record in place (S column-major; C^T, A^T as transposed views), accumulating
    // in the factor record: a tile is read back by the thread that wrote it, so the three products need no barrier
    {
      const MatV Pxx = colmajor(fr + fo.Vxx, nx), px = colmajor(fr + fo.vx, nx);
      const MatV Sm = colmajor(const_cast<double *>(k) + ko.S, nx);
      const MatV Ct = MatV{const_cast<double *>(k) + ko.C, nc, 1}, At = MatV{const_cast<double *>(k) + ko.A, nx2, 1};
      const MatV Rg = rowmajor(Rm, rld); // rows: u (nu), v (nc), lambda' (nx2), x' (nx2); columns: ff | fb | ft
      wg_gemm(w, nx, nx, nu, Sm, Rg.sub(0, 1), colmajor(const_cast<double *>(k) + ko.Q, nx), Pxx, 1.0);
      wg_gemm(w, nx, 1, nu, Sm, Rg.sub(0, 0), colmajor(const_cast<double *>(k) + ko.q, nx), px, 1.0);
      wg_gemm(w, nx, nx, nc, Ct, Rg.sub(o1, 1), Pxx, Pxx, 1.0);
      wg_gemm(w, nx, 1, nc, Ct, Rg.sub(o1, 0), px, px, 1.0);
      if (!term) {
        wg_gemm(w, nx, nx, nx2, At, Rg.sub(o2, 1), Pxx, Pxx, 1.0);
        wg_gemm(w, nx, 1, nx2, At, Rg.sub(o2, 0), px, px, 1.0);
      }
    }
    if (nth > 0) {
      const bool stored = (m.flags & GAR_KNOT_HAS_PARAM) != 0;
      // Pxt (nx x nth), Ptt (nth x nth), pt (nth): rows i < nx -> Pxt, nx <= i < nx+nth -> Ptt,
      // i = nx+nth -> pt; left factor = column (1+i) of R for Pxt/Ptt, column 0 for pt
      for (int e = w.tid; e < (nx + nth + 1) * nth; e += w.nthr) {
        const int c = e / (nx + nth + 1), i = e - c * (nx + nth + 1);
        const int col = (i < nx + nth) ? 1 + i : 0;
        double s = 0.0;
        if (stored) {
          if (i < nx)
            s = k[ko.Gx + i + c * nx];
          else if (i < nx + nth)
            s = k[ko.Gth + (i - nx) + c * nth];
          else
            s = k[ko.gamma + c];
          for (int r = 0; r < o1; ++r)
            s += Rm[r * rld + col] * k[ko.Gu + r + c * nu];
          for (int r = o1; r < o2; ++r)
            s += Rm[r * rld + col] * k[ko.Gv + (r - o1) + c * nc];
        }
        if (!term && Pxtn)
          for (int r = 0; r < nx2; ++r)
            s += Rm[(o3 + r) * rld + col] * Pxtn[r + c * nx2];
        if (i < nx)
          fr[fo.Vxt + i + c * nx] = s;
        else if (i < nx + nth)
          fr[fo.Vtt + (i - nx) + c * nth] = s;
        else
          fr[fo.vt + c] = s;
      }
    }
    wg_bar(w); // the record is read back by the next stage; LDS is reused
#undef KP
#undef RR
  }
  // initial stage (dense-riccati.hxx:67-88): the same kkt0 as ProximalRiccatiSolver's
  const gar_stage_meta m0 = P.meta[0];
  const gar_factor_offsets f0 = gar_factor_layout(m0.nx, m0.nu, m0.nc, 2 * m0.nx2, m0.nth);
  const double *r0 = fac + m0.fac_off;
  const int n0 = m0.nx + P.nc0;
  double *k0mat = sm, *k0rhs = k0mat + n0 * n0 + (n0 & 1), *k0sub = k0rhs + n0 * (1 + m0.nth) + ((n0 * (1 + m0.nth)) & 1);
  int *piv0 = (int *)(k0sub + n0 + (n0 & 1));
  failed |= initial_stage_ptr(w, P, b, m0.nx, m0.nth, r0 + f0.Vxx, r0 + f0.vx, r0 + f0.Vxt,
                              r0 + f0.Vtt, r0 + f0.vt, k0mat, k0rhs, k0sub, piv0, piv0 + 512);
  if (failed && w.tid == 0)
    atomicOr(&P.status[b], failed);
}

// forwardStep (dense-kernel.hpp:174-209) after xs[0], lbdas[0] from kkt0 (dense-riccati.hxx:99-104):
// thread r owns row r of [K; Z; L; Y]; u, v, lambda' go to the solution record, x' to LDS.
__global__ void __launch_bounds__(256) gar_forward_dense(GenericParams P) {
  const WG w = wg_self();
  double *sm = gar_smem;
  const int b = (int)blockIdx.x;
  const double *fac = P.fac + (long long)b * P.fac_stride;
  double *sol = P.sol + (long long)b * P.sol_stride;
  double *x = sm + P.lds.fx, *xn = sm + P.lds.fxn, *th = sm + P.lds.fth;
  const gar_stage_meta m0 = P.meta[0];
  const int nth0 = m0.nth;
  const bool have_theta = (P.theta != nullptr) && nth0 > 0;
  {
    const int nx = m0.nx, n0 = nx + P.nc0;
    const double *io = P.init + (long long)b * P.init_stride;
    if (have_theta)
      for (int e = w.tid; e < nth0; e += w.nthr)
        th[e] = P.theta[(long long)b * nth0 + e];
    wg_bar(w);
    for (int i = w.tid; i < n0; i += w.nthr) {
      double s = io[i];
      if (have_theta)
        for (int q = 0; q < nth0; ++q)
          s += io[n0 + i * nth0 + q] * th[q];
      if (i < nx) {
        x[i] = s;
        sol[m0.x_off + i] = s;
      } else {
        sol[m0.l_off + (i - nx)] = s;
      }
    }
    wg_bar(w);
  }
  for (int t = 0; t <= P.horizon; ++t) {
    const gar_stage_meta m = P.meta[t];
    const int nx = m.nx, nu = m.nu, nc = m.nc, nx2 = m.nx2, nth = m.nth;
    const bool term = (t == P.horizon);
    const int o2 = nu + nc, o3 = o2 + nx2;
    const int rows = term ? o2 : o3 + nx2;
    const gar_factor_offsets fo = gar_factor_layout(nx, nu, nc, 2 * nx2, nth);
    const double *fr = fac + m.fac_off;
    int xo = 0, lo = 0;
    if (!term) {
      xo = P.meta[t + 1].x_off;
      lo = P.meta[t + 1].l_off;
    }
    for (int r = w.tid; r < rows; r += w.nthr) {
      double s = fr[fo.ff + r];
      const double *row = fr + fo.fb + (long long)r * nx;
      for (int j = 0; j < nx; ++j)
        s += row[j] * x[j];
      if (have_theta && nth == nth0)
        for (int q = 0; q < nth; ++q)
          s += fr[fo.fth + r * nth + q] * th[q];
      if (r < nu)
        sol[m.u_off + r] = s;
      else if (r < o2)
        sol[m.v_off + (r - nu)] = s;
      else if (r < o3)
        sol[lo + (r - o2)] = s;
      else {
        xn[r - o3] = s;
        sol[xo + (r - o3)] = s;
      }
    }
    wg_bar(w);
    double *tmp = x;
    x = xn;
    xn = tmp;
  }
}

} // namespace gar
