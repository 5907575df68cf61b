// gar_condensed_cr.hpp -- BLOCK CYCLIC REDUCTION of the reduced condensed system on the any-dimension path.
//
// After gar_condensed_leg_eliminate (gar_generic.hpp) the condensed leg-boundary system of ParallelRiccatiSolver
// (assembleCondensedSystem, parallel-solver.hxx:85-129) is block tridiagonal in (lbd0, th_0 .. th_{J-2}): J blocks,
// block 0 of dimension nc0, the others of dimension n = nxb, negative definite.  The chain kernel
// (gar_condensed_generic, `reduced`) eliminates them one after the other on ONE workgroup: J dependent steps of a
// factorisation, a substitution and a product each.  Here the same system is reduced level by level -- O(log J)
// dependent steps, every block of a level on its own workgroup:
//
//   row i :  C_{i-h}^T z_{i-h} + S_i z_i + C_i z_{i+h} = r_i          (h = 1: S = diag, C = super, r = rhs)
//
//   level h = 1, 2, 4, ... (< J): the blocks j at ODD multiples of h leave the system,
//     (A) every such j at once:   S_j [A_j | B_j | y_j] = [C_{j-h}^T | C_j | r_j]     (one factorisation -- unpivoted
//         blocked L D L^T, the blocks being definite, Bunch-Kaufman otherwise -- and one substitution of 2n+1 columns)
//     (B) every survivor i (even multiples of h) at once folds its two neighbours in -- products only:
//           S_i -= C_i A_{i+h} + C_{i-h}^T B_{i-h}     r_i -= C_i y_{i+h} + C_{i-h}^T y_{i-h}     C_i <- -C_i B_{i+h}
//   back-substitution, ONE workgroup: z_0 = S_0^{-1} r_0 (block 0 survives every level), then level by level from
//     the top  z_j = y_j - A_j z_{j-h} - B_j z_{j+h}: matrix-vector products, all blocks of a level at once.
//
// No refinement and no residual here: gar_condensed_leg_states evaluates the residual of EVERY row of the full
// system from the tuples (the quantity parallel-solver.hxx:191 tests) and the full chain runs gated behind it, in the
// reference's order (block-tridiagonal.hpp:82-138 with refinement), for whatever misses the threshold; a block that
// will not factorise poisons its y_j with NaN, which reaches that residual.  Same standing as the cyclic reduction
// of the specialised families (gar_cyclic.hpp; DESIGN.md section 2, deviation 2).
//
// Scratch (CondensedParams; pitch nblk = 2 J blocks): S_i = diag[i], C_i = super[i], r_i = rhs[i] (i < J; updated in
// place), A_j = U[j], B_j = U[J + j], y_j = fsub[j]; the solution goes where the reduced chain leaves it
// (err[J ..]).  diag/super/facD/rhs/fsub[J + l] hold what gar_condensed_leg_eliminate left (P_l, Y_l, W_l, z_l, c_l).
#pragma once
#include "gar_generic.hpp"

namespace gar {

// (debug build -DGAR_CTRACE: cycles per phase of workgroup (0, 0) of each kernel, accumulated over the launches since
// the last read with gar_hip_debug_crtrace -- scripts/ctrace_condensed_cr.py)
#ifdef GAR_CTRACE
__device__ long long g_crtrace[16];
#define CRT0() long long tprev = clock64();
#define CRT(id)                                                                                                        \
  {                                                                                                                    \
    __syncthreads();                                                                                                   \
    const long long now_ = clock64();                                                                                  \
    if (w.tid == 0 && blockIdx.x == 0 && blockIdx.y == 0)                                                              \
      g_crtrace[id] += now_ - tprev;                                                                                   \
    tprev = now_;                                                                                                      \
  }
#else
#define CRT0()
#define CRT(id)
#endif

struct CrScratch {
  double *diag, *super, *facD, *U, *fsub, *rhs, *err, *info;
  __device__ CrScratch(const CondensedParams &P, int b) {
    const int n = P.nxb, nblk = 2 * P.num_legs;
    const long long bs = (long long)n * n;
    diag = P.scratch + (long long)b * P.scratch_stride;
    super = diag + nblk * bs;
    facD = super + nblk * bs;
    U = facD + nblk * bs;
    fsub = U + nblk * bs;
    rhs = fsub + nblk * n;
    err = rhs + nblk * n;
    info = err + 2 * nblk * n;
  }
};

// ---- the reduced system, one workgroup per block (the chain kernel assembles it serially) ---------------------
// grid (J, batch) x GAR_CONDENSED_THREADS (block 0 is three products: a tile per wave); LDS: nc0 x n doubles
__global__ void __launch_bounds__(GAR_CONDENSED_THREADS) gar_condensed_cr_assemble(CondensedParams P) {
  const WG w = wg_self();
  CRT0()
  const int i = (int)blockIdx.x, b = (int)blockIdx.y, J = P.num_legs;
  const int n = P.nxb, bs = n * n, nc0 = P.nc0;
  const CrScratch X(P, b);
  if (i == 0) {
    // diag(lbd0) = -G0 P_0 G0^T,  off(lbd0, th_0) = -G0 Y_0,  rhs = -g0 - G0 z_0
    double *prob = const_cast<double *>(P.prob) + (long long)b * P.prob_stride;
    const MatV G0 = colmajor(prob + P.G0_off, nc0);
    double *GP = gar_smem;
    wg_gemm(w, nc0, n, n, G0, colmajor(X.diag + (long long)J * bs, n), MatV{nullptr, 0, 0}, colmajor(GP, nc0), 1.0);
    __syncthreads();
    wg_gemm(w, nc0, nc0, n, colmajor(GP, nc0), G0.T(), MatV{nullptr, 0, 0}, colmajor(X.diag, nc0), -1.0);
    wg_gemm(w, nc0, n, n, G0, colmajor(X.super + (long long)J * bs, n), MatV{nullptr, 0, 0}, colmajor(X.super, nc0), -1.0);
    const int srow = w.tid >> 2, sq = w.tid & 3, srows = w.nthr >> 2;
    for (int i0 = 0; i0 < nc0; i0 += srows) { // rhs = -g0 - G0 z_0: four threads per row
      const int e = i0 + srow, ec = e < nc0 ? e : nc0 - 1;
      double sum = gar_sliced_dot(prob + P.G0_off + ec, nc0, X.rhs + J * n, n, sq);
      sum += __shfl_xor(sum, 1);
      sum += __shfl_xor(sum, 2);
      if (sq == 0 && e < nc0)
        X.rhs[e] = -prob[P.g0_off + e] - sum;
    }
    CRT(12)
    return;
  }
  // block i = th_l, l = i - 1:  diag = Vtt_l - W_l - P_{l+1},  off(th_l, th_{l+1}) = Y_{l+1},
  //                             rhs = -vt_l - c_l + z_{l+1}
  const int l = i - 1;
  const double *tup = cond_tuple(P, b, l);
  const double *Wl = X.facD + (long long)(J + l) * bs, *Pn = X.diag + (long long)(J + l + 1) * bs;
  const double *Yn = X.super + (long long)(J + l + 1) * bs;
  double *d = X.diag + (long long)i * bs, *s = X.super + (long long)i * bs;
  const bool coupled = i + 1 < J;
#pragma unroll 4
  for (int e = w.tid; e < bs; e += w.nthr) {
    d[e] = tup[2 * bs + e] - Wl[e] - Pn[e];
    if (coupled)
      s[e] = Yn[e];
  }
  for (int e = w.tid; e < n; e += w.nthr)
    X.rhs[i * n + e] = -tup[3 * bs + n + e] - X.fsub[(J + l) * n + e] + X.rhs[(J + l + 1) * n + e];
}

// ---- (A) the blocks that leave the system at level h ----------------------------------------------------------
// grid (number of odd multiples of h below J, batch) x GAR_CONDENSED_THREADS; LDS as gar_condensed_leg_eliminate
__global__ void __launch_bounds__(GAR_CONDENSED_THREADS) gar_condensed_cr_eliminate(CondensedParams P, int h) {
  const WG w = wg_self();
  CRT0()
  double *sm = gar_smem;
  const int j = (2 * (int)blockIdx.x + 1) * h, b = (int)blockIdx.y, J = P.num_legs;
  if (j >= J)
    return;
  const int n = P.nxb, bs = n * n;
  const int rl = (j - h == 0) ? P.nc0 : n; // dimension of the left neighbour
  const bool right = j + h < J;
  const CrScratch Xs(P, b);
  const double *Sj = Xs.diag + (long long)j * bs, *Cl = Xs.super + (long long)(j - h) * bs;
  const double *Cj = Xs.super + (long long)j * bs;
  const int ncol = rl + (right ? n : 0) + 1; // [C_{j-h}^T | C_j | r_j]
  double *X = sm, *R = X + bs, *wk = R + ((n * (2 * n + 1) + 1) & ~1), *sub = wk + GAR_LDL_PANEL * n;
  int *piv = (int *)(sub + n + (n & 1)), *ctrl = piv + n + 8;
  for (int e = w.tid; e < bs; e += w.nthr) {
    X[e] = Sj[e];
    if (right)
      R[rl * n + e] = Cj[e];
  }
  for (int e = w.tid; e < rl * n; e += w.nthr) { // C_{j-h} is rl x n, leading dimension rl (read along its columns)
    const int a = e % rl, bb = e / rl;
    R[a * n + bb] = Cl[e];
  }
  for (int e = w.tid; e < n; e += w.nthr)
    R[(ncol - 1) * n + e] = Xs.rhs[j * n + e];
  __syncthreads();
  CRT(0)
  int bad = 1;
  if (n >= 8 && n <= 64) {
    bad = wg_ldl_definite_factor(w, n, X, n, sub, piv, wk, ctrl);
    if (bad) {
      for (int e = w.tid; e < bs; e += w.nthr)
        X[e] = Sj[e];
      __syncthreads();
    }
  }
  if (bad)
    bad = wg_bk_factor(w, n, X, n, sub, piv, ctrl);
  CRT(1)
  wg_bk_solve(w, n, X, n, sub, piv, R, 1, n, ncol);
  __syncthreads();
  CRT(2)
  double *Aj = Xs.U + (long long)j * bs, *Bj = Xs.U + (long long)(J + j) * bs, *yj = Xs.fsub + j * n;
  for (int e = w.tid; e < rl * n; e += w.nthr)
    Aj[e] = R[e];
  if (right)
    for (int e = w.tid; e < bs; e += w.nthr)
      Bj[e] = R[rl * n + e];
  for (int e = w.tid; e < n; e += w.nthr) // a block that would not factorise: NaN reaches the residual gate
    yj[e] = bad ? __longlong_as_double(0x7ff8000000000000ll) : R[(ncol - 1) * n + e];
  CRT(3)
}

// ---- (B) the survivors of level h fold their neighbours in ----------------------------------------------------
// grid (number of multiples of 2h below J, batch) x GAR_CONDENSED_THREADS.  staged: the operands of a side (C, A or B
// of the neighbour, y) are brought into LDS with every load in flight at once and the products read them there -- a
// product whose operands sit in global memory pays a round trip per 16 columns of K and tile.  LDS: 4 n x n + n
// doubles (staged; the host decides by what fits a CU), else n x n.
__host__ __device__ inline int gar_condensed_cr_update_lds_doubles(int nxb, int staged) {
  return staged ? 4 * nxb * nxb + nxb + 2 : nxb * nxb;
}
__global__ void __launch_bounds__(GAR_CONDENSED_THREADS) gar_condensed_cr_update(CondensedParams P, int h, int staged) {
  const WG w = wg_self();
  CRT0()
  const int i = 2 * h * (int)blockIdx.x, b = (int)blockIdx.y, J = P.num_legs;
  if (i >= J)
    return;
  const int n = P.nxb, bs = n * n;
  const int r = i == 0 ? P.nc0 : n;
  const CrScratch X(P, b);
  double *Si = X.diag + (long long)i * bs, *Ci = X.super + (long long)i * bs, *ri = X.rhs + i * n;
  double *tmp = gar_smem; // the new coupling C_i (r x n) until every product has read the old one
  double *Lc = tmp + bs, *La = Lc + bs, *Lb = La + bs, *Ly = Lb + bs; // (staged only)
  const int jr = i + h, jl = i - h;
  if (jr < J) {
    const bool again = jr + h < J; // i keeps a right neighbour at the next level
    double *Ag = X.U + (long long)jr * bs, *Bg = X.U + (long long)(J + jr) * bs, *yg = X.fsub + jr * n;
    MatV C = colmajor(Ci, r), A = colmajor(Ag, n), B = colmajor(Bg, n), y = colmajor(yg, n);
    if (staged) {
#pragma unroll 4
      for (int e = w.tid; e < bs; e += w.nthr) {
        if (e < r * n) {
          Lc[e] = Ci[e];
          La[e] = Ag[e];
        }
        if (again)
          Lb[e] = Bg[e];
      }
      for (int e = w.tid; e < n; e += w.nthr)
        Ly[e] = yg[e];
      __syncthreads();
      C = colmajor(Lc, r), A = colmajor(La, n), B = colmajor(Lb, n), y = colmajor(Ly, n);
    }
    wg_gemm(w, r, r, n, C, A, colmajor(Si, r), colmajor(Si, r), -1.0);
    wg_gemm(w, r, 1, n, C, y, colmajor(ri, r), colmajor(ri, r), -1.0);
    if (again)
      wg_gemm(w, r, n, n, C, B, MatV{nullptr, 0, 0}, colmajor(tmp, r), -1.0);
    __syncthreads();
    if (again)
      for (int e = w.tid; e < r * n; e += w.nthr)
        Ci[e] = tmp[e];
  }
  CRT(4)
  if (jl >= 0) { // (i >= 2h: r == n; C_{jl} is n x n)
    double *Cg = X.super + (long long)jl * bs, *Bg = X.U + (long long)(J + jl) * bs, *yg = X.fsub + jl * n;
    MatV Ct = colmajor(Cg, n).T(), B = colmajor(Bg, n), y = colmajor(yg, n);
    if (staged) {
#pragma unroll 4
      for (int e = w.tid; e < bs; e += w.nthr) {
        Lc[e] = Cg[e];
        Lb[e] = Bg[e];
      }
      for (int e = w.tid; e < n; e += w.nthr)
        Ly[e] = yg[e];
      __syncthreads();
      Ct = colmajor(Lc, n).T(), B = colmajor(Lb, n), y = colmajor(Ly, n);
    }
    wg_gemm(w, n, n, n, Ct, B, colmajor(Si, n), colmajor(Si, n), -1.0);
    wg_gemm(w, n, 1, n, Ct, y, colmajor(ri, n), colmajor(ri, n), -1.0);
  }
  CRT(5)
}

// slice q of  sum_k a[k astride] x[k] (k < Ka) + sum_k b[k astride] y[k] (k < Kb): every load issued before the first
// product (gar_sliced_dot twice is two dependent round trips); Ka, Kb may be 0 (a, b must still point at memory)
__device__ __forceinline__ double gar_sliced_dot2(const double *a, const double *x, int Ka, const double *b, const double *y,
                                                  int Kb, int astride, int q) {
  double s = 0.0;
  const int K = Ka > Kb ? Ka : Kb;
  for (int k0 = 0; k0 < K; k0 += 64) {
    double av[16], bv[16], xv[16], yv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int k = k0 + q + 4 * j;
      const int ka = k < Ka ? k : (Ka > 0 ? Ka - 1 : 0), kb = k < Kb ? k : (Kb > 0 ? Kb - 1 : 0);
      av[j] = a[(long long)ka * astride];
      bv[j] = b[(long long)kb * astride];
      xv[j] = x[ka];
      yv[j] = y[kb];
      av[j] = k < Ka ? av[j] : 0.0;
      bv[j] = k < Kb ? bv[j] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
      s += av[j] * xv[j] + bv[j] * yv[j];
  }
  return s;
}

// ---- the top block and the back-substitution, one workgroup per problem ---------------------------------------
// LDS: X (n x n) | rhs (n) | wk (GAR_LDL_PANEL n) | sub (n) | piv, ctrl | z (J n)
__host__ __device__ inline int gar_condensed_cr_back_lds_doubles(int nxb, int J) {
  return nxb * nxb + nxb + GAR_LDL_PANEL * nxb + nxb + 2 + (nxb + 48) / 2 + 2 + J * nxb + 2;
}
__global__ void __launch_bounds__(GAR_CONDENSED_THREADS) gar_condensed_cr_back(CondensedParams P) {
  const WG w = wg_self();
  CRT0()
  double *sm = gar_smem;
  const int b = (int)blockIdx.x, J = P.num_legs;
  const int n = P.nxb, bs = n * n, n0 = P.nc0;
  const CrScratch Xs(P, b);
  double *X = sm, *rb = X + bs, *wk = rb + n, *sub = wk + GAR_LDL_PANEL * n;
  int *piv = (int *)(sub + n + (n & 1)), *ctrl = piv + n + 8;
  double *z = sub + n + 2 + (n + 48) / 2 + 2;
  int bad = 0;
  if (n0 > 0) {
    for (int e = w.tid; e < n0 * n0; e += w.nthr)
      X[e] = Xs.diag[e];
    for (int e = w.tid; e < n0; e += w.nthr)
      rb[e] = Xs.rhs[e];
    __syncthreads();
    CRT(7)
    bad = 1;
    if (n0 >= 8 && n0 <= 64) {
      bad = wg_ldl_definite_factor(w, n0, X, n0, sub, piv, wk, ctrl);
      if (bad) {
        for (int e = w.tid; e < n0 * n0; e += w.nthr)
          X[e] = Xs.diag[e];
        __syncthreads();
      }
    }
    if (bad)
      bad = wg_bk_factor(w, n0, X, n0, sub, piv, ctrl);
    CRT(8)
    wg_bk_solve(w, n0, X, n0, sub, piv, rb, 1, 0, 1);
    __syncthreads();
    CRT(9)
  }
  for (int e = w.tid; e < n; e += w.nthr)
    z[e] = e < n0 ? (bad ? __longlong_as_double(0x7ff8000000000000ll) : rb[e]) : 0.0;
  __syncthreads();
  int hmax = 1;
  while (2 * hmax < J)
    hmax *= 2;
  const int srow = w.tid >> 2, sq = w.tid & 3, srows = w.nthr >> 2;
  for (int h = hmax; h >= 1; h >>= 1) {
    const int m = (J - 1 + h) / (2 * h), rows = m * n; // the blocks eliminated at level h: j = (2 q + 1) h < J
    for (int i0 = 0; i0 < rows; i0 += srows) {
      const int idx = i0 + srow, ic = idx < rows ? idx : rows - 1;
      const int q = ic / n, row = ic - q * n, j = (2 * q + 1) * h;
      const int rl = (j - h == 0) ? n0 : n, rr = (j + h < J) ? n : 0;
      // (y_j and both rows requested before anything is consumed: one round trip per level, not three)
      const double yj = Xs.fsub[j * n + row];
      double sum = gar_sliced_dot2(Xs.U + (long long)j * bs + row, z + (j - h) * n, rl,
                                   Xs.U + (long long)(J + j) * bs + row, z + (rr ? j + h : j) * n, rr, n, sq);
      sum += __shfl_xor(sum, 1);
      sum += __shfl_xor(sum, 2);
      if (sq == 0 && idx < rows)
        z[j * n + row] = yj - sum;
    }
    __syncthreads();
  }
  CRT(10)
  double *sol = Xs.err + J * n; // (lbd0, th_0 .. th_{J-2}): where gar_condensed_leg_states reads it
  for (int e = w.tid; e < J * n; e += w.nthr)
    sol[e] = z[e];
  if (w.tid == 0) { // gar_condensed_leg_states accumulates the residual and the scale of every row of the full system
    Xs.info[0] = 0.0;
    Xs.info[1] = 0.0;
    Xs.info[2] = 0.0;
  }
  CRT(11)
}

} // namespace gar
