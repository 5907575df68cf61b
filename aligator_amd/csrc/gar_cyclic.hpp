// gar_cyclic.hpp -- the condensed (leg-boundary) system of ParallelRiccatiSolver solved by
// BLOCK CYCLIC REDUCTION: O(log J) dependent steps instead of the 2J-step elimination chain of
// symmetricBlockTridiagSolve (gar/block-tridiagonal.hpp:82-138), so that the number of legs (and
// with it the parallelism of the leg sweeps) is no longer paid for in the condensed solve.
//
// The system (assembleCondensedSystem, parallel-solver.hxx:85-129), unknowns (lambda_k, x_k) per
// leg k = 0..J-1 (lambda_0: multiplier of the initial condition):
//     lambda_0 row :  G0 x_0                                              = -g0
//     x_k row      :  E_k^T lambda_k + Vxx_k x_k + Vxt_k lambda_{k+1}    = -vx_k    (E_0 = G0, else -I)
//     lambda_{k+1} :  Vxt_k^T x_k + Vtt_k lambda_{k+1} - x_{k+1}         = -vt_k
//  1. setup (one wave per leg, all legs at once): Vxx_k is positive definite, so every x_k is
//     eliminated first.  With P = Vxx^{-1}, Q = P Vxt, p = P vx this leaves a block-tridiagonal
//     system in the multipliers alone,
//         S_0 = -G0 P_0 G0^T                         S_{k+1} = Vtt_k - Vxt_k^T Q_k - P_{k+1}
//         C_0 = -G0 Q_0  (row 0, column 1)           C_k     = Q_k                    (k >= 1)
//         r_0 = -g0 + G0 p_0                         r_{k+1} = -vt_k + Vxt_k^T p_k - p_{k+1}
//     whose diagonal blocks are negative definite (-P_{k+1} < 0, the rest <= 0), as are all their
//     Schur complements: any elimination order is safe.
//  2. reduce, level h = 1, 2, 4, ...: the blocks at odd multiples of h are eliminated; every
//     survivor i (multiple of 2h) folds in its two eliminated neighbours,
//         S_i -= C_i W_{i+h} C_i^T + C_{i-h}^T W_{i-h} C_{i-h}         (W_j = S_j^{-1})
//         C_i <- -C_i W_{i+h} C_{i+h}     r_i -= C_i W_{i+h} r_{i+h} + C_{i-h}^T W_{i-h} r_{i-h}
//     one wave per survivor, ONE launch per level, no wave writes what another reads in the
//     same level (a survivor inverts both neighbours itself).
//  3. back-substitution, z_0 = S_0^{-1} r_0 ;  z_j = W_j (r_j - Cl_j^T z_{j-h} - C_j z_{j+h})  level
//     by level: the top levels (a handful of blocks) in one workgroup, the wide ones a launch each;
//  4. recovery of the states, x_k = -p_k - P_k E_k^T lambda_k - Q_k lambda_{k+1}, and the residual
//     of the ORIGINAL system (the quantity parallel-solver.hxx:191 tests), one wave per leg; if it
//     exceeds the threshold the elimination-chain kernel (gar_condensed_wave, with the reference's
//     iterative refinement) re-solves the problem.
// Arithmetic differs from the reference's elimination order; parity is to the stated fp64
// tolerance, checked by the same residual the reference checks.
#pragma once
#include "gar_wave_leg.hpp"
#include "gar_ldl_blocked.hpp"

namespace gar {

struct CyclicParams {
  CondensedParams C; // tuples, problems, scratch, csol, status, dims
  int h;             // reduce / back-substitution level stride (top kernel: the last level it runs)
};

// scratch carve (doubles, per problem), J = num_legs, bs = NX*NX; fits in the 8*J*bs + ... of the
// elimination-chain kernel's scratch
template <int NX> struct CyclicScratch {
  static constexpr int bs = NX * NX;
  double *P, *Q, *S, *C, *W, *Cl, *r, *z, *p, *info;
  __device__ CyclicScratch(double *base, int J) {
    P = base;
    Q = P + (long long)J * bs;
    S = Q + (long long)J * bs;
    C = S + (long long)J * bs;
    W = C + (long long)J * bs;
    Cl = W + (long long)J * bs;
    r = Cl + (long long)J * bs;
    z = r + (long long)J * NX;
    p = z + (long long)J * NX;
    info = base + 8ll * J * bs + 8ll * J * NX; // same slot as the chain kernel's
  }
};

template <int NX> __device__ __forceinline__ double cyc_matvec(const double *Mc, double x, int row) {
  double s0 = 0.0, s1 = 0.0; // sum_k M(row, k) x_k, M column-major pitch NX
#pragma unroll
  for (int k = 0; k < NX; k += 2) {
    s0 = __builtin_fma(Mc[k * NX + row], lane_bcast(x, k), s0);
    s1 = __builtin_fma(Mc[(k + 1) * NX + row], lane_bcast(x, k + 1), s1);
  }
  return s0 + s1;
}
// sum_k |M(row, k)| |x_k| and its transpose: the denominators of the componentwise backward error
template <int NX> __device__ __forceinline__ double cyc_absmatvec(const double *Mc, double x, int row) {
  double s0 = 0.0;
  const double ax = fabs(x);
#pragma unroll
  for (int k = 0; k < NX; ++k)
    s0 = __builtin_fma(fabs(Mc[k * NX + row]), lane_bcast(ax, k), s0);
  return s0;
}
template <int NX> __device__ __forceinline__ double cyc_absmatvecT(const double *Mc, double x, int row) {
  double s0 = 0.0;
  const double ax = fabs(x);
#pragma unroll
  for (int k = 0; k < NX; ++k)
    s0 = __builtin_fma(fabs(Mc[row * NX + k]), lane_bcast(ax, k), s0);
  return s0;
}
template <int NX> __device__ __forceinline__ double cyc_matvecT(const double *Mc, double x, int row) {
  double s0 = 0.0, s1 = 0.0; // sum_k M(k, row) x_k
#pragma unroll
  for (int k = 0; k < NX; k += 2) {
    s0 = __builtin_fma(Mc[row * NX + k], lane_bcast(x, k), s0);
    s1 = __builtin_fma(Mc[row * NX + k + 1], lane_bcast(x, k + 1), s1);
  }
  return s0 + s1;
}

// LDS plan of the setup / reduce kernels (one wave)
template <int NX> struct CyclicLds {
  static constexpr int bs = NX * NX;
  static constexpr int oD = 0, oW = bs, oB = 2 * bs, oM = 3 * bs, oD2 = 4 * bs, oT = 5 * bs,
                       oDl = oT + 16 * NX, total = (oDl + NX + 1) & ~1;
};

// W = D^{-1} of the symmetric DEFINITE block in Dm (lower triangle read): unpivoted register
// LDL^T (every block this solver inverts is definite -- Vxx_k > 0, S_j < 0 and its Schur
// complements -- so no pivoting is needed for stability) + blocked inverse.  Returns 1 on a zero or
// non-finite pivot: the caller then poisons the residual, which hands the problem to the
// elimination-chain kernel (Bunch-Kaufman pivoting, refinement).
// Round 6: blocks wider than one DPP row (36, 32) are factorised in panels of 12 / 16 columns with the trailing
// matrix updated on MFMA tiles (gar_ldl_blocked.hpp; 630 v_readlane broadcast-FMA pairs at NX = 36 otherwise): L goes
// straight to LDS where the blocked inverse wants it.  Dm is consumed.
#ifndef GAR_CYC_BLOCKED_LDL
#define GAR_CYC_BLOCKED_LDL 1
#endif
template <int NX>
__device__ __forceinline__ int cyc_inverse(double *sm, int lane) {
  using L = CyclicLds<NX>;
  double *Dm = sm + L::oD, *Wm = sm + L::oW, *Mm = sm + L::oM, *Tm = sm + L::oT, *Dl = sm + L::oDl;
  if constexpr (GAR_CYC_BLOCKED_LDL && NX > 16 && (CondCfg<NX>::BS == 12 || CondCfg<NX>::BS == 16)) {
    for (int e = lane; e < NX * NX; e += 64) { // unit diagonal, zeros above it: the strictly lower part is the routine's
      const int j = e / NX, i = e - j * NX;
      if (i <= j)
        Wm[e] = (i == j) ? 1.0 : 0.0;
    }
    bool ff;
    const int failed = wave_ldl_blocked<NX, CondCfg<NX>::BS, true, LdlColMajor<NX>, LdlColMajor<NX>, true>(Dm, Tm, Wm, Dl, lane, ff, false);
    wave_lds_order();
    const double dl = -Dl[lane < NX ? lane : NX - 1]; // -1 / d_k -> 1 / d_k
    wave_lds_order();
    if (lane < NX)
      Dl[lane] = dl;
    cond_inverse_from_lds<NX>(Wm, Mm, Wm, Tm, Dl, lane);
    wave_sync();
    return failed;
  } else {
    double a_row[NX], nd[NX];
    const int failed = wave_ldl_fast<NX, true>(Dm, lane, a_row, nd);
    cond_inverse<NX>(a_row, nd, Wm, Mm, Wm, Tm, Dl, lane);
    wave_sync();
    return failed;
  }
}
// a failed block inverse: the residual slot is set to +inf (atomicMax on the bit pattern)
__device__ __forceinline__ void cyc_poison(double *info) {
  atomicMax(reinterpret_cast<unsigned long long *>(info), 0x7ff0000000000000ull);
}

// U = W op(B)^T (kept in accumulator registers) and Dd -= op(B) U (lower tiles), W, B, Dd
// column-major pitch-NX blocks in LDS; op(B) = B or B^T.  U's accumulator registers are the B
// operand of the second product (D -> B identity of the f64 MFMA).
template <int NX, bool TRANSB>
__device__ __forceinline__ void cyc_update(const double *Wm, const double *Bm, double *Dd,
                                           double4_t (&Ut)[CondCfg<NX>::TX][CondCfg<NX>::TX],
                                           int lane) {
  using K = CondCfg<NX>;
  constexpr int TX = K::TX, KS = K::KS;
  const int li = lane & 15, lk = lane >> 4;
  double opW[TX][KS], opB[TX][KS];
#pragma unroll
  for (int t = 0; t < TX; ++t) {
    const int c = (16 * t + li) < NX ? (16 * t + li) : NX - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      opW[t][s] = Wm[(4 * s + lk) * NX + c];                                // W(16t+li, 4s+lk)
      opB[t][s] = TRANSB ? Bm[c * NX + 4 * s + lk] : Bm[(4 * s + lk) * NX + c]; // op(B)(16t+li, 4s+lk)
    }
  }
#pragma unroll
  for (int ta = 0; ta < TX; ++ta)
#pragma unroll
    for (int tb = 0; tb < TX; ++tb) {
      double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < KS; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(opW[ta][s], opB[tb][s], acc, 0, 0, 0);
      Ut[ta][tb] = acc;
    }
#pragma unroll
  for (int ta = 0; ta < TX; ++ta)
#pragma unroll
    for (int tb = 0; tb <= ta; ++tb) {
      double4_t acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ii = 16 * ta + lk + 4 * r, jj = 16 * tb + li;
        acc[r] = (16 * ta + 4 * r < NX) ? Dd[(jj < NX ? jj : NX - 1) * NX + (ii < NX ? ii : NX - 1)] : 0.0;
      }
#pragma unroll
      for (int s = 0; s < KS; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-opB[ta][s], Ut[s >> 2][tb][s & 3], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * ta + 4 * r < NX) {
          const int ii = 16 * ta + lk + 4 * r, jj = 16 * tb + li;
          if (ii < NX && jj < NX)
            Dd[jj * NX + ii] = acc[r];
        }
    }
}

// U = W B^T alone, in the accumulator layout and MFMA order of cyc_update<NX, false>'s first half (bitwise the same U)
template <int NX>
__device__ __forceinline__ void cyc_form_u(const double *Wm, const double *Bm,
                                           double4_t (&Ut)[CondCfg<NX>::TX][CondCfg<NX>::TX], int lane) {
  using K = CondCfg<NX>;
  constexpr int TX = K::TX, KS = K::KS;
  const int li = lane & 15, lk = lane >> 4;
  double opW[TX][KS], opB[TX][KS];
#pragma unroll
  for (int t = 0; t < TX; ++t) {
    const int c = (16 * t + li) < NX ? (16 * t + li) : NX - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      opW[t][s] = Wm[(4 * s + lk) * NX + c];
      opB[t][s] = Bm[(4 * s + lk) * NX + c];
    }
  }
#pragma unroll
  for (int ta = 0; ta < TX; ++ta)
#pragma unroll
    for (int tb = 0; tb < TX; ++tb) {
      double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < KS; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(opW[ta][s], opB[tb][s], acc, 0, 0, 0);
      Ut[ta][tb] = acc;
    }
}

// Out = sgn * U^T Cm, U in accumulator registers (element (4s+lk, 16ta+li) of U is register s&3 of
// tile (s>>2, ta): the A operand of the transposed product), Cm a column-major block in LDS;
// the result goes to a column-major block in global memory
template <int NX>
__device__ __forceinline__ void cyc_ut_times(const double4_t (&Ut)[CondCfg<NX>::TX][CondCfg<NX>::TX],
                                             const double *Cm, double *out, double sgn, int lane) {
  using K = CondCfg<NX>;
  constexpr int TX = K::TX, KS = K::KS;
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int tb = 0; tb < TX; ++tb) {
    const int c = (16 * tb + li) < NX ? (16 * tb + li) : NX - 1;
    double opC[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
      opC[s] = Cm[c * NX + 4 * s + lk]; // C(4s+lk, 16tb+li)
#pragma unroll
    for (int ta = 0; ta < TX; ++ta) {
      double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < KS; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ut[s >> 2][ta][s & 3], opC[s], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * ta + 4 * r < NX) {
          if (16 * tb + 15 < NX || 16 * tb + li < NX)
            out[(16 * tb + li) * NX + 16 * ta + 4 * r + lk] = sgn * acc[r];
        }
    }
  }
}

// accumulator tiles -> column-major block in global memory
template <int NX>
__device__ __forceinline__ void cyc_store_tiles(const double4_t (&Ut)[CondCfg<NX>::TX][CondCfg<NX>::TX],
                                                double *out, int lane) {
  constexpr int TX = CondCfg<NX>::TX;
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int ta = 0; ta < TX; ++ta)
#pragma unroll
    for (int tb = 0; tb < TX; ++tb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * ta + 4 * r < NX) {
          if (16 * tb + 15 < NX || 16 * tb + li < NX)
            out[(16 * tb + li) * NX + 16 * ta + 4 * r + lk] = Ut[ta][tb][r];
        }
}

// LDS block -> global block (all reads first)
template <int NX>
__device__ __forceinline__ void cyc_store_block(double *dst, const double *src_lds, int lane) {
  constexpr int bs = NX * NX, NCH = (bs + 63) / 64;
  double tmp[NCH];
#pragma unroll
  for (int q = 0; q < NCH; ++q) {
    const int e = 64 * q + lane;
    tmp[q] = src_lds[(64 * q + 63 < bs || e < bs) ? e : bs - 1];
  }
#pragma unroll
  for (int q = 0; q < NCH; ++q) {
    const int e = 64 * q + lane;
    if (64 * q + 63 < bs || e < bs)
      dst[e] = tmp[q];
  }
}
// a block in flight: global loads issued now, written to LDS later (the latency hides behind
// whatever runs in between)
template <int NX> struct BlockRegs {
  static constexpr int bs = NX * NX, NCH = (bs + 63) / 64;
  double v[NCH];
  __device__ __forceinline__ void issue(const double *a, int lane) {
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int e = 64 * q + lane;
      v[q] = a[(64 * q + 63 < bs || e < bs) ? e : bs - 1];
    }
  }
  __device__ __forceinline__ void issue_diff(const double *a, const double *b, bool sub, int lane) {
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int e = 64 * q + lane;
      const int ec = (64 * q + 63 < bs || e < bs) ? e : bs - 1;
      const double x = a[ec], y = sub ? b[ec] : 0.0;
      v[q] = x - y;
    }
  }
  __device__ __forceinline__ void commit(double *dst_lds, int lane) const {
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int e = 64 * q + lane;
      if (64 * q + 63 < bs || e < bs)
        dst_lds[e] = v[q];
    }
  }
};

// G0 (nc0 x NX, column-major pitch nc0, in the problem record) -> an NX x NX column-major LDS block padded with zero
// rows; every load is issued before the first LDS write (a loop of load -> use iterations pays the memory latency
// once per iteration on a lone wave)
template <int NX>
__device__ __forceinline__ void cyc_stage_G0(double *dst_lds, const double *G0, int nc0, int lane) {
  constexpr int bs = NX * NX, NCH = (bs + 63) / 64;
  double tmp[NCH];
#pragma unroll
  for (int q = 0; q < NCH; ++q) {
    const int e = 64 * q + lane;
    const int ec = (64 * q + 63 < bs || e < bs) ? e : bs - 1;
    const int j = ec / NX, r = ec - j * NX;
    tmp[q] = r < nc0 ? G0[j * nc0 + r] : 0.0;
  }
#pragma unroll
  for (int q = 0; q < NCH; ++q) {
    const int e = 64 * q + lane;
    if (64 * q + 63 < bs || e < bs)
      dst_lds[e] = tmp[q];
  }
}
// sum_k M(row, k) x_k and sum_k M(k, row) x_k with ONE accumulator in the order k = 0, 1, ...: the order of the
// loops over the rows of G0 these replace (zero padding adds exact zeros)
template <int NX> __device__ __forceinline__ double cyc_matvec_seq(const double *Mc, double x, int row) {
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < NX; ++k)
    s = __builtin_fma(Mc[k * NX + row], lane_bcast(x, k), s);
  return s;
}
template <int NX> __device__ __forceinline__ double cyc_matvecT_seq(const double *Mc, double x, int row) {
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < NX; ++k)
    s = __builtin_fma(Mc[row * NX + k], lane_bcast(x, k), s);
  return s;
}
template <int NX> __device__ __forceinline__ double cyc_absmatvecT_seq(const double *Mc, double ax, int row) {
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < NX; ++k)
    s = __builtin_fma(fabs(Mc[row * NX + k]), lane_bcast(ax, k), s);
  return s;
}
template <int NX> __device__ __forceinline__ double cyc_absmatvec_seq(const double *Mc, double ax, int row) {
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < NX; ++k)
    s = __builtin_fma(fabs(Mc[k * NX + row]), lane_bcast(ax, k), s);
  return s;
}

// ---- 1. setup: eliminate the states, one wave per leg -----------------------------------------
// Grid J + 2: leg 0 also owns the initial condition's row (S_0, r_0, C_0), which used to trail behind its wave's
// regular work and made the launch a third longer than any other leg needs (37 us against 25 at NX = 36).  Two extra
// waves take it over, each inverting Vxx_0 again for itself: role 1 (blockIdx.x == J) forms S_0 and r_0, role 2
// (blockIdx.x == J + 1) C_0.  Same operations on the same operands: the results are bitwise what one wave produced.
// LDS: the plan of CyclicLds + one block (G0).
template <int NX>
__global__ void __launch_bounds__(64, 1) gar_cyclic_setup(CyclicParams Y) {
  using L = CyclicLds<NX>;
  using K = CondCfg<NX>;
  constexpr int bs = NX * NX, TX = K::TX;
  const CondensedParams &P = Y.C;
  const int lane = (int)threadIdx.x & 63;
  const int b = (int)blockIdx.y, J = P.num_legs;
  const int role = (int)blockIdx.x < J ? 0 : (int)blockIdx.x - J + 1;
  const int k = role == 0 ? (int)blockIdx.x : 0;
  if (role == 2 && J < 2)
    return; // no C_0 without a second leg
  double *sm = gar_smem;
  double *Dm = sm + L::oD, *Wm = sm + L::oW, *Bm = sm + L::oB, *D2 = sm + L::oD2;
  double *Gm = sm + L::total; // G0 padded to NX rows (roles 1, 2): a block of its own, the inverse works in the plan's
  CyclicScratch<NX> X(P.scratch + (long long)b * P.scratch_stride, J);
  const double *prob = P.prob + (long long)b * P.prob_stride;
  const double *tup = cond_tuple(P, b, k);
  const int row = lane < NX ? lane : NX - 1;
  const int nc0 = P.nc0;
  const bool has_next = (k + 1 < J);
  const bool need_q = has_next && role != 1;
  // everything this wave reads is requested before the inverse (Vxt_k, Vtt_k and G0 used to be fetched behind it: a
  // second round trip to memory on a lone wave); none of the LDS blocks they land in is touched by the inverse
  BlockRegs<NX> rxx, rxt, rtt;
  rxx.issue(tup, lane); // Vxx_k
  if (need_q) {
    rxt.issue(tup + bs, lane);     // Vxt_k
    rtt.issue(tup + 2 * bs, lane); // Vtt_k
  }
  const double vxr = tup[3 * bs + row];
  const double vtr = (role == 0 && has_next) ? tup[3 * bs + NX + row] : 0.0;
  if (role != 0)
    cyc_stage_G0<NX>(Gm, prob + P.G0_off, nc0, lane);
  rxx.commit(Dm, lane);
  if (need_q) {
    rxt.commit(Bm, lane);
    rtt.commit(D2, lane);
  }
  if (role == 1)
    for (int e = lane; e < bs; e += 64) {
      const int j = e / NX, r = e - j * NX;
      D2[e] = (r == j && r >= nc0) ? -1.0 : 0.0;
    }
  wave_sync();
  int failed = cyc_inverse<NX>(sm, lane); // Wm = P_k
  if (role == 0)
    cyc_store_block<NX>(X.P + (long long)k * bs, Wm, lane);
  double pk = 0.0;
  if (role != 2) {
    pk = cyc_matvec<NX>(Wm, vxr, row); // p_k = P vx
    if (role == 0 && lane < NX)
      X.p[k * NX + lane] = pk;
  }
  double4_t Qt[TX][TX];
  if (need_q) {
    // Q = P Vxt = W (Vxt^T)^T ; D2 = Vtt - Vxt^T Q  -> the part of S_{k+1} this leg owns
    cyc_update<NX, true>(Wm, Bm, D2, Qt, lane);
    if (role == 0) {
      cyc_store_tiles<NX>(Qt, X.Q + (long long)k * bs, lane);
      if (k >= 1)
        cyc_store_tiles<NX>(Qt, X.C + (long long)k * bs, lane); // C_k = Q_k
      wave_sync();
      cyc_store_block<NX>(X.S + (long long)(k + 1) * bs, D2, lane);
      // r_{k+1} (this leg's part) = -vt_k + Vxt^T p_k
      const double rk = -vtr + cyc_matvecT<NX>(Bm, pk, row);
      if (lane < NX)
        X.r[(k + 1) * NX + lane] = rk;
    }
  }
  if (role != 0) {
    // S_0 = -G0 P_0 G0^T (padded with -I), C_0 = -G0 Q_0, r_0 = -g0 + G0 p_0
    const double *Mm = Gm;
    if (role == 1) {
      double4_t Zt[TX][TX];
      cyc_update<NX, false>(Wm, Mm, D2, Zt, lane); // Z = P G0^T ; D2 -= G0 Z
      wave_sync();
      cyc_store_block<NX>(X.S, D2, lane);
      double r0 = cyc_matvec<NX>(Mm, pk, row);
      r0 += (row < nc0) ? -prob[P.g0_off + row] : 0.0;
      if (lane < NX)
        X.r[lane] = r0;
    }
    if (role == 2) {
      // C_0 = -G0 Q_0 = -(G0^T)^T Q_0: the tiles of Z^T... computed as (G0^T)^T Q with G0^T's
      // accumulator-layout copy: G0^T = Mm^T, so use U := G0^T in tile registers
      double4_t Gt[TX][TX];
      const int li = lane & 15, lk = lane >> 4;
#pragma unroll
      for (int ta = 0; ta < TX; ++ta)
#pragma unroll
        for (int tb = 0; tb < TX; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r) { // U(16ta+lk+4r, 16tb+li) = G0^T(.,.) = G0(16tb+li, 16ta+lk+4r)
            const int ii = 16 * ta + lk + 4 * r, jj = 16 * tb + li;
            Gt[ta][tb][r] = (ii < NX && jj < NX) ? Mm[ii * NX + jj] : 0.0;
          }
      // Q_0 as a column-major LDS block: write the tiles to D2 (S_0 already stored)
      wave_sync();
#pragma unroll
      for (int ta = 0; ta < TX; ++ta)
#pragma unroll
        for (int tb = 0; tb < TX; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ii = 16 * ta + lk + 4 * r, jj = 16 * tb + li;
            if (16 * ta + 4 * r < NX) {
              if (ii < NX && jj < NX)
                D2[jj * NX + ii] = Qt[ta][tb][r];
            }
          }
      wave_sync();
      cyc_ut_times<NX>(Gt, D2, X.C, -1.0, lane); // C_0 = -(G0^T)^T Q_0
    }
  }
  if (failed && lane == 0)
    cyc_poison(X.info);
}

// ---- 2. one reduction level -------------------------------------------------------------------
// Three waves per survivor: wave 0 folds in the right neighbour, wave 1 the left one (each inverts its neighbour in
// its own LDS slice), wave 2 forms the survivor's new coupling -C_i W_{i+h} C_{i+h} from wave 0's inverse while
// wave 0 updates S_i (it used to follow that update on wave 0: 7.8 k of a level's 59 k cycles); wave 0 then adds
// wave 1's contribution and stores.  Wave 2 repeats wave 0's U = W_j C_i^T for itself, instruction by instruction:
// the coupling is bitwise what wave 0 produced.
template <int NX> struct CyclicReduceLds {
  using L = CyclicLds<NX>;
  static constexpr int oX = 2 * L::total, oC = oX + 64, total = oC + NX * NX; // r_i exchange | C_j of wave 2
};
template <int NX>
__global__ void __launch_bounds__(192, 1) gar_cyclic_reduce(CyclicParams Y) {
  using L = CyclicLds<NX>;
  using K = CondCfg<NX>;
  constexpr int bs = NX * NX, TX = K::TX;
  const CondensedParams &P = Y.C;
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  const int h = Y.h, J = P.num_legs;
  const int i = 2 * h * (int)blockIdx.x, b = (int)blockIdx.y;
  if (i >= J)
    return;
  const bool first = (h == 1); // the -P_j / -p_j halves of S_j, r_j are still separate
  double *sm = gar_smem + (wave == 1 ? L::total : 0);
  double *xch = gar_smem + CyclicReduceLds<NX>::oX; // r_i contribution of wave 1
  double *Cx = gar_smem + CyclicReduceLds<NX>::oC;  // C_j (wave 2)
  double *Dm = sm + L::oD, *Wm = sm + L::oW, *Bm = sm + L::oB, *D2 = sm + L::oD2;
  CyclicScratch<NX> X(P.scratch + (long long)b * P.scratch_stride, J);
  const int row = lane < NX ? lane : NX - 1;
  const bool has_right = i + h < J, has_left = i - h >= 0, has_coupling = i + 2 * h < J;
  int failed = 0;
  auto S_of = [&](int j) { return X.S + (long long)j * bs; };
  auto r_of = [&](int j) {
    double v = X.r[j * NX + row];
    if (first && j >= 1)
      v -= X.p[j * NX + row];
    return v;
  };
  // debug: cycle stamps of wave 0 of the second survivor of the first level
#ifdef GAR_TRACE
  const bool tracing = P.trace != nullptr && b == 0 && h == 1 && blockIdx.x == 1 && threadIdx.x == 0;
#endif
#ifdef GAR_TRACE
#define GAR_YMARK(id)                                                          \
  __builtin_amdgcn_sched_barrier(0);                                           \
  if (tracing)                                                                 \
    P.trace[32 + (id)] = (long long)clock64();                                 \
  __builtin_amdgcn_sched_barrier(0);
#else
#define GAR_YMARK(id)
#endif
  GAR_YMARK(0)
  // ---- until the inverses are there ----
  double ri = 0.0;
  if (wave == 0) {
    BlockRegs<NX> own; // S_i: in flight while the neighbour is inverted
    own.issue_diff(S_of(i), X.P + (long long)i * bs, first && i >= 1, lane);
    ri = r_of(i);
    if (has_right) { // right eliminated neighbour j = i + h
      const int j = i + h;
      BlockRegs<NX> rs, rc;
      rs.issue_diff(S_of(j), X.P + (long long)j * bs, first && j >= 1, lane);
      rc.issue(X.C + (long long)i * bs, lane); // C_i (row i, column j)
      rs.commit(Dm, lane);
      rc.commit(Bm, lane);
      wave_sync();
      GAR_YMARK(1)
      failed |= cyc_inverse<NX>(sm, lane); // Wm = W_j
    }
    own.commit(D2, lane);
    wave_sync();
    GAR_YMARK(2)
  } else if (wave == 1) {
    for (int e = lane; e < bs; e += 64)
      D2[e] = 0.0;
    if (has_left) { // left eliminated neighbour j = i - h
      const int j = i - h;
      BlockRegs<NX> rs, rc;
      rs.issue_diff(S_of(j), X.P + (long long)j * bs, first && j >= 1, lane);
      rc.issue(X.C + (long long)j * bs, lane); // C_j (row j, column i)
      rs.commit(Dm, lane);
      rc.commit(Bm, lane);
      wave_sync();
      failed |= cyc_inverse<NX>(sm, lane);
    }
  } else if (has_coupling) {
    cond_copy_block<NX>(Cx, X.C + (long long)(i + h) * bs, lane); // C_j, for the new coupling
    wave_sync();
  }
  __syncthreads();
  // ---- products ----
  if (wave == 0 && has_right) {
    const int j = i + h;
    cyc_store_block<NX>(X.W + (long long)j * bs, Wm, lane);
    cyc_store_block<NX>(X.Cl + (long long)j * bs, Bm, lane); // the coupling j had to its left
    GAR_YMARK(3)
    double4_t Ut[TX][TX];
    cyc_update<NX, false>(Wm, Bm, D2, Ut, lane); // U = W_j C_i^T ; S_i -= C_i U
    GAR_YMARK(4)
    const double y = cyc_matvec<NX>(Wm, r_of(j), row);
    ri -= cyc_matvec<NX>(Bm, y, row);
    GAR_YMARK(5)
    GAR_YMARK(6)
    wave_sync();
    GAR_YMARK(7)
  }
  if (wave == 1 && has_left) {
    const int j = i - h;
    double4_t Ut[TX][TX];
    cyc_update<NX, true>(Wm, Bm, D2, Ut, lane); // U = W_j C_j ; S_i -= C_j^T U
    const double y = cyc_matvec<NX>(Wm, r_of(j), row);
    ri -= cyc_matvecT<NX>(Bm, y, row);
    wave_sync();
  }
  if (wave == 1 && lane < NX)
    xch[lane] = ri;
  if (wave == 2 && has_coupling) { // new coupling (i, i + 2h) = -C_i W_j C_j = -U^T C_j, from wave 0's W_j and C_i
    double4_t Ut[TX][TX];
    cyc_form_u<NX>(gar_smem + L::oW, gar_smem + L::oB, Ut, lane);
    cyc_ut_times<NX>(Ut, Cx, X.C + (long long)i * bs, -1.0, lane);
  }
  __syncthreads();
  GAR_YMARK(8)
  if (wave == 0) {
    const double *D2b = gar_smem + L::total + L::oD2;
    if (has_left) {
      for (int e = lane; e < bs; e += 64)
        D2[e] += D2b[e];
      ri += xch[row];
      wave_sync();
    }
    // first level: the stored S_i / r_i become the complete ones (S slot was missing -P_i)
    cyc_store_block<NX>(S_of(i), D2, lane);
    if (lane < NX)
      X.r[i * NX + lane] = ri;
  }
  GAR_YMARK(9)
#undef GAR_YMARK
  if (failed && lane == 0)
    cyc_poison(X.info);
}

// ---- 3. back-substitution ---------------------------------------------------------------------
// z_j = W_j (r_j - Cl_j^T z_{j-h} - C_j z_{j+h}) for one block eliminated at level h
template <int NX>
__device__ __forceinline__ void cyc_backsolve_block(const CyclicScratch<NX> &X, int j, int h, int J,
                                                    int lane) {
  constexpr int bs = NX * NX;
  const int row = lane < NX ? lane : NX - 1;
  double v = X.r[j * NX + row];
  if (h == 1)
    v -= X.p[j * NX + row];
  v -= cyc_matvecT<NX>(X.Cl + (long long)j * bs, X.z[(j - h) * NX + row], row);
  if (j + h < J)
    v -= cyc_matvec<NX>(X.C + (long long)j * bs, X.z[(j + h) * NX + row], row);
  const double zj = cyc_matvec<NX>(X.W + (long long)j * bs, v, row);
  if (lane < NX)
    X.z[j * NX + lane] = zj;
}

// the top of the tree in one workgroup: z_0 = S_0^{-1} r_0 (the only block never eliminated), then
// the levels hmax .. Y.h, which hold a handful of blocks each
template <int NX>
__global__ void __launch_bounds__(256) gar_cyclic_top(CyclicParams Y) {
  using L = CyclicLds<NX>;
  const CondensedParams &P = Y.C;
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
  const int b = (int)blockIdx.x, J = P.num_legs;
  double *sm = gar_smem;
  CyclicScratch<NX> X(P.scratch + (long long)b * P.scratch_stride, J);
  const int row = lane < NX ? lane : NX - 1;
  int hmax = 1;
  while (2 * hmax < J)
    hmax *= 2;
  if (wave == 0) {
    const double r0 = X.r[row]; // (requested with S_0, not behind the inverse)
    cond_copy_block<NX>(sm + L::oD, X.S, lane);
    wave_sync();
    const int failed = cyc_inverse<NX>(sm, lane);
    const double z0 = cyc_matvec<NX>(sm + L::oW, r0, row);
    if (lane < NX)
      X.z[lane] = z0;
    if (failed && lane == 0)
      cyc_poison(X.info);
  }
  __threadfence_block();
  __syncthreads();
  for (int h = hmax; h >= Y.h; h >>= 1) {
    for (int q = wave; (2 * q + 1) * h < J; q += nw)
      cyc_backsolve_block<NX>(X, (2 * q + 1) * h, h, J, lane);
    __threadfence_block();
    __syncthreads();
  }
}

// one lower level: a wave per eliminated block
template <int NX>
__global__ void __launch_bounds__(64) gar_cyclic_backlevel(CyclicParams Y) {
  const CondensedParams &P = Y.C;
  const int lane = (int)threadIdx.x & 63;
  const int b = (int)blockIdx.y, J = P.num_legs, h = Y.h;
  const int j = (2 * (int)blockIdx.x + 1) * h;
  if (j >= J)
    return;
  CyclicScratch<NX> X(P.scratch + (long long)b * P.scratch_stride, J);
  cyc_backsolve_block<NX>(X, j, h, J, lane);
}

// ---- 4. recovery of the states and the residual of the ORIGINAL system, a wave per leg --------
//   lambda_k = z_k ;  x_k = -p_k - P_k E_k^T lambda_k - Q_k lambda_{k+1}
//   rows 2k (lambda_k row) and 2k+1 (x_k row) of rhs - A sol (blockTridiagMatMul,
//   block-tridiagonal.hpp:52-75); x_{k-1} is recomputed locally so that one launch suffices;
//   the infinity norm is accumulated with atomicMax on the bit pattern (non-negative doubles
//   order like their bits, NaN above everything)
template <int NX>
__global__ void __launch_bounds__(64) gar_cyclic_recover(CyclicParams Y) {
  constexpr int bs = NX * NX;
  const CondensedParams &P = Y.C;
  const int lane = (int)threadIdx.x & 63;
  const int k = (int)blockIdx.x, b = (int)blockIdx.y, J = P.num_legs, nblk = 2 * J;
  CyclicScratch<NX> X(P.scratch + (long long)b * P.scratch_stride, J);
  const double *prob = P.prob + (long long)b * P.prob_stride;
  double *sol = P.csol + (long long)b * nblk * NX;
  const int row = lane < NX ? lane : NX - 1;
  const int nc0 = P.nc0;
  // G0 is read by the waves of legs 0 and 1 alone; they stage it in LDS (zero rows past nc0) with every load in
  // flight at once: five loops of `load a row of G0 -> use it` made leg 0's wave -- and with it the launch -- twice
  // as long as any other leg's
  double *G0p = gar_smem;
  if (k <= 1) {
    cyc_stage_G0<NX>(G0p, prob + P.G0_off, nc0, lane);
    wave_sync();
  }
  auto G0T = [&](double lam) { return cyc_matvecT_seq<NX>(G0p, lam, row); }; // (G0^T lam)(row)
  auto state = [&](int kk, double lam, double lamn) {
    double x = -X.p[kk * NX + row];
    if (kk == 0)
      x -= cyc_matvec<NX>(X.P, G0T(lam), row);
    else
      x += cyc_matvec<NX>(X.P + (long long)kk * bs, lam, row);
    if (kk + 1 < J)
      x -= cyc_matvec<NX>(X.Q + (long long)kk * bs, lamn, row);
    return x;
  };
  double lam = X.z[k * NX + row];
  if (k == 0 && lane >= nc0)
    lam = 0.0;
  const double lamn = (k + 1 < J) ? X.z[(k + 1) * NX + row] : 0.0;
  const double x = state(k, lam, lamn);
  if (lane < NX) {
    sol[(2 * k) * NX + lane] = lam;
    sol[(2 * k + 1) * NX + lane] = x;
  }
  const double *tup = cond_tuple(P, b, k);
  // x_k row: -vx_k - Vxx_k x_k - E_k^T lambda_k - Vxt_k lambda_{k+1}
  double rx = -tup[3 * bs + row] - cyc_matvec<NX>(tup, x, row);
  rx -= (k == 0) ? G0T(lam) : -lam;
  if (k + 1 < J)
    rx -= cyc_matvec<NX>(tup + bs, lamn, row);
  // |rhs| + |A| |sol| of the same rows; omega = max_i |r_i| / max_i (|rhs_i| + (|A| |sol|)_i) is the (row-scaled
  // normwise) backward error of the solve.
  // A residual at omega ~ n eps is all a backward-stable solve can deliver and all that refinement with fp64
  // residuals can reach: with value functions of order 1/mu (constrained knots) the ABSOLUTE threshold of
  // parallel-solver.hpp:92 is out of reach for ANY solver -- the reference then spends its maxRefinementSteps
  // without effect (parallel-solver.hxx:184-202) -- so the gate below also accepts on omega (info[2]).
  double dx = fabs(tup[3 * bs + row]) + cyc_absmatvec<NX>(tup, x, row);
  if (k == 0) {
    dx += cyc_absmatvecT_seq<NX>(G0p, fabs(lam), row);
  } else {
    dx += fabs(lam);
  }
  if (k + 1 < J)
    dx += cyc_absmatvec<NX>(tup + bs, lamn, row);
  // lambda_k row
  double rl, dl = 0.0;
  if (k == 0) { // -g0 - G0 x_0
    const double s = cyc_matvec_seq<NX>(G0p, x, row);
    rl = (row < nc0 ? -prob[P.g0_off + row] : 0.0) - s;
    if (lane >= nc0)
      rl = 0.0;
    const double sa = cyc_absmatvec_seq<NX>(G0p, fabs(x), row);
    dl = (row < nc0 ? fabs(prob[P.g0_off + row]) : 0.0) + sa;
  } else { // -vt_{k-1} - Vxt_{k-1}^T x_{k-1} - Vtt_{k-1} lambda_k + x_k
    double lamp = X.z[(k - 1) * NX + row];
    if (k - 1 == 0 && lane >= nc0)
      lamp = 0.0;
    const double xp = state(k - 1, lamp, lam);
    const double *tp = cond_tuple(P, b, k - 1);
    rl = -tp[3 * bs + NX + row] - cyc_matvecT<NX>(tp + bs, xp, row) -
         cyc_matvec<NX>(tp + 2 * bs, lam, row) + x;
    dl = fabs(tp[3 * bs + NX + row]) + cyc_absmatvecT<NX>(tp + bs, xp, row) + cyc_absmatvec<NX>(tp + 2 * bs, lam, row) +
         fabs(x);
  }
  double mx = fmax(fabs(rx), fabs(rl));
  if (rx != rx || rl != rl)
    mx = rx + rl; // NaN
  if (lane >= NX)
    mx = 0.0;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const double other = __shfl_xor(mx, o);
    mx = (other > mx || other != other) ? other : mx;
  }
  if (lane == 0)
    atomicMax(reinterpret_cast<unsigned long long *>(&X.info[0]),
              (unsigned long long)__double_as_longlong(fabs(mx)));
  double om = fmax(dx, dl); // (the ratio is formed by the reader: info[0] / info[2])
  if (om != om)
    om = 0.0;
  if (lane >= NX)
    om = 0.0;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const double other = __shfl_xor(om, o);
    om = (other > om || other != other) ? other : om;
  }
  if (lane == 0)
    atomicMax(reinterpret_cast<unsigned long long *>(&X.info[2]),
              (unsigned long long)__double_as_longlong(fabs(om)));
}

} // namespace gar
