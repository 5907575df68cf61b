// gar_mfma.hpp -- the specialised backward sweep for uniform, unconstrained,
// unparameterised problems (nc = nth = 0, every stage NX x NU, terminal knot
// nu = 0): the BASELINE.json north-star path (N=256, nx=36, nu=12, fp64).
//
// Same arithmetic as ProximalRiccatiKernel::stageKernelSolve
// (gar/riccati-kernel.hxx:209-277) but laid out for CDNA4:
//
//  * one 256-thread workgroup per problem; waves 0..2 are "column-tile workers",
//    wave 3 streams the next knot HBM -> LDS and does the vector recursions
//    (vplus, qhat, rhat, kff, yff, vx);
//  * P = V'[A B] and H = [Q S;S^T R] + [A B]^T P are v_mfma_f64_16x16x4 chains in
//    which the D registers of P are fed straight back as the B operand of H
//    (the f64 C/D map row=(l>>4)+4r, col=l&15 IS the B map k=l>>4, j=l&15), so
//    P and H never touch LDS and there is no barrier between the two products;
//  * only the lower tiles of H / Vxx are computed; Vxx is mirrored when it is
//    stored -- exactly the reference's `selfadjointView<Lower>` (:216);
//  * Rhat (NU x NU) is factorised redundantly by every wave in registers
//    (lane = row) as an unpivoted LDL^T while evaluating the Bunch-Kaufman
//    pivot rule (core/bunchkaufman.hpp:46-83) at every column; if the rule would
//    ever interchange (or hit a zero column) the stage falls back, uniformly
//    for the workgroup, to the generic device Bunch-Kaufman -- so pivot
//    decisions are always the reference's;
//  * 2 barriers per stage; LDS ~50 KB -> 3 workgroups per CU.
#pragma once
#include "gar_device.hpp"
#include "gar_layout.h"

namespace gar {

struct MfmaParams {
  const double *prob;
  double *fac;
  int *status;
  // diagnostics (bench.py's slow_path_stage_frac): slow[0] += stages whose Rhat failed the first
  // Bunch-Kaufman test somewhere (they leave the register LDL^T), slow[1] += those of them where
  // the complete rule really pivots (generic device Bunch-Kaufman)
  int *slow; // (constrained sweeps: slow[2] += coupled stages, slow[3] += stages on the LDS Bunch-Kaufman)
  int *resume; // constrained sweeps: per problem, the knot the second kernel takes over at (-1: none)
  long long prob_stride, fac_stride;
  long long in_off0, in_rec, in_offN; // knot record of stage t < N at in_off0 + t*in_rec
  long long fac_rec, fac_offN;        // factor record of stage t < N at t*fac_rec
  int horizon;
  long long *trace; // debug: per-wave cycle stamps of one stage of problem 0 (or null)
  // initial stage fused into the one-wave-per-problem sweep (gar_wave.hpp); init == null: the
  // host launches the separate initial-stage kernel instead
  double *init;
  long long init_stride, G0_off, g0_off;
  int nc0;
  double mueq; // constrained stages: the proximal weight of [Rhat D^T; D -mu I]
  int init_closed; // fused initial stage: closed form when G0 = +-I (0: always factorise kkt0)
  // MPC cycling (cycleAppend, proximal-riccati.hxx:79-86) as a RING: the records of the stages
  // t < horizon are never moved; logical stage t lives in slot (t + ring0) mod horizon
  int ring0;
  // plain stage (gar_wave2.hpp): a positive definite Rhat keeps the unpivoted register LDL^T even where
  // Bunch-Kaufman would interchange (wave_ldl_fast_neg_pre); 0: the reference's pivot rule literally
  int spd_accept;
  // pipelined sweep (gar_hip_set_pipeline): the launch carries no LDS for the fused initial stage's kkt0.  Closed
  // form (G0 = +-I) as ever; any other problem is left to gar_initial_wave, launched behind the sweep on the
  // problems with resume[b] != 0 (this kernel writes resume[b] for every problem then)
  int init_small;
  __host__ __device__ long long slot(int t) const {
    const int p = t + ring0;
    return (ring0 != 0 && p >= horizon) ? p - horizon : p;
  }
};

// NC > 0: every knot carries NC equality constraints C x + D u + d = mu v (the one-wave-per-problem
// backward sweep and the forward sweep; the 4-wave backward kernel is unconstrained only)
template <int NX, int NU, int NC = 0> struct MfmaCfg {
  static_assert(NX % 4 == 0 && NU % 4 == 0 && NC % 4 == 0, "NX, NU, NC must be multiples of 4");
  static_assert(NU >= 4 && NU <= 32 && NX >= NU, "4 <= NU <= 32 <= NX");
  static constexpr int NW = NX + NU;
  static constexpr int NK = NU + NC; // rows of the reduced KKT system [Rhat D^T; D -mu I]
  static constexpr int NR = NK + NX; // rows of ff / fb: [kff; zff; yff], [K; Z; Aff]
  static constexpr int TX = (NX + 15) / 16; // tiles over the state index
  static constexpr int TW = (NW + 15) / 16; // tiles over [x; u]
  static constexpr int KS = NX / 4;         // k-steps over the next-state index
  static constexpr int KU = NU / 4;         // k-steps over the control index
  // (the 4-wave kernel gar_backward_mfma needs TW <= 3: one column tile per worker wave, NW <= 64;
  // the wide shapes -- NW > 64, e.g. (56, 24) -- exist in the one-wave-per-problem family only)
  static constexpr bool WIDE = (NW > 64);
  // pitch of the k-fast buffers: p = 2 mod 4 makes "16 lanes stride p, 4 lane
  // groups +1" conflict-free for ds_read_b64 (bank = dword address mod 64)
  static constexpr int PK = NX + 2;
  static constexpr int FROWS = TW * 16;
  // pitch of G = [-rhat | -Shat^T]: 16 mod 32 -> "16 lanes +1, 4 groups stride p" conflict-free
  static constexpr int PG = ((NX + 1 + 15) / 32) * 32 + 16;
  static constexpr int NL = NU * (NU - 1) / 2; // packed strictly-lower L
  // LDS carve (doubles); +16 slack after buffers read with padded tile indices
  static constexpr int oV = 0;
  static constexpr int oFt0 = oV + NX * PK + 16;
  static constexpr int oFt1 = oFt0 + FROWS * PK + 16;
  static constexpr int oG = oFt1 + FROWS * PK + 16;
  static constexpr int oM = oG + NU * PG + 16;
  static constexpr int oL = oM + NU * NU;      // packed strictly-lower L
  static constexpr int oVec = oL + NL + (NL % 2);
  // vectors: vn[NX] fv[2][NX] qr[2][NW] vp[NX]
  static constexpr int oVn = oVec, oFv = oVn + NX, oQr = oFv + 2 * NX, oVp = oQr + 2 * NW;
  // G2 = solution [kff | K] in the layout of G (pitch PG); the Bunch-Kaufman
  // fallback solves in place there; sub (16) and piv+ctrl (16 doubles) follow
  static constexpr int oG2 = (oVp + NX + 1) & ~1;
  static constexpr int oBk = oG2 + NU * PG + 16;
  // (gar_backward_mfma, GAR_MFMA_EARLY_FACTOR: L of Rhat, 1 / d_k and the verdict, handed from the worker wave that
  // factorises to wave 3, which solves)
  static constexpr int oLf = oBk + 16 + 16;
  static constexpr int total = oLf + NU * NU + NU + 2;
  // record offsets (uniform stage / terminal knot)
  static constexpr int kQ = 0, kS = NX * NX, kR = kS + NX * NU, kq = kR + NU * NU, kr = kq + NX,
                       kA = kr + NU, kB = kA + NX * NX, kf = kB + NX * NU;
  static constexpr int kC = kf + NX, kD = kC + NC * NX, kd = kD + NC * NU; // C, D, d follow f
  static constexpr int tQ = 0, tq = NX * NX; // terminal: Q, q, A, f (nu = 0)
  static constexpr int tC = tq + NX + NX * NX + NX, td = tC + NC * NX; // ... then C, d
  static constexpr int fFF = 0, fFB = NR, fVxx = fFB + NR * NX, fvx = fVxx + NX * NX;
  // Device layout of fb = [K; Z; Aff] in this kernel family ("fbT2"): element (r, j),
  // r in [0, NR), j in [0, NX), lives at (j/2)*(2 NR) + 2 r + (j & 1): the forward
  // sweep reads 16 B per lane with lane = row r, consecutive lanes consecutive
  // addresses.  gar_hip_get_gains converts back to StageFactor's row-major fb.
  __host__ __device__ static constexpr int fbT2(int r, int j) {
    return (j >> 1) * (2 * NR) + 2 * r + (j & 1);
  }
  // terminal factor record: ff (NC + NX), fb ((NC + NX) x NX, ROW-major: the generic layout), Vxx, vx
  static constexpr int tVxx = (NC + NX) + (NC + NX) * NX, tvx = tVxx + NX * NX;
};

__device__ __forceinline__ double lane_bcast(double v, int src /*wave-uniform*/) {
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  return __hiloint2double(hi, lo);
}

// Element (a, b), a >= b, of the stage Hessian [Q S; S^T R] read from the knot
// record through its LOWER triangle (the reference's products only ever feed the
// lower triangle of Vxx forward, riccati-kernel.hxx:216).
template <int NX, int NU>
__device__ __forceinline__ double w_lower(const double *rec, int row, int col) {
  using C = MfmaCfg<NX, NU>;
  int a = row >= col ? row : col, b = row >= col ? col : row;
  a = a < C::NW ? a : C::NW - 1; // padded tile rows: any valid element (result unused)
  b = b < C::NW ? b : C::NW - 1;
  // one address, one load (no divergent branches around the HBM loads)
  const int iq = C::kQ + b * NX + a;
  const int is = C::kS + (a - NX) * NX + b;
  const int ir = C::kR + (b - NX) * NU + (a - NX);
  const int idx = a < NX ? iq : (b < NX ? is : ir);
  return rec[idx];
}

// Unpivoted LDL^T of the NU x NU matrix M (LDS, column-major, lower valid) in
// registers, lane i < NU owning row i, with the Bunch-Kaufman rule evaluated at
// every column.  On return a[j] (j < row) = L(row, j), dinv[k] = 1/d_k
// (wave-uniform).  Returns 0 if BK would have taken the 1x1 pivot kp = k at every
// column (then BK == this factorisation, operation for operation,
// bunchkaufman.hpp:104-121), 1 if it would interchange, 2 on a zero column.
template <int NU>
__device__ __forceinline__ int wave_ldl_bk_rule(const double *M, int lane, double (&a)[NU],
                                                double (&dinv)[NU]) {
  const double alpha = (1.0 + 4.123105625617661) / 8.0;
  const int row = lane < NU ? lane : NU - 1;
#pragma unroll
  for (int j = 0; j < NU; ++j)
    a[j] = (j <= row) ? M[j * NU + row] : 0.0;
  int verdict = 0;
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const double akk = lane_bcast(a[k], k);
    double xs[NU];
    double colmax = 0.0;
    int imax = k + 1;
#pragma unroll
    for (int j = k + 1; j < NU; ++j) {
      xs[j] = lane_bcast(a[k], j); // a(j,k), lower triangle
      const double v = fabs(xs[j]);
      if (v > colmax) {
        colmax = v;
        imax = j;
      }
    }
    const double abs_akk = fabs(akk);
    if (fmax(abs_akk, colmax) == 0.0) {
      verdict |= 2;
    } else if (!(abs_akk >= colmax * alpha)) {
      // rowmax over row imax of the trailing matrix (bunchkaufman.hpp:63-73)
      double t = 0.0; // own a[imax]
#pragma unroll
      for (int c = 0; c < NU; ++c)
        t = (c == imax) ? a[c] : t;
      double rowmax = 0.0;
#pragma unroll
      for (int j = k; j < NU; ++j) {
        const double e1 = lane_bcast(a[j], imax); // a(imax, j), valid for j < imax
        const double e2 = lane_bcast(t, j);       // a(j, imax), valid for j > imax
        if (j < imax)
          rowmax = fmax(rowmax, fabs(e1));
        else if (j > imax)
          rowmax = fmax(rowmax, fabs(e2));
      }
      if (!(abs_akk >= (alpha * colmax) * (colmax / rowmax)))
        verdict |= 1;
    }
    const double d11 = 1.0 / akk;
#pragma unroll
    for (int j = k + 1; j < NU; ++j) {
      const double d11xj = xs[j] * d11;
      if (lane >= j)
        a[j] -= d11xj * a[k];
    }
    if (lane > k)
      a[k] *= d11;
    dinv[k] = d11;
  }
  return verdict;
}

// 1/a to <= 1 ulp: v_rcp_f64 + two Newton steps (the IEEE division sequence is
// ~4x longer and sits on the pivot-to-pivot critical path of the factorisation)
__device__ __forceinline__ double fast_rcp(double a) {
  double r = __builtin_amdgcn_rcp(a);
  double e = __builtin_fma(-a, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-a, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}

// Lean variant of wave_ldl_bk_rule: same elimination, but only the first test of
// the Bunch-Kaufman rule (|a_kk| >= alpha * colmax, bunchkaufman.hpp:61) is
// evaluated.  Returns 0 when that test held at every column (so BK takes kp = k
// everywhere and the factorisation below IS Bunch-Kaufman's); otherwise the
// caller re-runs the full rule.  No lane predication: entries above the diagonal
// hold garbage and are never read.
template <int NU>
__device__ __forceinline__ int wave_ldl_lean(const double *M, int lane, double (&a)[NU],
                                             double (&dinv)[NU]) {
  const double alpha = (1.0 + 4.123105625617661) / 8.0;
  const int row = lane < NU ? lane : NU - 1;
#pragma unroll
  for (int j = 0; j < NU; ++j)
    a[j] = M[j * NU + (j <= row ? row : j)]; // lower triangle (clamped address above it)
  unsigned long long bad = 0ull;
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const double akk = lane_bcast(a[k], k);
    // the first test, |a_kk| >= alpha * colmax, holds iff it holds against every row below:
    // one compare per lane and a ballot instead of a max-reduction over broadcast entries
    const unsigned long long nok = __ballot(!(fabs(akk) >= alpha * fabs(a[k])) || akk == 0.0);
    const unsigned long long from_k = ((1ull << NU) - 1ull) & ~((1ull << k) - 1ull);
    bad |= nok & from_k;
    const double d = fast_rcp(akk);
    const double lik = a[k] * d; // L(i,k) = a(i,k) d11: for row j this IS the reference's d11xj
#pragma unroll
    for (int j = k + 1; j < NU; ++j)
      a[j] = __builtin_fma(-lane_bcast(lik, j), a[k], a[j]); // a(i,j) -= d11xj * a(i,k)
    a[k] = lik;
    dinv[k] = d;
  }
  return bad != 0ull;
}

// x <- (L D L^T)^{-1} x, lane = right-hand-side column, L(i, j) broadcast from lane
// i's register j (no LDS, no waits)
template <int NU>
__device__ __forceinline__ void ldl_solve_bcast(const double (&a)[NU], const double (&dinv)[NU],
                                                double (&x)[NU]) {
#pragma unroll
  for (int i = 1; i < NU; ++i) {
#pragma unroll
    for (int j = 0; j < i; ++j)
      x[i] = __builtin_fma(-lane_bcast(a[j], i), x[j], x[i]);
  }
#pragma unroll
  for (int i = 0; i < NU; ++i)
    x[i] *= dinv[i];
#pragma unroll
  for (int j = NU - 2; j >= 0; --j) {
#pragma unroll
    for (int i = j + 1; i < NU; ++i)
      x[j] = __builtin_fma(-lane_bcast(a[j], i), x[i], x[j]);
  }
}

typedef double double2_t __attribute__((ext_vector_type(2)));

// Vxx -> HBM, 16 B per lane.  PACK (the serial family with the gar_forward_mfma roll-out): the LOWER TRIANGLE of V
// (LDS, pitch NX), rectangular packed (gar_layout.h: gar_sym_index) -- half the bytes of the full block,
// contiguous.  Otherwise (the wide shapes, whose roll-out reads the full block): V as it is, linear.
template <int NX, bool PACK, int PK = NX> struct VxxOut { // PK: pitch of V in LDS
  static_assert(!PACK || NX % 4 == 0, "packed Vxx: nx (nx + 1) / 2 must be even (16-byte stores)");
  static_assert(NX % 2 == 0 && PK % 2 == 0, "16-byte pieces stay inside a column");
  static constexpr int NP2 = PACK ? NX * (NX + 1) / 4 : NX * NX / 2; // 16-byte pairs
  static constexpr int NCH = (NP2 + 63) / 64;                        // pairs per lane
  __device__ static __forceinline__ int lds_of(int p) { // LDS offset of packed element p
    const int c = p / (NX + 1), k = p - c * (NX + 1);
    const bool first = k < NX - c;
    const int j = first ? c : NX - 1 - c;
    const int i = first ? c + k : j + k - (NX - c);
    return i * PK + j;
  }
  __device__ static __forceinline__ double2_t read(const double *V, int q, int lane) { // chunk q of this lane
    const int e = 64 * q + lane, ec = (64 * q + 63 < NP2 || e < NP2) ? e : NP2 - 1;
    double2_t v;
    if constexpr (PACK) {
      v.x = V[lds_of(2 * ec)];
      v.y = V[lds_of(2 * ec + 1)];
    } else if constexpr (PK == NX) {
      v = *reinterpret_cast<const double2_t *>(&V[2 * ec]);
    } else { // column (2 ec) / NX of the record = column of V, rows 2 ec % NX, + 1
      const int col = (2 * ec) / NX, row = 2 * ec - col * NX;
      v = *reinterpret_cast<const double2_t *>(&V[col * PK + row]);
    }
    return v;
  }
  __device__ static __forceinline__ void write(double *dst, int q, int lane, double2_t v) {
    const int e = 64 * q + lane;
    if (64 * q + 63 < NP2 || e < NP2)
      *reinterpret_cast<double2_t *>(&dst[2 * e]) = v;
  }
};
template <int NX, bool PACK = true, int PK = NX>
__device__ __forceinline__ void wave_flush_vxx(const double *V, double *dst, int lane) {
  using VO = VxxOut<NX, PACK, PK>;
  double2_t vbuf[VO::NCH];
#pragma unroll
  for (int q = 0; q < VO::NCH; ++q) // all the LDS reads first (one latency), then the stores
    vbuf[q] = VO::read(V, q, lane);
#pragma unroll
  for (int q = 0; q < VO::NCH; ++q)
    VO::write(dst, q, lane, vbuf[q]);
}

#ifndef GAR_MFMA_EARLY_FACTOR
#define GAR_MFMA_EARLY_FACTOR 1
#endif
#ifndef GAR_MFMA_FLUSH16
#define GAR_MFMA_FLUSH16 1
#endif
// (GAR_MFMA_EARLY_H measured and NOT adopted: backward 1.786 against 1.741 ms at batch 256 -- the tiles' latency was
// hidden behind the first product already; profiles/r06_ab_mfma_4wave_early_hessian_tiles_not_kept.log)
#ifndef GAR_MFMA_EARLY_H
#define GAR_MFMA_EARLY_H 0
#endif
// (launch bounds: the kernel only ever runs with at most one workgroup per CU -- the library binds it while
// batch <= #CUs --, so it could take the registers two resident workgroups share; measured: (256, 1) = 304 registers, no
// scratch: 1.621 against 1.556 ms at batch 256, 1.494 against 1.528 at batch 1 -- kept at (256, 2) for the reporting point;
// profiles/r06_ab_mfma_4wave_launch_bounds_not_kept.log)
#ifndef GAR_MFMA_MIN_BLOCKS
#define GAR_MFMA_MIN_BLOCKS 2
#endif
template <int NX, int NU>
__global__ void __launch_bounds__(256, GAR_MFMA_MIN_BLOCKS) gar_backward_mfma(MfmaParams P) {
  using C = MfmaCfg<NX, NU>;
  static_assert(C::TW <= 3 && NU <= 16, "one column tile per worker wave (3 workers), NW <= 64");
  // Rhat (rows / columns NX .. NW-1) inside the LAST tile column alone: the worker that holds it factorises it
  constexpr bool EARLY_FACTOR = (GAR_MFMA_EARLY_FACTOR != 0) && (NX >= 16 * (C::TW - 1));
  constexpr int NW = C::NW, PK = C::PK, PG = C::PG;
  double *sm = gar_smem;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int b = (int)blockIdx.x;
  const double *prob = P.prob + (long long)b * P.prob_stride;
  double *fac = P.fac + (long long)b * P.fac_stride;
  const int N = P.horizon;
  double *V = sm + C::oV, *G = sm + C::oG, *Mm = sm + C::oM, *G2 = sm + C::oG2;
  double *vn = sm + C::oVn, *vp = sm + C::oVp;
  const bool tracing = P.trace != nullptr && b == 0 && lane == 0;
#define GAR_MARK(id)                                                           \
  if (tracing && t == (N >> 1))                                                \
    P.trace[wave * 16 + (id)] = (long long)clock64();

  // ---- terminal knot (terminalSolve, nu = 0, nc = 0, :175-178): Vxx = Q, vx = q
  {
    const double *rec = prob + P.in_offN;
    double *out = fac + P.fac_offN;
    for (int e = tid; e < NX * NX; e += 256) {
      const int j = e / NX, i = e - j * NX; // column-major element (i, j)
      const double v = (i >= j) ? rec[C::tQ + e] : rec[C::tQ + i * NX + j];
      V[i * PK + j] = v; // symmetrised from lower, as the consumer stage does (:216)
      if (!GAR_VXX_PACKED)
        out[C::tVxx + e] = v;
      else if (i >= j)
        out[C::tVxx + gar_sym_index(1, NX, i, j)] = v; // (packed lower triangle: gar_layout.h)
    }
    for (int e = tid; e < NX; e += 256) {
      const double v = rec[C::tq + e];
      vn[e] = v;
      out[C::tvx + e] = v;
    }
    if (N >= 1) { // prologue: knot N-1 -> Ft[(N-1)&1]
      const double *r1 = prob + P.in_off0 + P.slot(N - 1) * P.in_rec;
      double *Ft = sm + (((N - 1) & 1) ? C::oFt1 : C::oFt0);
      for (int e = tid; e < NX * NW; e += 256) {
        const int j = e / NX, k = e - j * NX;
        Ft[j * PK + k] = r1[C::kA + e]; // [A B] column-major == F^T rows with pitch PK
      }
      for (int e = tid; e < NX; e += 256)
        sm[C::oFv + ((N - 1) & 1) * NX + e] = r1[C::kf + e];
      for (int e = tid; e < NW; e += 256)
        sm[C::oQr + ((N - 1) & 1) * NW + e] = r1[C::kq + e];
    }
  }
  __syncthreads();

  if (wave < 3) {
    // =====================================================================
    // column-tile workers: the matrix recursions on MFMA
    // =====================================================================
    const int tj = wave;
    const int c = 16 * tj + li;               // this lane's column in tile tj
    const int cc = c < NX ? c : NX - 1;
    double4_t Hc[C::TW]; // H tiles (ti, tj), ti >= tj
    // C-init of H from the knot record (lower elements).  GAR_MFMA_EARLY_H: the tiles of knot t - 1 are requested at the
    // END of stage t, before its last workgroup barrier -- a fence the compiler does not move loads across -- instead
    // of at the start of stage t - 1, where their HBM latency had only the first product to hide behind
    auto load_h = [&](int t) {
      const double *rec = prob + P.in_off0 + P.slot(t) * P.in_rec;
      if (tj < C::TW) {
#pragma unroll
        for (int ti = 0; ti < C::TW; ++ti)
          if (ti >= tj) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              Hc[ti][r] = w_lower<NX, NU>(rec, 16 * ti + lk + 4 * r, c);
          }
      }
    };
    if (GAR_MFMA_EARLY_H && N > 0)
      load_h(N - 1);
    for (int t = N - 1; t >= 0; --t) {
      GAR_MARK(0)
      double *out = fac + P.slot(t) * P.fac_rec;
      const double *Ft = sm + ((t & 1) ? C::oFt1 : C::oFt0);
      if (!GAR_MFMA_EARLY_H)
        load_h(t);
      if (tj < C::TW) {
        // S1: P(:, tj) = V' F(:, tj)            (:216-221, AtV / BtV fused)
        double4_t Pt[C::TX];
#pragma unroll
        for (int tm = 0; tm < C::TX; ++tm)
          Pt[tm] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < C::KS; ++s) {
          const double bq = Ft[c * PK + 4 * s + lk];
#pragma unroll
          for (int tm = 0; tm < C::TX; ++tm) {
            const int ic = (16 * tm + li) < NX ? (16 * tm + li) : NX - 1;
            const double aq = V[ic * PK + 4 * s + lk];
            Pt[tm] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, bq, Pt[tm], 0, 0, 0);
          }
        }
        GAR_MARK(1)
        // S2: H(ti, tj) += F(:, ti)^T P(:, tj); B operand = P's D registers (:224-228)
#pragma unroll
        for (int ti = 0; ti < C::TW; ++ti)
          if (ti >= tj) {
#pragma unroll
            for (int s = 0; s < C::KS; ++s) {
              const double aq = Ft[(16 * ti + li) * PK + 4 * s + lk];
              Hc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, Pt[s >> 2][s & 3], Hc[ti], 0, 0, 0);
            }
          }
        GAR_MARK(2)
        // export the control rows: G(u, 1+j) = -Shat^T(u, j), M = Rhat (lower)
#pragma unroll
        for (int ti = 0; ti < C::TW; ++ti)
          if (ti >= tj) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = 16 * ti + lk + 4 * r;
              if (row >= NX && row < NW) {
                if (c < NX)
                  G[(row - NX) * PG + 1 + c] = -Hc[ti][r];
                else if (c <= row)
                  Mm[(c - NX) * NU + (row - NX)] = Hc[ti][r];
              }
            }
          }
      }
      if constexpr (EARLY_FACTOR) {
        // Rhat lies in this worker's tile alone (the last tile column): it is factorised HERE, right behind the export,
        // while the workers of the wider tile columns are still multiplying -- wave 3 then only solves behind barrier A
        // (the 12 x 12 factorisation used to sit between the two barriers with all three workers idle)
        if (tj == C::TW - 1) {
          wave_sync(); // (this wave's own export of Rhat is what it reads: LDS order within the wave)
          double a_row[NU], dinv[NU];
          int verdict = wave_ldl_lean<NU>(Mm, lane, a_row, dinv);
          if (verdict != 0) { // rare: evaluate the complete Bunch-Kaufman rule
            verdict = wave_ldl_bk_rule<NU>(Mm, lane, a_row, dinv);
            if (lane == 0) {
              atomicAdd(&P.slow[0], 1);
              if (verdict != 0)
                atomicAdd(&P.slow[1], 1);
            }
          }
          double *Lf = sm + C::oLf;
          if (lane < NU) {
#pragma unroll
            for (int j = 0; j < NU; ++j)
              Lf[lane * NU + j] = a_row[j];
          }
          if (lane == 0) {
#pragma unroll
            for (int k = 0; k < NU; ++k)
              Lf[NU * NU + k] = dinv[k];
            reinterpret_cast<int *>(Lf + NU * NU + NU)[0] = verdict;
          }
        }
      }
      GAR_MARK(3)
      __syncthreads(); // A: G, M complete -> wave 3 factors and solves
      GAR_MARK(4)
      __syncthreads(); // B: [kff | K] in G2
      GAR_MARK(5)
      if (tj < C::TX) {
        double Kb[C::KU]; // B operand K[u = 4s + lk][j = c]
#pragma unroll
        for (int s = 0; s < C::KU; ++s)
          Kb[s] = G2[(4 * s + lk) * PG + 1 + cc];
        // Aff^T(tj, ti) = A^T + K^T B^T (:267), i.e. Aff with the column index j on the
        // tile rows, so that every lane group stores 16 consecutive i (fbT2 layout)
#pragma unroll
        for (int ti = 0; ti < C::TX; ++ti) {
          double4_t acc;
          const int i = 16 * ti + li, icl = i < NX ? i : NX - 1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 16 * tj + lk + 4 * r;
            acc[r] = Ft[(j < NX ? j : NX - 1) * PK + icl]; // A(i, j) = F^T(j, i)
          }
#pragma unroll
          for (int s = 0; s < C::KU; ++s) {
            const double bq = Ft[(NX + 4 * s + lk) * PK + icl]; // B^T(u, i)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Kb[s], bq, acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 16 * tj + lk + 4 * r;
            if (i < NX && j < NX)
              out[C::fFB + C::fbT2(NU + i, j)] = acc[r];
          }
        }
        GAR_MARK(6)
        // Vxx(ti, tj), ti >= tj: Qhat + Shat K (:272-273); lower tiles only, mirrored
#pragma unroll
        for (int ti = 0; ti < C::TX; ++ti)
          if (ti >= tj) {
            double4_t acc = Hc[ti];
#pragma unroll
            for (int s = 0; s < C::KU; ++s) {
              const double aq = -G[(4 * s + lk) * PG + 1 + 16 * ti + li]; // Shat(i, u)
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, Kb[s], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * ti + lk + 4 * r;
              if (i < NX && c < NX && i >= c) {
                V[i * PK + c] = acc[r];
                V[c * PK + i] = acc[r];
              }
            }
          }
      }
      GAR_MARK(7)
      if (GAR_MFMA_EARLY_H && t > 0)
        load_h(t - 1);
      __syncthreads(); // C: V, vn, Ft[next] complete
      GAR_MARK(8)
#if GAR_MFMA_FLUSH16
      { // Vxx -> HBM (packed lower triangle: gar_layout.h) in 16-byte pieces, the chunks dealt over the three workers:
        // two stores per lane instead of seven 8-byte ones behind an index division (round 6)
        using VO = VxxOut<NX, GAR_VXX_PACKED != 0, PK>;
#pragma unroll
        for (int q0 = 0; q0 < VO::NCH; q0 += 3)
          if (q0 + wave < VO::NCH)
            VO::write(out + C::fVxx, q0 + wave, lane, VO::read(V, q0 + wave, lane));
      }
#else
      for (int e = tid; e < NX * NX; e += 192) { // Vxx -> HBM (packed lower triangle: gar_layout.h)
        const int j = e / NX, i = e - j * NX;
        if (!GAR_VXX_PACKED)
          out[C::fVxx + e] = V[i * PK + j];
        else if (i >= j)
          out[C::fVxx + gar_sym_index(1, NX, i, j)] = V[i * PK + j];
      }
#endif
      GAR_MARK(9)
    }
  } else {
    // =====================================================================
    // wave 3: streams knot t-1 into LDS, runs the vector recursions, factors
    // Rhat and solves for [kff | K]
    // =====================================================================
    const WG w1 = wave_self();
    int failed = 0;
    for (int t = N - 1; t >= 0; --t) {
      GAR_MARK(0)
      const double *rec = prob + P.in_off0 + P.slot(t) * P.in_rec;
      double *out = fac + P.slot(t) * P.fac_rec;
      const int cur = t & 1;
      const double *Ft = sm + (cur ? C::oFt1 : C::oFt0);
      double *Ftn = sm + (cur ? C::oFt0 : C::oFt1);
      const double *fv = sm + C::oFv + cur * NX;
      const double *qr = sm + C::oQr + cur * NW;
      const double *rn = prob + P.in_off0 + P.slot(t > 0 ? t - 1 : 0) * P.in_rec; // knot t-1 (t = 0: harmless re-read)
      // [A B] of knot t-1 moves HBM -> registers -> LDS in 3 chunks interleaved with
      // the arithmetic below; indices are clamped, never predicated, so the loads
      // stay in flight together
      constexpr int PFN = (NX * NW + 63) / 64, PFC = (PFN + 2) / 3;
      double pf[PFC];
#define GAR_PF_LOAD(ch)                                                        \
  _Pragma("unroll") for (int q = 0; q < PFC; ++q) {                            \
    const int e = lane + 64 * ((ch) * PFC + q);                                \
    pf[q] = rn[C::kA + (e < NX * NW ? e : NX * NW - 1)];                       \
  }
#define GAR_PF_STORE(ch)                                                       \
  _Pragma("unroll") for (int q = 0; q < PFC; ++q) {                            \
    const int e = lane + 64 * ((ch) * PFC + q);                                \
    const int ec = e < NX * NW ? e : NX * NW - 1;                              \
    const int j = ec / NX, k = ec - j * NX;                                    \
    Ftn[j * PK + k] = pf[q];                                                   \
  }
      GAR_PF_LOAD(0)
      const double pf_f = rn[C::kf + (lane < NX ? lane : NX - 1)];
      const double pf_qr = rn[C::kq + (lane < NW ? lane : NW - 1)];
      double qhat;
      // vplus = vx' + V' f (:217-218), lane i < NX
      {
        const int ic = lane < NX ? lane : NX - 1;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; k += 2) {
          s0 += V[ic * PK + k] * fv[k];
          s1 += V[ic * PK + k + 1] * fv[k + 1];
        }
        if (lane < NX)
          vp[lane] = vn[lane] + (s0 + s1);
      }
      wave_sync();
      GAR_MARK(1)
      // [qhat; rhat] = [q; r] + F^T vplus (:227-228), lane j < NW
      {
        const int jc = lane < NW ? lane : NW - 1;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < NX; k += 2) {
          s0 += Ft[jc * PK + k] * vp[k];
          s1 += Ft[jc * PK + k + 1] * vp[k + 1];
        }
        qhat = qr[jc] + (s0 + s1);
        if (lane >= NX && lane < NW)
          G[(lane - NX) * PG] = -qhat; // kff right-hand side (:248)
      }
      GAR_MARK(2)
      GAR_PF_STORE(0)
      GAR_PF_LOAD(1)
      GAR_MARK(3)
      __syncthreads(); // A
      GAR_MARK(4)
      // ---- factor Rhat in registers (lane = row) under the Bunch-Kaufman rule ----
      double a_row[NU], dinv[NU], x[NU];
      int verdict;
      if constexpr (EARLY_FACTOR) { // (done by the last worker wave before barrier A: its L, 1 / d, verdict from LDS)
        const double *Lf = sm + C::oLf;
        const int frow = lane < NU ? lane : NU - 1;
#pragma unroll
        for (int j = 0; j < NU; ++j)
          a_row[j] = Lf[frow * NU + j];
#pragma unroll
        for (int k = 0; k < NU; ++k)
          dinv[k] = Lf[NU * NU + k];
        verdict = reinterpret_cast<const int *>(Lf + NU * NU + NU)[0];
      } else {
        verdict = wave_ldl_lean<NU>(Mm, lane, a_row, dinv);
        if (verdict != 0) { // rare: evaluate the complete Bunch-Kaufman rule
          verdict = wave_ldl_bk_rule<NU>(Mm, lane, a_row, dinv);
          if (lane == 0) { // (this code runs in wave 3 only: threadIdx.x == 0 never gets here)
            atomicAdd(&P.slow[0], 1);
            if (verdict != 0)
              atomicAdd(&P.slow[1], 1);
          }
        }
      }
      GAR_MARK(5)
      const int col = lane <= NX ? lane : NX; // G column: 0 = kff, 1 + j = K(:, j)
      if (verdict == 0) {
#pragma unroll
        for (int k = 0; k < NU; ++k)
          x[k] = G[k * PG + col];
        ldl_solve_bcast<NU>(a_row, dinv, x);
        if (lane <= NX) {
#pragma unroll
          for (int k = 0; k < NU; ++k)
            G2[k * PG + col] = x[k];
        }
      } else {
        // Bunch-Kaufman would interchange (or met a zero column): do exactly what the
        // reference does, with the generic device BK run by this wave alone
        for (int e = lane; e < NU * PG; e += 64)
          G2[e] = G[e];
        double *sub = sm + C::oBk;
        int *piv = (int *)(sub + 16);
        wave_sync();
        failed |= wg_bk_factor(w1, NU, Mm, NU, sub, piv, piv + 16);
        wg_bk_solve(w1, NU, Mm, NU, sub, piv, G2, PG, 1, NX + 1);
#pragma unroll
        for (int k = 0; k < NU; ++k)
          x[k] = G2[k * PG + col];
      }
      GAR_MARK(6)
      __syncthreads(); // B: workers pick K up from G2
      GAR_MARK(7)
      GAR_PF_STORE(1)
      GAR_PF_LOAD(2)
      // K -> fb rows 0..NU-1 (fbT2 layout), lane = column
      if (lane >= 1 && lane <= NX) {
#pragma unroll
        for (int k = 0; k < NU; ++k)
          out[C::fFB + C::fbT2(k, lane - 1)] = x[k];
      }
      // kff = G2(:, 0); yff = f + B kff (:266), vx = qhat + Shat kff (:275-276)
      {
        const int ic = lane < NX ? lane : NX - 1;
        double yf = fv[ic], vxv = qhat;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const double ku = G2[u * PG];
          yf += Ft[(NX + u) * PK + ic] * ku;
          vxv -= G[u * PG + 1 + ic] * ku;
        }
        if (lane < NU)
          out[C::fFF + lane] = G2[lane * PG];
        if (lane < NX) {
          out[C::fFF + NU + lane] = yf;
          out[C::fvx + lane] = vxv;
          vn[lane] = vxv;
        }
      }
      GAR_PF_STORE(2)
      if (t > 0) { // f, q, r of knot t-1 -> LDS
        if (lane < NX)
          sm[C::oFv + (cur ^ 1) * NX + lane] = pf_f;
        if (lane < NW)
          sm[C::oQr + (cur ^ 1) * NW + lane] = pf_qr;
      }
      GAR_MARK(8)
      __syncthreads(); // C
      GAR_MARK(9)
#undef GAR_PF_LOAD
#undef GAR_PF_STORE
    }
    if (failed && lane == 0)
      atomicOr(&P.status[b], failed);
  }
#undef GAR_MARK
}

// ---------------------------------------------------------------------------
// Forward sweep for the same problem family (computeInitial + forwardImpl,
// riccati-kernel.hxx:195-207, 314-377, nc = nth = 0).  One wave per problem, no
// LDS: lane r < NW owns row r of [K; Aff], lane i < NX row i of Vxx'; the state
// is broadcast lane -> wave with v_readlane; every knot's gains are streamed
// once, 16 B per lane, straight into registers.  Latency is hidden by
// occupancy (many independent waves per CU), not by software pipelining.
// ---------------------------------------------------------------------------
struct MfmaFwdParams {
  const double *fac;  // factor records (fbT2 layout)
  const double *init; // kkt0.ff per problem: [x0; lbd0]
  double *sol;        // xs | us | vs | lbdas
  long long fac_stride, init_stride, sol_stride;
  long long fac_rec, fac_offN;
  int horizon, nc0;
  int sol_u, sol_l; // base offsets of us / lbdas inside a solution record
  int sol_v;        // ... and of vs (constrained stages)
  int ring0;        // logical stage t lives in factor slot (t + ring0) mod horizon (MfmaParams::ring0)
  __host__ __device__ long long slot(int t) const {
    const int p = t + ring0;
    return (ring0 != 0 && p >= horizon) ? p - horizon : p;
  }
};

template <int NX, int NC = 0> struct FwdStage {
  double2_t g[NX / 2];
  double2_t gz[NC > 0 ? NX / 2 : 1];
  // Vxx' of the next stage.  Packed records (gar_layout.h): this lane's 16-byte pieces of the packed lower triangle,
  // fetched linearly (whole lines, 6 requests per lane at nx = 36) -- they go through LDS to become rows in
  // fwd_step.  Full records (-DGAR_VXX_PACKED=0): row iv itself, gathered from the lower triangle.
  double2_t vp[GAR_VXX_PACKED ? VxxOut<NX, true>::NCH : 1];
  double vrow[GAR_VXX_PACKED ? 1 : NX];
  double ff, vxn, ffz;
};

// r: row of [K; Z; Aff] this lane owns in the first slot (a control row or a next-state row)
template <int NX, int NU, int NC = 0>
__device__ __forceinline__ void fwd_load(const MfmaFwdParams &P, const double *fac, int t, int r,
                                         int iv, int lane, FwdStage<NX, NC> &S) {
  using C = MfmaCfg<NX, NU, NC>;
  constexpr int NW = C::NR;
  const int N = P.horizon;
  const int tc = t < N ? t : N - 1; // past the end: harmless re-read of the last stage
  const double *rec = fac + P.slot(tc) * P.fac_rec;
  const double *recn = (tc + 1 < N) ? fac + P.slot(tc + 1) * P.fac_rec : fac + P.fac_offN;
  const int oVn = (tc + 1 < N) ? C::fVxx : C::tVxx, ovn = (tc + 1 < N) ? C::fvx : C::tvx;
#pragma unroll
  for (int m = 0; m < NX / 2; ++m)
    S.g[m] = *reinterpret_cast<const double2_t *>(rec + C::fFB + m * 2 * NW + 2 * r);
  if (NC > 0) {
    const int rz = NU + (lane < NC ? lane : NC - 1);
#pragma unroll
    for (int m = 0; m < NX / 2; ++m)
      S.gz[m] = *reinterpret_cast<const double2_t *>(rec + C::fFB + m * 2 * NW + 2 * rz);
    S.ffz = rec[C::fFF + rz];
  }
  if (GAR_VXX_PACKED) {
    using VO = VxxOut<NX, true>;
#pragma unroll
    for (int q = 0; q < VO::NCH; ++q) {
      const int e = 64 * q + lane, ec = (64 * q + 63 < VO::NP2 || e < VO::NP2) ? e : VO::NP2 - 1;
      S.vp[GAR_VXX_PACKED ? q : 0] = *reinterpret_cast<const double2_t *>(recn + oVn + 2 * ec);
    }
  } else { // full block: lower triangle only (column j, row iv for j <= iv; this lane's own column below)
#pragma unroll
    for (int j = 0; j < NX; ++j)
      S.vrow[GAR_VXX_PACKED ? 0 : j] = recn[oVn + (iv >= j ? j * NX + iv : iv * NX + j)];
  }
  S.ff = rec[C::fFF + r];
  S.vxn = recn[ovn + iv];
}

template <int NX, int NU, int NC = 0>
__device__ __forceinline__ double fwd_step(const MfmaFwdParams &P, double *sol, int t, int lane,
                                           double xs, const FwdStage<NX, NC> &S, double *vb, int iv) {
  using C = MfmaCfg<NX, NU, NC>;
  constexpr int NW = C::NW;
  if (GAR_VXX_PACKED) { // the packed triangle of Vxx' -> LDS (the rows are read back after the products below)
    using VO = VxxOut<NX, true>;
    wave_sync(); // (the previous stage's row reads are done)
#pragma unroll
    for (int q = 0; q < VO::NCH; ++q) {
      const int e = 64 * q + lane;
      if (64 * q + 63 < VO::NP2 || e < VO::NP2)
        *reinterpret_cast<double2_t *>(&vb[2 * e]) = S.vp[GAR_VXX_PACKED ? q : 0];
    }
    wave_sync();
  }
  // u = kff + K x ; x' = yff + Aff x   (:334-336, :360-361); two accumulators
  double acc = S.ff, acc1 = 0.0;
#pragma unroll
  for (int m = 0; m < NX / 2; ++m) {
    acc = __builtin_fma(S.g[m].x, lane_bcast(xs, NU + 2 * m), acc);
    acc1 = __builtin_fma(S.g[m].y, lane_bcast(xs, NU + 2 * m + 1), acc1);
  }
  acc += acc1;
  if (lane < NU)
    sol[P.sol_u + t * NU + lane] = acc;
  else if (lane < NW)
    sol[(t + 1) * NX + (lane - NU)] = acc;
  if (NC > 0) { // v = zff + Z x  (:338-340), off the state recursion's critical path
    double az = S.ffz, az1 = 0.0;
#pragma unroll
    for (int m = 0; m < NX / 2; ++m) {
      az = __builtin_fma(S.gz[m].x, lane_bcast(xs, NU + 2 * m), az);
      az1 = __builtin_fma(S.gz[m].y, lane_bcast(xs, NU + 2 * m + 1), az1);
    }
    if (lane < NC)
      sol[P.sol_v + t * NC + lane] = az + az1;
  }
  // lbd' = vx' + Vxx' x'  (:369-371); x'_j sits in lane NU + j
  double lam = S.vxn, lam1 = 0.0;
  if (GAR_VXX_PACKED) {
    // row iv of the symmetric matrix from its packed lower triangle (gar_sym_index): elements (iv, j), j <= iv, at
    // cj + iv (consecutive lanes, consecutive addresses); (j, iv), j > iv, at lowbase + j
    const int lowbase = 2 * iv < NX ? iv * NX : (NX - 1 - iv) * (NX + 1) + 1;
    double vr[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int cj = 2 * j < NX ? j * NX : (NX - 1 - j) * (NX + 1) + 1;
      vr[j] = vb[iv >= j ? cj + iv : lowbase + j];
    }
#pragma unroll
    for (int j = 0; j < NX; j += 2) {
      lam = __builtin_fma(vr[j], lane_bcast(acc, NU + j), lam);
      lam1 = __builtin_fma(vr[j + 1], lane_bcast(acc, NU + j + 1), lam1);
    }
  } else {
#pragma unroll
    for (int j = 0; j < NX; j += 2) {
      lam = __builtin_fma(S.vrow[GAR_VXX_PACKED ? 0 : j], lane_bcast(acc, NU + j), lam);
      lam1 = __builtin_fma(S.vrow[GAR_VXX_PACKED ? 0 : j + 1], lane_bcast(acc, NU + j + 1), lam1);
    }
  }
  lam += lam1;
  if (lane < NX)
    sol[P.sol_l + P.nc0 + t * NX + lane] = lam;
  return acc;
}

// The gains of stage t+1 are requested BEFORE stage t is evaluated (two register sets, the loop
// body instantiated twice): with one wave per SIMD -- all a 1 024-problem batch gives a 1 024-SIMD
// chip -- nothing else would cover the HBM latency of the next 25 KB.
template <int NX, int NU, int NC = 0>
__global__ void __launch_bounds__(64) gar_forward_mfma(MfmaFwdParams P) {
  using C = MfmaCfg<NX, NU, NC>;
  constexpr int NW = C::NW;
  const int lane = (int)threadIdx.x;
  const int b = (int)blockIdx.x;
  const double *fac = P.fac + (long long)b * P.fac_stride;
  double *sol = P.sol + (long long)b * P.sol_stride;
  const double *io = P.init + (long long)b * P.init_stride;
  const int N = P.horizon;
  // row of [K; Z; Aff] in the first slot: a control row or a next-state row (Z rows: second slot)
  const int r = lane < NU ? lane : (lane < NW ? lane + NC : C::NR - 1);
  const int iv = lane < NX ? lane : NX - 1; // row of Vxx'
  // the state lives in lanes NU .. NW-1 (where x' = yff + Aff x is produced)
  const int ix = (lane >= NU && lane < NW) ? lane - NU : 0;
  double *vb = gar_smem; // nx (nx + 1) / 2 doubles of dynamic LDS (packed Vxx records: gar_forward_mfma_lds_bytes)
  FwdStage<NX, NC> SA, SB;
  fwd_load<NX, NU, NC>(P, fac, 0, r, iv, lane, SA);
  double xs = io[ix]; // x0 from the initial-stage solve (kkt0.ff)
  if (lane >= NU && lane < NW)
    sol[ix] = xs;
  for (int e = lane; e < P.nc0; e += 64)
    sol[P.sol_l + e] = io[NX + e]; // lbd0
  int t = 0;
  for (; t + 1 < N; t += 2) {
    fwd_load<NX, NU, NC>(P, fac, t + 1, r, iv, lane, SB);
    xs = fwd_step<NX, NU, NC>(P, sol, t, lane, xs, SA, vb, iv);
    fwd_load<NX, NU, NC>(P, fac, t + 2, r, iv, lane, SA);
    xs = fwd_step<NX, NU, NC>(P, sol, t + 1, lane, xs, SB, vb, iv);
  }
  if (t < N)
    xs = fwd_step<NX, NU, NC>(P, sol, t, lane, xs, SA, vb, iv);
  if (NC > 0) { // terminal knot: v_N = zff + Z x_N (its record keeps the generic row-major layout)
    const double *recN = fac + P.fac_offN;
    const int i = lane < NC ? lane : NC - 1;
    double az = recN[i];
#pragma unroll
    for (int j = 0; j < NX; ++j)
      az = __builtin_fma(recN[(NC + NX) + i * NX + j], lane_bcast(xs, NU + j), az);
    if (lane < NC)
      sol[P.sol_v + N * NC + lane] = az;
  }
}

} // namespace gar
