// gar_wave.hpp -- ONE WAVE PER PROBLEM backward sweep for uniform, unconstrained,
// unparameterised problems (nc = nth = 0, every stage NX x NU, terminal knot
// nu = 0): the BASELINE.json north-star path (N=256, nx=36, nu=12, fp64).
//
// Same arithmetic as ProximalRiccatiKernel::stageKernelSolve
// (gar/riccati-kernel.hxx:209-277); what changes against gar_mfma.hpp is the
// mapping onto the machine, driven by two measurements on MI355X
// (scripts/ubench/ubench_f64.cpp, profiles/r01_ubench_f64.log):
//   * v_mfma_f64_16x16x4_f64 issues once per 64 cycles per SIMD, and while a wave
//     streams them NO other VALU instruction of ANY wave on that SIMD issues
//     (fp64 MFMA runs on the vector ALU's DP lanes): a SIMD's time is the plain sum
//     of its MFMA time (64 cyc each) and of every VALU instruction it executes;
//   * so a workgroup whose waves wait on each other at barriers wastes exactly the
//     SIMD time the HBM roofline needs (the 4-wave kernel of gar_mfma.hpp spends
//     ~2/3 of each stage with 3 of its 4 waves parked).
// Here a problem lives in ONE wave: no barriers, no inter-wave hand-offs, every
// SIMD runs its own independent problem(s), and the non-MFMA work is kept off
// the VALU wherever an LDS-port or memory-port instruction can do it:
//   * F = [A B] goes HBM -> registers directly in MFMA operand layout (lane
//     (li,lk) holds F[4s+lk][16t+li]); the same registers are the B operand of
//     P = V'F, the A operand of H = W + F^T P and -- because the operand rows
//     4s+lk ARE the C/D rows lk+4r of tile s>>2 -- the accumulator init of
//     Aff = A + B K, which is computed in place on top of them;
//   * P's D registers are the B operand of H (as in gar_mfma.hpp); the Shat^T rows
//     of H's D registers are the A operand of Vxx = Qhat + Shat K;
//   * V' lives in LDS (symmetric, k-fast) as the A operand of P = V'F;
//   * the next knot is loaded into the SAME registers as soon as the current one has
//     been consumed (F after Aff, the Hessian tiles after Vxx), most of a stage ahead of
//     its first use;
//   * wave-uniform operands of the VALU phases (the pivot column of the LDL^T, L in the
//     triangular solves, f in vplus) are broadcast with v_readlane from the register that
//     already holds them: measured 2.5x cheaper than wave-uniform LDS reads;
//   * the initial-stage KKT solve (proximal-riccati.hxx:42-60) runs at the end of the same
//     wave (packed lower triangle of kkt0 in the LDS the sweep no longer needs).
// Measured on the (32,12) shape, whose working set fits 256 registers: a second wave per SIMD
// buys ~5 % (MFMA and VALU of the two waves exclude each other), so the kernel is built for
// ONE wave per SIMD (all 512 registers), and hides its latencies with instruction-level
// parallelism instead.  LDS: 19 KB per wave.
#pragma once
#include "gar_mfma.hpp"

namespace gar {

// F-DMA (see WaveCfg::oF): measured and NOT adopted -- same box, alternating launches, batch 4 096: backward 11.32 ms
// with it against 10.58 ms without (profiles/r05_ab_f_operands_by_lds_dma_not_kept.log), results bit for bit the same.
// The 27 operand loads of sixteen 32-byte pieces each are not what the stage waits for.  `make fdma` builds it.
#ifndef GAR_F_DMA
#define GAR_F_DMA 0
#endif
#ifndef GAR_COUPLED_REFRESH_LANE
#define GAR_COUPLED_REFRESH_LANE 1
#endif
#ifndef GAR_SWEEP_REFRESH_LANE  // ... and for the unconstrained headline sweep
#define GAR_SWEEP_REFRESH_LANE 0
#endif
#ifndef GAR_CSTR_REFRESH_LANE   // the same for the decoupled constrained stage (first kernel of the chain): by itself
#define GAR_CSTR_REFRESH_LANE 1 // +5.5 % slower; together with GAR_CSTR_EARLY_C (gar_wave2.hpp), which needs the room, -9 %
#endif

template <int NX, int NU, int NC = 0> struct WaveCfg {
  using M = MfmaCfg<NX, NU, NC>;
  static constexpr int NW = NX + NU, TX = M::TX, TW = M::TW, KS = M::KS, KU = M::KU;
  static constexpr int NK = M::NK, NR = M::NR, KC = NC / 4; // KKT rows ; fb rows ; k-steps over constraints
  static constexpr int KSF = KS / 4, KST = KS % 4; // full double4 groups of k-steps, tail steps
  // NX = 16 (TX-1) + 4: the last row tile of P, Aff and Vxx holds FOUR valid rows.  Those tiles
  // run on v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks, 16 cycles) instead of
  // wasting 3/4 of a 64-cycle 16x16x4: with A_b[i][k] in lane 16k+4b+i, B_b[k][j] in lane
  // 16k+4b+j and D_b[i][j] in lane 16i+4b+j (measured, scripts/ubench/mfma4x4_probe.cpp) the B
  // operand and the result have exactly the 16x16x4 B-operand layout (column on lane&15, k or
  // row on lane>>4), so they drop into the same registers.
#ifdef GAR_NO_REM4
  static constexpr bool REM4 = false;
#else
  static constexpr bool REM4 = (NX % 16 == 4) && (KST == 1);
#endif
  // Pitch of V in LDS: NX (unpadded).  Its MFMA operand reads V[(16 t + li) * PK + 4 s + lk] are 2-way bank conflicts
  // (lanes li and li + 8); a pitch of NX + 2 makes them conflict-free ((PK li + lk) mod 32 is then a bijection) and
  // is supported (-DGAR_V_PITCH_PAD=1: every flush is a gather with the pitch) -- measured: SQ_LDS_BANK_CONFLICT
  // falls from 8.1 % to 4.3 % of the wave cycles (45 % -> 30 % of the LDS-active cycles) and SQ_WAVE_CYCLES does not
  // move (-0.3 %; profiles/r03_sq_lds_pitch.log); A/B on one box, alternating launches: backward 11.25 ms padded
  // against 11.07 ms unpadded (scripts/ab_vxx_packed.py).  The conflicts hide behind the 64-cycle MFMAs they feed.
  // Not used.
#ifndef GAR_V_PITCH_PAD
#define GAR_V_PITCH_PAD 0
#endif
  static constexpr int PK = (GAR_V_PITCH_PAD && NX % 4 == 0) ? NX + 2 : NX, PG = M::PG;
  // tile row / register of H holding Shat^T row u = 4s' + lk (rows NX + u)
  __host__ __device__ static constexpr int shTile(int s) { return (NX + 4 * s) >> 4; }
  __host__ __device__ static constexpr int shReg(int s) { return ((NX + 4 * s) & 15) >> 2; }
  // LDS carve (doubles), one slice per wave
  static constexpr int oV = 0;
  static constexpr int oG = oV + NX * PK + 16;    // [rhat | Shat^T ; d | C], NK x PG
  static constexpr int oG2 = oG;                  // [kff | K ; zff | Z]: the solve runs in place
  static constexpr int oM = oG + NK * PG + 16;    // Rhat (NC > 0: [Rhat D^T; D -mu I]), column-major lower
  // (NC > 0: the KKT matrix is kept as a packed lower triangle, GAR_PACKED_LOWER)
  static constexpr int oVn = oM + (NC > 0 ? NK * (NK + 1) / 2 : NK * NK); // vx' (NX)
  static constexpr int oVp = oVn + NX;            // vplus (NX)
  static constexpr int oLr = (oVp + NX + 1) & ~1; // L of Rhat = L D L^T, row-major (forward solve)
  static constexpr int oLc = oLr;                 // (the transposed solve reads L strided)
  static constexpr int oDi = oLr + NU * NU;       // -1/d_k
  static constexpr int oBk = (oDi + NU + 1) & ~1; // Bunch-Kaufman: sub(BKS) | piv, ctrl (BKS ints each)
  static constexpr int BKS = (NK + 15) & ~15; // (16 for the unconstrained shapes with NU <= 16)
  static constexpr int oDump = (oBk + 2 * BKS + 1) & ~1; // 2 doubles: target of masked-out LDS writes
  static constexpr int oFlag = oDump + 2;           // MODE 3: verdict of the factorisation (as a double)
  static constexpr int total = oDump + 4;
  // F-DMA (GAR_F_DMA, the serial one-wave sweep NC = 0): [A | B] of the NEXT knot, requested into LDS at the start
  // of the stage by global_load_lds (14 linear 1 KiB pieces at (36, 12) instead of 27 loads of sixteen 32-byte
  // pieces each, no registers), read as MFMA operands where the stage used to load them from HBM.  It lies behind
  // everything the stage uses -- where the fused initial stage's kkt0 goes once the sweep is over.
  static constexpr int oF = (total + 1) & ~1;
  static constexpr int f_doubles = NX * NW;
  static constexpr int total_fdma = oF + f_doubles;
  // fused initial stage (after the sweep): the packed lower triangle of kkt0 = [Vxx0 G0^T; G0 0]
  // and its right-hand side overlay everything but V
  static constexpr int oK0 = oG;
  __host__ __device__ static constexpr int k0_doubles(int n0) {
    return n0 * (n0 + 1) / 2 + 2 * n0 + (n0 + 24) / 2 + 4; // matrix | rhs | sub | piv, ctrl
  }
  __host__ __device__ static constexpr int total_with_init(int nc0) {
    return (oK0 + k0_doubles(NX + nc0)) > total ? (oK0 + k0_doubles(NX + nc0)) : total;
  }
  __host__ __device__ static constexpr int imax(int a, int b) { return a > b ? a : b; }
  // ---- parameterised legs (gar_wave_leg.hpp; nth = NX): the extra state of the recursion
  // (riccati-kernel.hxx:278-311) lives in LDS in LANE-PRIVATE layouts, slot (s, tj) at
  // ((s*TX + tj)*64 + lane):
  //   Xt : Vxt'[4s+lk][16tj+li]  -- the MFMA B operand of k-step s (== register s&3 of D tile s>>2)
  //   Tt : Vtt'^T accumulators, D-tile register (s&3) of tile (s>>2, tj)
  static constexpr int oXt = (total + 1) & ~1;
  static constexpr int oTt = oXt + KS * TX * 64;
  static constexpr int oGt = oTt + KS * TX * 64;   // [Ghat_u] then [Kth], NU x PG (solved in place)
  static constexpr int oVt = oGt + NU * PG + 16;   // vt (NX)
  static constexpr int oYf = oVt + NX;             // yff (NX)
  static constexpr int leg_total = (oYf + NX + 1) & ~1;
  // two waves per leg (gar_backward_wave_leg2): wave A publishes, per stage, [kff | K], Rhat, L,
  // 1/d, the verdict and yff for wave B; the block [oG, total) and yff exist twice (stage parity),
  // so that one workgroup barrier per stage suffices
  static constexpr int oPub1 = leg_total;
  static constexpr int pub_shift = oPub1 - oG;
  static constexpr int oYf1 = oPub1 + (total - oG);
  static constexpr int leg2_total = (oYf1 + NX + 1) & ~1;
  // factor record of a parameterised stage (gar_factor_layout(NX,NU,0,NX,NX)); fb and fth in the
  // fbT2 device order
  static constexpr int pFTH = NW + NW * NX, pVxx = pFTH + NW * NX, pvx = pVxx + NX * NX,
                       pVxt = pvx + NX, pVtt = pVxt + NX * NX, pvt = pVtt + NX * NX,
                       prec = pvt + NX;
};

// base + cst (doubles, compile-time: goes to the scalar base / the instruction's immediate) +
// a loop-invariant per-lane byte offset: one VGPR per access PATTERN, not per access
__device__ __forceinline__ double ldg_b(const double *base, int cst, unsigned lane_bytes) {
  return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base + cst) + lane_bytes);
}
__device__ __forceinline__ void stg_b(double *base, int cst, unsigned lane_bytes, double v) {
  *reinterpret_cast<double *>(reinterpret_cast<char *>(base + cst) + lane_bytes) = v;
}

// one knot's data in registers, in the layouts the stage body consumes
template <int NX, int NU> struct WaveStage {
  using C = WaveCfg<NX, NU>;
  // F[4s+lk][16t+li]: k-steps s = 4g+e < 4*KSF in Fo[t][g][e], the tail steps in FoT[t][s-4*KSF]
  double4_t Fo[C::TW][C::KSF > 0 ? C::KSF : 1];
  double FoT[C::TW][C::KST > 0 ? C::KST : 1];
  double4_t Hc[C::TW][C::TW];    // lower tiles of [Q S;S^T R], D layout (init of H)
  double fi, qri;                // f[lane], [q;r][lane]
  __device__ __forceinline__ double fo(int t, int s) const {
    return s < 4 * C::KSF ? Fo[t][s >> 2][s & 3] : FoT[t][s - 4 * C::KSF];
  }
};

// loop-invariant per-lane byte offsets into a knot record.  A column/row tile that lies
// entirely inside its block shares ONE lane offset with the other interior tiles (the tile
// origin is a compile-time constant that goes to the scalar base / the immediate); only a tile
// that straddles a block boundary or overhangs the matrix needs its own (clamped) offsets.
template <int NX, int NU, int NC = 0> struct WaveLane {
  using C = WaveCfg<NX, NU, NC>;
  unsigned fo0, foX;             // F operand F[lk][li] ; overhanging last column tile
  // element (16ti+lk+4r, 16tj+li) of [Q S;S^T R] at its NATURAL position (no mirroring:
  // the strictly-upper part of a diagonal tile only feeds results that are never used)
  unsigned hcx0, hcxX[C::TW];    // rows < NX: Q[lk][li] ; column tiles reaching past NX
  unsigned hcq[C::TW];           // QP (Q, R as packed lower triangles, gar_layout.h): rows < NX of column tile tj
  unsigned hcu0, hcuX[C::TW][C::KU]; // rows NX+4s'+lk: S^T[lk][li] ; tiles reaching past NX
  unsigned bop0, bopX;           // B[li][lk] ; overhanging last row tile
  unsigned bop4;                 // B[NX-4+(lane&3)][lane>>4]: A operand of the 4x4x4 blocks
  unsigned fi, qri;
  unsigned fbl;                  // fbT2 lane part: (li>>1)*2NR + 2lk + (li&1)
  __host__ __device__ static constexpr bool fo_in(int t) { return 16 * t + 15 < C::NW; }
  __host__ __device__ static constexpr bool x_in(int t) { return 16 * t + 15 < NX; }
};

// QP: the knot records keep Q and R as packed lower triangles (gar_layout.h: gar_lower_index).  Element (i, j) sits
// at i + g(j), g(j) = (2n - j - 1) j / 2: one lane offset per column tile, the row goes to the immediate.  The
// strictly-upper lanes of a diagonal tile then read some other element of the block -- they only feed results that
// are never used, as with the full blocks.
template <int NX, int NU, int NC, bool QP = false>
__device__ __forceinline__ void wave_lane_init(WaveLane<NX, NU, NC> &L, int lane) {
  using C = WaveCfg<NX, NU, NC>;
  using M = MfmaCfg<NX, NU, NC>;
  const int li = lane & 15, lk = lane >> 4;
  L.fo0 = 8u * (unsigned)(M::kA + li * NX + lk);
  {
    const int t = C::TW - 1, col = (16 * t + li) < C::NW ? (16 * t + li) : C::NW - 1;
    L.foX = 8u * (unsigned)(M::kA + col * NX + lk);
  }
  L.hcx0 = 8u * (unsigned)(M::kQ + li * NX + lk);
  L.hcu0 = 8u * (unsigned)(M::kS + lk * NX + li);
#pragma unroll
  for (int tj = 0; tj < C::TW; ++tj) {
    const int col = (16 * tj + li) < C::NW ? (16 * tj + li) : C::NW - 1;
    // x rows (row = lk + const): Q(row, col) = kQ + col*NX + row ; S(row, col-NX) = kS + (col-NX)*NX + row
    L.hcxX[tj] = 8u * (unsigned)((col < NX ? M::kQ + col * NX : M::kS + (col - NX) * NX) + lk);
    L.hcq[tj] = 8u * (unsigned)((col < NX ? M::kQ + ((2 * NX - col - 1) * col) / 2 : M::kS + (col - NX) * NX) + lk);
    // u rows (row = NX + u, u = 4s' + lk): S^T(u, col) = kS + u*NX + col ; R(u, col-NX) = kR + (col-NX)*NU + u
#pragma unroll
    for (int sp = 0; sp < C::KU; ++sp) {
      const int u = 4 * sp + lk;
      L.hcuX[tj][sp] = 8u * (unsigned)(col < NX ? M::kS + u * NX + col
                                                  : (QP ? M::kR + ((2 * NU - (col - NX) - 1) * (col - NX)) / 2 + u
                                                        : M::kR + (col - NX) * NU + u));
    }
  }
  L.bop0 = 8u * (unsigned)(M::kB + lk * NX + li);
  {
    const int ti = C::TX - 1, row = (16 * ti + li) < NX ? (16 * ti + li) : NX - 1;
    L.bopX = 8u * (unsigned)(M::kB + lk * NX + row);
  }
  L.bop4 = 8u * (unsigned)(M::kB + (lane >> 4) * NX + NX - 4 + (lane & 3));
  const int ir = lane < NX ? lane : NX - 1, iw = lane < C::NW ? lane : C::NW - 1;
  L.fbl = 8u * (unsigned)((li >> 1) * 2 * C::NR + 2 * lk + (li & 1));
  L.fi = 8u * (unsigned)(M::kf + ir);
  L.qri = 8u * (unsigned)(M::kq + iw);
}

// part A of a knot: the F operands and the vectors (needed from the start of the stage)
template <int NX, int NU, class LANE>
__device__ __forceinline__ void wave_load_a(const double *rec, const LANE &L, WaveStage<NX, NU> &S) {
  using C = WaveCfg<NX, NU>;
#pragma unroll
  for (int t = 0; t < C::TW; ++t)
#pragma unroll
    for (int s = 0; s < C::KS; ++s) {
      const double v = WaveLane<NX, NU>::fo_in(t) ? ldg_b(rec, 16 * t * NX + 4 * s, L.fo0)
                                                  : ldg_b(rec, 4 * s, L.foX);
      if (s < 4 * C::KSF)
        S.Fo[t][s >> 2][s & 3] = v;
      else
        S.FoT[t][s - 4 * C::KSF] = v;
    }
  S.fi = ldg_b(rec, 0, L.fi);
  S.qri = ldg_b(rec, 0, L.qri);
}
// part B: the Hessian tiles (first needed by H's accumulation) and B as the A operand of Aff
template <int NX, int NU, class LANE, bool QP = false>
__device__ __forceinline__ void wave_load_b(const double *rec, const LANE &L, WaveStage<NX, NU> &S) {
  using C = WaveCfg<NX, NU>;
#pragma unroll
  for (int ti = 0; ti < C::TW; ++ti)
#pragma unroll
    for (int tj = 0; tj <= ti; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row0 = 16 * ti + 4 * r; // + lk
        if (row0 + 3 < NX)
          S.Hc[ti][tj][r] = QP ? ldg_b(rec, row0, L.hcq[tj])
                               : (WaveLane<NX, NU>::x_in(tj) ? ldg_b(rec, 16 * tj * NX + row0, L.hcx0)
                                                             : ldg_b(rec, row0, L.hcxX[tj]));
        else if (row0 >= NX && row0 + 3 < C::NW)
          S.Hc[ti][tj][r] = WaveLane<NX, NU>::x_in(tj)
                                ? ldg_b(rec, (row0 - NX) * NX + 16 * tj, L.hcu0)
                                : ldg_b(rec, 0, L.hcuX[tj][(row0 - NX) >> 2]);
        else
          S.Hc[ti][tj][r] = 0.0; // rows past NW (padding)
      }
}

// x <- -(L D L^T)^{-1} x, lane = right-hand-side column; L and -1/d are read from LDS with
// wave-uniform addresses (LDS port, broadcast) instead of v_readlane pairs (VALU port)
// A zero-instruction scheduling fence: returns 0 in a register the compiler believes to depend
// on `after`, so loads at `base + fence0(after)` cannot be hoisted above the instruction that
// produces `after` -- without it the scheduler issues ALL the reads of an unrolled phase first.
// (An offset, not the pointer, goes through the asm: the pointer keeps its LDS address space.)
__device__ __forceinline__ int fence0(double after) {
  int z = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(z) : "v"(after));
#else
  (void)after;
#endif
  return z;
}

template <int NU>
__device__ __forceinline__ void ldl_solve_lds(const double *Lr, const double *ndi, double (&x)[NU]) {
  // column-oriented substitutions: the FMAs of one column are independent of each other;
  // L is read one column (forward) / one row (transposed) at a time with wave-uniform
  // addresses (LDS port, broadcast)
#pragma unroll
  for (int j = 0; j < NU - 1; ++j) {
    const double *Lp = Lr + fence0(x[j > 2 ? j - 3 : 0]);
    double lc[NU];
#pragma unroll
    for (int i = j + 1; i < NU; ++i)
      lc[i] = Lp[i * NU + j];
#pragma unroll
    for (int i = j + 1; i < NU; ++i)
      x[i] = __builtin_fma(-lc[i], x[j], x[i]);
  }
  {
    const double *dp = ndi + fence0(x[NU - 2]);
#pragma unroll
    for (int i = 0; i < NU; ++i)
      x[i] *= dp[i];
  }
#pragma unroll
  for (int i = NU - 1; i >= 1; --i) {
    const double *Lp = Lr + fence0(x[i < NU - 3 ? i + 3 : NU - 1]);
    double lr[NU];
#pragma unroll
    for (int j = 0; j < i; ++j)
      lr[j] = Lp[i * NU + j];
#pragma unroll
    for (int j = i - 1; j >= 0; --j) // x[i-1], the next pivot, first
      x[j] = __builtin_fma(-lr[j], x[i], x[j]);
  }
}

// Unpivoted LDL^T of the NU x NU matrix M (LDS, column-major, lower valid), lane i < NU owning
// row i, checking at every column the first test of the Bunch-Kaufman rule
// (|a_kk| >= alpha * colmax, bunchkaufman.hpp:61) lane-parallel (one compare + ballot).  When
// the test holds everywhere BK takes kp = k at every column and its elimination
// (bunchkaufman.hpp:104-121: d11xj = a(j,k) d11; a(i,j) -= d11xj a(i,k)) is what runs here,
// product for product.  Writes L row-major / column-major and -1/d to LDS; returns 0 if the
// test held at every column (and no pivot was zero).
// DEFINITE = true: the caller knows the matrix is definite (any diagonal pivot order is then
// stable); only an exactly-zero or non-finite pivot is reported.
template <int NU, bool DEFINITE = false>
__device__ __forceinline__ int wave_ldl_fast(const double *M, int lane, double (&a)[NU],
                                             double (&nd)[NU]) {
  const double alpha = (1.0 + 4.123105625617661) / 8.0;
  const int row = lane < NU ? lane : NU - 1;
#pragma unroll
  for (int j = 0; j < NU; ++j)
    a[j] = M[j * NU + row]; // row i of Rhat; entries above the diagonal are stale LDS, never used
  unsigned long long bad = 0ull;
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const double akk = lane_bcast(a[k], k);
    // lanes i > k hold a(i,k): the column below the pivot
    // rows i >= k: |a_kk| >= alpha |a(i,k)| (lane k itself: a non-zero pivot)
    if (DEFINITE) {
      bad |= (akk == 0.0 || !(fabs(akk) <= 1.79e308)) ? 1ull : 0ull; // wave-uniform
    } else {
      const unsigned long long nok = __ballot(!(fabs(akk) >= alpha * fabs(a[k])) || akk == 0.0);
      const unsigned long long from_k = ((1ull << NU) - 1ull) & ~((1ull << k) - 1ull);
      bad |= nok & from_k;
    }
    const double d = fast_rcp(akk);
    const double lik = a[k] * d; // L(i,k) = a(i,k) d11  (== the reference's d11xj for row i)
#pragma unroll
    for (int j = k + 1; j < NU; ++j)
      a[j] = __builtin_fma(-lane_bcast(lik, j), a[k], a[j]); // a(i,j) -= d11xj * a(i,k)
    a[k] = lik;
    nd[k] = -d; // wave-uniform
  }
  return bad != 0ull;
}

// x <- -(L D L^T)^{-1} x, lane = right-hand-side column; L(i,j) is broadcast from lane i's
// register j with v_readlane (no LDS round trip on the way from the factorisation to the solve)
template <int NU>
__device__ __forceinline__ void ldl_solve_regs_bcast(const double (&a)[NU], const double (&nd)[NU],
                                                     double (&x)[NU]) {
#pragma unroll
  for (int j = 0; j < NU - 1; ++j) {
#pragma unroll
    for (int i = j + 1; i < NU; ++i)
      x[i] = __builtin_fma(-lane_bcast(a[j], i), x[j], x[i]);
  }
#pragma unroll
  for (int i = 0; i < NU; ++i)
    x[i] *= nd[i];
#pragma unroll
  for (int i = NU - 1; i >= 1; --i) {
#pragma unroll
    for (int j = i - 1; j >= 0; --j)
      x[j] = __builtin_fma(-lane_bcast(a[j], i), x[i], x[j]);
  }
}

// two right-hand-side sets per lane (the gains and the parameter gains, :262 and :291) behind ONE
// set of broadcasts
template <int NU>
__device__ __forceinline__ void ldl_solve_regs_bcast2(const double (&a)[NU], const double (&nd)[NU],
                                                      double (&x)[NU], double (&y)[NU]) {
#pragma unroll
  for (int j = 0; j < NU - 1; ++j) {
#pragma unroll
    for (int i = j + 1; i < NU; ++i) {
      const double l = lane_bcast(a[j], i);
      x[i] = __builtin_fma(-l, x[j], x[i]);
      y[i] = __builtin_fma(-l, y[j], y[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < NU; ++i) {
    x[i] *= nd[i];
    y[i] *= nd[i];
  }
#pragma unroll
  for (int i = NU - 1; i >= 1; --i) {
#pragma unroll
    for (int j = i - 1; j >= 0; --j) {
      const double l = lane_bcast(a[j], i);
      x[j] = __builtin_fma(-l, x[i], x[j]);
      y[j] = __builtin_fma(-l, y[i], y[j]);
    }
  }
}

// Rare path, kept out of line so that its registers do not weigh on the sweep: the first
// Bunch-Kaufman test failed somewhere (or a pivot was zero).  Evaluate the complete rule;
// if BK still takes kp = k everywhere redo the unpivoted factorisation, else run the generic
// device Bunch-Kaufman exactly as the reference would (interchanges, 2x2 pivots), solving
// [kff | K] into G2.  Returns 1 if the factorisation failed (zero pivot column).
// PIVOTS_KNOWN (the round-2 stage, which has evaluated the complete rule in line): straight to the
// device Bunch-Kaufman, and the 1 + NX right-hand sides on the blocked MFMA solve.
template <int NX, int NU, bool PARAM = false, bool PIVOTS_KNOWN = false>
__device__ __attribute__((noinline)) int wave_slow_factor_solve(double *sm, int lane, int *slow) {
  using C = WaveCfg<NX, NU>;
  constexpr int PG = C::PG;
  double *G = sm + C::oG, *G2 = sm + C::oG2, *Mm = sm + C::oM;
  double *Gt = sm + C::oGt; // PARAM: [Ghat_u] -> [Kth], NX columns
  double *Lr = sm + C::oLr, *ndi = sm + C::oDi;
  int failed = 0;
  int verdict = 1;
  if (!PIVOTS_KNOWN) {
    double a_row[NU], dinv[NU];
    verdict = wave_ldl_bk_rule<NU>(Mm, lane, a_row, dinv);
    if (verdict == 0) {
      if (lane < NU) {
#pragma unroll
        for (int j = 0; j < NU; ++j) {
          Lr[lane * NU + j] = a_row[j];
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NU; ++j)
          ndi[j] = -dinv[j];
      }
    }
  }
  if (lane == 0) { // which factorisation the block holds: 0 = L, 1/d (unpivoted) ; 2 = Bunch-Kaufman
    sm[C::oFlag] = verdict == 0 ? 0.0 : 2.0;
    if (slow != nullptr) {
      atomicAdd(&slow[0], 1);
      if (verdict != 0)
        atomicAdd(&slow[1], 1);
    }
  }
  wave_sync();
  const int col = lane <= NX ? lane : NX;
  if (verdict == 0) {
    double x[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k)
      x[k] = G[k * PG + col];
    ldl_solve_lds<NU>(Lr, ndi, x);
    if (lane <= NX) {
#pragma unroll
      for (int k = 0; k < NU; ++k)
        G2[k * PG + col] = x[k];
    }
    if (PARAM) {
      const int ct = lane < NX ? lane : NX - 1;
#pragma unroll
      for (int k = 0; k < NU; ++k)
        x[k] = Gt[k * PG + ct];
      ldl_solve_lds<NU>(Lr, ndi, x);
      if (lane < NX) {
#pragma unroll
        for (int k = 0; k < NU; ++k)
          Gt[k * PG + ct] = x[k];
      }
    }
  } else {
    for (int e = lane; e < NU * PG; e += 64)
      G2[e] = -G[e]; // in place (G2 aliases G)
    if (PARAM)
      for (int e = lane; e < NU * PG; e += 64)
        Gt[e] = -Gt[e];
    double *sub = sm + C::oBk;
    int *piv = (int *)(sub + C::BKS);
    const WG w1 = wave_self();
    wave_sync();
    failed |= wg_bk_factor(w1, NU, Mm, NU, sub, piv, piv + C::BKS);
    if constexpr (PIVOTS_KNOWN && NU % 4 == 0 && NX + 1 <= 48)
      wave_bk_solve_mfma<NU, NX + 1, GAR_COLMAJOR>(Mm, sub, piv, G2, PG, lane);
    else
      wg_bk_solve(w1, NU, Mm, NU, sub, piv, G2, PG, 1, NX + 1);
    if (PARAM)
      wg_bk_solve(w1, NU, Mm, NU, sub, piv, Gt, PG, 1, NX);
  }
  wave_sync();
  return failed;
}

// Constrained stages: Bunch-Kaufman factorisation of the reduced KKT matrix (packed lower triangle in
// LDS) and the solve of its right-hand sides, out of line (the stage keeps ~400 registers live).
// The triangular solves run as blocked MFMA updates (wave_bk_solve_mfma).  (A fully unrolled
// column-in-registers substitution was 1.8x SLOWER than even the LDS-resident column loop of
// wg_bk_solve: 1 900 hoisted LDS reads spill.)
template <int NK, int BKS, int PG, int NCOLS>
__device__ __attribute__((noinline)) int wave_kkt_factor_solve(double *Mm, double *sub, double *G,
                                                               int lane) {
  int *piv = (int *)(sub + BKS);
  const WG w1 = wave_self();
  const int failed = wg_bk_factor<GAR_PACKED_LOWER>(w1, NK, Mm, NK, sub, piv, piv + BKS);
  wave_bk_solve_mfma<NK, NCOLS, GAR_PACKED_LOWER>(Mm, sub, piv, G, PG, lane);
  return failed;
}

// One stage t of the backward sweep, entirely inside one wave.  On entry S holds knot t
// (F operands, vectors, Hessian tiles); on exit it holds knot t-1.
// MODE 0: the plain stage (stageKernelSolve, :209-277).
// MODE 1: a stage of a parameterised leg (ParallelRiccatiSolver, nth = NX, implicit
//         Gx = Gu = Gth = gamma = 0): the plain stage plus :278-311 --
//           Ghat_u = B^T Vxt' ; Kth = -Rhat^{-1} Ghat_u ; Yth = B Kth ; vt = vt' + Vxt'^T yff ;
//           Vxt = Aff^T Vxt' ; Vtt = Vtt' + Ghat_u^T Kth (== Vxt'^T Yth, :310, in NU instead of
//           NX products per entry).
// MODE 2: the leg-end knot (terminalSolve with nu > 0, :146-192, under configure_knot's
//         Gx = A^T, Gu = B^T, Gth = 0, gamma = f, parallel-solver.hxx:136-141): no V', so
//         H = W; Kth = -R^{-1} B^T ; Vxt = A^T + K^T B^T ; Vtt = B Kth ; vt = f + B kff.
// MODE 3: MODE 0's arithmetic for wave A of a two-wave leg (gar_backward_wave_leg2): record
//         layout of a parameterised stage, the per-stage LDS block selected by the stage parity
//         PAR, [kff | K], Rhat, L, 1/d, the verdict and yff published for wave B, which does the
//         parameter part (wave_param_stage) after the workgroup barrier that ends the publication.
// NC > 0 (MODE 0 only): the constrained stage (:232-262, :272-276 with C, D, d): the reduced KKT
//         system [Rhat D^T; D -mu I] [K; Z] = -[Shat^T; C] is assembled in LDS and factorised by the
//         wave-scope Bunch-Kaufman (it is indefinite: there is no unpivoted fast path);
//         Vxx += C^T Z, vx += C^T zff; ff = [kff; zff; yff], fb = [K; Z; Aff].
template <int NX, int NU, int MODE = 0, int PAR = 0, int NC = 0, bool PACKV = false>
__device__ __forceinline__ void wave_stage(const MfmaParams &P, double *sm, const double *prob,
                                           double *fac, int t, int lane,
                                           const WaveLane<NX, NU, NC> &L, WaveStage<NX, NU> &S,
                                           int &failed, const bool tracing) {
  static_assert(NC == 0 || MODE == 0, "constrained stages: plain recursion only");
  using C = WaveCfg<NX, NU, NC>;
  using M = MfmaCfg<NX, NU, NC>;
  constexpr int NK = C::NK, NR = C::NR, KC = C::KC;
  constexpr int NW = C::NW, PK = C::PK, PG = C::PG, TX = C::TX, TW = C::TW, KS = C::KS,
                KU = C::KU;
  const int li = lane & 15, lk = lane >> 4;
  constexpr bool PRM = (MODE == 1 || MODE == 2); // this wave does the parameter part itself
  double *sb = sm + (MODE == 3 ? PAR * C::pub_shift : 0); // this stage's block [oG, total)
  double *V = sm + C::oV, *G = sb + C::oG, *Mm = sb + C::oM;
  double *vn = sb + C::oVn, *vp = sb + C::oVp;
  // vx' was written by the previous stage: into the other block when the blocks alternate
  const double *vnr = sm + (MODE == 3 ? (1 - PAR) * C::pub_shift : 0) + C::oVn;
  double *Xt = sm + C::oXt + lane, *Tt = sm + C::oTt + lane; // lane-private slots, stride 64
  double *Gt = sm + C::oGt, *vtl = sm + C::oVt, *yfl = sm + (MODE == 3 && PAR ? C::oYf1 : C::oYf);
  constexpr int oVxx = MODE ? C::pVxx : M::fVxx, ovx = MODE ? C::pvx : M::fvx;
  const unsigned lkx = 8u * (unsigned)(lk * NX + li); // element (lk, li) of a pitch-NX block
  double *out = fac + P.slot(t) * P.fac_rec;
  const double *rec = prob + P.in_off0 + P.slot(t) * P.in_rec;
  const double *recn = prob + P.in_off0 + P.slot(t > 0 ? t - 1 : 0) * P.in_rec; // knot t-1 (t = 0: harmless re-read)
// phase boundaries are pinned (sched_barrier) so that the per-phase cycle stamps of
// scripts/trace_wave.py mean what they say; measured: un-pinning them changes nothing
#define GAR_WMARK(id)                                                          \
  __builtin_amdgcn_sched_barrier(0);                                           \
  if (tracing && t == (P.horizon >> 1))                                        \
    P.trace[(id)] = (long long)clock64();                                      \
  __builtin_amdgcn_sched_barrier(0);
  GAR_WMARK(0)
  // ---- vplus = vx' + V' f (:217-218), lane i < NX ------------------------------
  const int ir = lane < NX ? lane : NX - 1;
  if (MODE != 2) {
    // f[k] is broadcast from lane k's register (v_readlane: measured 2.5x cheaper here than
    // wave-uniform LDS reads), V's row from LDS
    double s0 = 0.0, s1 = 0.0;
    double s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int k = 0; k < NX; k += 4) {
      s0 = __builtin_fma(V[ir * PK + k], lane_bcast(S.fi, k), s0);
      s1 = __builtin_fma(V[ir * PK + k + 1], lane_bcast(S.fi, k + 1), s1);
      s2 = __builtin_fma(V[ir * PK + k + 2], lane_bcast(S.fi, k + 2), s2);
      s3 = __builtin_fma(V[ir * PK + k + 3], lane_bcast(S.fi, k + 3), s3);
    }
    s0 += s2;
    s1 += s3;
    if (lane < NX)
      vp[lane] = vnr[lane] + (s0 + s1);
  }
  wave_sync();
  GAR_WMARK(1)
  // ---- [qhat; rhat] = [q; r] + F^T vplus (:227-228) from the operand registers:
  // lane (li,lk) sums its rows 4s+lk, the four lk groups are added by two xor-shuffles
  double hq; // lane j < NW: [qhat; rhat][j]
  const double fi = S.fi;
  if (MODE == 2) { // no next knot inside the leg: qhat = q, rhat = r
    hq = S.qri;
    if (lane >= NX && lane < NW)
      G[(lane - NX) * PG] = hq;
  } else {
    double vps[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
      vps[s] = vp[4 * s + lk];
    double part[TW];
#pragma unroll
    for (int tt = 0; tt < TW; ++tt) {
      double a0 = 0.0, a1 = 0.0;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s & 1)
          a1 = __builtin_fma(S.fo(tt, s), vps[s], a1);
        else
          a0 = __builtin_fma(S.fo(tt, s), vps[s], a0);
      }
      double a = a0 + a1;
      a += __shfl_xor(a, 16);
      a += __shfl_xor(a, 32);
      part[tt] = a; // column 16 tt + li, replicated over lk
    }
    double sel = part[0];
#pragma unroll
    for (int tt = 1; tt < TW; ++tt)
      sel = (lk == tt) ? part[tt] : sel; // lane 16 tt + li picks its own column
    hq = S.qri + sel;
    if (lane >= NX && lane < NW)
      G[(lane - NX) * PG] = hq; // rhat: right-hand side of kff (:248), sign applied in the solve
  }
  GAR_WMARK(2)
  // ---- P = V' F, H = W + F^T P (:216-228), column tile by column tile ----------
  constexpr int TXF = C::REM4 ? TX - 1 : TX; // row tiles of P on the 16x16x4 instruction
  const int i4 = lane & 3, k4 = lane >> 4;   // 4x4x4 A operand: row i4 of the block, k = k4
  // MODE 1: Ghat_u = B^T Vxt' (:286-287).  The rows u of F^T Vxt' are rows NX+u of the tile rows
  // shTile(0..KU-1) -- the same (tile, register) as the Shat^T rows of H; A operand = the F
  // operand registers of those column tiles, B operand = Vxt' (LDS, lane-private)
  constexpr int shLoT = C::shTile(0), shHiT = C::shTile(KU - 1);
  double4_t Gh[shHiT - shLoT + 1][TX];
  if (MODE == 1) {
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
  }
#pragma unroll
  for (int tj = 0; tj < (MODE == 2 ? 0 : TW); ++tj) {
    double4_t Pt[TX];
    double p4 = 0.0; // REM4: P[NX-4+lk][16tj+li], the rows of the last (4-row) tile
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
#pragma unroll
    for (int ti = tj; ti < TW; ++ti) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const double pq = (C::REM4 && (s >> 2) == TX - 1) ? p4 : Pt[s >> 2][s & 3];
        S.Hc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(S.fo(ti, s), pq, S.Hc[ti][tj], 0, 0, 0);
      }
    }
  }
  GAR_WMARK(3)
  // ---- export the control rows: G(u, 1+j) = Shat^T(u, j), M = Rhat (lower) -----
#pragma unroll
  for (int ti = 0; ti < TW; ++ti)
#pragma unroll
    for (int tj = 0; tj <= ti; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + lk + 4 * r, c = 16 * tj + li;
        if (16 * ti + 4 * r >= NX && 16 * ti + 4 * r < NW) { // compile-time: control rows
          if (c < NX)
            G[(row - NX) * PG + 1 + c] = S.Hc[ti][tj][r];
          else if (c <= row) {
            Mm[NC > 0 ? bk_idx<GAR_PACKED_LOWER>(row - NX, c - NX, NK) : (c - NX) * NK + (row - NX)] =
                S.Hc[ti][tj][r];
            if (NC > 0) // pitch-NU copy of Rhat for the decoupled (D = 0) path
              (sb + C::oLr)[(c - NX) * NU + (row - NX)] = S.Hc[ti][tj][r];
          }
        }
      }
  bool d_is_zero = false; // NC > 0: D == 0 on this knot (wave-uniform)
  if (NC > 0) {
    // rows NU.. of G: [d | C]; the KKT matrix [Rhat D^T; D -mu I] (lower, column-major pitch NK):
    // Rhat was placed above with pitch NU -- re-place it, add D and the -mu diagonal
    // (all the loads first, then the LDS writes: the compiler cannot prove that the two do not
    // alias and would otherwise serialise one round trip per element)
    constexpr int NCC = NC > 0 ? (NC * NX + 63) / 64 : 1, NCD = NC > 0 ? (NC * NU + 63) / 64 : 1;
    constexpr int NC1 = NC > 0 ? NC : 1; // (this block is compiled, never run, for NC = 0)
    double tc[NCC], td[NCD];
#pragma unroll
    for (int q = 0; q < NCC; ++q) {
      const int e = 64 * q + lane;
      tc[q] = rec[M::kC + ((64 * q + 63 < NC * NX || e < NC * NX) ? e : NC * NX - 1)];
    }
#pragma unroll
    for (int q = 0; q < NCD; ++q) {
      const int e = 64 * q + lane;
      td[q] = rec[M::kD + ((64 * q + 63 < NC * NU || e < NC * NU) ? e : NC * NU - 1)];
    }
    const double dv = rec[M::kd + (lane < NC ? lane : NC - 1)];
    {
      bool nz = false;
#pragma unroll
      for (int q = 0; q < NCD; ++q)
        nz |= (td[q] != 0.0);
      d_is_zero = (__ballot(nz) == 0ull);
    }
#pragma unroll
    for (int q = 0; q < NCC; ++q) { // C (NC x NX), column-major
      const int e = 64 * q + lane;
      const int c = e / NC1, i = e - c * NC1;
      if (64 * q + 63 < NC * NX || e < NC * NX)
        G[(NU + i) * PG + 1 + c] = tc[q];
    }
    if (lane < NC)
      G[(NU + lane) * PG] = dv;
#pragma unroll
    for (int q = 0; q < NCD; ++q) { // D (NC x NU), column-major
      const int e = 64 * q + lane;
      const int j = e / NC1, i = e - j * NC1;
      if (64 * q + 63 < NC * NU || e < NC * NU)
        Mm[bk_idx<GAR_PACKED_LOWER>(NU + i, j, NK)] = td[q];
    }
    for (int e = lane; e < NC * NC; e += 64) { // -mu I (lower part; BunchKaufman reads Lower)
      const int j = e / NC1, i = e - j * NC1;
      if (i >= j)
        Mm[bk_idx<GAR_PACKED_LOWER>(NU + i, NU + j, NK)] = (i == j) ? -P.mueq : 0.0;
    }
  }
  if (MODE == 1) { // Gt(u, c) = Ghat_u(u, c): the right-hand sides of Kth
#pragma unroll
    for (int sp = 0; sp < KU; ++sp)
#pragma unroll
      for (int tj = 0; tj < TX; ++tj)
        if (16 * tj + 15 < NX || 16 * tj + li < NX)
          Gt[(4 * sp + lk) * PG + 16 * tj + li] = Gh[C::shTile(sp) - shLoT][tj][C::shReg(sp)];
  }
  wave_sync();
  // B of this knot as the A operand of Aff = A + B K: needed a factorisation from now
  double Bop[TX][KU]; // B[16ti+li][4s'+lk]
  double Bop4[KU];    // REM4: B[NX-4+i4][4s'+k4], the A operand of the 4x4x4 blocks
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
  if (MODE == 2) { // Gu = B^T: Gt(u, c) = B(c, u)
#pragma unroll
    for (int sp = 0; sp < KU; ++sp)
#pragma unroll
      for (int tj = 0; tj < TX; ++tj)
        if (16 * tj + 15 < NX || 16 * tj + li < NX)
          Gt[(4 * sp + lk) * PG + 16 * tj + li] = Bop[tj][sp];
    wave_sync();
  }
  GAR_WMARK(4)
  // ---- factor Rhat (lane = row) under the Bunch-Kaufman rule; solve [kff | K] ---
  if (NC > 0) {
    // [kff K; zff Z] = -KKT^{-1} [rhat Shat^T; d C]  (:232-262).
    bool done = false;
    if (d_is_zero) {
      // D = 0 (what the reference's own generator produces, tests/gar/test_util.cpp:42-43): the KKT
      // matrix is block diagonal, Bunch-Kaufman on it is Bunch-Kaufman on Rhat (the columns of the
      // -mu I block never pivot) and [zff | Z] = (-[d | C]) * (1 / -mu) = [d | C] * (1/mu).
      double a_row[NU], nd[NU];
      const int verdict = wave_ldl_fast<NU>(sb + C::oLr, lane, a_row, nd);
      if (verdict == 0) {
        const int col = lane <= NX ? lane : NX;
        double x[NU];
#pragma unroll
        for (int k = 0; k < NU; ++k)
          x[k] = G[k * PG + col];
        ldl_solve_regs_bcast<NU>(a_row, nd, x);
        if (lane <= NX) {
#pragma unroll
          for (int k = 0; k < NU; ++k)
            G[k * PG + col] = x[k];
        }
        const double imu = -(1.0 / -P.mueq); // the stored inverse pivot of the -mu block, negated
#pragma unroll
        for (int i = 0; i < NC; ++i)
          if (lane <= NX)
            G[(NU + i) * PG + col] *= imu;
        wave_sync();
        done = true;
      }
    }
    if (!done) { // Bunch-Kaufman with interchanges and 2x2 pivots: the reference's kktChol
      for (int e = lane; e < NK * PG; e += 64)
        G[e] = -G[e];
      wave_sync();
      failed |= wave_kkt_factor_solve<NK, C::BKS, PG, NX + 1>(Mm, sb + C::oBk, G, lane);
    }
  } else {
    double a_row[NU], nd[NU];
    const int verdict = wave_ldl_fast<NU>(Mm, lane, a_row, nd);
    GAR_WMARK(5)
    if (verdict == 0) {
      const int col = lane <= NX ? lane : NX; // G column: 0 = kff, 1 + j = K(:, j)
      double x[NU];
#pragma unroll
      for (int k = 0; k < NU; ++k)
        x[k] = G[k * PG + col];
      GAR_WMARK(11)
      if (!PRM) {
        ldl_solve_regs_bcast<NU>(a_row, nd, x); // [kff | K] = -Rhat^{-1} [rhat | Shat^T]  (:248-262)
      } else {
        const int ct = lane < NX ? lane : NX - 1; // Kth = -Rhat^{-1} Ghat_u (:288-291)
        double y[NU];
#pragma unroll
        for (int k = 0; k < NU; ++k)
          y[k] = Gt[k * PG + ct];
        ldl_solve_regs_bcast2<NU>(a_row, nd, x, y);
        if (lane < NX) {
#pragma unroll
          for (int k = 0; k < NU; ++k)
            Gt[k * PG + ct] = y[k];
        }
      }
      GAR_WMARK(12)
      if (lane <= NX) {
#pragma unroll
        for (int k = 0; k < NU; ++k)
          G[k * PG + col] = x[k]; // in place: G now holds [kff | K]
      }
      if (MODE == 3) { // the factorisation, for wave B's Kth
        double *Lr = sb + C::oLr, *ndi = sb + C::oDi;
        if (lane < NU) {
#pragma unroll
          for (int j = 0; j < NU; ++j)
            Lr[lane * NU + j] = a_row[j];
        }
        if (lane == 0) {
#pragma unroll
          for (int j = 0; j < NU; ++j)
            ndi[j] = nd[j];
          sb[C::oFlag] = 0.0;
        }
      }
      wave_sync();
    } else {
      failed |= wave_slow_factor_solve<NX, NU, PRM>(sb, lane, P.slow);
    }
  }
  GAR_WMARK(6)
  // ---- K as the B operand of Aff and Vxx: Kb[tj][s'] = K[4s'+lk][16tj+li] --------
  double Kb[TX][KU];
#pragma unroll
  for (int tj = 0; tj < TX; ++tj) {
    const int cc = (16 * tj + li) < NX ? (16 * tj + li) : NX - 1;
#pragma unroll
    for (int s = 0; s < KU; ++s) {
      Kb[tj][s] = G[(4 * s + lk) * PG + 1 + cc];
      if (16 * tj + li < NX) // fbT2(4s+lk, 16tj+li) = 8tj*2NW + 8s + [(li>>1)*2NW + 2lk + (li&1)]
        stg_b(out, M::fFB + 8 * tj * 2 * NR + 8 * s, L.fbl, Kb[tj][s]);
    }
  }
  double Kthb[TX][KU]; // PRM: Kth[4s'+lk][16tj+li], fth rows 0..NU-1 (same device order as fb)
  if (PRM) {
#pragma unroll
    for (int tj = 0; tj < TX; ++tj) {
      const int cc = (16 * tj + li) < NX ? (16 * tj + li) : NX - 1;
#pragma unroll
      for (int s = 0; s < KU; ++s) {
        Kthb[tj][s] = Gt[(4 * s + lk) * PG + cc];
        if (16 * tj + li < NX)
          stg_b(out, C::pFTH + 8 * tj * 2 * NW + 8 * s, L.fbl, Kthb[tj][s]);
      }
    }
  }
  // ---- kff; yff = f + B kff (:266); vx = qhat + Shat kff (:275-276) ----------------
  // B and Shat are read from the operand registers (rows on li, u = 4s'+lk): partial sums
  // over this lane's u, the four lk groups added by two xor-shuffles
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
      py[ti] = a; // (B kff)[16 ti + li]
      pv[ti] = c; // (Shat kff)[16 ti + li]
    }
    double sy = py[0], sv = pv[0];
#pragma unroll
    for (int ti = 1; ti < TX; ++ti) {
      sy = (lk == ti) ? py[ti] : sy;
      sv = (lk == ti) ? pv[ti] : sv;
    }
    double cz = 0.0; // NC > 0: (C^T zff)[lane]  (:275-276)
    if (NC > 0) {
      const int xc = lane < NX ? lane : NX - 1;
      double c0 = 0.0, c1 = 0.0;
#pragma unroll
      for (int i = 0; i < NC; i += 2) {
        c0 = __builtin_fma(rec[M::kC + xc * NC + i], G[(NU + i) * PG], c0);
        c1 = __builtin_fma(rec[M::kC + xc * NC + i + 1], G[(NU + i + 1) * PG], c1);
      }
      cz = c0 + c1;
      if (lane < NC)
        out[M::fFF + NU + lane] = G[(NU + lane) * PG]; // zff
      // Z rows of fb (device order fbT2)
      for (int e = lane; e < NC * NX; e += 64) {
        const int j = e / NC, i = e - j * NC;
        out[M::fFB + M::fbT2(NU + i, j)] = G[(NU + i) * PG + 1 + j];
      }
    }
    const double yf = fi + sy, vxv = hq + sv + cz;
    if (lane < NU)
      out[M::fFF + lane] = G[lane * PG];
    if (lane < NX) {
      out[M::fFF + NK + lane] = (MODE == 2) ? 0.0 : yf; // terminalSolve leaves yff untouched (zero)
      out[ovx + lane] = vxv;
      vn[lane] = vxv;
    }
    if (MODE == 2) { // vt = gamma + Gu^T kff = f + B kff (:189-190)
      if (lane < NX) {
        vtl[lane] = yf;
        out[C::pvt + lane] = yf;
      }
    }
    if (MODE == 3) { // published for wave B's vt = vt' + Vxt'^T yff
      if (lane < NX)
        yfl[lane] = yf;
    }
    if (MODE == 1) { // vt = vt' + Vxt'^T yff (:298-301): Vxt' from the operand slots, yff via LDS
      if (lane < NX)
        yfl[lane] = yf;
      wave_sync();
      double ys[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s)
        ys[s] = yfl[4 * s + lk];
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
        pt[tj] = a; // (Vxt'^T yff)[16 tj + li]
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
  }
  if (MODE == 3)
    __syncthreads(); // [kff | K], the factorisation and yff are published: wave B goes on
  GAR_WMARK(7)
  // ---- Aff = A + B K (:267), in place on the F operand registers ------------------
  // A[16ti+lk+4r][16tj+li] IS the F operand F[4s+lk][16tj+li] with s = 4ti + r.  The MFMAs of
  // one k-step go round all the tiles (independent accumulators, back-to-back issue); the
  // stores follow when every tile is done.
  double4_t accT[TX]; // the tile row made of the tail k-steps (rows 16*KSF ..)
  if (C::KST > 0 && !C::REM4) {
#pragma unroll
    for (int tj = 0; tj < TX; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        accT[tj][r] = (r < C::KST) ? S.FoT[tj][r] : 0.0;
  }
#pragma unroll
  for (int s = 0; s < (MODE == 2 ? 0 : KU); ++s)
#pragma unroll
    for (int tj = 0; tj < TX; ++tj)
#pragma unroll
      for (int ti = 0; ti < TX; ++ti) {
        if (ti < C::KSF)
          S.Fo[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bop[ti][s], Kb[tj][s], S.Fo[tj][ti], 0, 0, 0);
        else if (C::REM4) // rows NX-4 .. NX-1: four 4x4x4 blocks, in place on the tail operand
          S.FoT[tj][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(Bop4[s], Kb[tj][s], S.FoT[tj][0], 0, 0, 0);
        else
          accT[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(Bop[ti][s], Kb[tj][s], accT[tj], 0, 0, 0);
      }
#pragma unroll
  for (int tj = 0; tj < TX; ++tj)
#pragma unroll
    for (int ti = 0; ti < TX; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * ti + lk + 4 * r, j = 16 * tj + li;
        if (16 * ti + 4 * r < NX) { // compile-time
          if (i < NX && j < NX) // fbT2(NU+i, j), i = 16ti+4r+lk
            stg_b(out, M::fFB + 8 * tj * 2 * NR + 2 * (NK + 16 * ti + 4 * r), L.fbl,
                  MODE == 2 ? 0.0 // terminalSolve leaves the Aff rows untouched (zero)
                            : (ti < C::KSF ? S.Fo[tj][ti][r] : (C::REM4 ? S.FoT[tj][0] : accT[tj][r])));
        }
      }
  if (PRM) {
    // ---- Yth = B Kth (:295), fth rows NU.. (MODE 2: those rows stay zero) ------------------
#pragma unroll
    for (int tj = 0; tj < TX; ++tj)
#pragma unroll
      for (int ti = 0; ti < TX; ++ti) {
        double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
        if (MODE == 1) {
#pragma unroll
          for (int s = 0; s < KU; ++s)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Bop[ti][s], Kthb[tj][s], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * ti + lk + 4 * r, j = 16 * tj + li;
          if (16 * ti + 4 * r < NX) {
            if (i < NX && j < NX)
              stg_b(out, C::pFTH + 8 * tj * 2 * NW + 2 * (NU + 16 * ti + 4 * r), L.fbl, acc[r]);
          }
        }
      }
    // ---- Vxt (NX x nth), one parameter column tile at a time; the new tile column replaces
    // the old one in the lane-private operand slots once all its row tiles are done
    //   MODE 1: Vxt = Aff^T Vxt' (:304-306): A operand = the Aff registers (operand layout of
    //           the F they were computed on), B operand = Vxt'
    //   MODE 2: Vxt = A^T + K^T B^T (:185-186): A operand = K as loaded for Aff, B operand = B
    //           as loaded for Aff, accumulator initialised with A^T from the knot record
#pragma unroll
    for (int tj = 0; tj < TX; ++tj) {
      double4_t nv[TX];
#pragma unroll
      for (int ti = 0; ti < TX; ++ti) {
        double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
        if (MODE == 1) {
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            // Aff(4s+lk, 16ti+li): in place on the F operand, except the tail k-steps of a
            // shape without the 4-row remainder tile, which sit in accT
            const double aq = (s < 4 * C::KSF || C::REM4) ? S.fo(ti, s) : accT[ti][s - 4 * C::KSF];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, Xt[(s * TX + tj) * 64], acc, 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) // A^T(16ti+lk+4r, 16tj+li) = A(16tj+li, 16ti+4r+lk)
            if (16 * ti + 4 * r < NX)
              acc[r] = ldg_b(rec, M::kA + (16 * ti + 4 * r) * NX + (16 * tj + 15 < NX ? 16 * tj : 0),
                             16 * tj + 15 < NX ? lkx : 8u * (unsigned)(lk * NX + ((16 * tj + li) < NX ? 16 * tj + li : NX - 1)));
#pragma unroll
          for (int s = 0; s < KU; ++s)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Kb[ti][s], Bop[tj][s], acc, 0, 0, 0);
        }
        nv[ti] = acc;
      }
#pragma unroll
      for (int ti = 0; ti < TX; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * ti + 4 * r < NX) { // rows 16ti+4r+lk < NX: k-step 4ti+r of the next stage
            Xt[((4 * ti + r) * TX + tj) * 64] = nv[ti][r];
            if (16 * tj + 15 < NX || 16 * tj + li < NX) // column-major (x, theta)
              stg_b(out, C::pVxt + 16 * tj * NX + 16 * ti + 4 * r, 8u * (unsigned)(li * NX + lk), nv[ti][r]);
          }
    }
    // ---- Vtt = Gth + Vtt' + Gu^T Kth + Vxt'^T Yth (:308-310) as Vtt' + Ghat_u^T Kth; computed
    // transposed (D' = Kth^T Ghat_u), so that the accumulator row index runs along the lanes that
    // are contiguous in the column-major record
#pragma unroll
    for (int ti = 0; ti < TX; ++ti)
#pragma unroll
      for (int tj = 0; tj < TX; ++tj) {
        double4_t acc = double4_t{0.0, 0.0, 0.0, 0.0};
        if (MODE == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * ti + 4 * r < NX)
              acc[r] = Tt[((4 * ti + r) * TX + tj) * 64];
        }
#pragma unroll
        for (int s = 0; s < KU; ++s)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(
              Kthb[ti][s], MODE == 1 ? Gh[C::shTile(s) - shLoT][tj][C::shReg(s)] : Bop[tj][s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * ti + 4 * r < NX) {
            Tt[((4 * ti + r) * TX + tj) * 64] = acc[r];
            // D'(i, j) = Vtt(j, i): column-major element (j, i) at i*NX + j
            if (16 * tj + 15 < NX || 16 * tj + li < NX)
              stg_b(out, C::pVtt + (16 * ti + 4 * r) * NX + 16 * tj, lkx, acc[r]);
          }
      }
  }
  // ---- knot t-1: the F operands and vectors go into the registers Aff just released
  wave_load_a<NX, NU>(recn, L, S);
  GAR_WMARK(8)
  // ---- Vxx = Qhat + Shat K (:272-273), lower tiles, mirrored into LDS --------------
  // Shat(16ti+li, 4s'+lk) is register shReg(s') of H tile (shTile(s'), ti).  Tiles of those
  // tile rows accumulate in copies (their other registers are still operands); every other
  // tile accumulates in place.  One k-step goes round all the tiles.
  constexpr int shLo = C::shTile(0);
  double4_t accS[TX][TX];
  double acc4[TX]; // REM4: rows NX-4 .. NX-1 of Vxx (row on lane>>4, column 16tj + lane&15)
  double sh4[KU];  // REM4: Shat(NX-4+i4, 4s'+k4) = H(NX+4s'+k4, NX-4+i4), from lane 16 k4 + (NX-4)%16 + i4
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
  if (NC > 0) { // Vxx += C^T Z (:272-274): A operand C^T from the knot record, B operand Z from G
#pragma unroll
    for (int sc = 0; sc < KC; ++sc) {
      double Zb[TX], Cop[TX], Cop4 = 0.0;
#pragma unroll
      for (int tj = 0; tj < TX; ++tj) {
        const int cc = (16 * tj + li) < NX ? (16 * tj + li) : NX - 1;
        Zb[tj] = G[(NU + 4 * sc + lk) * PG + 1 + cc];   // Z(4sc+lk, 16tj+li)
        Cop[tj] = rec[M::kC + cc * NC + 4 * sc + lk];   // C(4sc+lk, 16tj+li) = C^T(16tj+li, 4sc+lk)
      }
      if (C::REM4)
        Cop4 = rec[M::kC + (NX - 4 + i4) * NC + 4 * sc + k4];
#pragma unroll
      for (int tj = 0; tj < TX; ++tj)
#pragma unroll
        for (int ti = tj; ti < TX; ++ti) {
          if (C::REM4 && ti == TX - 1)
            acc4[tj] = __builtin_amdgcn_mfma_f64_4x4x4f64(Cop4, Zb[tj], acc4[tj], 0, 0, 0);
          else if (ti >= shLo)
            accS[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(Cop[ti], Zb[tj], accS[ti][tj], 0, 0, 0);
          else
            S.Hc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(Cop[ti], Zb[tj], S.Hc[ti][tj], 0, 0, 0);
        }
    }
  }
  // branch-free: a lane whose element is outside the lower triangle writes to a dump slot
  // (its address is a loop-invariant select), so the whole phase is one scheduling region
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
  // ---- knot t-1: its Hessian tiles replace H  (NC > 0 is the serial constrained chain, whose records keep Q and R
  // as packed lower triangles like the plain sweep's: gar_layout.h; the leg kernels run this stage with NC = 0)
  wave_load_b<NX, NU, WaveLane<NX, NU, NC>, (GAR_QR_PACKED && NC > 0 && !MfmaCfg<NX, NU, NC>::WIDE)>(recn, L, S);
  GAR_WMARK(9)
  // ---- Vxx -> HBM, 16 B per lane: column-major and symmetric, or (PACKV: the serial family, gar_layout.h) the
  // packed lower triangle
  wave_flush_vxx<NX, PACKV, PK>(V, out + oVxx, lane);
  GAR_WMARK(10)
#undef GAR_WMARK
}

} // namespace gar
#include "gar_wave2.hpp"
namespace gar {

// Constrained sweeps (NC > 0) are a chain of three kernels on one stream, PHASE = 0, 1, 2:
//   0  the decoupled stage (gar_wave2.hpp: D = 0, Bunch-Kaufman keeps Rhat's natural order);
//   1  the coupled stage (any D; register LDL^T of the (NU+NC) x (NU+NC) reduced KKT matrix while
//      Bunch-Kaufman keeps its natural order);
//   2  the stage with the LDS Bunch-Kaufman (interchanges, 2x2 pivots: the reference's kktChol).
// A kernel that meets a knot it does not serve records the knot in P.resume[b] and leaves -- nothing
// irreversible has happened: the deferred flush of the previous Vxx is complete by then -- and the next
// kernel picks the problem up at that knot (V', vx' from the record) and keeps it to the end or to the
// first knot IT does not serve.  Problems that are finished cost the later kernels an early exit; no host
// synchronisation.  One kernel with a branch per stage was measured first: the register allocation of
// the union (724 B of scratch per lane) made a decoupled stage 63 k cycles instead of 37 k.
template <int NX, int NU, int NC, int PHASE>
__device__ __forceinline__ void gar_backward_wave_body(const MfmaParams &P, int batch) {
  using C = WaveCfg<NX, NU, NC>;
  using M = MfmaCfg<NX, NU, NC>;
  constexpr int PK = C::PK;
  const int lane = (int)threadIdx.x & 63;
  // wave-uniform by construction; readfirstlane tells the compiler, so that every global
  // access is "scalar base + 32-bit lane offset + immediate" (no 64-bit VALU address math)
  // one 64-thread workgroup per problem: the LDS slice starts at the (link-time constant) base
  // of the dynamic LDS, so every ds address is "lane pattern + immediate"
  const int b = (int)blockIdx.x;
  if (b >= batch)
    return;
  double *sm = gar_smem;
  const double *prob = P.prob + (long long)b * P.prob_stride;
  double *fac = P.fac + (long long)b * P.fac_stride;
  const int N = P.horizon;
  double *V = sm + C::oV, *vn = sm + C::oVn;
  // cycle stamps (scripts/trace_wave.py) only in the debug build (make trace, -DGAR_TRACE): the
  // run-time test alone costs 1 % of the sweep
#ifdef GAR_TRACE
  const bool tracing = P.trace != nullptr && b == 0 && lane == 0;
#else
  const bool tracing = false;
#endif

  constexpr bool BK_RESUME = PHASE > 0;
  static_assert(PHASE == 0 || NC > 0, "the unconstrained stage handles its own pivoting");
  int tstart = N - 1;
  if constexpr (BK_RESUME) {
    tstart = P.resume[b];
    if (tstart < 0)
      return;
  }
  // Q, R of the knots as packed lower triangles (gar_layout.h): the one-wave serial sweeps
  constexpr bool QP = GAR_QR_PACKED && !M::WIDE; // (the constrained chain NC > 0 too)
  WaveLane<NX, NU, NC> L;
  wave_lane_init<NX, NU, NC, QP>(L, lane);
  WaveStage<NX, NU> S;
  if (N > 0) {
    wave_load_a<NX, NU>(prob + P.in_off0 + P.slot(tstart) * P.in_rec, L, S);
    wave_load_b<NX, NU, WaveLane<NX, NU, NC>, QP>(prob + P.in_off0 + P.slot(tstart) * P.in_rec, L, S);
  }

  // ---- terminal knot (terminalSolve, nu = 0, :146-149, :175-178): Z = C/mu, zff = d/mu,
  // Vxx = Q + C^T Z, vx = q + C^T zff (nc = 0: Vxx = Q, vx = q)
  if (BK_RESUME && tstart < N - 1) {
    // V' = Vxx, vx' = vx of knot tstart+1: complete in its record (the first kernel's deferred flush of
    // that Vxx runs at the start of the stage it then abandoned)
    const double *rn = fac + P.slot(tstart + 1) * P.fac_rec;
    for (int e = lane; e < NX * NX; e += 64) {
      const int j = e / NX, i = e - j * NX;
      V[i * PK + j] = rn[M::fVxx + gar_sym_index(GAR_VXX_PACKED && !M::WIDE, NX, i, j)]; // (packed lower triangle: gar_layout.h)
    }
    if (lane < NX)
      vn[lane] = rn[M::fvx + lane];
  } else {
    const double *rec = prob + P.in_offN;
    double *out = fac + P.fac_offN;
    for (int e = lane; e < NX * NX; e += 64) {
      const int j = e / NX, i = e - j * NX; // column-major element (i, j)
      double v = (i >= j) ? rec[M::tQ + e] : rec[M::tQ + i * NX + j];
      for (int k = 0; k < NC; ++k) // (C^T Z)(i, j), Z = C / mu
        v = __builtin_fma(rec[M::tC + i * NC + k], rec[M::tC + j * NC + k] / P.mueq, v);
      V[i * PK + j] = v; // symmetrised from lower, as the consumer stage does (:216)
      if (M::WIDE || !GAR_VXX_PACKED)
        out[M::tVxx + e] = v;
      else if (i >= j)
        out[M::tVxx + gar_sym_index(1, NX, i, j)] = v; // (packed lower triangle: gar_layout.h)
    }
    if (lane < NX) {
      double v = rec[M::tq + lane];
      for (int k = 0; k < NC; ++k)
        v = __builtin_fma(rec[M::tC + lane * NC + k], rec[M::td + k] / P.mueq, v);
      vn[lane] = v;
      out[M::tvx + lane] = v;
    }
    if (NC > 0) { // ff = [zff; 0], fb = [Z; 0] (row-major: the terminal record keeps the generic layout)
      for (int e = lane; e < NC + NX; e += 64)
        out[e] = e < NC ? rec[M::td + e] / P.mueq : 0.0;
      for (int e = lane; e < (NC + NX) * NX; e += 64) {
        const int i = e / NX, j = e - i * NX;
        out[(NC + NX) + e] = i < NC ? rec[M::tC + j * NC + i] / P.mueq : 0.0;
      }
    }
  }
  wave_sync();
  int failed = 0;
  // gar_wave2.hpp: a stage leaves its Vxx in LDS and the NEXT stage copies it to HBM behind its
  // MFMAs; the first stage re-writes the terminal Vxx (same values), the last one is flushed below
  [[maybe_unused]] double *vflush = (BK_RESUME && tstart < N - 1) ? fac + P.slot(tstart + 1) * P.fac_rec + M::fVxx
                                                                   : fac + P.fac_offN + M::tVxx;
  for (int t = tstart; t >= 0; --t) {
    if constexpr (NC == 0) {
#if GAR_SWEEP_REFRESH_LANE
      const int lane_t = lane + fence0(S.fi);
      WaveLane<NX, NU, NC> Lt;
      wave_lane_init<NX, NU, NC, QP>(Lt, lane_t);
      wave_stage2<NX, NU, 0, false, (GAR_F_DMA != 0) && !M::WIDE>(P, sm, prob, fac, t, lane_t, Lt, S, failed, vflush, tracing);
#else
      wave_stage2<NX, NU, 0, false, (GAR_F_DMA != 0) && !M::WIDE>(P, sm, prob, fac, t, lane, L, S, failed, vflush, tracing);
#endif
    } else {
      if constexpr (PHASE == 2) {
        if (lane == 0)
          atomicAdd(&P.slow[3], 1);
        wave_stage<NX, NU, 0, 0, NC, GAR_VXX_PACKED != 0>(P, sm, prob, fac, t, lane, L, S, failed, tracing);
      } else {
        if (PHASE == 1 && lane == 0)
          atomicAdd(&P.slow[2], 1);
#if GAR_COUPLED_REFRESH_LANE
        // The coupled stage held every one of the 512 registers AND 768 bytes of scratch per lane: the ~100 loop-invariant
        // lane offsets of WaveLane (and everything else the compiler derives from the lane index once, outside the stage
        // loop) lived in scratch and came back through vmcnt(0) waits inside the stage -- 1.42 x the algorithmic bytes
        // in FETCH_SIZE (round 6, profiles/r06_pmc_and_sq_secondary_shapes.json).  Re-deriving them per stage from a
        // lane index the compiler cannot prove loop-invariant costs ~100 integer instructions per stage and leaves the
        // kernel at 432 registers, no scratch: backward 12.63 -> 7.98 ms at batch 1 024, N = 256 (0.198 -> 0.314 of the
        // HBM roofline), results BITWISE the same (profiles/r06_ab_coupled_lane_offsets_rederived_per_stage.log).
        constexpr bool REFRESH = PHASE == 1 || (GAR_CSTR_REFRESH_LANE != 0);
        const int lane_t = REFRESH ? lane + fence0(S.fi) : lane;
        WaveLane<NX, NU, NC> Lt;
        if constexpr (REFRESH)
          wave_lane_init<NX, NU, NC, QP>(Lt, lane_t);
        if (!wave_stage2<NX, NU, NC, PHASE == 1>(P, sm, prob, fac, t, lane_t, REFRESH ? Lt : L, S, failed, vflush, tracing)) {
#else
        if (!wave_stage2<NX, NU, NC, PHASE == 1>(P, sm, prob, fac, t, lane, L, S, failed, vflush, tracing)) {
#endif
          if (lane == 0) { // over to the next kernel of the chain from this knot on
            P.resume[b] = t;
            if (PHASE == 1)
              atomicAdd(&P.slow[2], -1);
            if (failed)
              atomicOr(&P.status[b], failed);
          }
          return;
        }
      }
    }
  }
  if constexpr (NC > 0 && PHASE < 2) {
    if (lane == 0)
      P.resume[b] = -1;
  }
  if constexpr (PHASE < 2) {
    if (N > 0)
      wave_flush_vxx<NX, GAR_VXX_PACKED && !M::WIDE, PK>(V, vflush, lane);
  }
  // ---- initial stage (proximal-riccati.hxx:42-60), fused: kkt0 = [Vxx0 G0^T; G0 0] is
  // Bunch-Kaufman-factorised by this wave right away (packed lower triangle in LDS, read from
  // the V and vx this wave still holds) and solved for kkt0.ff = -kkt0^{-1} [vx0; g0]
  if (P.init != nullptr) {
    const int nc0 = P.nc0, n0 = NX + nc0;
    const double vx0 = vn[lane < NX ? lane : NX - 1];
    wave_sync();
    const double *G0 = prob + P.G0_off, *g0 = prob + P.g0_off;
    // The initial condition of every trajectory-optimisation use is "x0 given": G0 = -I (or +I),
    // g0 = -+x0 (tests/gar/test_util.cpp:72-74, riccati.cpp:56-57).  Then the KKT system
    // [Vxx0 G0^T; G0 0] [x0; lbd0] = -[vx0; g0] has the closed form x0 = -s g0,
    // lbd0 = -s (vx0 + Vxx0 x0), s = +-1: detected at run time, 36 FMAs instead of a 72 x 72
    // Bunch-Kaufman factorisation (0.45 M cycles, 6 % of a wave's sweep).  P.init_closed = 0
    // (GAR_HIP_INIT=bk) keeps the factorisation for every G0.
    double sgn = 0.0;
    if (P.init_closed && nc0 == NX) {
      const double g00 = G0[0];
      bool ok = (g00 == 1.0 || g00 == -1.0);
      for (int e = lane; e < NX * NX; e += 64) {
        const int j = e / NX, i = e - j * NX;
        ok &= (G0[e] == (i == j ? g00 : 0.0));
      }
      if (__ballot(!ok) == 0ull)
        sgn = g00;
    }
    if (sgn != 0.0) {
      const int ix = lane < NX ? lane : NX - 1;
      const double x0 = -sgn * g0[ix];
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int k = 0; k < NX; k += 2) {
        s0 = __builtin_fma(V[ix * PK + k], lane_bcast(x0, k), s0);
        s1 = __builtin_fma(V[ix * PK + k + 1], lane_bcast(x0, k + 1), s1);
      }
      double *io = P.init + (long long)b * P.init_stride;
      if (lane < NX) {
        io[lane] = x0;
        io[NX + lane] = -sgn * (vx0 + (s0 + s1));
      }
      if (failed && lane == 0)
        atomicOr(&P.status[b], failed);
      if (P.init_small && lane == 0)
        P.resume[b] = 0;
      return;
    }
    if (P.init_small) { // no room for kkt0 in this launch's LDS: gar_initial_wave takes the problem (resume[b] != 0)
      if (lane == 0) {
        P.resume[b] = 1;
        if (failed)
          atomicOr(&P.status[b], failed);
      }
      return;
    }
    double *k0 = sm + C::oK0, *rhs = k0 + n0 * (n0 + 1) / 2, *sub = rhs + n0;
    int *piv0 = (int *)(sub + n0);
    for (int j = 0; j < n0; ++j) // lower triangle, by columns
      for (int i = j + lane; i < n0; i += 64) {
        double v = 0.0;
        if (j < NX)
          v = (i < NX) ? V[i * PK + j] : G0[j * nc0 + (i - NX)];
        k0[bk_idx<GAR_PACKED_LOWER>(i, j, n0)] = v;
      }
    if (lane < NX)
      rhs[lane] = -vx0;
    for (int e = lane; e < nc0; e += 64)
      rhs[NX + e] = -g0[e];
    const WG w1 = wave_self();
    wave_sync();
    if (tracing)
      P.trace[13] = (long long)clock64();
    failed |= 2 * wg_bk_factor<GAR_PACKED_LOWER>(w1, n0, k0, n0, sub, piv0, piv0 + n0);
    if (tracing)
      P.trace[14] = (long long)clock64();
    wg_bk_solve<GAR_PACKED_LOWER>(w1, n0, k0, n0, sub, piv0, rhs, 1, 0, 1);
    if (tracing)
      P.trace[15] = (long long)clock64();
    double *io = P.init + (long long)b * P.init_stride;
    for (int e = lane; e < n0; e += 64)
      io[e] = rhs[e];
  }
  if (failed && lane == 0)
    atomicOr(&P.status[b], failed);
}

template <int NX, int NU, int NC = 0>
__global__ void __launch_bounds__(64, 1) gar_backward_wave(MfmaParams P, int batch) {
  gar_backward_wave_body<NX, NU, NC, 0>(P, batch);
}
// the same sweep under a name of its own for the HALF-batch launches of the pipelined schedule (gar_hip_set_pipeline):
// a profile of a run that uses both schedules then lists the two launch populations apart
template <int NX, int NU>
__global__ void __launch_bounds__(64, 1) gar_backward_wave_half(MfmaParams P, int batch) {
  gar_backward_wave_body<NX, NU, 0, 0>(P, batch);
}
// The plain unconstrained shapes are instantiated ONCE, in gar_wave_sweep.cpp (a translation unit of its own: see
// there); every other translation unit sees them as extern templates.
#define GAR_SWEEP_SHAPES(X) X(36, 12) X(32, 12) X(16, 8) X(12, 8) X(12, 4) X(8, 4)
// constrained sweeps, second and third kernel of the chain (see gar_backward_wave_body)
template <int NX, int NU, int NC>
__global__ void __launch_bounds__(64, 1) gar_backward_wave_coupled(MfmaParams P, int batch) {
  gar_backward_wave_body<NX, NU, NC, 1>(P, batch);
}
template <int NX, int NU, int NC>
__global__ void __launch_bounds__(64, 1) gar_backward_wave_bk(MfmaParams P, int batch) {
  gar_backward_wave_body<NX, NU, NC, 2>(P, batch);
}

} // namespace gar
