// gar_forward_lean.hpp -- the roll-out (computeInitial + forwardImpl, riccati-kernel.hxx:195-207, 314-377,
// nc = nth = 0) cut to fit BESIDE a backward sweep: the pipelined schedule of gar_hip_set_pipeline runs the forward
// sweep of one half of the batch while the backward sweep of the other half holds the chip (gar_pipeline.hpp).
//
// gar_backward_wave<36,12> allocates 432 of a SIMD's 512 registers and 18.5 KB of LDS per wave.  A forward wave that
// fits in the remaining 80 registers shares the SIMD with it:
//
//  * one 256-thread workgroup = four independent waves, one problem each (no workgroup barrier anywhere);
//  * the stage's factor record never passes through registers on its way in: `global_load_lds_dwordx4` (gfx950)
//    copies [ff | fb] (13.9 KB, the fbT2 image as it lies in HBM) and the next stage's packed Vxx' | vx' (5.6 KB)
//    straight into this wave's LDS slice, 1 KiB per instruction; each buffer is refilled for the NEXT stage the
//    moment its rows have been consumed, so a whole stage (20 KB) is in flight per wave with no register cost;
//  * rows come back with ds_read_b128 (lane = row, conflict-free) in the SAME order and with the SAME two
//    accumulators as gar_forward_mfma: the solutions of the two kernels are bitwise identical (66 registers);
//  * the workgroup ASKS for more than half of a CU's LDS (85 KB; it uses 84), so at most ONE such workgroup lives on a
//    CU -- one forward wave per SIMD, never two (two would take the backward wave's registers) -- and 4 backward waves
//    + 1 forward workgroup fill the CU's 160 KB exactly (gar_pipeline.hpp: pipe_plan).
//
// Measured (profiles/r05_*): alone it is as fast as gar_forward_mfma to 0.5 % on the same records (the
// roll-out's time is the memory system's); resident beside a backward sweep both slow down by what the other adds --
// the two sweeps contend for the CU's vector-memory path, the pipelined schedule gains the tails only (+1 ... 3 %).
#pragma once
#include "gar_mfma.hpp"

namespace gar {

template <int NX, int NU> struct LeanFwdCfg {
  using C = MfmaCfg<NX, NU, 0>;
  static constexpr int NW = C::NW;
  // LDS slice of one wave (bytes): [ff | fb] image, then the packed Vxx' image, then vx'
  static constexpr int FB_BYTES = 8 * (C::NR + C::NR * NX);                       // 14 208 at (36, 12)
  static constexpr int FB_PIECES = (FB_BYTES + 1023) / 1024;                      // 1 KiB per DMA instruction
  static constexpr int VX_BYTES = 8 * (NX * (NX + 1) / 2);                        // packed lower triangle
  static constexpr int VX_PIECES = (VX_BYTES + 1023) / 1024;
  static constexpr int oFB = 0;
  static constexpr int oVX = FB_PIECES * 1024;
  static constexpr int ovx = oVX + VX_PIECES * 1024;                              // vx' (NX doubles, one piece)
  static constexpr int SLICE = ovx + 1024;
  static_assert(FB_BYTES % 16 == 0 && VX_BYTES % 16 == 0 && (8 * NX) % 16 == 0, "16-byte DMA pieces");
  static_assert(8 * NX <= 1024, "vx' is one piece");
  static constexpr int WAVES = 4;
  static constexpr int USED = WAVES * SLICE;
};

// a value the compiler must take as new in every iteration: keeps the per-lane LDS addresses derived from it
// from being hoisted out of the stage loop (36 + 18 loop-invariant addresses would cost the registers the
// kernel exists to do without)
__device__ __forceinline__ int lean_opaque(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#endif
  return v;
}

// values pinned to a point of the instruction stream (an empty volatile asm statement that "rewrites" them): what
// consumes them cannot be scheduled above it, what produced them not below; volatile statements keep their order
__device__ __forceinline__ void lean_pin(double &a) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(a));
#endif
}
__device__ __forceinline__ void lean_pin(double &a, double &b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(a), "+v"(b));
#endif
}
__device__ __forceinline__ void lean_pin(double2_t &a) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(a));
#endif
}

// (80 registers: what gar_backward_wave<36,12> leaves of a SIMD's 512 -- six waves per SIMD is the same budget)
#if defined(__HIPCC__)
#define GAR_LEAN_BUDGET __attribute__((amdgpu_waves_per_eu(6, 6)))
#else
#define GAR_LEAN_BUDGET
#endif
template <int NX, int NU>
__global__ void __launch_bounds__(256) GAR_LEAN_BUDGET gar_forward_lean(MfmaFwdParams P, int nb) {
  using C = MfmaCfg<NX, NU, 0>;
  using L = LeanFwdCfg<NX, NU>;
  constexpr int NW = C::NW;
  const int lane = (int)threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int b = (int)blockIdx.x * L::WAVES + wv;
  if (b >= nb)
    return;
  const double *fac = P.fac + (long long)b * P.fac_stride;
  double *sol = P.sol + (long long)b * P.sol_stride;
  const double *io = P.init + (long long)b * P.init_stride;
  const int N = P.horizon;
  char *sl = reinterpret_cast<char *>(gar_smem) + wv * L::SLICE;
  // row of [K; Aff] this lane owns (a control row or a next-state row), row of Vxx'
  const int r0 = lane < NW ? lane : C::NR - 1;
  const int iv0 = lane < NX ? lane : NX - 1;
  const int ix = (lane >= NU && lane < NW) ? lane - NU : 0;

  // (past the end: a harmless re-read of the last stage -- every stage issues the same number of pieces)
  auto fill_fb = [&](int t) {
    const int tc = t < N ? t : N - 1;
    wave_dma<L::FB_BYTES>(fac + P.slot(tc) * P.fac_rec + C::fFF, sl + L::oFB, lane);
  };
  auto fill_vx = [&](int t) { // value function of stage t + 1 (the terminal knot's when t + 1 >= N)
    const double *recn = (t + 1 < N) ? fac + P.slot(t + 1) * P.fac_rec : fac + P.fac_offN;
    const int oV = (t + 1 < N) ? C::fVxx : C::tVxx, ov = (t + 1 < N) ? C::fvx : C::tvx;
    wave_dma<L::VX_BYTES>(recn + oV, sl + L::oVX, lane);
    wave_dma<8 * NX>(recn + ov, sl + L::ovx, lane);
  };
  constexpr int FBP = L::FB_PIECES, VXP = L::VX_PIECES + 1; // vector-memory instructions per refill
  if (N > 0) {
    fill_fb(0);
    fill_vx(0);
  }
  double xs = io[ix]; // x0 from the initial-stage solve (kkt0.ff)
  if (lane >= NU && lane < NW)
    sol[ix] = xs;
  for (int e = lane; e < P.nc0; e += 64)
    sol[P.sol_l + e] = io[NX + e]; // lbd0
  GAR_WAIT_VMCNT(0); // stage 0 has landed

  // Vector-memory instructions of a stage, in program order: FBP pieces of [ff | fb](t+1), WAIT, the store of
  // u / x', VXP pieces of Vxx' | vx' (t+2), WAIT, the store of lbd'.  gfx9 retires them in order on ONE counter, so
  // "everything but the k youngest instructions has completed" is s_waitcnt vmcnt(k).  Each wait sits right behind
  // the pieces it lets fly (k = their number) and right in front of a store: what it waits for is the OTHER buffer's
  // pieces, requested half a stage ago, and the store issued behind the previous wait -- however many instructions
  // the compiler makes of that store.
  for (int t = 0; t < N; ++t) {
    // ---- u = kff + K x ; x' = yff + Aff x   (:334-336, :360-361); two accumulators (as gar_forward_mfma)
    wave_sync();
    const int r = lean_opaque(r0);
    const double *fbb = reinterpret_cast<const double *>(sl + L::oFB); // [ff | fb] of this stage
    double acc = fbb[C::fFF + r], acc1 = 0.0;
    {
      // column pairs in chunks of CH, software-pipelined by hand: the reads of chunk k + 1 are issued, then chunk k
      // is evaluated; the pins keep it that way (left alone, the compiler issues every read of the stage first
      // and spills them)
      constexpr int CH = 6, NCH = (NX / 2 + CH - 1) / CH;
      double2_t g[2][CH];
      auto rd = [&](int k, double2_t *dst) {
        const int rk = lean_opaque(r);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int m = k * CH + c;
          if (m < NX / 2)
            dst[c] = *reinterpret_cast<const double2_t *>(fbb + C::fFB + m * 2 * C::NR + 2 * rk);
        }
      };
      rd(0, g[0]);
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (k + 1 < NCH)
          rd(k + 1, g[(k + 1) & 1]);
#pragma unroll
        for (int c = 0; c < CH; ++c)
          if (k * CH + c < NX / 2)
            lean_pin(g[k & 1][c]);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int m = k * CH + c;
          if (m < NX / 2) {
            acc = __builtin_fma(g[k & 1][c].x, lane_bcast(xs, NU + 2 * m), acc);
            acc1 = __builtin_fma(g[k & 1][c].y, lane_bcast(xs, NU + 2 * m + 1), acc1);
          }
        }
        lean_pin(acc, acc1);
      }
    }
    acc += acc1;
    GAR_WAIT_LGKMCNT0(); // every row is in registers: the buffer is free
    wave_sync();
    fill_fb(t + 1);
    GAR_WAIT_VMCNT(FBP); // Vxx' | vx' of stage t + 1 (requested a stage ago) have landed
    __builtin_amdgcn_sched_barrier(0);
    if (lane < NW)
      sol[lane < NU ? P.sol_u + t * NU + lane : (t + 1) * NX + (lane - NU)] = acc;
    wave_sync();
    // ---- lbd' = vx' + Vxx' x'  (:369-371); x'_j sits in lane NU + j.  Row iv of the symmetric matrix from its
    // packed lower triangle (gar_sym_index): (iv, j), j <= iv, at cj + iv; (j, iv), j > iv, at lowbase + j
    const int iv = lean_opaque(iv0);
    const double *vb = reinterpret_cast<const double *>(sl + L::oVX);
    const double *va = vb + iv;                                                        // + cj
    const double *vl = vb + (2 * iv < NX ? iv * NX : (NX - 1 - iv) * (NX + 1) + 1);    // + j
    double lam = reinterpret_cast<const double *>(sl + L::ovx)[iv], lam1 = 0.0;
    {
      constexpr int CH = 6, NCH = (NX + CH - 1) / CH; // elements per chunk (even)
      static_assert(CH % 2 == 0 && NX % 2 == 0, "elements go in pairs");
      double va_[2][CH], vl_[2][CH];
      auto rd = [&](int k, double *da, double *dl) {
        const int z = lean_opaque(0);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int j = k * CH + c;
          if (j < NX) {
            da[c] = va[z + (2 * j < NX ? j * NX : (NX - 1 - j) * (NX + 1) + 1)];
            dl[c] = vl[z + j];
          }
        }
      };
      rd(0, va_[0], vl_[0]);
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (k + 1 < NCH)
          rd(k + 1, va_[(k + 1) & 1], vl_[(k + 1) & 1]);
#pragma unroll
        for (int c = 0; c < CH; ++c)
          if (k * CH + c < NX)
            lean_pin(va_[k & 1][c], vl_[k & 1][c]);
#pragma unroll
        for (int c = 0; c < CH; c += 2) {
          const int j = k * CH + c;
          if (j < NX) {
            const double v0 = iv >= j ? va_[k & 1][c] : vl_[k & 1][c];
            const double v1 = iv >= j + 1 ? va_[k & 1][c + 1] : vl_[k & 1][c + 1];
            lam = __builtin_fma(v0, lane_bcast(acc, NU + j), lam);
            lam1 = __builtin_fma(v1, lane_bcast(acc, NU + j + 1), lam1);
          }
        }
        lean_pin(lam, lam1);
      }
    }
    lam += lam1;
    GAR_WAIT_LGKMCNT0();
    wave_sync();
    fill_vx(t + 1);
    GAR_WAIT_VMCNT(VXP); // [ff | fb] of stage t + 1 has landed
    __builtin_amdgcn_sched_barrier(0);
    if (lane < NX)
      sol[P.sol_l + P.nc0 + t * NX + lane] = lam;
    xs = acc;
  }
}

} // namespace gar
