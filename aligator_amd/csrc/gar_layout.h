// gar_layout.h -- packed HBM layout of the LQ problem and of the Riccati
// factors, shared by the host packer, the C-ABI and the HIP kernels.
//
// Replaces the reference's per-matrix heap allocations
//   * LqrKnotTpl      (include/aligator/gar/lqr-problem.hpp:34-103): 16
//     separately allocated column-major ArenaMatrix buffers per knot;
//   * StageFactor     (include/aligator/gar/riccati-kernel.hpp:30-102).
// with ONE contiguous record per knot so a workgroup streams a stage with
// fully coalesced 16-byte loads.
//
// Input record of knot t (doubles, every block column-major exactly as the
// reference stores it, so host packing is 16 memcpy's):
//   Q(nx,nx) S(nx,nu) R(nu,nu) q(nx) r(nu) A(nx2,nx) B(nx2,nu) f(nx2)
//   C(nc,nx) D(nc,nu) d(nc) [Gth(nth,nth) Gx(nx,nth) Gu(nu,nth) Gv(nc,nth) gamma(nth)]
// The G* blocks are present only when the knot is user-parameterised
// (GAR_KNOT_HAS_PARAM).  In leg mode (ParallelRiccatiSolver) the
// parameterisation the reference writes INTO the caller's knots
// (parallel-solver.hxx:52-60,136-147: zeros, and Gx=A^T, Gu=B^T, gamma=f on
// the leg-end knot) is implicit: it is never stored or streamed.
//
// Factor record of knot t (mirrors StageFactor's surviving outputs, same
// storage orders as the reference so download is a memcpy):
//   ff(nr) fb(nr,nx; ROW-major) fth(nr,nth; ROW-major)
//   Vxx(nx,nx) vx(nx) Vxt(nx,nth) Vtt(nth,nth) vt(nth)        nr = nu+nc+nx2
#ifndef GAR_LAYOUT_H
#define GAR_LAYOUT_H

#include <stdint.h>

// The offset helpers are shared by plain-C host code (gcc) and by the kernels.
#if defined(__HIPCC__)
#define GAR_HD __host__ __device__
#else
#define GAR_HD
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define GAR_KNOT_HAS_PARAM 1 /* knot record carries Gth,Gx,Gu,Gv,gamma       */
#define GAR_KNOT_LEG_PARAM 2 /* implicit leg parameterisation (all zero)      */
#define GAR_KNOT_LEG_END 4   /* implicit Gx=A^T, Gu=B^T, Gth=0, gamma=f       */

// Per-stage descriptor, identical for every problem of the batch.
typedef struct gar_stage_meta {
  int32_t nx, nu, nc, nx2, nth; // nth = effective parameter dim in the kernel
  int32_t flags;
  int32_t leg;      // leg index owning this stage
  int32_t reserved;
  int64_t in_off;   // offset (doubles) of the knot record inside one problem
  int64_t fac_off;  // offset (doubles) of the factor record inside one problem
  int32_t x_off, u_off, v_off, l_off; // offsets inside the solution record
} gar_stage_meta;

GAR_HD static inline int64_t gar_knot_doubles(int nx, int nu, int nc, int nx2, int nth_stored) {
  int64_t n = (int64_t)nx * nx + (int64_t)nx * nu + (int64_t)nu * nu + nx + nu;
  n += (int64_t)nx2 * nx + (int64_t)nx2 * nu + nx2;
  n += (int64_t)nc * nx + (int64_t)nc * nu + nc;
  if (nth_stored > 0)
    n += (int64_t)nth_stored * nth_stored + (int64_t)nx * nth_stored +
         (int64_t)nu * nth_stored + (int64_t)nc * nth_stored + nth_stored;
  return n;
}

GAR_HD static inline int64_t gar_factor_doubles(int nx, int nu, int nc, int nx2, int nth) {
  int64_t nr = (int64_t)nu + nc + nx2;
  return nr + nr * nx + nr * nth + (int64_t)nx * nx + nx + (int64_t)nx * nth +
         (int64_t)nth * nth + nth;
}

// offsets inside a knot record
typedef struct gar_knot_offsets {
  int32_t Q, S, R, q, r, A, B, f, C, D, d, Gth, Gx, Gu, Gv, gamma, total;
} gar_knot_offsets;

GAR_HD static inline gar_knot_offsets gar_knot_layout(int nx, int nu, int nc, int nx2, int nth_stored) {
  gar_knot_offsets o;
  int32_t p = 0;
  o.Q = p; p += nx * nx;
  o.S = p; p += nx * nu;
  o.R = p; p += nu * nu;
  o.q = p; p += nx;
  o.r = p; p += nu;
  o.A = p; p += nx2 * nx;
  o.B = p; p += nx2 * nu;
  o.f = p; p += nx2;
  o.C = p; p += nc * nx;
  o.D = p; p += nc * nu;
  o.d = p; p += nc;
  o.Gth = p; p += nth_stored * nth_stored;
  o.Gx = p; p += nx * nth_stored;
  o.Gu = p; p += nu * nth_stored;
  o.Gv = p; p += nc * nth_stored;
  o.gamma = p; p += nth_stored;
  o.total = p;
  return o;
}

// ---- packed storage of a symmetric block ---------------------------------------------------------------------
// The one-wave-per-problem serial kernels (gar_wave*.hpp, gar_mfma.hpp: the solvers whose roll-out is
// gar_forward_mfma) keep only the LOWER TRIANGLE of Vxx in the Vxx block of their factor records, in rectangular
// packed order: the block's first nx (nx + 1) / 2 doubles are nx / 2 "super-columns" of nx + 1 doubles, super-column
// c holding column c's rows c .. nx-1 followed by column nx-1-c's rows nx-1-c .. nx-1 (nx even).  The rest of the
// block is not touched: the sweep writes, and the roll-out reads, 5.3 KB instead of 10.4 KB per stage at nx = 36.
// Every other family stores the full column-major matrix.  gar_hip_get_value unpacks.
// (-DGAR_VXX_PACKED=0 builds the library with full blocks everywhere: the A/B of scripts/ab_vxx_packed.sh)
#ifndef GAR_VXX_PACKED
#define GAR_VXX_PACKED 1
#endif
//   element (i, j) of the block, either order:
GAR_HD static inline int gar_sym_index(int packed, int n, int i, int j) {
  if (!packed)
    return j * n + i;
  const int a = i >= j ? i : j, b = i >= j ? j : i; // a >= b
  return 2 * b < n ? b * (n + 1) + (a - b) : (n - 1 - b) * (n + 1) + (b + 1) + (a - b);
}
GAR_HD static inline int gar_sym_packed_doubles(int n) { return n * (n + 1) / 2; }

// The input side of the same idea (round 4): the headline one-wave-per-problem sweep gar_backward_wave<NX, NU>
// (unconstrained, serial in time, batch > CUs) only ever uses the LOWER triangle of Q and R -- Vxx is symmetrised
// from its lower triangle by the consuming stage (riccati-kernel.hxx:216), the reduced KKT matrix from its own
// (:232), so the upper triangles of Q-hat and R-hat are never formed.  Its knot records keep Q and R as their
// lower triangles in LAPACK "L" packed order (column after column, column j holding rows j .. n-1) in the FIRST
// n (n + 1) / 2 doubles of the Q / R block; the rest of the block is not touched, offsets and record pitch are
// unchanged.  The sweep reads 23.9 KB instead of 29.5 KB per knot at (36, 12).  The terminal knot stays full.
// gar_hip_upload_stage / gar_hip_upload_packed / gar_hip_update_lq_subproblem_device pack, gar_hip_download_packed
// and gar_hip_get_kkt unpack; gar_hip_device_sizes reports the format to device-resident producers.
// (-DGAR_QR_PACKED=0: full blocks everywhere, the A/B of scripts/ab_vxx_packed.py)
#ifndef GAR_QR_PACKED
#define GAR_QR_PACKED 1
#endif
//   element (i, j), i >= j, of an n x n block
GAR_HD static inline int gar_lower_index(int n, int i, int j) { return i + ((2 * n - j - 1) * j) / 2; }

// offsets inside a factor record
typedef struct gar_factor_offsets {
  int32_t ff, fb, fth, Vxx, vx, Vxt, Vtt, vt, total;
} gar_factor_offsets;

GAR_HD static inline gar_factor_offsets gar_factor_layout(int nx, int nu, int nc, int nx2, int nth) {
  gar_factor_offsets o;
  int32_t nr = nu + nc + nx2, p = 0;
  o.ff = p; p += nr;
  o.fb = p; p += nr * nx;
  o.fth = p; p += nr * nth;
  o.Vxx = p; p += nx * nx;
  o.vx = p; p += nx;
  o.Vxt = p; p += nx * nth;
  o.Vtt = p; p += nth * nth;
  o.vt = p; p += nth;
  o.total = p;
  return o;
}

// ---- derivative record (input of the device-resident updateLQSubproblem) -------------------
// What SolverProxDDPTpl::updateLQSubproblem reads for stage t (solver-proxddp.hxx:746-785),
// packed in the knot record's own order so that assembling the LQ problem is a fused copy:
//   Lxx Lxu Luu Lx Lu | Jx Ju slack | Cx Cu Lv      <- same shapes/order as Q S R q r A B f C D d
//   Hxx Hxu Huu  (dynamics Hessians, used when hess_exact)  | lx_corr lu_corr
typedef struct gar_deriv_offsets {
  int32_t Hxx, Hxu, Huu, lxc, luc, total; // the knot-shaped part starts at 0
} gar_deriv_offsets;

GAR_HD static inline gar_deriv_offsets gar_deriv_layout(int nx, int nu, int nc, int nx2) {
  gar_deriv_offsets o;
  int32_t p = (int32_t)gar_knot_doubles(nx, nu, nc, nx2, 0);
  o.Hxx = p; p += nx * nx;
  o.Hxu = p; p += nx * nu;
  o.Huu = p; p += nu * nu;
  o.lxc = p; p += nx;
  o.luc = p; p += nu;
  o.total = p;
  return o;
}

// get_work (parallel-solver.hxx:23-28): leg i of J owns [beg, end)
GAR_HD static inline void gar_get_work(int horz, int leg, int nlegs, int *beg, int *end) {
  *beg = (int)((int64_t)leg * (horz + 1) / nlegs);
  *end = (int)((int64_t)(leg + 1) * (horz + 1) / nlegs);
}

#ifdef __cplusplus
}
#endif
#endif
