// gar_generic.hpp -- dimension-generic gar kernels (any per-stage nx,nu,nc,nx2,
// nth that fits one CU's LDS).  One 256-thread workgroup owns one (problem,
// leg) pair and walks its stages; the value function (Vxx, vx, Vxt, Vtt, vt)
// stays resident in LDS across stages (ping-pong), each knot record is streamed
// once from HBM, each factor record written once.
//
// Restates, with MFMA tiles instead of Eigen expressions:
//   gar_backward_generic : ProximalRiccatiKernel::backwardImpl / terminalSolve /
//                          stageKernelSolve (gar/riccati-kernel.hxx:104-312) and the
//                          initial-stage block of ProximalRiccatiSolver::backward
//                          (gar/proximal-riccati.hxx:42-60); in leg mode the
//                          per-leg body of ParallelRiccatiSolver::backward
//                          (gar/parallel-solver.hxx:136-164).
//   gar_forward_generic  : computeInitial + forwardImpl (riccati-kernel.hxx:195-207,
//                          314-377), per leg as parallel-solver.hxx:215-240.
//   gar_condensed_generic: assembleCondensedSystem + symmetricBlockTridiagSolve +
//                          refinement (parallel-solver.hxx:85-129,169-202,
//                          block-tridiagonal.hpp:52-182).
#pragma once
#include "gar_device.hpp"
#include "gar_layout.h"

namespace gar {

// LDS plan (offsets in doubles), computed on the host from the maximum
// per-stage dimensions (gar_hip.cpp: plan_generic_lds).
struct LdsPlan {
  int V[2], v[2], Vxt[2], Vtt[2], vt[2];
  int lean; // 1: one buffer for the parameter blocks too; stage t reads Vxt', Vtt', vt' from stage t+1's record
  int H, h, F, fv, P, vp, CD, dd, Gu, Guh, Gv, M, msub, piv, G, Yth, yff;
  int bkw; // panel workspace of the blocked Bunch-Kaufman (nk x GAR_BK_PANEL), -1: none (nk > 128)
  int k0mat, k0rhs, k0sub, k0piv; // initial-stage KKT (aliases the stage buffers)
  int total;                      // doubles (backward kernel)
  int fx, fxn, fth, ftotal;       // forward kernel: x, x', theta
};

// LDS of the stage-dense kernels (gar_dense.hpp): KKT matrix, right-hand sides, subdiagonal, pivots
struct DensePlan {
  int K, R, sub, piv, total;
  int wk; // panel workspace of the blocked Bunch-Kaufman (n x GAR_BK_PANEL), -1: none (n > 128 or no room)
};

struct GenericParams {
  int init_closed;            // gar_initial_wave: closed form when G0 = +-I (0: always factorise)
  const gar_stage_meta *meta; // horizon+1 entries (device)
  const double *prob;         // packed problems
  double *fac;                // factor records
  double *sol;                // xs|us|vs|lbdas
  double *init;               // per problem: kkt0.ff | kkt0.fth | thGrad | thHess
  int *status;                // per problem: 1 if a factorisation failed
  double *boundary;           // leg mode: [problem][local leg][tuple]
  const double *csol;         // leg mode: condensed solution [problem][2*legs][nx]
  const double *theta;        // forward: device theta [problem][ntheta] or null
  long long prob_stride, fac_stride, sol_stride, init_stride, boundary_stride;
  long long G0_off, g0_off;
  int horizon, nc0, nx0, nth0;
  int num_legs, leg_begin, local_legs;
  int tuple_doubles, nxb; // boundary tuple size, boundary block dim
  double mueq;
  LdsPlan lds;
  DensePlan dense;
  const int *only; // folded solvers (gar_fold.hpp): sweep only the problems with only[b] != 0 (null: all)
  int vxx_packed;  // the factor records keep the lower triangle of Vxx, packed (gar_layout.h: gar_sym_index)
};

// (gar_smem, the one dynamic-LDS region: declared in gar_device.hpp)


// ---------------------------------------------------------------------------
// Initial stage of ProximalRiccatiSolver::backward (proximal-riccati.hxx:42-60):
// kkt0 = [Vxx0 G0^T; G0 0] (Bunch-Kaufman, reads the LOWER triangle only, so
// Vxx0 is not symmetrised first), kkt0.ff = -[vx0; g0], kkt0.fth = -[Vxt0; 0],
// thGrad, thHess.  The value function of stage 0 sits in LDS buffer `fin`.
// ---------------------------------------------------------------------------
__device__ inline int initial_stage_ptr(const WG &w, const GenericParams &P, int b, int nx, int nth,
                                        const double *V0p, const double *v0, const double *Vxt0p,
                                        const double *Vtt0, const double *vt0, double *k0mat,
                                        double *k0rhs, double *k0sub, int *piv0, int *ctrl0, int v0_packed = 0) {
  const double *prob = P.prob + (long long)b * P.prob_stride;
  int failed = 0;
  const int nc0 = P.nc0, n0 = nx + nc0;
  const double *G0 = prob + P.G0_off, *g0 = prob + P.g0_off;
  MatV K0 = colmajor(k0mat, n0);
  const int rld = 1 + nth;
  MatV R0 = rowmajor(k0rhs, rld); // [ff | fth]
  for (int e = w.tid; e < n0 * n0; e += w.nthr) {
    const int j = e / n0, i = e - j * n0;
    double v = 0.0;
    if (j < nx)
      v = (i < nx) ? V0p[gar_sym_index(v0_packed, nx, i, j)] : G0[j * nc0 + (i - nx)];
    else if (i < nx)
      v = G0[i * nc0 + (j - nx)];
    K0(i, j) = v;
  }
  for (int e = w.tid; e < n0 * rld; e += w.nthr) {
    const int i = e / rld, j = e - i * rld;
    double v;
    if (j == 0)
      v = (i < nx) ? -v0[i] : -g0[i - nx];
    else
      v = (i < nx) ? -Vxt0p[(j - 1) * nx + i] : 0.0;
    R0(i, j) = v;
  }
  wg_bar(w);
  failed |= 2 * wg_bk_factor(w, n0, K0.p, n0, k0sub, piv0, ctrl0);
  wg_bk_solve(w, n0, K0.p, n0, k0sub, piv0, R0.p, rld, 1, rld);
  double *io = P.init + (long long)b * P.init_stride;
  for (int e = w.tid; e < n0; e += w.nthr)
    io[e] = R0(e, 0);
  for (int e = w.tid; e < n0 * nth; e += w.nthr) {
    const int i = e / nth, j = e - i * nth;
    io[n0 + e] = R0(i, 1 + j);
  }
  // thGrad = vt + Vxt^T x0 ; thHess = Vtt + Vxt^T fth_x   (:56-59)
  for (int i = w.tid; i < nth; i += w.nthr) {
    double s = 0.0;
    for (int k = 0; k < nx; ++k)
      s += Vxt0p[i * nx + k] * R0(k, 0);
    io[n0 + n0 * nth + i] = vt0[i] + s;
  }
  for (int e = w.tid; e < nth * nth; e += w.nthr) {
    const int j = e / nth, i = e - j * nth;
    double s = 0.0;
    for (int k = 0; k < nx; ++k)
      s += Vxt0p[i * nx + k] * R0(k, 1 + j);
    io[n0 + n0 * nth + nth + e] = Vtt0[e] + s;
  }
  return failed;
}

__device__ inline int initial_stage(const WG &w, const GenericParams &P, int b, double *sm,
                                    int fin, const gar_stage_meta &m0) {
  const LdsPlan &L = P.lds;
  // V[fin], v[fin], Vxt[fin], Vtt[fin], vt[fin] are outside the range the kkt0 buffers alias
  int *piv0 = (int *)(sm + L.k0piv);
  return initial_stage_ptr(w, P, b, m0.nx, m0.nth, sm + L.V[fin], sm + L.v[fin], sm + L.Vxt[fin],
                           sm + L.Vtt[fin], sm + L.vt[fin], sm + L.k0mat, sm + L.k0rhs,
                           sm + L.k0sub, piv0, piv0 + 512);
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------
// (GAR_BACKWARD_THREADS threads per (leg, problem): copies, product tiles and substitution strips spread over 16 waves)
#ifndef GAR_BACKWARD_THREADS
#define GAR_BACKWARD_THREADS 1024
#endif
__global__ void __launch_bounds__(GAR_BACKWARD_THREADS) gar_backward_generic(GenericParams P) {
  const WG w = wg_self();
  double *sm = gar_smem;
  const int leg = (int)blockIdx.x + P.leg_begin;
  const int b = (int)blockIdx.y;
  if (P.only != nullptr && P.only[b] == 0)
    return;
  const double *prob = P.prob + (long long)b * P.prob_stride;
  double *fac = P.fac + (long long)b * P.fac_stride;
  int t_beg, t_end;
  gar_get_work(P.horizon, leg, P.num_legs, &t_beg, &t_end);
  const LdsPlan &L = P.lds;
  int *piv = (int *)(sm + L.piv);
  int *ctrl = piv + 512;
  int cur = 0; // ping-pong index of the value function being produced
  int failed = 0;

  for (int t = t_end - 1; t >= t_beg; --t) {
    const gar_stage_meta m = P.meta[t];
    const int nx = m.nx, nu = m.nu, nc = m.nc, nx2 = m.nx2, nth = m.nth;
    const int nw = nx + nu, nk = nu + nc, nr = nk + nx2;
    const bool terminal = (t == t_end - 1);
    const int nth_st = (m.flags & GAR_KNOT_HAS_PARAM) ? nth : 0;
    const gar_knot_offsets ko = gar_knot_layout(nx, nu, nc, nx2, nth_st);
    const gar_factor_offsets fo = gar_factor_layout(nx, nu, nc, nx2, nth);
    const double *rec = prob + m.in_off;
    double *out = fac + m.fac_off;
    const int nxt = cur ^ 1; // value function of stage t+1
    MatV H = colmajor(sm + L.H, nw);
    double *h = sm + L.h;
    MatV F = colmajor(sm + L.F, nx2); // [A B], nx2 x nw
    double *fv = sm + L.fv;
    MatV Pm = colmajor(sm + L.P, nx2); // V' [A B]
    double *vp = sm + L.vp;
    MatV CD = colmajor(sm + L.CD, nc); // [C D], nc x nw
    double *dd = sm + L.dd;
    MatV Gu = colmajor(sm + L.Gu, nu), Gv = colmajor(sm + L.Gv, nc);
    MatV Guh = colmajor(sm + L.Guh, nu);
    MatV Mk = colmajor(sm + L.M, nk);
    const int gld = 1 + nx + nth;
    MatV G = rowmajor(sm + L.G, gld); // [kff K Kth; zff Z Zth]
    MatV Vn = colmajor(sm + L.V[nxt], nx2), Vc = colmajor(sm + L.V[cur], nx);
    double *vn = sm + L.v[nxt], *vc = sm + L.v[cur];
    MatV Vxtn = colmajor(sm + L.Vxt[nxt], nx2), Vxtc = colmajor(sm + L.Vxt[cur], nx);
    MatV Vttn = colmajor(sm + L.Vtt[nxt], nth), Vttc = colmajor(sm + L.Vtt[cur], nth);
    double *vtn = sm + L.vt[nxt], *vtc = sm + L.vt[cur];
    if (L.lean && !terminal && nth > 0) {
      // the plan that keeps both generations of the parameter blocks does not fit a CU's LDS (nx = 36,
      // nc = 32, nth = 36: 183 KB): Vxt', Vtt', vt' are read where this workgroup stored them one
      // iteration ago -- stage t+1's record (L2; the barrier at the end of the stage orders the two)
      const gar_stage_meta mn = P.meta[t + 1];
      const gar_factor_offsets fn = gar_factor_layout(mn.nx, mn.nu, mn.nc, mn.nx2, mn.nth);
      double *recn = fac + mn.fac_off;
      Vxtn = colmajor(recn + fn.Vxt, nx2);
      Vttn = colmajor(recn + fn.Vtt, nth);
      vtn = recn + fn.vt;
    }
    MatV Aff = rowmajor(sm + L.P, nx); // aliases P (dead once H is formed)
    MatV Yth = rowmajor(sm + L.Yth, nth);
    double *yff = sm + L.yff;

    // ---- S0: stream the knot record HBM -> LDS ---------------------------
    // H = [Q S; S^T R] (lqr-problem.hpp:16-22), h = [q; r]
    wg_load_colmajor(w, rec + ko.Q, nx, nx, H);
    for (int e = w.tid; e < nx * nu; e += w.nthr) {
      const int j = e / nx, i = e - j * nx;
      const double s = rec[ko.S + e];
      H(i, nx + j) = s;
      H(nx + j, i) = s;
    }
    wg_load_colmajor(w, rec + ko.R, nu, nu, H.sub(nx, nx));
    for (int e = w.tid; e < nw; e += w.nthr)
      h[e] = rec[ko.q + e]; // q then r are contiguous in the record
    for (int e = w.tid; e < nx2 * nw; e += w.nthr)
      F.p[e] = rec[ko.A + e]; // A then B contiguous == [A B] column-major
    for (int e = w.tid; e < nx2; e += w.nthr)
      fv[e] = rec[ko.f + e];
    for (int e = w.tid; e < nc * nw; e += w.nthr)
      CD.p[e] = rec[ko.C + e]; // C then D contiguous == [C D] column-major
    for (int e = w.tid; e < nc; e += w.nthr)
      dd[e] = rec[ko.d + e];
    if (nth > 0) { // Gx -> Vxt(cur), Gth -> Vtt(cur), gamma -> vt(cur)
      if (m.flags & GAR_KNOT_HAS_PARAM) {
        wg_load_colmajor(w, rec + ko.Gx, nx, nth, Vxtc);
        wg_load_colmajor(w, rec + ko.Gth, nth, nth, Vttc);
        wg_load_colmajor(w, rec + ko.Gu, nu, nth, Gu);
        wg_load_colmajor(w, rec + ko.Gv, nc, nth, Gv);
        for (int e = w.tid; e < nth; e += w.nthr)
          vtc[e] = rec[ko.gamma + e];
      } else if (m.flags & GAR_KNOT_LEG_END) {
        // configure_knot (parallel-solver.hxx:136-141): Gx=A^T Gu=B^T Gth=0 gamma=f
        for (int e = w.tid; e < nx * nth; e += w.nthr) {
          const int j = e / nx, i = e - j * nx;
          Vxtc(i, j) = rec[ko.A + i * nx2 + j];
        }
        for (int e = w.tid; e < nu * nth; e += w.nthr) {
          const int j = e / nu, i = e - j * nu;
          Gu(i, j) = rec[ko.B + i * nx2 + j];
        }
        wg_fill(w, Vttc.p, nth * nth, 0.0);
        wg_fill(w, Gv.p, nc * nth, 0.0);
        for (int e = w.tid; e < nth; e += w.nthr)
          vtc[e] = rec[ko.f + e];
      } else { // addParameterization zeros (lqr-problem.hxx:232-241)
        wg_fill(w, Vxtc.p, nx * nth, 0.0);
        wg_fill(w, Vttc.p, nth * nth, 0.0);
        wg_fill(w, Gu.p, nu * nth, 0.0);
        wg_fill(w, Gv.p, nc * nth, 0.0);
        wg_fill(w, vtc, nth, 0.0);
      }
    }
    __syncthreads();

    if (!terminal) {
      // ---- S1: P = V' [A B]; vplus = vx' + V' f  (riccati-kernel.hxx:216-221)
      wg_gemm(w, nx2, nw, nx2, Vn, F, MatV{nullptr, 0, 0}, Pm, 1.0);
      wg_gemv(w, nx2, nx2, Vn, fv, 1, vn, 1, vp, 1, 1.0);
      if (nth > 0) // Ghat_u = Gu + B^T Vxt'  (:286-287; Ghat_x is unused downstream)
        wg_gemm(w, nu, nth, nx2, F.sub(0, nx).T(), Vxtn, Gu, Guh, 1.0);
      __syncthreads();
      // ---- S2: H += [A B]^T P ; h += [A B]^T vplus        (:224-228)
      wg_gemm(w, nw, nw, nx2, F.T(), Pm, H, H, 1.0);
      wg_gemv(w, nw, nx2, F.T(), vp, 1, h, 1, h, 1, 1.0);
      __syncthreads();
    }

    // ---- S3: reduced KKT [Rhat D^T; D -mu I] and its right-hand sides ------
    // lower triangle only: BunchKaufman::compute reads Lower (bunchkaufman.hpp:670)
    for (int e = w.tid; e < nk * nk; e += w.nthr) {
      const int j = e / nk, i = e - j * nk;
      double v;
      if (j < nu)
        v = (i < nu) ? H(nx + i, nx + j) : CD(i - nu, nx + j);
      else
        v = (i < nu) ? CD(j - nu, nx + i) : ((i == j) ? -P.mueq : 0.0);
      Mk(i, j) = v;
    }
    // G = -[rhat Shat^T Ghat_u ; d C Gv]   (:248-256, :288-291)
    for (int e = w.tid; e < nk * (1 + nx); e += w.nthr) {
      const int i = e / (1 + nx), j = e - i * (1 + nx);
      double v;
      if (i < nu)
        v = (j == 0) ? h[nx + i] : H(j - 1, nx + i); // Shat^T from Shat = S + AtV B
      else
        v = (j == 0) ? dd[i - nu] : CD(i - nu, j - 1);
      G(i, j) = -v;
    }
    if (nth > 0) { // Kth rhs = -Ghat_u (Ghat_u = Gu at the terminal knot), Zth rhs = -Gv
      for (int e = w.tid; e < nk * nth; e += w.nthr) {
        const int i = e / nth, j = e - i * nth;
        double v;
        if (i < nu)
          v = terminal ? Gu(i, j) : Guh(i, j);
        else
          v = terminal ? 0.0 : Gv(i - nu, j); // terminalSolve: Zth.setZero() (:168)
        G(i, 1 + nx + j) = -v;
      }
    }
    __syncthreads();

    // ---- S4: factor + solve ------------------------------------------------
    if (nu == 0) { // terminalSolve, nu == 0 branch (:146-149): Z = C/mu, zff = d/mu
      for (int e = w.tid; e < nc * gld; e += w.nthr) {
        const int i = e / gld, j = e - i * gld;
        double v = 0.0;
        if (j == 0)
          v = dd[i] / P.mueq;
        else if (j <= nx)
          v = CD(i, j - 1) / P.mueq;
        G(i, j) = v;
      }
      __syncthreads();
    } else {
      // (from 24 columns on the panel-blocked factorisation pays: one wave runs the pivot search, the trailing matrix
      // is updated once per panel on MFMA tiles)
      if (L.bkw >= 0 && nk >= 24 && nk <= 128)
        failed |= wg_bk_factor_blocked(w, nk, Mk.p, nk, sm + L.msub, piv, ctrl, sm + L.bkw);
      else
        failed |= wg_bk_factor(w, nk, Mk.p, nk, sm + L.msub, piv, ctrl);
      wg_bk_solve(w, nk, Mk.p, nk, sm + L.msub, piv, G.p, gld, 1, gld);
    }

    // ---- S5: closed loop + value function  (:266-277) ----------------------
    MatV K = G.sub(0, 1), Z = G.sub(nu, 1);
    MatV Kth = G.sub(0, 1 + nx);
    if (!terminal) {
      // yff = f + B kff ; Aff = A + B K
      wg_gemv(w, nx2, nu, F.sub(0, nx), G.p, gld, fv, 1, yff, 1, 1.0);
      wg_gemm(w, nx2, nx, nu, F.sub(0, nx), K, F, Aff, 1.0);
      if (nth > 0) // Yth = B Kth (:295)
        wg_gemm(w, nx2, nth, nu, F.sub(0, nx), Kth, MatV{nullptr, 0, 0}, Yth, 1.0);
    }
    // Vxx = Qhat + Shat K ; vx = qhat + Shat kff
    wg_gemm(w, nx, nx, nu, H.sub(0, nx), K, H, Vc, 1.0);
    wg_gemv(w, nx, nu, H.sub(0, nx), G.p, gld, h, 1, vc, 1, 1.0);
    __syncthreads();
    if (nc > 0) { // + C^T Z, + C^T zff
      wg_gemm(w, nx, nx, nc, CD.T(), Z, Vc, Vc, 1.0);
      wg_gemv(w, nx, nc, CD.T(), G.p + nu * gld, gld, vc, 1, vc, 1, 1.0);
      __syncthreads();
    }
    if (nth > 0) { // (:298-310) / terminal (:185-192)
      // vt = gamma [+ vt'] + Gu^T kff [+ Vxt'^T yff]
      for (int i = w.tid; i < nth; i += w.nthr) {
        double s = vtc[i];
        if (!terminal)
          s += vtn[i];
        double a1 = 0.0;
        for (int k = 0; k < nu; ++k)
          a1 += Gu(k, i) * G(k, 0);
        s += a1;
        if (!terminal) {
          double a2 = 0.0;
          for (int k = 0; k < nx2; ++k)
            a2 += Vxtn(k, i) * yff[k];
          s += a2;
        }
        vtc[i] = s;
      }
      // Vxt = Gx + K^T Gu [+ Aff^T Vxt']
      wg_gemm(w, nx, nth, nu, K.T(), Gu, Vxtc, Vxtc, 1.0);
      __syncthreads();
      if (!terminal) {
        wg_gemm(w, nx, nth, nx2, Aff.T(), Vxtn, Vxtc, Vxtc, 1.0);
        // Vtt = Gth + Vtt' (elementwise) -- then the two products
        for (int e = w.tid; e < nth * nth; e += w.nthr)
          Vttc.p[e] += Vttn.p[e];
        __syncthreads();
      }
      wg_gemm(w, nth, nth, nu, Gu.T(), Kth, Vttc, Vttc, 1.0);
      __syncthreads();
      if (!terminal) {
        wg_gemm(w, nth, nth, nx2, Vxtn.T(), Yth, Vttc, Vttc, 1.0);
        __syncthreads();
      }
    }

    // ---- S6: symmetrise (as the consumer stage does, :216) and write out ----
    if (t > t_beg) {
      for (int e = w.tid; e < nx * nx; e += w.nthr) {
        const int j = e / nx, i = e - j * nx;
        if (i < j)
          Vc(i, j) = Vc(j, i);
      }
    }
    // ff = [kff; zff; yff], fb = [K; Z; Aff], fth = [Kth; Zth; Yth]
    for (int e = w.tid; e < nr; e += w.nthr)
      out[fo.ff + e] = (e < nk) ? G(e, 0) : (terminal ? 0.0 : yff[e - nk]);
    for (int e = w.tid; e < nr * nx; e += w.nthr) {
      const int i = e / nx, j = e - i * nx;
      out[fo.fb + e] = (i < nk) ? G(i, 1 + j) : (terminal ? 0.0 : Aff(i - nk, j));
    }
    for (int e = w.tid; e < nr * nth; e += w.nthr) {
      const int i = e / nth, j = e - i * nth;
      out[fo.fth + e] = (i < nk) ? G(i, 1 + nx + j) : (terminal ? 0.0 : Yth(i - nk, j));
    }
    __syncthreads(); // symmetrisation visible before Vxx is stored
    for (int e = w.tid; e < nx * nx; e += w.nthr)
      out[fo.Vxx + e] = Vc.p[e];
    for (int e = w.tid; e < nx; e += w.nthr)
      out[fo.vx + e] = vc[e];
    for (int e = w.tid; e < nx * nth; e += w.nthr)
      out[fo.Vxt + e] = Vxtc.p[e];
    for (int e = w.tid; e < nth * nth; e += w.nthr)
      out[fo.Vtt + e] = Vttc.p[e];
    for (int e = w.tid; e < nth; e += w.nthr)
      out[fo.vt + e] = vtc[e];
    cur ^= 1;
    __syncthreads();
  }

  // value function of the leg's first stage now sits in buffer (cur ^ 1)
  const int fin = cur ^ 1;
  const gar_stage_meta m0 = P.meta[t_beg];
  if (P.num_legs > 1) {
    // leg-boundary tuple: Vxx | Vxt | Vtt | vx | vt, blocks of nxb
    const int nxb = P.nxb;
    double *tup = P.boundary + (long long)b * P.boundary_stride +
                  (long long)(leg - P.leg_begin) * P.tuple_doubles;
    const int nx = m0.nx, nth = m0.nth;
    for (int e = w.tid; e < nxb * nxb; e += w.nthr) {
      const int j = e / nxb, i = e - j * nxb;
      tup[e] = (i < nx && j < nx) ? sm[L.V[fin] + j * nx + i] : 0.0;
      tup[nxb * nxb + e] = (i < nx && j < nth) ? sm[L.Vxt[fin] + j * nx + i] : 0.0;
      tup[2 * nxb * nxb + e] = (i < nth && j < nth) ? sm[L.Vtt[fin] + j * nth + i] : 0.0;
    }
    for (int e = w.tid; e < nxb; e += w.nthr) {
      tup[3 * nxb * nxb + e] = (e < nx) ? sm[L.v[fin] + e] : 0.0;
      tup[3 * nxb * nxb + nxb + e] = (e < nth) ? sm[L.vt[fin] + e] : 0.0;
    }
  } else {
    failed |= initial_stage(w, P, b, sm, fin, m0);
  }
  if (failed && w.tid == 0)
    atomicOr(&P.status[b], failed);
}


// Stand-alone initial stage for backward kernels that do not carry it
// (gar_mfma.hpp): stage 0's value function is re-read from its factor record.
__global__ void __launch_bounds__(256) gar_initial_generic(GenericParams P) {
  const WG w = wg_self();
  double *sm = gar_smem;
  const int b = (int)blockIdx.x;
  const LdsPlan &L = P.lds;
  const gar_stage_meta m0 = P.meta[0];
  const gar_factor_offsets fo = gar_factor_layout(m0.nx, m0.nu, m0.nc, m0.nx2, m0.nth);
  const double *rec = P.fac + (long long)b * P.fac_stride + m0.fac_off;
  for (int e = w.tid; e < m0.nx * m0.nx; e += w.nthr)
    sm[L.V[0] + e] = rec[fo.Vxx + gar_sym_index(P.vxx_packed, m0.nx, e % m0.nx, e / m0.nx)];
  for (int e = w.tid; e < m0.nx; e += w.nthr)
    sm[L.v[0] + e] = rec[fo.vx + e];
  for (int e = w.tid; e < m0.nx * m0.nth; e += w.nthr)
    sm[L.Vxt[0] + e] = rec[fo.Vxt + e];
  for (int e = w.tid; e < m0.nth * m0.nth; e += w.nthr)
    sm[L.Vtt[0] + e] = rec[fo.Vtt + e];
  for (int e = w.tid; e < m0.nth; e += w.nthr)
    sm[L.vt[0] + e] = rec[fo.vt + e];
  __syncthreads();
  const int failed = initial_stage(w, P, b, sm, 0, m0);
  if (failed && w.tid == 0)
    atomicOr(&P.status[b], failed);
}

// The same initial stage run by ONE wave per problem, reading the stage-0 value function
// straight from the factor record: only the kkt0 buffers live in LDS ((nx+nc0)^2 + ...), so
// several problems share a CU, and the Bunch-Kaufman steps synchronise with wave barriers
// instead of workgroup barriers.  Dynamic LDS: gar_initial_wave_lds_doubles(n0, nth).
__host__ __device__ inline int gar_initial_wave_lds_doubles(int n0, int nth) {
  return n0 * n0 + n0 * (1 + nth) + n0 + (n0 & 1) + (n0 + 16 + 1) / 2 * 1 + 8;
}
__global__ void __launch_bounds__(64) gar_initial_wave(GenericParams P) {
  const WG w = wave_self();
  double *sm = gar_smem;
  const int b = (int)blockIdx.x;
  if (P.only && !P.only[b]) // (pipelined sweeps: only the problems the fused initial stage left behind)
    return;
  const gar_stage_meta m0 = P.meta[0];
  const gar_factor_offsets fo = gar_factor_layout(m0.nx, m0.nu, m0.nc, m0.nx2, m0.nth);
  const double *rec = P.fac + (long long)b * P.fac_stride + m0.fac_off;
  const int n0 = m0.nx + P.nc0, nth = m0.nth;
  // "x0 given" (G0 = +-I, nc0 = nx, no parameters): the closed form of the fused initial stage of
  // gar_backward_wave -- x0 = -s g0, lbd0 = -s (vx0 + Vxx0 x0) -- instead of the (2 nx)^2
  // Bunch-Kaufman factorisation (0.18 ms of a 2.1 ms single-problem sweep)
  if (P.init_closed && nth == 0 && P.nc0 == m0.nx && m0.nx <= 64) {
    const int nx = m0.nx;
    const double *prob = P.prob + (long long)b * P.prob_stride;
    const double *G0 = prob + P.G0_off, *g0 = prob + P.g0_off;
    const double g00 = G0[0];
    bool ok = (g00 == 1.0 || g00 == -1.0);
    for (int e = w.lane; e < nx * nx; e += 64) {
      const int j = e / nx, i = e - j * nx;
      ok &= (G0[e] == (i == j ? g00 : 0.0));
    }
    if (__ballot(!ok) == 0ull) {
      const int ix = w.lane < nx ? w.lane : nx - 1;
      const double x0 = -g00 * g0[ix];
      double *xs = sm; // x0 through LDS: the state dimension is a run-time value here
      if (w.lane < nx)
        xs[w.lane] = x0;
      wave_sync();
      double acc = rec[fo.vx + ix];
      for (int k = 0; k < nx; ++k)
        acc += rec[fo.Vxx + gar_sym_index(P.vxx_packed, nx, ix, k)] * xs[k]; // Vxx0 symmetric: row ix
      double *io = P.init + (long long)b * P.init_stride;
      if (w.lane < nx) {
        io[w.lane] = x0;
        io[nx + w.lane] = -g00 * acc;
      }
      return;
    }
  }
  double *k0mat = sm, *k0rhs = k0mat + n0 * n0, *k0sub = k0rhs + n0 * (1 + nth);
  int *piv0 = (int *)(k0sub + n0 + (n0 & 1));
  const int failed = initial_stage_ptr(w, P, b, m0.nx, nth, rec + fo.Vxx, rec + fo.vx, rec + fo.Vxt,
                                       rec + fo.Vtt, rec + fo.vt, k0mat, k0rhs, k0sub, piv0,
                                       piv0 + n0, P.vxx_packed);
  if (failed && w.tid == 0)
    atomicOr(&P.status[b], failed);
}

// ---------------------------------------------------------------------------
// cycleAppend (proximal-riccati.hxx:79-86: rotate_vec_left of the stage data) on the device: the
// `nrec` uniform records starting at `off` slide down by one, the last one is zeroed.  Thread e
// owns element e of every record (no synchronisation: it only ever touches its own offsets),
// eight records in flight per round trip.  grid (ceil(rec/256), batch).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gar_rotate_records(double *base, long long stride, long long off,
                                                          long long rec, int nrec) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= rec)
    return;
  double *a = base + (long long)blockIdx.y * stride + off + e;
  int t = 0;
  for (; t + 8 < nrec; t += 8) {
    double tmp[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      tmp[q] = a[(t + 1 + q) * rec];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      a[(t + q) * rec] = tmp[q];
  }
  for (; t + 1 < nrec; ++t)
    a[t * rec] = a[(t + 1) * rec];
  a[(long long)(nrec - 1) * rec] = 0.0;
}

// A small record straight into pinned host memory, by the kernel's own stores: it does not queue on the copy
// engine behind a large device-to-host copy that is in flight (the solution next to the gains' read-back).
__global__ void __launch_bounds__(256) gar_store_to_host(double *dst_host, const double *src, long long n) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(src[e], dst_host + e);
}

// ---------------------------------------------------------------------------
// Bulk read-back of the gains (the loop of SolverProxDDPTpl::computeDirection that copies
// getFeedforward(i) / getFeedback(i) of every stage, solver-proxddp.hxx:620-632): ff and fb of all
// stages of ONE problem are gathered into two dense arrays in the reference's own storage order
// (fb ROW-major, riccati-kernel.hpp:96-97; the specialised kernel families keep it in the fbT2
// device order) so that the host needs ONE copy instead of 2(N+1).  grid (N+1), 256 threads.
// goff[2t], goff[2t+1]: offsets of stage t inside ff_all / fb_all.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gar_gather_gains(const gar_stage_meta *meta, const double *fac,
                                                        double *ff_all, double *fb_all,
                                                        const long long *goff, int horizon, int t2, int dense,
                                                        int unx, int unu, int t0) {
  // unx > 0: the solver is padded (gar_hip.cpp, gar_hip_solver::padded) -- the caller's arrays hold the rows of the
  // real controls [0, unu) and states [nu, nu + unx) and the first unx columns only; goff are the caller's offsets
  // t0: first stage of the range gathered (one process, several devices: each device gathers its own stages)
  const int t = t0 + (int)blockIdx.x;
  const gar_stage_meta m = meta[t];
  const int nx2r = dense ? 2 * m.nx2 : m.nx2;
  const gar_factor_offsets o = gar_factor_layout(m.nx, m.nu, m.nc, nx2r, m.nth);
  const int nr = m.nu + m.nc + nx2r, nx = m.nx;
  const int un = unx > 0 ? (m.nu > 0 ? unu : 0) : m.nu;      // control rows kept
  const int onr = unx > 0 ? un + unx : nr, onx = unx > 0 ? unx : nx;
  const double *rec = fac + m.fac_off;
  double *ff = ff_all + goff[2 * t], *fb = fb_all + goff[2 * t + 1];
  for (int e = (int)threadIdx.x; e < onr; e += 256)
    ff[e] = rec[o.ff + (e < un ? e : e - un + m.nu)];
  const bool tr = t2 && t < horizon; // fbT2: element (r, j) at (j/2) 2nr + 2r + (j&1)
  for (int e = (int)threadIdx.x; e < onr * onx; e += 256) {
    const int ro = e / onx, j = e - ro * onx, r = ro < un ? ro : ro - un + m.nu;
    fb[e] = rec[o.fb + (tr ? (j >> 1) * (2 * nr) + 2 * r + (j & 1) : r * nx + j)];
  }
}

// ---------------------------------------------------------------------------
// Device-resident SolverProxDDPTpl::updateLQSubproblem (solver-proxddp.hxx:734-805): one
// workgroup per (stage, problem) turns the stage's derivative record (gar_layout.h) into its
// knot record -- Q = Lxx + preg I [+ Hxx], S = Lxu [+ Hxu], R = Luu + preg I [+ Huu],
// q = Lx + lx_corr, r = Lu + lu_corr, A = Jx, B = Ju, f = slack, C, D, d copied; the terminal
// knot takes Q, q, C, d only (:787-797); G0 = init Jx, g0 = init value, and stage 0's Q gets the
// initial condition's Hessian (:799-804).  Same order of additions as the reference.  HBM-bound:
// one read and one write of a knot record per stage.
// ---------------------------------------------------------------------------
struct UpdateParams {
  const gar_stage_meta *meta;
  const double *deriv; // [problem][header: G0 | g0 | init Hxx][stage records]
  double *prob;
  long long deriv_stride, prob_stride, G0_off, g0_off;
  const long long *deriv_off; // horizon+1 stage-record offsets inside one derivative buffer
  long long d_G0, d_g0, d_iH;
  int horizon, nc0, nx0, hess_exact;
  double preg;
  int qr_packed; // knots t < horizon keep Q and R as packed lower triangles (gar_layout.h)
};

__global__ void __launch_bounds__(256) gar_update_lq(UpdateParams P) {
  const int t = (int)blockIdx.x, b = (int)blockIdx.y, tid = (int)threadIdx.x;
  const gar_stage_meta m = P.meta[t];
  const int nx = m.nx, nu = m.nu, nc = m.nc, nx2 = m.nx2;
  const gar_knot_offsets ko = gar_knot_layout(nx, nu, nc, nx2, 0);
  const gar_deriv_offsets dof = gar_deriv_layout(nx, nu, nc, nx2);
  const double *dv = P.deriv + (long long)b * P.deriv_stride;
  const double *d = dv + P.deriv_off[t];
  double *k = P.prob + (long long)b * P.prob_stride + m.in_off;
  const bool term = (t == P.horizon);
  const bool exact = P.hess_exact && !term;
  const bool pk = P.qr_packed && !term;
  for (int e = tid; e < nx * nx; e += 256) { // Q (:763, :768, :773, :804)
    const int j = e / nx, i = e - j * nx;
    double v = d[ko.Q + e];
    if (i == j)
      v += P.preg;
    if (exact)
      v += d[dof.Hxx + e];
    if (t == 0)
      v += dv[P.d_iH + e];
    if (!pk)
      k[ko.Q + e] = v;
    else if (i >= j)
      k[ko.Q + gar_lower_index(nx, i, j)] = v;
  }
  for (int e = tid; e < nx; e += 256) // q (:766, :783)
    k[ko.q + e] = d[ko.q + e] + d[dof.lxc + e];
  for (int e = tid; e < nc * nx; e += 256) // C
    k[ko.C + e] = d[ko.C + e];
  for (int e = tid; e < nc; e += 256) // d
    k[ko.d + e] = d[ko.d + e];
  if (!term) {
    for (int e = tid; e < nx * nu; e += 256) // S (:764, :774)
      k[ko.S + e] = exact ? d[ko.S + e] + d[dof.Hxu + e] : d[ko.S + e];
    for (int e = tid; e < nu * nu; e += 256) { // R (:765, :769, :775)
      const int j = e / nu, i = e - j * nu;
      double v = d[ko.R + e];
      if (i == j)
        v += P.preg;
      if (exact)
        v += d[dof.Huu + e];
      if (!pk)
        k[ko.R + e] = v;
      else if (i >= j)
        k[ko.R + gar_lower_index(nu, i, j)] = v;
    }
    for (int e = tid; e < nu; e += 256) // r (:767, :784)
      k[ko.r + e] = d[ko.r + e] + d[dof.luc + e];
    for (int e = tid; e < nx2 * (nx + nu) + nx2; e += 256) // A, B, f contiguous (:759-761)
      k[ko.A + e] = d[ko.A + e];
    for (int e = tid; e < nc * nu; e += 256) // D
      k[ko.D + e] = d[ko.D + e];
  }
  if (t == 0) { // G0, g0 (:800-801)
    double *pb = P.prob + (long long)b * P.prob_stride;
    for (int e = tid; e < P.nc0 * P.nx0; e += 256)
      pb[P.G0_off + e] = dv[P.d_G0 + e];
    for (int e = tid; e < P.nc0; e += 256)
      pb[P.g0_off + e] = dv[P.d_g0 + e];
  }
}

// The same on a PADDED solver (gar_hip.cpp, gar_hip_solver::padded): the derivative records speak the CALLER's
// dimensions (unx, unu; uniform, unconstrained -- what a padded solver is), the knot records the device's (NX, NU).
// Real rows / columns: the reference's sums in the reference's order, as above; the dummy controls get R = I, S = 0,
// B = 0, r = 0, the dummy states Q = I, A = 0, f = 0, q = 0 and the extra rows [0 -I] x0 = 0 of the initial
// constraint -- exactly what gar_hip_upload_stage / gar_hip_set_init write.  deriv_off / d_G0 / d_g0 / d_iH: the
// CALLER's derivative layout; unc0 rows of the caller's G0.
__global__ void __launch_bounds__(256) gar_update_lq_padded(UpdateParams P, int unx, int unu, int unc0) {
  const int t = (int)blockIdx.x, b = (int)blockIdx.y, tid = (int)threadIdx.x;
  const gar_stage_meta m = P.meta[t];
  const int NX = m.nx, NU = m.nu; // device; nc = 0, nx2 = NX
  const bool term = (t == P.horizon);
  const int nu = term ? 0 : unu;
  const gar_knot_offsets KO = gar_knot_layout(NX, NU, 0, NX, 0), ko = gar_knot_layout(unx, nu, 0, unx, 0);
  const gar_deriv_offsets dof = gar_deriv_layout(unx, nu, 0, unx);
  const double *dv = P.deriv + (long long)b * P.deriv_stride;
  const double *d = dv + P.deriv_off[t];
  double *k = P.prob + (long long)b * P.prob_stride + m.in_off;
  const bool exact = P.hess_exact && !term;
  const bool pk = P.qr_packed && !term;
  for (int e = tid; e < NX * NX; e += 256) { // Q
    const int j = e / NX, i = e - j * NX;
    double v = i == j ? 1.0 : 0.0;
    if (i < unx && j < unx) {
      const int s = j * unx + i;
      v = d[ko.Q + s];
      if (i == j)
        v += P.preg;
      if (exact)
        v += d[dof.Hxx + s];
      if (t == 0)
        v += dv[P.d_iH + s];
    }
    if (!pk)
      k[KO.Q + e] = v;
    else if (i >= j)
      k[KO.Q + gar_lower_index(NX, i, j)] = v;
  }
  for (int e = tid; e < NX; e += 256) // q
    k[KO.q + e] = e < unx ? d[ko.q + e] + d[dof.lxc + e] : 0.0;
  if (!term) {
    for (int e = tid; e < NX * NU; e += 256) { // S
      const int j = e / NX, i = e - j * NX;
      double v = 0.0;
      if (i < unx && j < nu) {
        const int s = j * unx + i;
        v = exact ? d[ko.S + s] + d[dof.Hxu + s] : d[ko.S + s];
      }
      k[KO.S + e] = v;
    }
    for (int e = tid; e < NU * NU; e += 256) { // R
      const int j = e / NU, i = e - j * NU;
      double v = i == j ? 1.0 : 0.0;
      if (i < nu && j < nu) {
        const int s = j * nu + i;
        v = d[ko.R + s];
        if (i == j)
          v += P.preg;
        if (exact)
          v += d[dof.Huu + s];
      }
      if (!pk)
        k[KO.R + e] = v;
      else if (i >= j)
        k[KO.R + gar_lower_index(NU, i, j)] = v;
    }
    for (int e = tid; e < NU; e += 256) // r
      k[KO.r + e] = e < nu ? d[ko.r + e] + d[dof.luc + e] : 0.0;
    for (int e = tid; e < NX * NX; e += 256) { // A
      const int j = e / NX, i = e - j * NX;
      k[KO.A + e] = (i < unx && j < unx) ? d[ko.A + j * unx + i] : 0.0;
    }
    for (int e = tid; e < NX * NU; e += 256) { // B
      const int j = e / NX, i = e - j * NX;
      k[KO.B + e] = (i < unx && j < nu) ? d[ko.B + j * unx + i] : 0.0;
    }
    for (int e = tid; e < NX; e += 256) // f
      k[KO.f + e] = e < unx ? d[ko.f + e] : 0.0;
  }
  if (t == 0) { // [G0 0; 0 -I], [g0; 0]
    double *pb = P.prob + (long long)b * P.prob_stride;
    const int nc0 = P.nc0; // device rows = unc0 + (NX - unx)
    for (int e = tid; e < nc0 * NX; e += 256) {
      const int j = e / nc0, i = e - j * nc0;
      double v = 0.0;
      if (i < unc0 && j < unx)
        v = dv[P.d_G0 + j * unc0 + i];
      else if (i >= unc0 && j >= unx && i - unc0 == j - unx)
        v = -1.0;
      pb[P.G0_off + e] = v;
    }
    for (int e = tid; e < nc0; e += 256)
      pb[P.g0_off + e] = e < unc0 ? dv[P.d_g0 + e] : 0.0;
  }
}

// One K-slice of a dot product: sum over k = q, q + 4, q + 8, ... < K of a[k astride] x[k], sixteen products per
// round trip, every load unconditional from a clamped address (a branch per load would serialise the round trips).
__device__ __forceinline__ double gar_sliced_dot(const double *a, int astride, const double *x, int K, int q) {
  double s = 0.0;
  for (int k0 = 0; k0 < K; k0 += 64) {
    double av[16], xv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int k = k0 + q + 4 * j, kc = k < K ? k : K - 1;
      av[j] = a[(long long)kc * astride];
      xv[j] = x[kc];
      av[j] = k < K ? av[j] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
      s += av[j] * xv[j];
  }
  return s;
}

#define GAR_FORWARD_THREADS 1024
// ---------------------------------------------------------------------------
// forward: x0/lbd0 from kkt0 (serial) or the condensed solution (legs), then
// the closed-loop roll-out.  LDS: x (nxM), xn (nxM), theta (nthM).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(GAR_FORWARD_THREADS) gar_forward_generic(GenericParams P) {
  const WG w = wg_self();
  double *sm = gar_smem;
  const int leg = (int)blockIdx.x + P.leg_begin;
  const int b = (int)blockIdx.y;
  if (P.only != nullptr && P.only[b] == 0)
    return;
  const double *fac = P.fac + (long long)b * P.fac_stride;
  double *sol = P.sol + (long long)b * P.sol_stride;
  int t_beg, t_end;
  gar_get_work(P.horizon, leg, P.num_legs, &t_beg, &t_end);
  const gar_stage_meta m0 = P.meta[t_beg];
  double *x = sm + P.lds.fx;   // current state
  double *xn = sm + P.lds.fxn; // next state
  double *th = sm + P.lds.fth; // theta
  bool have_theta = false;
  if (P.num_legs == 1) { // computeInitial (riccati-kernel.hxx:195-207)
    const int nx = m0.nx, nc0 = P.nc0, n0 = nx + nc0, nth = m0.nth;
    const double *io = P.init + (long long)b * P.init_stride;
    have_theta = (P.theta != nullptr) && nth > 0;
    if (have_theta)
      for (int e = w.tid; e < nth; e += w.nthr)
        th[e] = P.theta[(long long)b * nth + e];
    __syncthreads();
    for (int i = w.tid; i < n0; i += w.nthr) {
      double s = io[i];
      if (have_theta)
        for (int k = 0; k < nth; ++k)
          s += io[n0 + i * nth + k] * th[k];
      if (i < nx) {
        x[i] = s;
        sol[m0.x_off + i] = s;
      } else {
        sol[m0.l_off + (i - nx)] = s;
      }
    }
  } else { // scatter of the condensed solution (parallel-solver.hxx:215-220)
    const int nxb = P.nxb;
    const double *cs = P.csol + (long long)b * (2 * P.num_legs) * nxb;
    const int nx = m0.nx;
    const int nl0 = (leg == 0) ? P.nc0 : nx;
    for (int i = w.tid; i < nl0; i += w.nthr)
      sol[m0.l_off + i] = cs[(2 * leg) * nxb + i];
    for (int i = w.tid; i < nx; i += w.nthr) {
      const double s = cs[(2 * leg + 1) * nxb + i];
      x[i] = s;
      sol[m0.x_off + i] = s;
    }
    if (leg < P.num_legs - 1) { // theta = lbdas[end] (:234-236)
      have_theta = true;
      for (int e = w.tid; e < m0.nth; e += w.nthr)
        th[e] = cs[(2 * (leg + 1)) * nxb + e];
    }
  }
  __syncthreads();

  // ---- the state chain (riccati-kernel.hxx:343-365, the rows of x'):  x_{t+1} = yff + Aff x_t (+ Yth th).
  // It is the only sequential part of the roll-out.  Thread (row, slice) = (tid / 4, tid % 4) takes every fourth
  // column of its row, all its loads in flight at once; the four slices meet through two lane shuffles.
  const int srow = w.tid >> 2, sq = w.tid & 3, srows = w.nthr >> 2;
  for (int t = t_beg; t + 1 < t_end; ++t) {
    const gar_stage_meta m = P.meta[t];
    const int nx = m.nx, nx2 = m.nx2, nth = m.nth, nk = m.nu + m.nc;
    const gar_factor_offsets fo = gar_factor_layout(nx, m.nu, m.nc, nx2, nth);
    const double *rec = fac + m.fac_off;
    const bool use_th = have_theta && nth > 0;
    const gar_stage_meta mn = P.meta[t + 1];
    for (int i0 = 0; i0 < nx2; i0 += srows) {
      const int i = i0 + srow, ic = i < nx2 ? i : nx2 - 1;
      double sum = gar_sliced_dot(rec + fo.fb + (long long)(nk + ic) * nx, 1, x, nx, sq);
      if (use_th)
        sum += gar_sliced_dot(rec + fo.fth + (long long)(nk + ic) * nth, 1, th, nth, sq);
      sum += __shfl_xor(sum, 1);
      sum += __shfl_xor(sum, 2);
      if (sq == 0 && i < nx2) {
        const double v = sum + rec[fo.ff + nk + i];
        xn[i] = v;
        sol[mn.x_off + i] = v;
      }
    }
    __syncthreads();
    double *tmp = x;
    x = xn;
    xn = tmp;
  }
  __syncthreads();
  // ---- everything else is a function of the states just computed and independent from stage to stage:
  //   u = kff + K x (+ Kth th),  v = zff + Z x (+ Zth th)          (:343-365, rows of u and v)
  //   lbd' = vx' + Vxx' x' (+ Vxt' th)                               (:369-374)
  // A wave takes the stages t = t_beg + wave, + nwaves, ...; 16 rows per pass, lane (row, slice) = (lane / 4, lane % 4).
  {
    const int lrow = w.lane >> 2, lq = w.lane & 3;
    for (int t = t_beg + w.wave; t < t_end; t += w.nwaves) {
      const gar_stage_meta m = P.meta[t];
      const int nx = m.nx, nu = m.nu, nx2 = m.nx2, nth = m.nth, nk = m.nu + m.nc;
      const gar_factor_offsets fo = gar_factor_layout(nx, nu, m.nc, nx2, nth);
      const double *rec = fac + m.fac_off;
      const bool use_th = have_theta && nth > 0;
      const double *xt = sol + m.x_off;
      for (int i0 = 0; i0 < nk; i0 += 16) {
        const int i = i0 + lrow, ic = i < nk ? i : nk - 1;
        double sum = gar_sliced_dot(rec + fo.fb + (long long)ic * nx, 1, xt, nx, lq);
        if (use_th)
          sum += gar_sliced_dot(rec + fo.fth + (long long)ic * nth, 1, th, nth, lq);
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        if (lq == 0 && i < nk) {
          const double v = sum + rec[fo.ff + i];
          if (i < nu)
            sol[m.u_off + i] = v;
          else
            sol[m.v_off + (i - nu)] = v;
        }
      }
      if (t + 1 < t_end) {
        const gar_stage_meta mn = P.meta[t + 1];
        const gar_factor_offsets fn = gar_factor_layout(mn.nx, mn.nu, mn.nc, mn.nx2, mn.nth);
        const double *recn = fac + mn.fac_off, *xt1 = sol + mn.x_off;
        for (int i0 = 0; i0 < nx2; i0 += 16) {
          const int i = i0 + lrow, ic = i < nx2 ? i : nx2 - 1;
          double sum = gar_sliced_dot(recn + fn.Vxx + ic, nx2, xt1, nx2, lq);
          if (use_th)
            sum += gar_sliced_dot(recn + fn.Vxt + ic, nx2, th, nth, lq);
          sum += __shfl_xor(sum, 1);
          sum += __shfl_xor(sum, 2);
          if (lq == 0 && i < nx2)
            sol[mn.l_off + i] = sum + recn[fn.vx + i];
        }
      }
    }
  }
}

} // namespace gar

namespace gar {

// ---------------------------------------------------------------------------
// condensed (leg-boundary) system: assemble, factor, solve, refine.
// One workgroup per problem; every rank runs it redundantly on the gathered
// boundary tuples, so no second collective is needed.
// ---------------------------------------------------------------------------
// componentwise backward error at which a cyclic-reduction solve of the condensed system stands even though the
// reference's ABSOLUTE residual threshold is out of reach (gar_cyclic_recover): ~ 100 n eps at n = 72
// (round 4: 1e-12 -> 1e-13.  The forward error this gate admits is cond * omega, and a soak draw -- (8, 4, 4), 8 legs,
// mu = 1.6e-9, cyclic reduction's omega = 9.4e-13 -- kept a solution whose multipliers were 150 x farther from LAPACK
// than any CPU solver; the chain with refinement reaches omega = 4e-17 on it.  The bench shapes sit at 4-6e-15.)
#define GAR_CONDENSED_BACKWARD_OK 1e-13
struct CondensedParams {
  const double *ball;  // gathered tuples: [rank][problem][legs_per_rank][tuple]; rank r owns legs
                       // [r J / W, (r+1) J / W) of J = num_legs over W = world ranks (any W <= J:
                       // legs_per_rank = ceil(J / W) is the chunk pitch, short chunks leave their tail unused)
  const double *prob;  // packed problems (G0, g0)
  double *scratch;     // per problem: see offsets below
  double *csol;        // per problem: [2*num_legs][nxb]
  int *status;
  long long prob_stride, scratch_stride, G0_off, g0_off;
  int batch, num_legs, legs_per_rank, tuple_doubles, nxb, nc0, nx0;
  int world; // ranks the legs are split over (1 on a single-GPU solver)
  double backward_ok; // componentwise backward error at which a cyclic-reduction solve stands (0: never)
  int max_refinement;
  double threshold;
  long long *trace; // debug: cycle stamps of two elimination steps of problem 0 (or null)
  int gated;        // 1: skip the problems whose cyclic-reduction residual (scratch info[0]) met the threshold
  int reduced;      // 1: the chain runs on the REDUCED system (the leg states eliminated leg-parallel beforehand:
                    //    gar_condensed_leg_eliminate / gar_condensed_leg_states below)
  // scratch layout (doubles, per problem): nblk = 2*num_legs, bs = nxb*nxb
  //   diag[nblk][bs] super[nblk][bs] facD[nblk][bs] U[nblk][bs]
  //   fsub[nblk][nxb] rhs[nblk][nxb] err[nblk][nxb] fpiv[nblk][nxb] (ints in doubles)
  //   info[2] : residual, refinement steps
};

__device__ __forceinline__ const double *cond_tuple(const CondensedParams &P, int b, int leg) {
  // owner of a leg under the floor partition r J / W (equals leg / legs_per_rank when W divides J)
  const int rank = ((leg + 1) * P.world - 1) / P.num_legs, ll = leg - rank * P.num_legs / P.world;
  return P.ball + (((long long)rank * P.batch + b) * P.legs_per_rank + ll) * P.tuple_doubles;
}

// ---------------------------------------------------------------------------------------------------------------
// The condensed system with the leg STATES eliminated first, leg-parallel.  Unknowns (lbd0 | x_0, th_0, x_1, th_1,
// ..., x_{J-1}); the rows of x_l couple only to its neighbours:
//     Vxx_l x_l + Vxt_l th_l - th_{l-1} = -vx_l        (l = 0: G0^T lbd0 in place of -th_{-1}; l = J-1: no th_l)
// so x_l = z_l + P_l th_{l-1} - Y_l th_l with P_l = Vxx_l^{-1}, Y_l = P_l Vxt_l, z_l = -P_l vx_l: one factorisation and
// one substitution of [I | Vxt | -vx] per leg, all legs at once (gar_condensed_leg_eliminate, grid (J, batch)).
// What is left is a block-tridiagonal system in (lbd0, th_0 .. th_{J-2}) -- J blocks instead of 2 J:
//     diag(th_l) = Vtt_l - Vxt_l^T Y_l - P_{l+1},  off(th_l, th_{l+1}) = Y_{l+1},  rhs = -vt_l - Vxt_l^T z_l + z_{l+1}
//     diag(lbd0) = -G0 P_0 G0^T,  off(lbd0, th_0) = -G0 Y_0,  rhs = -g0 - G0 z_0
// solved by the chain below in `reduced` mode (same elimination, refinement and residual code, half the sequential
// steps), after which the states follow leg-parallel together with the residual of THEIR rows
// (gar_condensed_leg_states).  The reference eliminates the 2 J blocks in order (block-tridiagonal.hpp:82-138); this
// is the same system in another elimination order: its result is checked like the cyclic reduction's -- the chain
// kernel on the full system runs `gated` afterwards and re-solves in the reference's order what misses the
// residual threshold / the backward-error bound (or whose Vxx_l would not factorise).
// Scratch (beyond the chain's first J blocks): P_l = diag[J + l], Y_l = super[J + l], W_l = Vxt_l^T Y_l = facD[J + l],
// z_l = rhs[J + l], c_l = Vxt_l^T z_l = fsub[J + l]; the reduced solution lives in err[J ..].
// ---------------------------------------------------------------------------------------------------------------
// (GAR_CONDENSED_THREADS threads: 16 waves share the copies, the tiles of the products and the strips of the
// substitutions; the panel factorisation stays one wave's work)
#ifndef GAR_CONDENSED_THREADS
#define GAR_CONDENSED_THREADS 1024
#endif
__device__ __forceinline__ double gar_inf() { return __longlong_as_double(0x7ff0000000000000ll); }
__device__ __forceinline__ void gar_atomic_max_nonneg(double *addr, double v) { // v >= 0 (or NaN -> +inf)
  if (!(v == v))
    v = gar_inf();
  atomicMax(reinterpret_cast<unsigned long long *>(addr), (unsigned long long)__double_as_longlong(v));
}

// LDS: X (bs) | R (nxb x (2 nxb + 1)) | wk (GAR_LDL_PANEL nxb) | sub (nxb) | piv, ctrl
__host__ __device__ inline int gar_condensed_leg_lds_doubles(int nxb) {
  return nxb * nxb + nxb * (2 * nxb + 1) + 1 + GAR_LDL_PANEL * nxb + nxb + 2 + (nxb + 48) / 2 + 2;
}
__global__ void __launch_bounds__(GAR_CONDENSED_THREADS) gar_condensed_leg_eliminate(CondensedParams P) {
  const WG w = wg_self();
  double *sm = gar_smem;
  const int leg = (int)blockIdx.x, b = (int)blockIdx.y, J = P.num_legs;
  const int n = P.nxb, bs = n * n, nblk = 2 * J;
  const bool inner = leg + 1 < J;
  double *S = P.scratch + (long long)b * P.scratch_stride;
  double *diag = S, *super = diag + (long long)nblk * bs, *facD = super + (long long)nblk * bs;
  double *fsub = facD + 2ll * nblk * bs, *rhs = fsub + nblk * n;
  double *Pg = diag + (long long)(J + leg) * bs, *Yg = super + (long long)(J + leg) * bs, *Wg = facD + (long long)(J + leg) * bs;
  double *zg = rhs + (J + leg) * n, *cg = fsub + (J + leg) * n;
  const double *tup = cond_tuple(P, b, leg);
  const int ncol = n + (inner ? n : 0) + 1; // [I | Vxt | -vx]
  double *X = sm, *R = X + bs, *wk = R + ((n * (2 * n + 1) + 1) & ~1), *sub = wk + GAR_LDL_PANEL * n;
  int *piv = (int *)(sub + n + (n & 1)), *ctrl = piv + n + 8;
  for (int e = w.tid; e < bs; e += w.nthr) {
    X[e] = tup[e];
    R[e] = (e / n == e % n) ? 1.0 : 0.0;
    if (inner)
      R[bs + e] = tup[bs + e];
  }
  for (int e = w.tid; e < n; e += w.nthr)
    R[(ncol - 1) * n + e] = -tup[3 * bs + e];
  __syncthreads();
  int bad = 1;
  if (n >= 8 && n <= 64) {
    bad = wg_ldl_definite_factor(w, n, X, n, sub, piv, wk, ctrl);
    if (bad) {
      for (int e = w.tid; e < bs; e += w.nthr)
        X[e] = tup[e];
      __syncthreads();
    }
  }
  if (bad)
    bad = wg_bk_factor(w, n, X, n, sub, piv, ctrl);
  wg_bk_solve(w, n, X, n, sub, piv, R, 1, n, ncol);
  if (bad) { // a Vxx that would not factorise: poison z_l -- the residual gate then hands the problem to the full chain
    for (int e = w.tid; e < n; e += w.nthr)
      R[(ncol - 1) * n + e] = __longlong_as_double(0x7ff8000000000000ll);
    __syncthreads();
  }
  for (int e = w.tid; e < bs; e += w.nthr) {
    Pg[e] = R[e];
    if (inner)
      Yg[e] = R[bs + e];
  }
  for (int e = w.tid; e < n; e += w.nthr)
    zg[e] = R[(ncol - 1) * n + e];
  if (inner) { // [W | c] = Vxt^T [Y | z]
    const MatV Vxt = colmajor(const_cast<double *>(tup) + bs, n);
    wg_gemm(w, n, n, n, Vxt.T(), colmajor(R + bs, n), MatV{nullptr, 0, 0}, colmajor(Wg, n), 1.0);
    wg_gemm(w, n, 1, n, Vxt.T(), colmajor(R + (ncol - 1) * n, n), MatV{nullptr, 0, 0}, colmajor(cg, n), 1.0);
  }
}

// x_l from the reduced solution, and the residual / scale of the rows of x_l (everything the reduced chain did not
// check itself), accumulated into info[0] / info[2] over the legs.  grid (J, batch) x 256.
__global__ void __launch_bounds__(256) gar_condensed_leg_states(CondensedParams P) {
  const WG w = wg_self();
  double *sm = gar_smem; // thp (n) | thn (n) | x (n) | x_{l+1} (n) | th_{l+1} (n)
  const int leg = (int)blockIdx.x, b = (int)blockIdx.y, J = P.num_legs;
  const int n = P.nxb, bs = n * n, nblk = 2 * J;
  const bool inner = leg + 1 < J;
  double *S = P.scratch + (long long)b * P.scratch_stride;
  double *diag = S, *super = diag + (long long)nblk * bs;
  double *fsub = super + 3ll * nblk * bs, *rhs = fsub + nblk * n, *err = rhs + nblk * n;
  double *info = rhs + 3ll * nblk * n;
  const double *Pg = diag + (long long)(J + leg) * bs, *Yg = super + (long long)(J + leg) * bs, *zg = rhs + (J + leg) * n;
  const double *rsol = err + J * n; // (lbd0, th_0 .. th_{J-2})
  double *sol = P.csol + (long long)b * nblk * n;
  const double *tup = cond_tuple(P, b, leg);
  const double *G0 = P.prob + (long long)b * P.prob_stride + P.G0_off;
  double *thp = sm, *thn = sm + n, *x = sm + 2 * n;
  const int nc0 = P.nc0;
  // the reduced solution goes to its places in the full one: lbd0 -> block 0, th_l -> block 2 l + 2
  if (leg == 0)
    for (int e = w.tid; e < nc0; e += w.nthr)
      sol[e] = rsol[e];
  if (inner)
    for (int e = w.tid; e < n; e += w.nthr)
      sol[(2 * leg + 2) * n + e] = rsol[(leg + 1) * n + e];
  // thp = the left neighbour's contribution to the rows of x_l:  +th_{l-1}, or -G0^T lbd0 for the first leg
  for (int e = w.tid; e < n; e += w.nthr) {
    double v = 0.0;
    if (leg > 0) {
      v = rsol[leg * n + e];
    } else {
      for (int k = 0; k < nc0; ++k)
        v -= G0[e * nc0 + k] * rsol[k]; // G0 (nc0 x nx0) column-major: G0^T(e, k) = G0[k + e nc0]
    }
    thp[e] = v;
    thn[e] = inner ? rsol[(leg + 1) * n + e] : 0.0;
  }
  __syncthreads();
  const int srow = w.tid >> 2, sq = w.tid & 3, srows = w.nthr >> 2;
  for (int i0 = 0; i0 < n; i0 += srows) { // x = z + P thp - Y thn   (P, Y symmetric resp. general: rows read strided)
    const int i = i0 + srow, ic = i < n ? i : n - 1;
    double sum = gar_sliced_dot(Pg + ic, n, thp, n, sq);
    if (inner)
      sum -= gar_sliced_dot(Yg + ic, n, thn, n, sq);
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    if (sq == 0 && i < n) {
      const double v = zg[i] + sum;
      x[i] = v;
      sol[(2 * leg + 1) * n + i] = v;
    }
  }
  __syncthreads();
  // residual of the rows of x_l:  -vx - Vxx x - Vxt thn + thp, and their scale |vx| + |Vxx| |x| + |Vxt| |thn| + |thp|
  double rmax = 0.0, smax = 0.0;
  for (int i0 = 0; i0 < n; i0 += srows) {
    const int i = i0 + srow, ic = i < n ? i : n - 1;
    double r = 0.0, a = 0.0;
    for (int k = sq; k < n; k += 4) {
      const double vxx = tup[(long long)k * n + ic], vxt = inner ? tup[bs + (long long)k * n + ic] : 0.0;
      r += vxx * x[k] + vxt * thn[k];
      a += fabs(vxx) * fabs(x[k]) + fabs(vxt) * fabs(thn[k]);
    }
    r += __shfl_xor(r, 1);
    r += __shfl_xor(r, 2);
    a += __shfl_xor(a, 1);
    a += __shfl_xor(a, 2);
    if (sq == 0 && i < n) {
      const double vx = tup[3 * bs + i];
      const double res = fabs(-vx - r + thp[i]);
      rmax = fmax(rmax, res == res ? res : gar_inf());
      smax = fmax(smax, fabs(vx) + a + fabs(thp[i]));
    }
  }
  // ... and of the rows of th_l, of the FULL system (the reduced system was formed from P, Y, W in floating point: its
  // own residual does not see what forming it lost):  -vt_l - Vxt_l^T x_l - Vtt_l th_l + x_{l+1}, with x_{l+1}
  // recomputed here (its own workgroup writes it); and of the rows of lbd0: -g0 - G0 x_0
  if (inner) {
    double *xn = sm + 3 * n, *thnn = sm + 4 * n;
    const bool inner2 = leg + 2 < J;
    const double *Pn = Pg + bs, *Yn = Yg + bs, *zn = zg + n; // leg + 1
    for (int e = w.tid; e < n; e += w.nthr)
      thnn[e] = inner2 ? rsol[(leg + 2) * n + e] : 0.0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += srows) {
      const int i = i0 + srow, ic = i < n ? i : n - 1;
      double sum = gar_sliced_dot(Pn + ic, n, thn, n, sq);
      if (inner2)
        sum -= gar_sliced_dot(Yn + ic, n, thnn, n, sq);
      sum += __shfl_xor(sum, 1);
      sum += __shfl_xor(sum, 2);
      if (sq == 0 && i < n)
        xn[i] = zn[i] + sum;
    }
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += srows) {
      const int i = i0 + srow, ic = i < n ? i : n - 1;
      double r = 0.0, a = 0.0;
      for (int k = sq; k < n; k += 4) {
        const double vxt = tup[bs + (long long)ic * n + k], vtt = tup[2 * bs + (long long)k * n + ic]; // Vxt(k, i), Vtt(i, k)
        r += vxt * x[k] + vtt * thn[k];
        a += fabs(vxt) * fabs(x[k]) + fabs(vtt) * fabs(thn[k]);
      }
      r += __shfl_xor(r, 1);
      r += __shfl_xor(r, 2);
      a += __shfl_xor(a, 1);
      a += __shfl_xor(a, 2);
      if (sq == 0 && i < n) {
        const double vt = tup[3 * bs + n + i];
        const double res = fabs(-vt - r + xn[i]);
        rmax = fmax(rmax, res == res ? res : gar_inf());
        smax = fmax(smax, fabs(vt) + a + fabs(xn[i]));
      }
    }
  }
  if (leg == 0) {
    const double *g0 = P.prob + (long long)b * P.prob_stride + P.g0_off;
    for (int i = w.tid; i < nc0; i += w.nthr) {
      double r = 0.0, a = 0.0;
      for (int k = 0; k < n; ++k) {
        r += G0[i + k * nc0] * x[k];
        a += fabs(G0[i + k * nc0]) * fabs(x[k]);
      }
      const double res = fabs(-g0[i] - r);
      rmax = fmax(rmax, res == res ? res : gar_inf());
      smax = fmax(smax, fabs(g0[i]) + a);
    }
  }
  rmax = wave_max_f64(rmax);
  smax = wave_max_f64(smax);
  if (w.lane == 0) {
    gar_atomic_max_nonneg(&info[0], rmax);
    gar_atomic_max_nonneg(&info[2], smax);
  }
}

// LDS: blk[nxb*nxb] ublk[nxb*nxb] sub[nxb] piv/ctrl
// (debug build -DGAR_CTRACE: cycles per phase of workgroup 0, read with gar_hip_debug_ctrace -- scripts/ctrace_condensed.py)
#ifdef GAR_CTRACE
__device__ long long g_ctrace[16];
#define CT(id)                                                                                                         \
  {                                                                                                                    \
    __syncthreads();                                                                                                   \
    const long long now_ = clock64();                                                                                  \
    if (w.tid == 0 && blockIdx.x == 0)                                                                                 \
      g_ctrace[id] += now_ - tprev;                                                                                    \
    tprev = now_;                                                                                                      \
  }
#else
#define CT(id)
#endif
__global__ void __launch_bounds__(GAR_CONDENSED_THREADS) gar_condensed_generic(CondensedParams P) {
#ifdef GAR_CTRACE
  long long tprev = clock64();
#endif
  const WG w = wg_self();
  double *sm = gar_smem;
  const int b = (int)blockIdx.x;
  // nblkS: blocks of the FULL system (the scratch arrays' pitch); nblk: blocks of the system this launch solves --
  // the full one, or (P.reduced) the one in (lbd0, th_0 .. th_{J-2}) left by gar_condensed_leg_eliminate
  const int nxb = P.nxb, bs = nxb * nxb, nblkS = 2 * P.num_legs, nblk = P.reduced ? P.num_legs : nblkS, N = nblk - 1;
  const bool alt = !P.reduced; // the full system's super-diagonal alternates Vxt, -I
  if (P.gated) { // already solved to the residual threshold by the cyclic-reduction / reduced-system kernels?
    const double *inf = P.scratch + (long long)b * P.scratch_stride + 4ll * nblkS * bs + 4ll * nblkS * nxb;
    // (refinement disabled: the cyclic-reduction result stands -- unless a block inverse failed
    // outright, which poisons the residual with +inf)
    double *inf_w = P.scratch + (long long)b * P.scratch_stride + 4ll * nblkS * bs + 4ll * nblkS * nxb;
    if (inf[0] <= P.threshold || (P.max_refinement == 0 && inf[0] <= 1.79e308) ||
        (inf[0] <= 1.79e308 && inf[0] <= P.backward_ok * inf[2])) { // (see gar_cyclic_recover)
      if (w.tid == 0)
        inf_w[3] = 0.0; // the pre-solver's result stands (gar_hip_condensed_resolved)
      return;
    }
    if (w.tid == 0)
      inf_w[3] = 1.0; // re-solved here, in the reference's order
  } else if (w.tid == 0 && !P.reduced) {
    (P.scratch + (long long)b * P.scratch_stride + 4ll * nblkS * bs + 4ll * nblkS * nxb)[3] = 0.0;
  }
  double *S = P.scratch + (long long)b * P.scratch_stride;
  double *diag = S, *super = diag + (long long)nblkS * bs, *facD = super + (long long)nblkS * bs;
  double *U = facD + (long long)nblkS * bs;
  double *fsub = U + (long long)nblkS * bs;
  double *rhs = fsub + nblkS * nxb, *err = rhs + nblkS * nxb;
  int *fpiv = (int *)(err + nblkS * nxb);
  double *info = err + 2 * nblkS * nxb;
  double *sol = P.reduced ? err + P.num_legs * nxb : P.csol + (long long)b * nblkS * nxb;
  const double *prob = P.prob + (long long)b * P.prob_stride;
  // LDS: blk (n x (n + 1)), ublk (n x (r + 1); >= 9 nxb doubles for the slices of the back-substitution), bsup (r x n)
  double *blk = sm, *ublk = blk + bs + nxb, *bsup = ublk + bs + nxb, *lsub = bsup + bs;
  int *lpiv = (int *)(lsub + nxb + (nxb & 1));
  int *ctrl = lpiv + nxb + 8;
  // block i has dimension dim(i): nc0 for i == 0, else nxb  (rhsDims_, :68-73)
#define DIM(i) ((i) == 0 ? P.nc0 : nxb)

  CT(0)
  if (P.reduced) {
    // ---- the reduced system from what gar_condensed_leg_eliminate left (see there): P_l = diag[J + l],
    // Y_l = super[J + l], W_l = facD[J + l], z_l = rhs[J + l], c_l = fsub[J + l]
    const int J = P.num_legs, nc0 = P.nc0, nx0 = P.nx0;
    const MatV G0 = colmajor(const_cast<double *>(prob) + P.G0_off, nc0);
    double *GP = blk; // G0 P_0 (nc0 x nx0), in LDS
    wg_gemm(w, nc0, nx0, nx0, G0, colmajor(diag + (long long)J * bs, nxb), MatV{nullptr, 0, 0}, colmajor(GP, nc0), 1.0);
    __syncthreads();
    wg_gemm(w, nc0, nc0, nx0, colmajor(GP, nc0), G0.T(), MatV{nullptr, 0, 0}, colmajor(diag, nc0), -1.0);
    wg_gemm(w, nc0, nx0, nx0, G0, colmajor(super + (long long)J * bs, nxb), MatV{nullptr, 0, 0}, colmajor(super, nc0), -1.0);
    wg_gemm(w, nc0, 1, nx0, G0, colmajor(rhs + J * nxb, nxb), MatV{nullptr, 0, 0}, colmajor(rhs, nc0), -1.0);
    __syncthreads();
    for (int e = w.tid; e < nc0; e += w.nthr)
      rhs[e] -= prob[P.g0_off + e]; // rhs[0] = -g0 - G0 z_0
    for (int l = 0; l + 1 < J; ++l) {
      const double *tup = cond_tuple(P, b, l);
      const int i = l + 1;
      const double *Wl = facD + (long long)(J + l) * bs, *Pn = diag + (long long)(J + l + 1) * bs;
      const double *Yn = super + (long long)(J + l + 1) * bs;
#pragma unroll 4
      for (int e = w.tid; e < bs; e += w.nthr) {
        diag[(long long)i * bs + e] = tup[2 * bs + e] - Wl[e] - Pn[e];
        if (i < N)
          super[(long long)i * bs + e] = Yn[e];
      }
      for (int e = w.tid; e < nxb; e += w.nthr)
        rhs[i * nxb + e] = -tup[3 * bs + nxb + e] - fsub[(J + l) * nxb + e] + rhs[(J + l + 1) * nxb + e];
    }
    __syncthreads();
  } else {
  // ---- assembleCondensedSystem (parallel-solver.hxx:85-129), blocks stored column-major with leading dimension
  // DIM(row block).  Every block that is read later is written exactly once (diag, super and rhs stay as they are:
  // the refinement's residual is taken against them); facD and U are produced by the elimination itself.
  {
    const int nc0 = P.nc0, nx0 = P.nx0;
    for (int e = w.tid; e < nc0 * nc0; e += w.nthr)
      diag[e] = 0.0;
    for (int e = w.tid; e < nc0 * nx0; e += w.nthr) // super[0] = G0 (nc0 x nx0)
      super[e] = prob[P.G0_off + e];
    for (int e = w.tid; e < nc0; e += w.nthr)
      rhs[e] = -prob[P.g0_off + e];
  }
  for (int leg = 0; leg < P.num_legs; ++leg) {
    const double *tup = cond_tuple(P, b, leg);
    const bool inner = leg + 1 < P.num_legs;
    double *d1 = diag + (long long)(2 * leg + 1) * bs, *s1 = super + (long long)(2 * leg + 1) * bs;
    // diag[2 leg + 1] = Vxx(leg); super[2 leg + 1] = Vxt(leg); diag[2 leg + 2] = Vtt(leg); super[2 leg + 2] = -I is
    // never stored: the elimination, the residual and the refinement step apply it as what it is
#pragma unroll 4
    for (int e = w.tid; e < bs; e += w.nthr) {
      d1[e] = tup[e];
      if (inner) {
        s1[e] = tup[bs + e];
        d1[bs + e] = tup[2 * bs + e];
      }
    }
    for (int e = w.tid; e < nxb; e += w.nthr) { // rhs[2 leg + 1] = -vx(leg), rhs[2 leg + 2] = -vt(leg)
      rhs[(2 * leg + 1) * nxb + e] = -tup[3 * bs + e];
      if (inner)
        rhs[(2 * leg + 2) * nxb + e] = -tup[3 * bs + nxb + e];
    }
  }
  __syncthreads();
  }

  CT(1)
  int failed = 0;
  // ---- symmetricBlockTridiagSolve, up-looking (block-tridiagonal.hpp:82-138).  The block being eliminated lives
  // in LDS together with its right-hand side: blk = [facD[ib] | rhs[ib]] (n x (n + 1)); eliminating it leaves
  // [facD[i] | rhs[i]] = [diag[i] | rhs[i]] - super[i] D^{-1} [super[i]^T | rhs[ib]] in the same place.
  {
    const int n = DIM(N);
    for (int e = w.tid; e < n * n; e += w.nthr)
      blk[e] = diag[(long long)N * bs + e];
    for (int e = w.tid; e < n; e += w.nthr)
      blk[n * n + e] = rhs[N * nxb + e];
    __syncthreads();
  }
  for (int i = N - 1; i >= -1; --i) {
    const int ib = i + 1, n = DIM(ib);
    // the unfactorised block, for the factorisation that has to start over
    for (int e = w.tid; e < n * n; e += w.nthr)
      facD[(long long)ib * bs + e] = blk[e];
    CT(2)
    // the Schur complements of this elimination are definite, of alternating sign (Vxx - Vxt S^-1 Vxt^T > 0,
    // Vtt - S^-1 < 0, ...): blocked elimination without pivoting (ublk is free here: its panel workspace); a
    // pivot of the wrong sign or a zero sends the block to the reference's Bunch-Kaufman
    int indefinite = 1;
    if (n >= 8 && n <= 64) {
      indefinite = wg_ldl_definite_factor(w, n, blk, n, lsub, lpiv, ublk, ctrl);
      if (indefinite) {
        for (int e = w.tid; e < n * n; e += w.nthr)
          blk[e] = facD[(long long)ib * bs + e];
        __syncthreads();
      }
    }
    if (indefinite)
      failed |= wg_bk_factor(w, n, blk, n, lsub, lpiv, ctrl);
    CT(3)
    for (int e = w.tid; e < n * n; e += w.nthr)
      facD[(long long)ib * bs + e] = blk[e];
    for (int e = w.tid; e < n; e += w.nthr) {
      fsub[ib * nxb + e] = lsub[e];
      fpiv[ib * nxb + e] = lpiv[e];
    }
    if (i < 0) {
      wg_bk_solve(w, n, blk, n, lsub, lpiv, blk + n * n, 1, 0, 1);
      for (int e = w.tid; e < n; e += w.nthr)
        sol[e] = blk[n * n + e];
      break;
    }
    const int r = DIM(i);
    const bool minus_identity = alt && i >= 2 && (i & 1) == 0; // super[2 leg + 2] = -I
    const double *Bg = super + (long long)i * bs;        // r x n
    // ublk = [super[i]^T | rhs[ib]]  (n x (r + 1)), then <- D^{-1} ublk: one blocked substitution for both
    if (minus_identity) {
      for (int e = w.tid; e < n * r; e += w.nthr)
        ublk[e] = (e / n == e % n) ? -1.0 : 0.0;
    } else {
      for (int e = w.tid; e < n * r; e += w.nthr) { // (read along the columns of super: coalesced)
        const int a = e % r, bb = e / r;            // super(a, bb)
        const double v = Bg[e];
        ublk[a * n + bb] = v;
        bsup[e] = v;
      }
    }
    for (int e = w.tid; e < n; e += w.nthr)
      ublk[n * r + e] = blk[n * n + e];
    __syncthreads();
    CT(4)
    wg_bk_solve(w, n, blk, n, lsub, lpiv, ublk, 1, n, r + 1);
    CT(5)
    for (int e = w.tid; e < n * r; e += w.nthr)
      U[(long long)i * bs + e] = ublk[e];
    for (int e = w.tid; e < n; e += w.nthr)
      sol[ib * nxb + e] = ublk[n * r + e];
    CT(6)
    // [facD[i] | rhs[i]] = [diag[i] | rhs[i]] - super[i] ublk, into blk (r x (r + 1))
    if (minus_identity) { // (r == n)
      for (int e = w.tid; e < r * r; e += w.nthr)
        blk[e] = diag[(long long)i * bs + e] + ublk[e];
      for (int e = w.tid; e < r; e += w.nthr)
        blk[r * r + e] = rhs[i * nxb + e] + ublk[n * r + e];
    } else {
      for (int e = w.tid; e < r * r; e += w.nthr)
        blk[e] = diag[(long long)i * bs + e];
      for (int e = w.tid; e < r; e += w.nthr)
        blk[r * r + e] = rhs[i * nxb + e];
      __syncthreads();
      CT(7)
      wg_gemm(w, r, r + 1, n, colmajor(bsup, r), colmajor(ublk, n), colmajor(blk, r), colmajor(blk, r), -1.0);
    }
    __syncthreads();
    CT(8)
  }
  __syncthreads();
  CT(9)
  // :131-134  sol[i + 1] -= U[i] sol[i]; the products are cut into K-slices over the whole workgroup
  for (int i = 0; i < N; ++i) {
    const int r = DIM(i), n = DIM(i + 1);
    double *xin = ublk, *part = ublk + nxb; // sol[i] in LDS, partial sums
    for (int e = w.tid; e < r; e += w.nthr)
      xin[e] = sol[i * nxb + e];
    __syncthreads();
    const int slices = n > 0 && w.nthr / n >= 1 ? (w.nthr / n > 8 ? 8 : w.nthr / n) : 1;
    if (n > 0 && w.tid < n * slices) {
      const int sl = w.tid / n, row = w.tid - sl * n;
      const double *Ug = U + (long long)i * bs + row;
      double acc = 0.0;
#pragma unroll 8
      for (int k = sl; k < r; k += slices)
        acc += Ug[(long long)k * n] * xin[k];
      part[sl * n + row] = acc;
    }
    __syncthreads();
    for (int row = w.tid; row < n; row += w.nthr) {
      double acc = 0.0;
      for (int sl = 0; sl < slices; ++sl)
        acc += part[sl * n + row];
      sol[(i + 1) * nxb + row] -= acc;
    }
    __syncthreads();
  }

  CT(10)
  // ---- iterative refinement (parallel-solver.hxx:184-202, with the residual
  // computed from the true right-hand side; see DESIGN.md "refinement")
  int steps = 0;
  double resdl = 0.0;
  for (int it = 0; it < P.max_refinement; ++it) {
    // err = rhs - A sol   (blockTridiagMatMul, :52-75); sol is read from LDS when it fits, the -I blocks of the
    // super-diagonal are applied as such, and the infinity norm is reduced as the rows are produced
    const bool xs_lds = nblk * nxb <= 3 * bs + 2 * nxb;
    const double *xs = sol;
    if (xs_lds) {
      for (int e = w.tid; e < nblk * nxb; e += w.nthr)
        sm[e] = sol[e];
      xs = sm;
      __syncthreads();
    }
    double mx = 0.0;
    for (int e = w.tid; e < nblk * nxb; e += w.nthr) {
      const int i = e / nxb, a = e - i * nxb;
      const int n = DIM(i);
      double s = 0.0;
      if (a < n) {
        s = rhs[e];
        const double *Dg = diag + (long long)i * bs + a, *xi = xs + i * nxb;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
        for (int k = 0; k < n; ++k)
          s0 += Dg[k * n] * xi[k];
        if (i > 0) { // sub[i-1] = super[i-1]^T
          const int r = DIM(i - 1);
          if (alt && i - 1 >= 2 && ((i - 1) & 1) == 0) { // super[i-1] = -I
            s1 -= xs[(i - 1) * nxb + a];
          } else {
            const double *Bp = super + (long long)(i - 1) * bs + a * r, *xp = xs + (i - 1) * nxb;
#pragma unroll 8
            for (int k = 0; k < r; ++k)
              s1 += Bp[k] * xp[k];
          }
        }
        if (i < N) {
          const int c = DIM(i + 1);
          if (alt && i >= 2 && (i & 1) == 0) { // super[i] = -I
            s1 -= xs[(i + 1) * nxb + a];
          } else {
            const double *Bn = super + (long long)i * bs + a, *xn = xs + (i + 1) * nxb;
#pragma unroll 8
            for (int k = 0; k < c; ++k)
              s1 += Bn[k * n] * xn[k];
          }
        }
        s -= s0 + s1;
      }
      err[e] = s;
      const double v = fabs(s);
      mx = fmax(mx, v == v ? v : gar_inf()); // (a NaN counts as +inf)
    }
    mx = wave_max_f64(mx);
    __syncthreads(); // (sol's copy in LDS has been read by everyone)
    if (w.lane == 0)
      sm[w.wave] = mx; // (the copy of sol in LDS is dead)
    __syncthreads();
    for (int q = 0; q < w.nwaves; ++q)
      mx = fmax(mx, sm[q]);
    __syncthreads();
    resdl = mx;
    if (resdl <= P.threshold)
      break;
    // blockTridiagRefinementStep (:147-182) on err
    for (int i = N - 1; i >= -1; --i) {
      const int ib = i + 1, n = DIM(ib);
      __syncthreads();
      wg_bk_solve(w, n, facD + (long long)ib * bs, n, fsub + ib * nxb, fpiv + ib * nxb,
                  err + ib * nxb, 1, 0, 1);
      if (i < 0)
        break;
      const int r = DIM(i);
      if (alt && i >= 2 && (i & 1) == 0) { // super[i] = -I
        __syncthreads();
        for (int e = w.tid; e < r; e += w.nthr)
          err[i * nxb + e] += err[ib * nxb + e];
      } else {
        wg_gemv(w, r, n, colmajor(super + (long long)i * bs, r), err + ib * nxb, 1, err + i * nxb, 1,
                err + i * nxb, 1, -1.0);
      }
    }
    for (int i = 0; i < N; ++i) {
      const int r = DIM(i), n = DIM(i + 1);
      __syncthreads();
      wg_gemv(w, n, r, colmajor(U + (long long)i * bs, n), err + i * nxb, 1, err + (i + 1) * nxb,
              1, err + (i + 1) * nxb, 1, -1.0);
    }
    __syncthreads();
    for (int e = w.tid; e < nblk * nxb; e += w.nthr)
      sol[e] += err[e];
    steps = it + 1;
    __syncthreads();
  }
  CT(11)
  if (w.tid == 0) {
    info[1] = (double)steps;
    if (P.reduced) { // gar_condensed_leg_states adds the rows of the states; a failure here goes to the full chain
      info[0] = failed ? gar_inf() : resdl;
      info[2] = 0.0;
    } else {
      info[0] = resdl;
      if (failed)
        atomicOr(&P.status[b], 4);
    }
  }
#undef DIM
}

// collapseFeedback (parallel-solver.hpp:41-51): K0 -= Kth0 * Vxt(b0)^T on the
// factor record of stage 0 (subdiagonal[1] holds the UNFACTORED Vxt(b0)^T after
// backward()).
__global__ void gar_collapse_feedback(const gar_stage_meta *meta, double *fac,
                                      long long fac_stride, int batch, const int *flags, int want) {
  const int b = (int)blockIdx.x;
  if (b >= batch || (flags != nullptr && (flags[b] != 0) != (want != 0)))
    return;
  const gar_stage_meta m = meta[0];
  const gar_factor_offsets fo = gar_factor_layout(m.nx, m.nu, m.nc, m.nx2, m.nth);
  double *rec = fac + (long long)b * fac_stride + m.fac_off;
  // (eight products per round trip, every load unconditional from a clamped address: one dependent load per product
  // made this 54 us on the (56, 24) shape -- a fifth of it is the launch)
  const double *fth = rec + fo.fth, *Vxt = rec + fo.Vxt;
  const int nth = m.nth;
  for (int e = (int)threadIdx.x; e < m.nu * m.nx; e += (int)blockDim.x) {
    const int i = e / m.nx, j = e - i * m.nx;
    double s = 0.0;
    for (int k0 = 0; k0 < nth; k0 += 8) {
      double a[8], v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = k0 + q, kc = k < nth ? k : nth - 1;
        a[q] = fth[i * nth + kc];
        v[q] = Vxt[kc * m.nx + j];
        a[q] = k < nth ? a[q] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
        s += a[q] * v[q];
    }
    rec[fo.fb + e] -= s;
  }
}

// StageFactor::kktMat (gar/riccati-kernel.hpp:30-102; read from Python as datas[t].kktMat,
// bindings/python/src/gar/expose-prox-riccati.cpp:30-31): the reduced KKT matrix [Rhat D^T; D -mu I] of one
// stage, Rhat = R + B^T Vxx' B (riccati-kernel.hxx:224-226, :232-247).  The sweeps never store it (Rhat lives
// in registers / LDS for the length of a stage); this kernel forms it on request from what IS resident -- the
// knot and stage t+1's Vxx (symmetrised from its lower triangle, as the consuming stage does, :216).
// One workgroup; Vn == nullptr: no value-function term (terminal knot, last knot of a leg).
__global__ void __launch_bounds__(256) gar_kkt_matrix(const double *knot, gar_knot_offsets ko, const double *Vn,
                                                      int nx2, int nu, int nc, double mueq, double *out,
                                                      int vn_packed, int r_packed) {
  const int nk = nu + nc;
  for (int e = (int)threadIdx.x; e < nk * nk; e += (int)blockDim.x) {
    const int j = e / nk, i = e - j * nk;
    double v;
    if (i < nu && j < nu) {
      // (r_packed: the knot keeps R as its packed lower triangle, gar_layout.h)
      v = r_packed ? knot[ko.R + gar_lower_index(nu, i >= j ? i : j, i >= j ? j : i)] : knot[ko.R + j * nu + i];
      if (Vn != nullptr) {
        double acc = 0.0;
        for (int l = 0; l < nx2; ++l) {
          double w = 0.0; // (Vxx' B)(l, j)
          for (int k = 0; k < nx2; ++k)
            w = __builtin_fma(Vn[gar_sym_index(vn_packed, nx2, k >= l ? k : l, k >= l ? l : k)], knot[ko.B + j * nx2 + k], w);
          acc = __builtin_fma(knot[ko.B + i * nx2 + l], w, acc);
        }
        v += acc;
      }
    } else if (i >= nu && j < nu) {
      v = knot[ko.D + j * nc + (i - nu)];
    } else if (i < nu) {
      v = knot[ko.D + i * nc + (j - nu)];
    } else {
      v = (i == j) ? -mueq : 0.0;
    }
    out[e] = v;
  }
}

// Measurement aid (bench.py's roofline.stream_ceiling): the bytes of a serial-in-time backward sweep WITHOUT
// its arithmetic.  One wave per problem walks the horizon backwards, reads `in_pieces` 16-byte pieces per
// stage (the knot; the next one is requested while the current one is consumed), carries a 72-FMA dependent
// chain per stage and writes `out_pieces` pieces (the factor record).  What this kernel reaches is what HBM
// sustains for the sweep's read/write mix and walk; no product path calls it.
typedef double gar_double2 __attribute__((ext_vector_type(2)));
// AHEAD = 2: the same walk with TWO records requested ahead of the one being consumed (what a deeper prefetch of
// the knots would buy the sweep: roofline.stream_ceiling.two_ahead in the bench line)
template <int IN_PL, int OUT_PL>
__global__ void __launch_bounds__(64) gar_stream_sweep2(const gar_double2 *in, gar_double2 *out, double *sink,
                                                        int nrec, int in_pieces, int out_pieces) {
  const int b = (int)blockIdx.x, lane = (int)threadIdx.x;
  gar_double2 r0[IN_PL], r1[IN_PL], r2[IN_PL];
  const gar_double2 *pin = in + (size_t)b * nrec * in_pieces;
  gar_double2 *pout = out + (size_t)b * nrec * out_pieces;
  auto load = [&](const gar_double2 *p, gar_double2 (&r)[IN_PL]) {
#pragma unroll
    for (int q = 0; q < IN_PL; ++q) {
      const int e = 64 * q + lane;
      r[q] = p[e < in_pieces ? e : in_pieces - 1];
    }
  };
  auto consume = [&](const gar_double2 (&c)[IN_PL], int t, double &acc) {
#pragma unroll
    for (int q = 0; q < IN_PL; ++q)
      acc += c[q].x * 1.0000001 + c[q].y;
#pragma unroll 8
    for (int i = 0; i < 72; ++i)
      acc = __builtin_fma(acc, 0.999999, 1e-9);
#pragma unroll
    for (int q = 0; q < OUT_PL; ++q) {
      const int e = 64 * q + lane;
      if (e < out_pieces)
        pout[(size_t)t * out_pieces + e] = gar_double2{acc, c[q < IN_PL ? q : 0].x};
    }
  };
  double acc = 0.0;
  int t = nrec - 1;
  load(pin + (size_t)t * in_pieces, r0);
  if (t >= 1)
    load(pin + (size_t)(t - 1) * in_pieces, r1);
  // three records rotate through r0 (consumed) <- r1 <- r2 (just requested); unrolled by three so that the
  // rotation is a renaming, not register moves
  while (t >= 0) {
    if (t >= 2) load(pin + (size_t)(t - 2) * in_pieces, r2);
    consume(r0, t, acc);
    if (--t < 0) break;
    if (t >= 2) load(pin + (size_t)(t - 2) * in_pieces, r0);
    consume(r1, t, acc);
    if (--t < 0) break;
    if (t >= 2) load(pin + (size_t)(t - 2) * in_pieces, r1);
    consume(r2, t, acc);
    --t;
  }
  sink[(size_t)b * 64 + lane] = acc;
}

template <int IN_PL, int OUT_PL>
__global__ void __launch_bounds__(64) gar_stream_sweep(const gar_double2 *in, gar_double2 *out, double *sink,
                                                       int nrec, int in_pieces, int out_pieces, int stage_major) {
  // stage_major: record t of problem b at (t * batch + b) * pieces instead of (b * nrec + t) * pieces -- the
  // waves of a launch then walk ONE window of memory together (layout probe)
  const int b = (int)blockIdx.x, lane = (int)threadIdx.x;
  gar_double2 cur[IN_PL], nxt[IN_PL];
  const size_t nb = gridDim.x;
  const size_t rin = stage_major ? nb * in_pieces : (size_t)in_pieces, rout = stage_major ? nb * out_pieces : (size_t)out_pieces;
  const gar_double2 *pin = in + (stage_major ? (size_t)b * in_pieces : (size_t)b * nrec * in_pieces);
  gar_double2 *pout = out + (stage_major ? (size_t)b * out_pieces : (size_t)b * nrec * out_pieces);
  auto load = [&](const gar_double2 *p, gar_double2 (&r)[IN_PL]) {
#pragma unroll
    for (int q = 0; q < IN_PL; ++q) {
      const int e = 64 * q + lane;
      r[q] = p[e < in_pieces ? e : in_pieces - 1];
    }
  };
  load(pin + (size_t)(nrec - 1) * rin, cur);
  double acc = 0.0;
  for (int t = nrec - 1; t >= 0; --t) {
    if (t > 0)
      load(pin + (size_t)(t - 1) * rin, nxt);
#pragma unroll
    for (int q = 0; q < IN_PL; ++q)
      acc += cur[q].x * 1.0000001 + cur[q].y;
#pragma unroll 8
    for (int i = 0; i < 72; ++i)
      acc = __builtin_fma(acc, 0.999999, 1e-9);
#pragma unroll
    for (int q = 0; q < OUT_PL; ++q) {
      const int e = 64 * q + lane;
      if (e < out_pieces)
        pout[(size_t)t * rout + e] = gar_double2{acc, cur[q < IN_PL ? q : 0].x};
    }
#pragma unroll
    for (int q = 0; q < IN_PL; ++q)
      cur[q] = nxt[q];
  }
  sink[(size_t)b * 64 + lane] = acc;
}

// Measurement aid (bench.py, roofline.stream_ceiling.plain_copy): the plain grid-stride copy the guide's
// "achievable" HBM figure refers to -- 16 B per lane, every byte read once and written once, all waves resident --
// to put beside gar_stream_sweep (the sweep's own walk: one record in flight per wave).  No product path calls it.
// (four independent nontemporal 16-byte loads in flight per lane, 64 workgroups per CU: the best of the variants
// of scripts/ubench/copy_variants.cpp on this pool -- 6.2 TB/s where one load in flight and 8 workgroups per CU
// give 5.1)
__global__ void __launch_bounds__(256) gar_plain_copy(const gar_double2 *__restrict__ src, gar_double2 *__restrict__ dst,
                                                      long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    gar_double2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      v[u] = __builtin_nontemporal_load(&src[i + u * stride]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      __builtin_nontemporal_store(v[u], &dst[i + u * stride]);
  }
  for (; i < n; i += stride)
    dst[i] = src[i];
}

} // namespace gar
