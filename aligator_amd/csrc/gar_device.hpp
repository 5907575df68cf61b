// gar_device.hpp -- workgroup-level dense building blocks for the gar kernels
// (gfx950 / CDNA4, wave64).  Everything here runs on operands resident in LDS
// (or, for the rare large case, in HBM through the same generic pointers).
//
//  * wg_gemm      : D = C0 + sgn * A * B on v_mfma_f64_16x16x4_f64 tiles, the
//                   output tiles round-robined over the workgroup's waves.
//                   Replaces every Eigen `noalias() +=` product of
//                   riccati-kernel.hxx:216-311.
//  * wg_bk_factor : Bunch-Kaufman LDL^T with 1x1/2x2 pivots (alpha =
//                   (1+sqrt 17)/8), same pivot rule and same stored
//                   representation (unit-lower L, INVERSE D blocks, signed
//                   pivots) as core/bunchkaufman.hpp:23-169,348-420.
//  * wg_bk_solve  : in-place solve, core/bunchkaufman.hpp:451-518.
#pragma once
#include <hip/hip_runtime.h>

namespace gar {

// the ONE dynamic-LDS region of every kernel (each translation unit declares it; its size is the launch's)
extern __shared__ double gar_smem[];

typedef double double4_t __attribute__((ext_vector_type(4)));

// strided view: element (i,j) at p[i*rs + j*cs]
struct MatV {
  double *p;
  int rs, cs;
  __device__ __forceinline__ double &operator()(int i, int j) const {
    return p[i * rs + j * cs];
  }
  __device__ __forceinline__ MatV T() const { return MatV{p, cs, rs}; }
  __device__ __forceinline__ MatV sub(int i, int j) const {
    return MatV{p + i * rs + j * cs, rs, cs};
  }
};
__device__ __forceinline__ MatV colmajor(double *p, int ld) { return MatV{p, 1, ld}; }
__device__ __forceinline__ MatV rowmajor(double *p, int ld) { return MatV{p, ld, 1}; }

struct WG { // who am I inside the cooperating group (a workgroup, or one wave)
  int tid, nthr, lane, wave, nwaves;
  int wave_scope; // 1: the group is a single wave (tid = lane, nthr = 64)
};
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// ---- LDS-DMA (gfx950 global_load_lds_dwordx4): HBM -> LDS without passing through registers ---------------------
// one DMA piece: every active lane copies 16 bytes from its own source address to (wave-uniform) dst + 16 lane
__device__ __forceinline__ void wave_dma16(const char *src_lane, char *dst_wave) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src_lane,
                                   (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
#else
  __builtin_amdgcn_global_load_lds(src_lane, dst_wave, 16, 0, 0);
#endif
}
// BYTES (a multiple of 16) from src to the LDS image dst, 1 KiB per instruction: exactly ceil(BYTES / 1024)
// vector-memory instructions (the s_waitcnt arithmetic of the kernel counts on it)
template <int BYTES>
__device__ __forceinline__ void wave_dma(const double *src, char *dst, int lane) {
  constexpr int PIECES = (BYTES + 1023) / 1024;
  const char *s = reinterpret_cast<const char *>(src) + 16 * lane;
#pragma unroll
  for (int p = 0; p < PIECES; ++p) {
    const int left = BYTES - 1024 * p; // bytes of this piece and the ones behind it
    if (left >= 1024 || 16 * lane < left)
      wave_dma16(s + 1024 * p, dst + 1024 * p);
  }
}
// s_waitcnt with one field set (gfx9 encoding: vmcnt [3:0] | [15:14], expcnt [6:4], lgkmcnt [11:8])
#define GAR_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | ((((n) >> 4) & 3) << 14))
#define GAR_WAIT_LGKMCNT0() __builtin_amdgcn_s_waitcnt(0xC07F)
// barrier of the cooperating group
__device__ __forceinline__ void wg_bar(const WG &w) {
  if (w.wave_scope)
    wave_sync();
  else
    __syncthreads();
}
// workgroup barrier that orders LDS traffic only: global loads and stores in flight STAY in flight across it (a
// __syncthreads waits for vmcnt(0) -- every outstanding store's acknowledgement -- on each side)
__device__ __forceinline__ void wg_lds_bar() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// put(e, get(e)) for e < n over the group, EIGHT loads in flight per thread before the first store (a plain
// `for (e...) dst[e] = src[e]` compiles to load -> wait -> store per iteration)
template <class Get, class Put>
__device__ __forceinline__ void wg_move8(const WG &w, int n, Get get, Put put) {
  for (int e0 = w.tid; e0 < n; e0 += 8 * w.nthr) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = e0 + q * w.nthr;
      v[q] = get(e < n ? e : n - 1);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = e0 + q * w.nthr;
      if (e < n)
        put(e, v[q]);
    }
  }
}
// Maximum of a double over the wave, in every lane.  Six DPP stages (row_shr 1, 2, 4, 8, then
// row_bcast 15 and 31) reduce into lane 63, v_readlane broadcasts it: ~150 cycles.  The same
// reduction through __shfl_xor costs 12 ds_bpermute round trips (the LDS crossbar, ~100 cycles
// each), which is what a Bunch-Kaufman pivot search was spending most of its time on.
__device__ __forceinline__ double wave_max_f64(double x) {
#define GAR_DPP_MAX(ctrl, rmask)                                                                  \
  {                                                                                               \
    const long long xb = __double_as_longlong(x);                                                 \
    const int lo = (int)xb, hi = (int)(xb >> 32);                                                 \
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, ctrl, rmask, 0xf, false);                 \
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, ctrl, rmask, 0xf, false);                 \
    x = fmax(x, __longlong_as_double(((long long)hi2 << 32) | (long long)(unsigned)lo2));         \
  }
  GAR_DPP_MAX(0x111, 0xf)
  GAR_DPP_MAX(0x112, 0xf)
  GAR_DPP_MAX(0x114, 0xf)
  GAR_DPP_MAX(0x118, 0xf)
  GAR_DPP_MAX(0x142, 0xa)
  GAR_DPP_MAX(0x143, 0xc)
#undef GAR_DPP_MAX
  const long long xb = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readlane((int)xb, 63), hi = __builtin_amdgcn_readlane((int)(xb >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}

__device__ __forceinline__ WG wg_self() {
  WG w;
  w.tid = (int)threadIdx.x;
  w.nthr = (int)blockDim.x;
  w.lane = w.tid & 63;
  w.wave = w.tid >> 6;
  w.nwaves = w.nthr >> 6;
  w.wave_scope = 0;
  return w;
}
__device__ __forceinline__ WG wave_self() { // this wave as its own group
  WG w;
  w.lane = (int)threadIdx.x & 63;
  w.tid = w.lane;
  w.nthr = 64;
  w.wave = 0;
  w.nwaves = 1;
  w.wave_scope = 1;
  return w;
}

// D(MxN) = C0(MxN) + sgn * A(MxK) * B(KxN).  C0.p may be null (zero).
// f64 MFMA 16x16x4 operand maps (cdna_hip_programming.md section 3):
//   A: lane l holds A[i = l&15][k = l>>4];  B: lane l holds B[k = l>>4][j = l&15]
//   C/D: lane l, reg r holds D[row = (l>>4) + 4r][col = l&15].
// Out-of-range rows/cols/k are fed as zeros, so any M,N,K works.
// D may alias C0; D must not alias A or B.  No barrier inside.
__device__ inline void wg_gemm(const WG &w, int M, int N, int K, MatV A, MatV B, MatV C0,
                               MatV D, double sgn) {
  if (M <= 0 || N <= 0)
    return;
  const int tN = (N + 15) >> 4;
  const int nt = ((M + 15) >> 4) * tN;
  const int li = w.lane & 15, lk = w.lane >> 4;
  int ti = 0, tj = w.wave; // tile (ti, tj) of this wave: t = ti tN + tj, advanced without divisions
  while (tj >= tN) {
    tj -= tN;
    ++ti;
  }
  for (int t = w.wave; t < nt; t += w.nwaves) {
    const int i0 = ti << 4, j0 = tj << 4;
    const int col = j0 + li;
    // Every load is unconditional, from a clamped address, and selected afterwards: a conditional load is a branch
    // (the compiler may not speculate it), and a branch per operand serialises the round trips.
    const int colc = col < N ? col : N - 1;
    double4_t acc;
    if (C0.p != nullptr) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i0 + lk + 4 * r;
        acc[r] = C0(row < M ? row : M - 1, colc);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[r] = 0.0;
    }
    const int ai = i0 + li, aic = ai < M ? ai : M - 1;
    const bool aok = ai < M, bok = col < N;
    const double *ap = &A(aic, 0), *bp = &B(0, colc);
    // (the operands of four k-steps are requested together: one LDS / L2 round trip per 16 columns of A, not four)
    for (int k0 = 0; k0 < K; k0 += 16) {
      double a[4], b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = k0 + 4 * q + lk, kc = k < K ? k : K - 1;
        a[q] = ap[kc * A.cs];
        b[q] = bp[kc * B.rs];
        a[q] = (aok && k < K) ? sgn * a[q] : 0.0;
        b[q] = (bok && k < K) ? b[q] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (k0 + 4 * q < K)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + lk + 4 * r;
      if (row < M && col < N)
        D(row, col) = acc[r];
    }
    tj += w.nwaves; // next tile of this wave
    while (tj >= tN) {
      tj -= tN;
      ++ti;
    }
  }
}

// y(M) = y0(M) + sgn * A(MxK) x(K); one thread per output row.
__device__ inline void wg_gemv(const WG &w, int M, int K, MatV A, const double *x, int xs,
                               const double *y0, int y0s, double *y, int ys, double sgn) {
  for (int i = w.tid; i < M; i += w.nthr) {
    double s = 0.0;
    for (int k = 0; k < K; ++k)
      s += A(i, k) * x[k * xs];
    y[i * ys] = (y0 ? y0[i * y0s] : 0.0) + sgn * s;
  }
}

// copy src (contiguous, n doubles) -> dst view, element e -> (e % rows, e / rows)
// i.e. src is a column-major rows x cols block
__device__ inline void wg_load_colmajor(const WG &w, const double *src, int rows, int cols,
                                        MatV dst) {
  const int n = rows * cols;
  for (int e = w.tid; e < n; e += w.nthr) {
    const int j = e / rows, i = e - j * rows;
    dst(i, j) = src[e];
  }
}
__device__ inline void wg_fill(const WG &w, double *dst, int n, double v) {
  for (int e = w.tid; e < n; e += w.nthr)
    dst[e] = v;
}

// Storage of the symmetric matrix handed to the Bunch-Kaufman routines: every access they make
// is to the lower triangle (i >= j), so besides plain column-major (leading dimension lda) the
// lower triangle may be PACKED by columns (lda = n): half the LDS.
enum { GAR_COLMAJOR = 0, GAR_PACKED_LOWER = 1 };
template <int MODE> __device__ __forceinline__ int bk_idx(int i, int j, int lda) {
  return MODE == GAR_PACKED_LOWER ? j * lda - ((j * (j - 1)) >> 1) + (i - j) : j * lda + i;
}

// Row i of the 1x1 elimination step k (bunchkaufman.hpp:104-121): a(i,j) -= (a(j,k) d11) a(i,k)
// for j = k+1 .. i.  Loads are issued eight at a time before the stores that follow them (the
// compiler cannot prove that a(i,j+1) does not alias the store to a(i,j), and would otherwise
// serialise one LDS round trip per element); indices advance incrementally.
template <int MODE>
__device__ __forceinline__ void bk_update_row_range(double *a, int lda, int i, int k, double aik,
                                                    double d11, int jbeg, int jend) {
  int pij = bk_idx<MODE>(i, jbeg, lda), pjk = bk_idx<MODE>(jbeg, k, lda);
  int j = jbeg;
  for (; j + 8 <= jend; j += 8) {
    double wv[8], v[8];
    int pp[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      wv[q] = a[pjk + q];
      pp[q] = pij;
      v[q] = a[pij];
      pij += (MODE == GAR_PACKED_LOWER) ? lda - (j + q) - 1 : lda;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
      v[q] -= (wv[q] * d11) * aik;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      a[pp[q]] = v[q];
    pjk += 8;
  }
  for (; j < jend; ++j) {
    const double d11xj = a[pjk] * d11;
    a[pij] -= d11xj * aik;
    pij += (MODE == GAR_PACKED_LOWER) ? lda - j - 1 : lda;
    pjk += 1;
  }
}
template <int MODE>
__device__ __forceinline__ void bk_update_row(double *a, int lda, int i, int k, double aik,
                                              double d11) {
  bk_update_row_range<MODE>(a, lda, i, k, aik, d11, k + 1, i + 1);
}

// ---------------------------------------------------------------------------
// Bunch-Kaufman, whole workgroup, matrix `a` (n x n, lower triangle used,
// column-major with leading dimension lda) factorised in place.
//   piv[k] >= 0 : 1x1 pivot, row interchanged with piv[k]
//   piv[k] = piv[k+1] = -1-p : 2x2 pivot, row k+1 interchanged with p
// ctrl: >= 4 ints of scratch.  Returns 0 on success, 1 on an exactly-zero
// pivot column (NumericalIssue, bunchkaufman.hpp:58-59).  Ends with a barrier.
template <int MODE = GAR_COLMAJOR>
__device__ inline int wg_bk_factor(const WG &w, int n, double *a, int lda, double *subdiag,
                                   int *piv, int *ctrl) {
#define GA(i, j) a[bk_idx<MODE>((i), (j), lda)]
  const double alpha = (1.0 + 4.123105625617661) / 8.0; // (1+sqrt(17))/8, :29
  if (n == 0)
    return 0;
  if (n == 1) { // :36-43
    wg_bar(w);
    int bad = (fabs(GA(0, 0)) == 0.0);
    wg_bar(w);
    if (w.tid == 0) {
      if (!bad)
        GA(0, 0) = 1.0 / GA(0, 0);
      piv[0] = 0;
      subdiag[0] = 0.0;
    }
    wg_bar(w);
    return bad;
  }
  int k = 0;
  int info = 0;
  int r_step = 0, r_kp = 0, r_fail = 0; // decision taken by every thread itself (workgroup search)
  while (k < n) {
    // (workgroup scope: no barrier between two pivot steps -- the scaling of column k that ends a
    // step touches nothing the next step's search, interchange or update reads)
    if (w.wave_scope || k == 0)
      wg_bar(w);
    bool in_regs = false;
    if (w.wave_scope && n <= 128) {
      // Common case inside one wave, decided without a reduction or a round trip through
      // thread 0: the first test of the rule, |a_kk| >= alpha * colmax (:61), holds iff
      // |a_kk| >= alpha |a(i,k)| for every row below -- one compare per lane and a ballot.
      // Then kp = k, a 1x1 pivot (:104-121), eliminated from registers.
      const double akk = GA(k, k);
      const int i0 = k + 1 + w.lane, i1 = i0 + 64;
      const double a0 = i0 < n ? GA(i0, k) : 0.0, a1 = i1 < n ? GA(i1, k) : 0.0;
      const double abs_akk = fabs(akk);
      const bool bad = !(abs_akk >= fabs(a0) * alpha) || !(abs_akk >= fabs(a1) * alpha);
      if (__ballot(bad) == 0ull && akk != 0.0) {
        const double d11 = 1.0 / akk;
        if (i0 < n)
          bk_update_row<MODE>(a, lda, i0, k, a0, d11);
        if (i1 < n)
          bk_update_row<MODE>(a, lda, i1, k, a1, d11);
        wg_bar(w); // every row has read column k: now scale it (:118-120)
        if (i0 < n)
          GA(i0, k) = a0 * d11;
        if (i1 < n)
          GA(i1, k) = a1 * d11;
        if (w.tid == 0) {
          GA(k, k) = d11;
          piv[k] = k;
        }
        k += 1;
        continue;
      }
    }
    // pivot search, :46-83.  The column maximum (first row attaining it, as the reference's
    // strict ">" scan) is found by the whole group when the column is long: every thread scans
    // its rows, waves reduce with xor-shuffles, thread 0 combines the per-wave results that
    // were parked in `subdiag` (free until the end of the factorisation).
    if (w.wave_scope && n <= 64) {
      // One wave, n <= 64: the whole decision lane-parallel, no LDS parking, no barrier -- lane i
      // looks at row i of column k, xor-shuffles reduce (value, first index); the row maximum of
      // the candidate (:62-70) the same way.  Every lane ends with the same (k_step, kp, fail).
      const int i = w.lane;
      double bv = (i > k && i < n) ? fabs(GA(i, k)) : -1.0;
      int bi = (i > k && i < n) ? i : 0x7fffffff;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        if (ov > bv || (ov == bv && oi < bi)) {
          bv = ov;
          bi = oi;
        }
      }
      const double abs_akk = fabs(GA(k, k));
      const double colmax = bv < 0.0 ? 0.0 : bv;
      const int imax = bv < 0.0 ? k + 1 : bi;
      int k_step = 1, kp = k, fail = 0;
      if (fmax(abs_akk, colmax) == 0.0) {
        fail = 1;
      } else if (!(abs_akk >= colmax * alpha)) {
        double rv = 0.0; // |a(imax, j)|, j in [k, imax) ; |a(i, imax)|, i in (imax, n)
        if (i >= k && i < imax)
          rv = fabs(GA(imax, i));
        else if (i > imax && i < n)
          rv = fabs(GA(i, imax));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
          rv = fmax(rv, __shfl_xor(rv, off));
        const double rowmax = rv;
        if (abs_akk >= (alpha * colmax) * (colmax / rowmax)) {
          kp = k;
        } else if (fabs(GA(imax, imax)) >= alpha * rowmax) {
          kp = imax;
        } else {
          kp = imax;
          k_step = 2;
        }
      }
      if (w.tid == 0) {
        ctrl[0] = k_step;
        ctrl[1] = kp;
        ctrl[2] = fail;
      }
      wg_bar(w);
    } else {
    // (rows are dealt one per thread: only the first pw = ceil(n / 64) waves can hold any -- they alone park results in
    // `subdiag`, 3 pw <= n entries whatever the size of the workgroup)
    const int pw = ((n + 63) >> 6) < w.nwaves ? ((n + 63) >> 6) : w.nwaves;
    const bool par_search = (n - k - 1) >= 16 && n >= 3 * pw;
    if (par_search) {
      double bv = -1.0;
      int bi = 0x7fffffff;
      for (int i = k + 1 + w.tid; i < n; i += w.nthr) {
        const double v = fabs(GA(i, k));
        if (v > bv) {
          bv = v;
          bi = i;
        }
      }
      if (n - k - 1 <= w.nthr) {
        // one row per thread, rows ascending with the lane: the wave maximum by DPP, then the
        // first row attaining it is the lowest lane that holds it
        const double mv = wave_max_f64(bv);
        const unsigned long long hit = __ballot(bv == mv);
        bi = __builtin_amdgcn_readlane(bi, (int)__builtin_ctzll(hit));
        bv = mv;
      } else {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          const double ov = __shfl_xor(bv, off);
          const int oi = __shfl_xor(bi, off);
          if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
          }
        }
      }
      if (w.lane == 0 && w.wave < pw) {
        subdiag[2 * w.wave] = bv;
        subdiag[2 * w.wave + 1] = (double)bi;
      }
      wg_bar(w);
      // Every thread combines the per-wave results (a handful of broadcast reads) and takes the
      // decision itself; when the first test of the rule fails, the row maximum of the candidate
      // (:62-70) is found by the whole group as well -- one more barrier instead of a serial scan
      // of up to n elements by thread 0.
      double colmax = subdiag[0];
      int imax = (int)subdiag[1];
      for (int q = 1; q < pw; ++q) {
        const double v = subdiag[2 * q];
        const int iq = (int)subdiag[2 * q + 1];
        if (v > colmax || (v == colmax && iq < imax)) {
          colmax = v;
          imax = iq;
        }
      }
      if (colmax < 0.0) {
        colmax = 0.0;
        imax = k + 1;
      }
      const double abs_akk = fabs(GA(k, k));
      int k_step = 1, kp = k, fail = 0;
      if (fmax(abs_akk, colmax) == 0.0) {
        fail = 1;
      } else if (!(abs_akk >= colmax * alpha)) {
        double rv = 0.0;
        // (read before the barrier below: the interchange of a faster thread may move it)
        const double abs_aii = fabs(GA(imax, imax));
        for (int i = k + w.tid; i < n; i += w.nthr) {
          if (i < imax)
            rv = fmax(rv, fabs(GA(imax, i)));
          else if (i > imax)
            rv = fmax(rv, fabs(GA(i, imax)));
        }
        rv = wave_max_f64(rv);
        if (w.lane == 0 && w.wave < pw)
          subdiag[2 * pw + w.wave] = rv; // (3 pw <= n)
        wg_bar(w);
        double rowmax = subdiag[2 * pw];
        for (int q = 1; q < pw; ++q)
          rowmax = fmax(rowmax, subdiag[2 * pw + q]);
        if (abs_akk >= (alpha * colmax) * (colmax / rowmax)) {
          kp = k;
        } else if (abs_aii >= alpha * rowmax) {
          kp = imax;
        } else {
          kp = imax;
          k_step = 2;
        }
      }
      // every thread holds the same decision: no round trip through `ctrl`, no barrier
      r_step = k_step;
      r_kp = kp;
      r_fail = fail;
      in_regs = true;
    } else if (w.tid == 0) {
      int k_step = 1, kp, fail = 0;
      const double abs_akk = fabs(GA(k, k));
      int imax = k + 1;
      double colmax = 0.0;
      if (k + 1 < n) {
        colmax = fabs(GA(k + 1, k));
        for (int i = k + 2; i < n; ++i) {
          const double v = fabs(GA(i, k));
          if (v > colmax) {
            colmax = v;
            imax = i;
          }
        }
      }
      if (fmax(abs_akk, colmax) == 0.0) {
        fail = 1;
        kp = k;
      } else if (abs_akk >= colmax * alpha) {
        kp = k;
      } else {
        double rowmax = 0.0;
        for (int j = k; j < imax; ++j)
          rowmax = fmax(rowmax, fabs(GA(imax, j)));
        for (int i = imax + 1; i < n; ++i)
          rowmax = fmax(rowmax, fabs(GA(i, imax)));
        if (abs_akk >= (alpha * colmax) * (colmax / rowmax)) {
          kp = k;
        } else if (fabs(GA(imax, imax)) >= alpha * rowmax) {
          kp = imax;
        } else {
          kp = imax;
          k_step = 2;
        }
      }
      ctrl[0] = k_step;
      ctrl[1] = kp;
      ctrl[2] = fail;
    }
    if (!par_search)
      wg_bar(w);
    }
    const int k_step = in_regs ? r_step : ctrl[0], kp = in_regs ? r_kp : ctrl[1];
    if (in_regs ? r_fail : ctrl[2]) { // NumericalIssue: keep the remaining pivots in range and stop
      for (int i = k + w.tid; i < n; i += w.nthr)
        piv[i] = i;
      info = 1;
      break;
    }
    const int kk = k + k_step - 1;
    if (kp != kk) { // symmetric interchange, :86-102 (disjoint element pairs)
      for (int i = kk + 1 + w.tid; i < n; i += w.nthr) {
        if (i < kp) {
          const double t = GA(i, kk);
          GA(i, kk) = GA(kp, i);
          GA(kp, i) = t;
        } else if (i > kp) {
          const double t = GA(i, kk);
          GA(i, kk) = GA(i, kp);
          GA(i, kp) = t;
        }
      }
      if (w.tid == 0) {
        const double t = GA(kk, kk);
        GA(kk, kk) = GA(kp, kp);
        GA(kp, kp) = t;
        if (k_step == 2) {
          const double t2 = GA(k + 1, k);
          GA(k + 1, k) = GA(kp, k);
          GA(kp, k) = t2;
        }
      }
      wg_bar(w);
    }
    if (k_step == 1) { // :104-121
      const int m = n - k - 1;
      const double d11 = 1.0 / GA(k, k);
      // one thread per trailing row i: a(i,j) -= (a(j,k) d11) a(i,k) for j <= i.  For a fixed j
      // the threads touch consecutive i (conflict-free), a(j,k) is a broadcast read, and there
      // is no integer division in the loop
      // (incremental indices: element (i, j+1) is lda [- j - 1 when packed] past (i, j))
      // (a short trailing block leaves most of a workgroup idle with one thread per row: each row's
      // j-range is then cut into g pieces -- every element still gets its one update of this step)
      int g = m > 0 ? w.nthr / m : 1;
      g = g < 1 ? 1 : (g > 8 ? 8 : g);
      for (int idx = w.tid; idx < m * g; idx += w.nthr) {
        const int part = idx / m, i = k + 1 + (idx - part * m), len = i - k;
        bk_update_row_range<MODE>(a, lda, i, k, GA(i, k), d11, k + 1 + (len * part) / g,
                                  k + 1 + (len * (part + 1)) / g);
      }
      wg_bar(w);
      for (int i = w.tid; i < m; i += w.nthr)
        GA(k + 1 + i, k) *= d11;
      if (w.tid == 0) {
        GA(k, k) = d11;
        piv[k] = kp;
      }
    } else { // 2x2 pivot, :122-149
      const double d21_abs = fabs(GA(k + 1, k));
      const double d21_inv = 1.0 / d21_abs;
      const double d11 = d21_inv * GA(k + 1, k + 1);
      const double d22 = d21_inv * GA(k, k);
      const double t = 1.0 / ((d11 * d22) - 1.0);
      const double d = t * d21_inv;
      const double d21 = GA(k + 1, k) * d21_inv;
      {
        const int m2 = n - k - 2;
        int g = m2 > 0 ? w.nthr / m2 : 1;
        g = g < 1 ? 1 : (g > 8 ? 8 : g);
        for (int idx = w.tid; idx < m2 * g; idx += w.nthr) {
          const int part = idx / m2, i = k + 2 + (idx - part * m2), len = i - k - 1;
          const double aik = GA(i, k), aik1 = GA(i, k + 1);
          const int j1 = k + 2 + (len * (part + 1)) / g;
          int j = k + 2 + (len * part) / g;
          // four elements per round trip (loads first, then the stores: see bk_update_row_range)
          for (; j + 4 <= j1; j += 4) {
            double ajk[4], ajk1[4], aij[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              ajk[q] = GA(j + q, k);
              ajk1[q] = GA(j + q, k + 1);
              aij[q] = GA(i, j + q);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const double wk = ((ajk[q] * d11) - (ajk1[q] * d21)) * d;
              const double wkp1 = ((ajk1[q] * d22) - (ajk[q] * d21)) * d;
              GA(i, j + q) = aij[q] - (aik * wk + aik1 * wkp1);
            }
          }
          for (; j < j1; ++j) {
            const double wk = ((GA(j, k) * d11) - (GA(j, k + 1) * d21)) * d;
            const double wkp1 = ((GA(j, k + 1) * d22) - (GA(j, k) * d21)) * d;
            GA(i, j) -= aik * wk + aik1 * wkp1;
          }
        }
      }
      double wk_r[2] = {0.0, 0.0}, wkp1_r[2] = {0.0, 0.0};
      // each thread owns rows j = k+2+tid (+nthr): n <= 2*nthr supported
      for (int q = 0; q < 2; ++q) {
        const int j = k + 2 + w.tid + q * w.nthr;
        if (j < n) {
          wk_r[q] = ((GA(j, k) * d11) - (GA(j, k + 1) * d21)) * d;
          wkp1_r[q] = ((GA(j, k + 1) * d22) - (GA(j, k) * d21)) * d;
        }
      }
      wg_bar(w);
      for (int q = 0; q < 2; ++q) {
        const int j = k + 2 + w.tid + q * w.nthr;
        if (j < n) {
          GA(j, k) = wk_r[q];
          GA(j, k + 1) = wkp1_r[q];
        }
      }
      if (w.tid == 0) {
        GA(k, k) = d11 * d;
        GA(k + 1, k) = -d21 * d;
        GA(k + 1, k + 1) = d22 * d;
        piv[k] = -1 - kp;
        piv[k + 1] = -1 - kp;
      }
    }
    k += k_step;
  }
  wg_bar(w);
  // subdiag extraction (:393-404) and row interchanges of the L part (:406-417).
  // Thread c applies the interchange sequence to column c of L.
  for (int c = w.tid; c < n; c += w.nthr) {
    int kq = 0;
    while (kq < n) {
      int p = piv[kq];
      int row, step;
      if (p < 0) {
        p = -1 - p;
        row = kq + 1;
        step = 2;
      } else {
        row = kq;
        step = 1;
      }
      if (c < kq && row != p) {
        const double t = GA(row, c);
        GA(row, c) = GA(p, c);
        GA(p, c) = t;
      }
      kq += step;
    }
  }
  wg_bar(w);
  for (int kq = w.tid; kq < n; kq += w.nthr)
    subdiag[kq] = 0.0;
  wg_bar(w);
  if (w.tid == 0) { // pairs are rare; serial pass keeps the pairing unambiguous
    int kq = 0;
    while (kq < n) {
      if (piv[kq] < 0) {
        subdiag[kq] = GA(kq + 1, kq);
        subdiag[kq + 1] = 0.0;
        GA(kq + 1, kq) = 0.0;
        kq += 2;
      } else {
        kq += 1;
      }
    }
  }
  wg_bar(w);
  return info;
#undef GA
}

// In-place solve of (L D L^T with interchanges) X = X for an n x ncols block X
// with strides (xrs, xcs); bunchkaufman.hpp:451-518.  One thread per column of
// X (columns are independent); ends with a barrier.
// Few right-hand sides (ncols < 16, e.g. the single column of the initial-stage KKT): one
// thread per column would leave the whole group idle behind a serial n^2 substitution, so the
// group works on ONE column at a time, column-oriented: x_j is final, every thread updates
// its rows i > j (i < j for the transposed solve).  Same operations as the per-column path,
// summed in the same order per row.
__device__ __forceinline__ double bk_bcast(double v, int src /*wave-uniform*/) {
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  return __hiloint2double(hi, lo);
}

// One wave, n <= 128, no 2x2 pivots: x lives in REGISTERS (lane l owns x_l and x_{l+64}); the
// pivot entry is broadcast with v_readlane (the column index is wave-uniform), so a substitution
// step is one conflict-free LDS read and one FMA per lane, with no barrier and no LDS round trip
// for x.  Same operations, same order per row, as wg_bk_solve_few.
template <int MODE>
__device__ inline void wave_bk_solve_regs(const WG &w, int n, const double *a, int lda,
                                          const int *piv, double *xc, int xrs) {
#define GA(i, j) a[bk_idx<MODE>((i), (j), lda)]
#define GX(i) xc[(i) * xrs]
  const int i0 = w.lane, i1 = w.lane + 64;
  wg_bar(w);
  if (w.tid == 0) { // forward interchanges (:458-468); 1x1 pivots only
    for (int k = 0; k < n; ++k) {
      const int p = piv[k];
      if (k != p) {
        const double t = GX(k);
        GX(k) = GX(p);
        GX(p) = t;
      }
    }
  }
  wg_bar(w);
  double x0 = i0 < n ? GX(i0) : 0.0, x1 = i1 < n ? GX(i1) : 0.0;
  for (int j = 0; j + 1 < n; ++j) { // unit-lower solve (:472), axpy form
    const double xj = j < 64 ? bk_bcast(x0, j) : bk_bcast(x1, j - 64);
    if (i0 > j && i0 < n)
      x0 -= GA(i0, j) * xj;
    if (i1 > j && i1 < n)
      x1 -= GA(i1, j) * xj;
  }
  if (i0 < n) // inverse-D multiply (:474-502), 1x1 blocks: D^{-1} is stored on the diagonal
    x0 *= GA(i0, i0);
  if (i1 < n)
    x1 *= GA(i1, i1);
  for (int i = n - 1; i >= 1; --i) { // unit-upper (L^T) solve (:504), axpy form
    const double xi = i < 64 ? bk_bcast(x0, i) : bk_bcast(x1, i - 64);
    if (i0 < i)
      x0 -= GA(i, i0) * xi;
    if (i1 < i)
      x1 -= GA(i, i1) * xi;
  }
  if (i0 < n)
    GX(i0) = x0;
  if (i1 < n)
    GX(i1) = x1;
  wg_bar(w);
  if (w.tid == 0) { // reverse interchanges (:506-517)
    for (int k = n - 1; k >= 0; --k) {
      const int p = piv[k];
      if (k != p) {
        const double t = GX(k);
        GX(k) = GX(p);
        GX(p) = t;
      }
    }
  }
  wg_bar(w);
#undef GA
#undef GX
}

template <int MODE = GAR_COLMAJOR>
__device__ inline void wg_bk_solve_few(const WG &w, int n, const double *a, int lda,
                                       const double *subdiag, const int *piv, double *x, int xrs,
                                       int xcs, int ncols) {
#define GA(i, j) a[bk_idx<MODE>((i), (j), lda)]
#define GX(i) xc[(i) * xrs]
  if (w.wave_scope && n <= 128) {
    // any 2x2 pivot (negative entry)?  lane-parallel scan + ballot
    const bool neg = (w.lane < n && piv[w.lane] < 0) || (w.lane + 64 < n && piv[w.lane + 64] < 0);
    if (__ballot(neg) == 0ull) {
      for (int c = 0; c < ncols; ++c)
        wave_bk_solve_regs<MODE>(w, n, a, lda, piv, x + c * xcs, xrs);
      return;
    }
  }
  for (int c = 0; c < ncols; ++c) {
    double *xc = x + c * xcs;
    wg_bar(w);
    if (w.tid == 0) { // forward interchanges (:458-468)
      int k = 0;
      while (k < n) {
        int p = piv[k];
        int row = k;
        if (p < 0) {
          p = -1 - p;
          row = k + 1;
          k += 2;
        } else {
          k += 1;
        }
        if (row != p) {
          const double t = GX(row);
          GX(row) = GX(p);
          GX(p) = t;
        }
      }
    }
    for (int j = 0; j + 1 < n; ++j) { // unit-lower solve (:472), axpy form
      wg_bar(w);
      const double xj = GX(j);
      for (int i = j + 1 + w.tid; i < n; i += w.nthr)
        GX(i) -= GA(i, j) * xj;
    }
    wg_bar(w);
    if (w.tid == 0) { // inverse-D multiply (:474-502)
      int k = 0;
      while (k < n) {
        if (piv[k] < 0) {
          const double akp1k = subdiag[k], ak = GA(k, k), akp1 = GA(k + 1, k + 1);
          const double xk = GX(k), xkp1 = GX(k + 1);
          GX(k) = xk * ak + xkp1 * akp1k;
          GX(k + 1) = xkp1 * akp1 + xk * akp1k;
          k += 2;
        } else {
          GX(k) *= GA(k, k);
          k += 1;
        }
      }
    }
    for (int i = n - 1; i >= 1; --i) { // unit-upper (L^T) solve (:504), axpy form
      wg_bar(w);
      const double xi = GX(i);
      for (int j = w.tid; j < i; j += w.nthr)
        GX(j) -= GA(i, j) * xi;
    }
    wg_bar(w);
    if (w.tid == 0) { // reverse interchanges (:506-517)
      int k = n;
      while (k > 0) {
        k -= 1;
        int p = piv[k];
        if (p < 0) {
          p = -1 - p;
          if (k != p) {
            const double t = GX(k);
            GX(k) = GX(p);
            GX(p) = t;
          }
          k -= 1;
        } else if (k != p) {
          const double t = GX(k);
          GX(k) = GX(p);
          GX(p) = t;
        }
      }
    }
  }
  wg_bar(w);
#undef GA
#undef GX
}

// Bunch-Kaufman solve of NCOLS (<= 48) right-hand sides by ONE wave with the triangular solves as
// BLOCKED MFMA updates: X (N x NCOLS, row i at X[i*xrs + c]) is held in accumulator layout (tile
// (ti, tj) register r = X(16ti + (lane>>4) + 4r, 16tj + (lane&15))); a k-step is a block of four
// rows, which are ONE register across the four lane groups: its 4x4 triangular part is solved
// with three shuffles + FMAs, and the rows beyond it are updated by v_mfma_f64_16x16x4 with the
// block's register as B operand (D -> B identity) and L read from LDS as A operand.
// The interchanges and the (1x1 / 2x2) D^{-1} step run in LDS, lane = column, exactly as
// wg_bk_solve (bunchkaufman.hpp:451-518).  N % 4 == 0.  Sums are accumulated four products at a
// time (MFMA order), not in the reference's dot-product order: equal to rounding.
template <int N, int NCOLS, int MODE>
__device__ inline void wave_bk_solve_mfma(const double *a, const double *subdiag, const int *piv,
                                          double *X, int xrs, int lane) {
#define GA(i, j) a[bk_idx<MODE>((i), (j), N)]
  static_assert(N % 4 == 0 && NCOLS <= 48, "unsupported block");
  constexpr int TR = (N + 15) / 16, TC = (NCOLS + 15) / 16, KSN = N / 4;
  const int li = lane & 15, lk = lane >> 4;
  double *xc = X + (lane < NCOLS ? lane : NCOLS - 1);
  if (lane < NCOLS) { // forward interchanges (:458-468)
    int k = 0;
    while (k < N) {
      int p = piv[k];
      int row = k;
      if (p < 0) {
        p = -1 - p;
        row = k + 1;
        k += 2;
      } else {
        k += 1;
      }
      if (row != p) {
        const double t = xc[row * xrs];
        xc[row * xrs] = xc[p * xrs];
        xc[p * xrs] = t;
      }
    }
  }
  wave_sync();
  double4_t Xt[TR][TC];
  auto load_tiles = [&]() {
#pragma unroll
    for (int ti = 0; ti < TR; ++ti)
#pragma unroll
      for (int tj = 0; tj < TC; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * ti + lk + 4 * r;
          Xt[ti][tj][r] = (16 * ti + 4 * r < N) ? X[row * xrs + 16 * tj + li] : 0.0;
        }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int ti = 0; ti < TR; ++ti)
#pragma unroll
      for (int tj = 0; tj < TC; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * ti + 4 * r < N)
            X[(16 * ti + lk + 4 * r) * xrs + 16 * tj + li] = Xt[ti][tj][r];
  };
  load_tiles();
  // ---- unit-lower solve (:472), k-step blocks top down -----------------------------------------
#pragma unroll
  for (int s = 0; s < KSN; ++s) {
    const int ti = s >> 2, r = s & 3, k0 = 4 * s;
    // the 4x4 triangular part: row k0 + lk lives in lane group lk
    const double c0 = lk >= 1 ? GA(k0 + lk, k0) : 0.0;
    const double c1 = lk >= 2 ? GA(k0 + lk, k0 + 1) : 0.0;
    const double c2 = lk == 3 ? GA(k0 + 3, k0 + 2) : 0.0;
#pragma unroll
    for (int tj = 0; tj < TC; ++tj) {
      double v = Xt[ti][tj][r];
      v = __builtin_fma(-c0, __shfl(v, li), v);
      v = __builtin_fma(-c1, __shfl(v, 16 + li), v);
      v = __builtin_fma(-c2, __shfl(v, 32 + li), v);
      Xt[ti][tj][r] = v;
    }
    // rows beyond the block: X(i, :) -= L(i, k0 .. k0+3) X(k0 .. k0+3, :)
#pragma unroll
    for (int t2 = ti; t2 < TR; ++t2) {
      const int row = 16 * t2 + li;
      const double aq = (row > k0 + 3 && row < N) ? -GA(row < N ? row : N - 1, k0 + lk) : 0.0;
#pragma unroll
      for (int tj = 0; tj < TC; ++tj)
        Xt[t2][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, Xt[ti][tj][r], Xt[t2][tj], 0, 0, 0);
    }
  }
  store_tiles();
  wave_sync();
  if (lane < NCOLS) { // inverse-D multiply (:474-502)
    int k = 0;
    while (k < N) {
      if (piv[k] < 0) {
        const double akp1k = subdiag[k], ak = GA(k, k), akp1 = GA(k + 1, k + 1);
        const double xk = xc[k * xrs], xkp1 = xc[(k + 1) * xrs];
        xc[k * xrs] = xk * ak + xkp1 * akp1k;
        xc[(k + 1) * xrs] = xkp1 * akp1 + xk * akp1k;
        k += 2;
      } else {
        xc[k * xrs] *= GA(k, k);
        k += 1;
      }
    }
  }
  wave_sync();
  load_tiles();
  // ---- unit-upper (L^T) solve (:504), k-step blocks bottom up ----------------------------------
#pragma unroll
  for (int s = KSN - 1; s >= 0; --s) {
    const int ti = s >> 2, r = s & 3, k0 = 4 * s;
    const double c3 = lk < 3 ? GA(k0 + 3, k0 + lk) : 0.0;
    const double c2 = lk < 2 ? GA(k0 + 2, k0 + lk) : 0.0;
    const double c1 = lk < 1 ? GA(k0 + 1, k0) : 0.0;
#pragma unroll
    for (int tj = 0; tj < TC; ++tj) {
      double v = Xt[ti][tj][r];
      v = __builtin_fma(-c3, __shfl(v, 48 + li), v);
      v = __builtin_fma(-c2, __shfl(v, 32 + li), v);
      v = __builtin_fma(-c1, __shfl(v, 16 + li), v);
      Xt[ti][tj][r] = v;
    }
    // rows above the block: X(i, :) -= L(k0 .. k0+3, i)^T X(k0 .. k0+3, :)
#pragma unroll
    for (int t2 = 0; t2 <= ti; ++t2) {
      const int row = 16 * t2 + li;
      const double aq = (row < k0) ? -GA(k0 + lk, row) : 0.0;
#pragma unroll
      for (int tj = 0; tj < TC; ++tj)
        Xt[t2][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq, Xt[ti][tj][r], Xt[t2][tj], 0, 0, 0);
    }
  }
  store_tiles();
  wave_sync();
  if (lane < NCOLS) { // reverse interchanges (:506-517)
    int k = N;
    while (k > 0) {
      k -= 1;
      int p = piv[k];
      if (p < 0) {
        p = -1 - p;
        if (k != p) {
          const double t = xc[k * xrs];
          xc[k * xrs] = xc[p * xrs];
          xc[p * xrs] = t;
        }
        k -= 1;
      } else if (k != p) {
        const double t = xc[k * xrs];
        xc[k * xrs] = xc[p * xrs];
        xc[p * xrs] = t;
      }
    }
  }
  wave_sync();
#undef GA
}

// Many right-hand sides, whole workgroup, the triangular solves BLOCKED: a step finishes NB rows of X (their
// NB x NB unit-triangular part by one thread per column), then the rows still open are updated with
// X(open, :) -= L(open, blk) X(blk, :) on f64 MFMA tiles shared out over the waves -- 2 n / NB pairs of barriers
// where the row-at-a-time substitution above needs 2 n, and a quarter of its LDS traffic.  Interchanges and the
// (1x1 / 2x2) D^{-1} step exactly as bunchkaufman.hpp:451-518; the stored form keeps L(k+1, k) = 0 inside a 2x2
// pivot, so the blocks need not respect pivot boundaries.  Sums are accumulated four products at a time (MFMA
// order), not in the reference's dot-product order: equal to rounding.
template <int MODE, int NB, bool COLX>
__device__ __forceinline__ void ldl_block_update(const WG &w, const double *a, int lda, double *x, int xld, int ncols,
                                                 int row0, int M, int p, int nb, bool transposed) {
  // X(row0 + i, :) -= sum_k Lop(i, k) X(p + k, :),  Lop(i, k) = L(row0 + i, p + k)  [ L(p + k, row0 + i) transposed ]
  // X(i, c) at x[i + c xld] (COLX) or x[i xld + c].  A wave owns a 16-column strip of X (its NB rows of the
  // finished block are fetched ONCE) and walks down the tile rows; with fewer strips than waves the tile rows are
  // dealt out among the waves of a strip.  Every load is unconditional, from a clamped address: a branch per load
  // would serialise the round trips.
  if (M <= 0)
    return;
  const int tN = (ncols + 15) >> 4, tM = (M + 15) >> 4;
  const int li = w.lane & 15, lk = w.lane >> 4;
  int share = 1, strip0 = w.wave, part = 0; // waves per strip, first strip, this wave's part of the tile rows
  if (tN < w.nwaves) {
    share = w.nwaves / tN;
    part = 0;
    while (strip0 >= tN) {
      strip0 -= tN;
      ++part;
    }
    if (part >= share)
      return;
  }
  const int sstep = tN < w.nwaves ? tN : w.nwaves;
  for (int tj = strip0; tj < tN; tj += sstep) {
    const int col = (tj << 4) + li;
    const bool cok = col < ncols;
    const int colc = cok ? col : ncols - 1;
    double *xc = COLX ? x + colc * xld : x + colc; // this lane's column
    double bv[NB / 4];
#pragma unroll
    for (int q = 0; q < NB / 4; ++q) {
      const int k = 4 * q + lk, kc = k < nb ? k : nb - 1;
      const double v = COLX ? xc[p + kc] : xc[(p + kc) * xld];
      bv[q] = (cok && k < nb) ? v : 0.0;
    }
    for (int ti = part; ti < tM; ti += share) {
      const int i0 = ti << 4;
      double4_t acc;
      int ra[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i0 + lk + 4 * r;
        ra[r] = COLX ? row0 + (row < M ? row : M - 1) : (row0 + (row < M ? row : M - 1)) * xld;
        acc[r] = xc[ra[r]];
      }
      const int ai = i0 + li, aic = ai < M ? ai : M - 1;
      double av[NB / 4];
#pragma unroll
      for (int q = 0; q < NB / 4; ++q) {
        const int k = 4 * q + lk, kc = k < nb ? k : nb - 1;
        const double l = transposed ? a[bk_idx<MODE>(p + kc, row0 + aic, lda)] : a[bk_idx<MODE>(row0 + aic, p + kc, lda)];
        av[q] = (ai < M && k < nb) ? -l : 0.0;
      }
#pragma unroll
      for (int q = 0; q < NB / 4; ++q)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (i0 + lk + 4 * r < M && cok)
          xc[ra[r]] = acc[r];
    }
  }
}

template <int MODE, int NB, bool COLX>
__device__ inline void wg_bk_solve_mfma_impl(const WG &w, int n, const double *a, int lda, const double *subdiag,
                                             const int *piv, double *x, int xld, int ncols) {
#define GA(i, j) a[bk_idx<MODE>((i), (j), lda)]
#define GXC(i, c) x[COLX ? (c) * xld + (i) : (c) + (i) * xld]
  wg_bar(w);
  bool moved = false; // any interchange or 2x2 pivot?  (every wave looks at every entry: uniform)
  for (int k = w.lane; k < n; k += 64)
    moved |= piv[k] != k;
  const bool pivoted = __ballot(moved) != 0ull;
  if (pivoted) {
    for (int c = w.tid; c < ncols; c += w.nthr) { // forward interchanges (:458-468)
      int k = 0;
      while (k < n) {
        int p = piv[k];
        int row = k;
        if (p < 0) {
          p = -1 - p;
          row = k + 1;
          k += 2;
        } else {
          k += 1;
        }
        if (row != p) {
          const double t = GXC(row, c);
          GXC(row, c) = GXC(p, c);
          GXC(p, c) = t;
        }
      }
    }
    wg_bar(w);
  }
  for (int p = 0; p < n; p += NB) { // unit-lower solve (:472)
    const int nb = n - p < NB ? n - p : NB;
    if (w.tid < ncols) { // (the block of L and the rows of X are fetched before the first store: one round trip;
      double lb[NB][NB]; //  unconditional loads from clamped addresses)
      if (nb == NB) {
#pragma unroll
        for (int r = 1; r < NB; ++r)
#pragma unroll
          for (int q = 0; q < r; ++q)
            lb[r][q] = GA(p + r, p + q);
      } else {
#pragma unroll
        for (int r = 1; r < NB; ++r)
#pragma unroll
          for (int q = 0; q < r; ++q)
            lb[r][q] = GA(p + (r < nb ? r : 0), p + (r < nb ? q : 0)) * (r < nb ? 1.0 : 0.0);
      }
      for (int c = w.tid; c < ncols; c += w.nthr) {
        double *xb = COLX ? x + c * xld + p : x + c + p * xld;
        double xv[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r)
          xv[r] = COLX ? xb[r < nb ? r : 0] : xb[(r < nb ? r : 0) * xld];
#pragma unroll
        for (int r = 1; r < NB; ++r) {
          double s = xv[r];
#pragma unroll
          for (int q = 0; q < r; ++q)
            s -= lb[r][q] * xv[q];
          xv[r] = s;
        }
#pragma unroll
        for (int r = 1; r < NB; ++r)
          if (r < nb) {
            if (COLX)
              xb[r] = xv[r];
            else
              xb[r * xld] = xv[r];
          }
      }
    }
    wg_bar(w);
    ldl_block_update<MODE, NB, COLX>(w, a, lda, x, xld, ncols, p + nb, n - p - nb, p, nb, false);
    wg_bar(w);
  }
  if (pivoted) {
    for (int c = w.tid; c < ncols; c += w.nthr) { // inverse-D multiply (:474-502)
      int k = 0;
      while (k < n) {
        if (piv[k] < 0) {
          const double akp1k = subdiag[k], ak = GA(k, k), akp1 = GA(k + 1, k + 1);
          const double xk = GXC(k, c), xkp1 = GXC(k + 1, c);
          GXC(k, c) = xk * ak + xkp1 * akp1k;
          GXC(k + 1, c) = xkp1 * akp1 + xk * akp1k;
          k += 2;
        } else {
          GXC(k, c) *= GA(k, k);
          k += 1;
        }
      }
    }
  } else if (COLX) { // thread = row, columns one after the other: the pivot is read once, no division
    for (int k = w.tid; k < n; k += w.nthr) {
      const double dk = GA(k, k);
      for (int c = 0; c < ncols; ++c)
        x[c * xld + k] *= dk;
    }
  } else { // thread = column
    for (int c = w.tid; c < ncols; c += w.nthr)
      for (int k = 0; k < n; ++k)
        x[c + k * xld] *= GA(k, k);
  }
  wg_bar(w);
  for (int p = ((n - 1) / NB) * NB; p >= 0; p -= NB) { // unit-upper (L^T) solve (:504)
    const int nb = n - p < NB ? n - p : NB;
    if (w.tid < ncols) {
      double lb[NB][NB];
      if (nb == NB) {
#pragma unroll
        for (int q = 1; q < NB; ++q)
#pragma unroll
          for (int r = 0; r < q; ++r)
            lb[q][r] = GA(p + q, p + r);
      } else {
#pragma unroll
        for (int q = 1; q < NB; ++q)
#pragma unroll
          for (int r = 0; r < q; ++r)
            lb[q][r] = GA(p + (q < nb ? q : 0), p + (q < nb ? r : 0)) * (q < nb ? 1.0 : 0.0);
      }
      for (int c = w.tid; c < ncols; c += w.nthr) {
        double *xb = COLX ? x + c * xld + p : x + c + p * xld;
        double xv[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r)
          xv[r] = COLX ? xb[r < nb ? r : 0] : xb[(r < nb ? r : 0) * xld];
#pragma unroll
        for (int r = NB - 2; r >= 0; --r) {
          double s = xv[r];
#pragma unroll
          for (int q = r + 1; q < NB; ++q)
            s -= lb[q][r] * xv[q];
          xv[r] = s;
        }
#pragma unroll
        for (int r = 0; r < NB - 1; ++r)
          if (r < nb - 1) {
            if (COLX)
              xb[r] = xv[r];
            else
              xb[r * xld] = xv[r];
          }
      }
    }
    wg_bar(w);
    ldl_block_update<MODE, NB, COLX>(w, a, lda, x, xld, ncols, 0, p, p, nb, true);
    wg_bar(w);
  }
  if (pivoted) {
    for (int c = w.tid; c < ncols; c += w.nthr) { // reverse interchanges (:506-517)
      int k = n;
      while (k > 0) {
        k -= 1;
        int p = piv[k];
        if (p < 0) {
          p = -1 - p;
          if (k != p) {
            const double t = GXC(k, c);
            GXC(k, c) = GXC(p, c);
            GXC(p, c) = t;
          }
          k -= 1;
        } else if (k != p) {
          const double t = GXC(k, c);
          GXC(k, c) = GXC(p, c);
          GXC(p, c) = t;
        }
      }
    }
    wg_bar(w);
  }
#undef GA
#undef GXC
}
// X(i, c) at x[i xrs + c xcs]: one of the strides is 1 at every call site (column-major or row-major X); the two
// layouts are compiled separately so that the unit stride folds into the instructions' immediate offsets
#ifndef GAR_SOLVE_NB
#define GAR_SOLVE_NB 8
#endif
template <int MODE, int NB = GAR_SOLVE_NB>
__device__ inline void wg_bk_solve_mfma(const WG &w, int n, const double *a, int lda, const double *subdiag,
                                        const int *piv, double *x, int xrs, int xcs, int ncols) {
  if (xrs == 1)
    wg_bk_solve_mfma_impl<MODE, NB, true>(w, n, a, lda, subdiag, piv, x, xcs, ncols);
  else
    wg_bk_solve_mfma_impl<MODE, NB, false>(w, n, a, lda, subdiag, piv, x, xrs, ncols);
}

template <int MODE = GAR_COLMAJOR>
__device__ inline void wg_bk_solve(const WG &w, int n, const double *a, int lda,
                                   const double *subdiag, const int *piv, double *x, int xrs,
                                   int xcs, int ncols) {
  if (ncols < 16 && n >= 16) {
    if (!w.wave_scope && n <= 128) {
      // a workgroup with few columns: if there are no 2x2 pivots, wave 0 alone runs the
      // register-resident substitution (no workgroup barrier per step); every wave
      // evaluates the same ballot, so the choice is uniform
      const bool neg = (w.lane < n && piv[w.lane] < 0) || (w.lane + 64 < n && piv[w.lane + 64] < 0);
      if (__ballot(neg) == 0ull) {
        wg_bar(w);
        if (w.wave == 0) {
          WG w0 = w;
          w0.tid = w.lane;
          w0.nthr = 64;
          w0.nwaves = 1;
          w0.wave_scope = 1;
          for (int c = 0; c < ncols; ++c)
            wave_bk_solve_regs<MODE>(w0, n, a, lda, piv, x + c * xcs, xrs);
        }
        wg_bar(w);
        return;
      }
    }
    wg_bk_solve_few<MODE>(w, n, a, lda, subdiag, piv, x, xrs, xcs, ncols);
    return;
  }
  if (!w.wave_scope && ncols >= 16 && (xrs == 1 || xcs == 1)) {
    wg_bk_solve_mfma<MODE>(w, n, a, lda, subdiag, piv, x, xrs, xcs, ncols);
    return;
  }
#define GA(i, j) a[bk_idx<MODE>((i), (j), lda)]
#define GX(i) xc[(i) * xrs]
  for (int c = w.tid; c < ncols; c += w.nthr) {
    double *xc = x + c * xcs;
    int k = 0;
    while (k < n) { // :458-468
      int p = piv[k];
      int row = k;
      if (p < 0) {
        p = -1 - p;
        row = k + 1;
        k += 2;
      } else {
        k += 1;
      }
      if (row != p) {
        const double t = GX(row);
        GX(row) = GX(p);
        GX(p) = t;
      }
    }
    for (int i = 1; i < n; ++i) { // unit-lower solve (:472)
      double s = GX(i);
      for (int j = 0; j < i; ++j)
        s -= GA(i, j) * GX(j);
      GX(i) = s;
    }
    k = 0;
    while (k < n) { // inverse-D multiply (:474-502)
      if (piv[k] < 0) {
        const double akp1k = subdiag[k], ak = GA(k, k), akp1 = GA(k + 1, k + 1);
        const double xk = GX(k), xkp1 = GX(k + 1);
        GX(k) = xk * ak + xkp1 * akp1k;
        GX(k + 1) = xkp1 * akp1 + xk * akp1k;
        k += 2;
      } else {
        GX(k) *= GA(k, k);
        k += 1;
      }
    }
    for (int j = n - 2; j >= 0; --j) { // unit-upper (L^T) solve (:504)
      double s = GX(j);
      for (int i = j + 1; i < n; ++i)
        s -= GA(i, j) * GX(i);
      GX(j) = s;
    }
    k = n;
    while (k > 0) { // reverse interchanges (:506-517)
      k -= 1;
      int p = piv[k];
      if (p < 0) {
        p = -1 - p;
        if (k != p) {
          const double t = GX(k);
          GX(k) = GX(p);
          GX(p) = t;
        }
        k -= 1;
      } else if (k != p) {
        const double t = GX(k);
        GX(k) = GX(p);
        GX(p) = t;
      }
    }
  }
  wg_bar(w);
#undef GA
#undef GX
}


// ---------------------------------------------------------------------------------------------------------------
// DEFINITE matrices (all pivots of one sign: Vxx > 0, Rhat > 0, the alternating-sign Schur complements of the
// condensed leg-boundary system): blocked L D L^T WITHOUT pivoting.  A panel of NBF columns is factorised by wave 0
// alone IN REGISTERS (lane = row; the pivot and the multipliers of a column reach the other lanes by v_readlane:
// no LDS round trip, no barrier inside a panel), the lower tiles of the trailing block take
// A22 -= (L21 D) L21^T on f64 MFMA from the whole workgroup.  Elimination of a definite matrix is backward stable
// without pivoting, so the result agrees with Bunch-Kaufman's (which may still interchange on such a matrix) to
// cond * eps; the stored form is Bunch-Kaufman's with piv[k] = k (unit-lower L below the diagonal, INVERSE pivots
// on it, subdiag = 0), so wg_bk_solve and every consumer of a factorised block work unchanged.  Only the lower
// triangle is read or written.  Returns 0 when every pivot is finite, nonzero and of the sign of the first one;
// otherwise 1 with `a` DESTROYED: the caller restores the block and runs wg_bk_factor (the reference's rule).
// n <= 64, column-major, wk: NBF n doubles of LDS (GAR_LDL_PANEL = NBF), ctrl: >= 1 int.
#define GAR_LDL_PANEL 8
__device__ inline int wg_ldl_definite_factor(const WG &w, int n, double *a, int lda, double *subdiag, int *piv,
                                             double *wk, int *ctrl) {
  constexpr int NBF = GAR_LDL_PANEL;
  if (w.tid == 0)
    ctrl[0] = 0;
  __syncthreads();
  if (n == 0)
    return 0;
  const bool positive = a[0] > 0.0;
  const int li = w.lane & 15, lk = w.lane >> 4;
  for (int p = 0; p < n; p += NBF) {
    const int nb = n - p < NBF ? n - p : NBF;
    if (w.wave == 0) {
      const int l = w.lane, i = p + l; // this lane's row
      double v[NBF], ld[NBF];
      const int ic = i < n ? i : n - 1; // (unconditional loads from clamped addresses; only the lower triangle counts)
#pragma unroll
      for (int c = 0; c < NBF; ++c) {
        const double t = a[ic + (p + (c < nb ? c : 0)) * lda];
        v[c] = (c < nb && l >= c && i < n) ? t : 0.0;
      }
      bool bad = false;
#pragma unroll
      for (int c = 0; c < NBF; ++c)
        if (c < nb) {
          const double d = bk_bcast(v[c], c);
          bad |= !(fabs(d) <= 1.7e308) || d == 0.0 || ((d > 0.0) != positive);
          const double rd = 1.0 / d;
          ld[c] = l > c ? v[c] : 0.0; // (L D)(i, k)
          const double lik = ld[c] * rd;
          v[c] = l == c ? rd : (l > c ? lik : v[c]);
#pragma unroll
          for (int j = c + 1; j < NBF; ++j)
            if (j < nb) { // rest of the panel: a(i, p+j) -= l(i, k) (L D)(p+j, k), rows i >= p+j
              const double ldj = bk_bcast(ld[c], j);
              if (l >= j)
                v[j] -= lik * ldj;
            }
        }
#pragma unroll
      for (int c = 0; c < NBF; ++c)
        if (c < nb && l >= c && i < n) {
          a[i + (p + c) * lda] = v[c];
          if (l >= nb)
            wk[i + c * n] = ld[c];
        }
      if (bad && l == 0)
        ctrl[0] = 1;
    }
    __syncthreads();
    if (ctrl[0])
      return 1;
    const int m = n - p - nb, r0 = p + nb; // trailing block: rows / columns r0 ..
    if (m > 0) {
      const int tM = (m + 15) >> 4, nt = tM * (tM + 1) / 2;
      for (int t = w.wave; t < nt; t += w.nwaves) {
        int ti = 0, rest = t; // lower tiles, row by row: (0,0) (1,0) (1,1) (2,0) ...
        while (rest > ti) {
          rest -= ti + 1;
          ++ti;
        }
        const int i0 = ti << 4, col = (rest << 4) + li, ai = i0 + li;
        double4_t acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = i0 + lk + 4 * r;
          acc[r] = (row < m && col < m) ? a[(r0 + row) + (r0 + col) * lda] : 0.0;
        }
        double av[NBF / 4], bv[NBF / 4];
#pragma unroll
        for (int q = 0; q < NBF / 4; ++q) {
          const int k = 4 * q + lk;
          av[q] = (ai < m && k < nb) ? -wk[(r0 + ai) + k * n] : 0.0;          // -(L D)(row, k)
          bv[q] = (col < m && k < nb) ? a[(r0 + col) + (p + k) * lda] : 0.0;  // L(col, k)
        }
#pragma unroll
        for (int q = 0; q < NBF / 4; ++q)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = i0 + lk + 4 * r;
          if (row < m && col < m && row >= col)
            a[(r0 + row) + (r0 + col) * lda] = acc[r];
        }
      }
      __syncthreads();
    }
  }
  for (int k = w.tid; k < n; k += w.nthr) {
    piv[k] = k;
    subdiag[k] = 0.0;
  }
  __syncthreads();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// BLOCKED Bunch-Kaufman, n <= 128 (core/bunchkaufman.hpp:172-344 is the reference's blocked path; LAPACK's dlasyf
// is the scheme): the same pivot rule and the same stored form as wg_bk_factor, with the trailing matrix updated
// once per PANEL instead of once per column.
//   panel   wave 0 alone, lane l owns rows l and l + 64, everything in LDS, wave barriers only.  Column k is brought
//           up to date on demand, w = A(k:, k) - L(k:, panel) W(k, panel)^T (W = L D: the columns as they were when
//           they were eliminated); the pivot search looks at w (and, when the diagonal fails the first test, at the
//           candidate column brought up to date the same way); interchanges swap rows of the stored A, of EVERY
//           column of L to the left (so that no pass over L is left for the end) and of W; the pivot column(s) are
//           scaled into L, kept unscaled in W.
//   update  all waves: A22 -= L21 W21^T on the lower tiles, f64 MFMA (K = panel width <= 8).
// Sums are dot products here and a sequence of rank-1 updates in the unblocked code: the factors agree to rounding
// -- and a pivot decision can differ where two candidates tie to rounding (either is a valid Bunch-Kaufman step).
// wk: n x GAR_BK_PANEL doubles of LDS.  Returns 0, or 1 on an exactly-zero pivot column.  Ends with a barrier.
#define GAR_BK_PANEL 8
template <int MODE = GAR_COLMAJOR>
__device__ inline int wg_bk_factor_blocked(const WG &w, int n, double *a, int lda, double *subdiag, int *piv, int *ctrl,
                                           double *wk) {
#define GA(i, j) a[bk_idx<MODE>((i), (j), lda)]
#define WK(i, c) wk[(i) + (c) * n]
  constexpr int NBP = GAR_BK_PANEL;
  const double alpha = (1.0 + 4.123105625617661) / 8.0; // (1+sqrt(17))/8
  if (w.tid == 0) {
    ctrl[0] = 0; // next column
    ctrl[1] = 0; // columns of W filled by the panel just finished
    ctrl[2] = 0; // info
  }
  __syncthreads();
  const int li = w.lane & 15, lk = w.lane >> 4;
  int k = 0;
  while (k < n) {
    const int p = k; // first column of this panel
    if (w.wave == 0) {
      const int l = w.lane;
      int kw = 0, info = 0;
      while (k < n && kw < NBP - 1) {
        // ---- column k brought up to date: wv[q] = row l + 64 q of w (rows >= k)
        // (every operand of the panel-wide dot products is requested before the first is used: unconditional loads
        // from clamped addresses, selected afterwards)
        double wv[2], w2[2], lrow[2][NBP - 1];
        {
          double wrow[NBP - 1], a0[2];
#pragma unroll
          for (int c = 0; c < NBP - 1; ++c)
            wrow[c] = WK(k, c < kw ? c : 0);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int i = l + 64 * q;
            const int ic = (i >= k && i < n) ? i : k;
            a0[q] = GA(ic, k);
#pragma unroll
            for (int c = 0; c < NBP - 1; ++c)
              lrow[q][c] = GA(ic, p + (c < kw ? c : 0)); // this lane's rows of the panel's L: used again below
          }
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int i = l + 64 * q;
            double s = a0[q];
#pragma unroll
            for (int c = 0; c < NBP - 1; ++c)
              s -= c < kw ? lrow[q][c] * wrow[c] : 0.0;
            wv[q] = (i >= k && i < n) ? s : 0.0;
          }
        }
        // ---- pivot search (bunchkaufman.hpp:46-83)
        const double wkk = bk_bcast(k < 64 ? wv[0] : wv[1], k & 63);
        const double abs_akk = fabs(wkk);
        double bv = -1.0;
        int bi = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int i = l + 64 * q;
          if (i > k && i < n && fabs(wv[q]) > bv) {
            bv = fabs(wv[q]);
            bi = i;
          }
        }
        double colmax = wave_max_f64(bv);
        int imax = k + 1;
        if (colmax < 0.0) {
          colmax = 0.0;
        } else { // first row attaining it (the reference's strict ">" scan)
          unsigned long long hit = __ballot(bv == colmax);
          imax = 0x7fffffff;
          while (hit) {
            const int src = (int)__builtin_ctzll(hit);
            const int cand = __builtin_amdgcn_readlane(bi, src);
            imax = cand < imax ? cand : imax;
            hit &= hit - 1;
          }
        }
        int k_step = 1, kp = k;
        bool use_w2 = false; // the pivot column (kp = imax, 1x1) / the second pivot column (2x2) is w2
        if (fmax(abs_akk, colmax) == 0.0) {
          info = 1;
        } else if (!(abs_akk >= colmax * alpha)) {
          // the candidate column imax of the symmetric trailing matrix, brought up to date
          double wrow[NBP - 1];
#pragma unroll
          for (int c = 0; c < NBP - 1; ++c)
            wrow[c] = WK(imax, c < kw ? c : 0);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int i = l + 64 * q;
            const bool in = i >= k && i < n;
            const int ic = in ? i : k;
            double s = GA(ic >= imax ? ic : imax, ic >= imax ? imax : ic);
#pragma unroll
            for (int c = 0; c < NBP - 1; ++c)
              s -= c < kw ? lrow[q][c] * wrow[c] : 0.0;
            w2[q] = in ? s : 0.0;
          }
          double rv = 0.0;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int i = l + 64 * q;
            if (i >= k && i < n && i != imax)
              rv = fmax(rv, fabs(w2[q]));
          }
          const double rowmax = wave_max_f64(rv);
          const double abs_aii = fabs(bk_bcast(imax < 64 ? w2[0] : w2[1], imax & 63));
          if (abs_akk >= (alpha * colmax) * (colmax / rowmax)) {
            kp = k;
          } else if (abs_aii >= alpha * rowmax) {
            kp = imax;
            use_w2 = true;
          } else {
            kp = imax;
            k_step = 2;
            use_w2 = true;
          }
        }
        if (info) { // NumericalIssue: keep the remaining pivots in range and stop
          for (int i = k + l; i < n; i += 64)
            piv[i] = i;
          break;
        }
        if (k_step == 1 && use_w2) { // the pivot column is column imax
          wv[0] = w2[0];
          wv[1] = w2[1];
        }
        const int kk = k + k_step - 1;
        if (kp != kk) { // ---- symmetric interchange of kk and kp (:86-102), lazily updated storage
          wave_sync();
          // the stored (not yet updated) column kk moves to where row / column kp lives
          for (int i = kk + 1 + l; i < n; i += 64) {
            if (i < kp)
              GA(kp, i) = GA(i, kk);
            else if (i > kp)
              GA(i, kp) = GA(i, kk);
          }
          if (l == 0)
            GA(kp, kp) = GA(kk, kk);
          // rows kk and kp of every column of L to the left, and of W
          for (int c = l; c < k; c += 64) {
            const double t = GA(kk, c);
            GA(kk, c) = GA(kp, c);
            GA(kp, c) = t;
          }
          if (l < kw) {
            const double t = WK(kk, l);
            WK(kk, l) = WK(kp, l);
            WK(kp, l) = t;
          }
          // ... and of the working columns
          {
            const double akk0 = bk_bcast(kk < 64 ? wv[0] : wv[1], kk & 63), akp0 = bk_bcast(kp < 64 ? wv[0] : wv[1], kp & 63);
            if (l == (kk & 63))
              wv[kk >> 6] = akp0;
            if (l == (kp & 63))
              wv[kp >> 6] = akk0;
            if (k_step == 2) {
              const double bkk0 = bk_bcast(kk < 64 ? w2[0] : w2[1], kk & 63), bkp0 = bk_bcast(kp < 64 ? w2[0] : w2[1], kp & 63);
              if (l == (kk & 63))
                w2[kk >> 6] = bkp0;
              if (l == (kp & 63))
                w2[kp >> 6] = bkk0;
            }
          }
          wave_sync();
        }
        // ---- the pivot column(s): L into A, the unscaled columns into W
        if (k_step == 1) { // :104-121
          const double d = bk_bcast(k < 64 ? wv[0] : wv[1], k & 63);
          const double d11 = 1.0 / d;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int i = l + 64 * q;
            if (i >= k && i < n) {
              WK(i, kw) = wv[q];
              GA(i, k) = i == k ? d11 : wv[q] * d11;
            }
          }
          if (l == 0)
            piv[k] = kp;
          kw += 1;
        } else { // 2x2 pivot, :122-149: columns k (wv) and k + 1 (w2)
          const double akk = bk_bcast(k < 64 ? wv[0] : wv[1], k & 63);
          const double ak1k = bk_bcast(k + 1 < 64 ? wv[0] : wv[1], (k + 1) & 63);
          const double ak1k1 = bk_bcast(k + 1 < 64 ? w2[0] : w2[1], (k + 1) & 63);
          const double d21_abs = fabs(ak1k), d21_inv = 1.0 / d21_abs;
          const double d11 = d21_inv * ak1k1, d22 = d21_inv * akk;
          const double t = 1.0 / ((d11 * d22) - 1.0), d = t * d21_inv, d21 = ak1k * d21_inv;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int i = l + 64 * q;
            if (i >= k && i < n) {
              WK(i, kw) = wv[q];
              WK(i, kw + 1) = i > k ? w2[q] : ak1k; // (row k of column k + 1: the symmetric element)
              if (i > k + 1) {
                GA(i, k) = ((wv[q] * d11) - (w2[q] * d21)) * d;
                GA(i, k + 1) = ((w2[q] * d22) - (wv[q] * d21)) * d;
              }
            }
          }
          if (l == 0) { // the inverse of the 2x2 block, as wg_bk_factor stores it
            GA(k, k) = d11 * d;
            GA(k + 1, k) = -d21 * d;
            GA(k + 1, k + 1) = d22 * d;
            piv[k] = -1 - kp;
            piv[k + 1] = -1 - kp;
          }
          kw += 2;
        }
        k += k_step;
        wave_sync();
      }
      if (l == 0) {
        ctrl[0] = info ? n : k;
        ctrl[1] = kw;
        ctrl[2] |= info;
      }
    }
    __syncthreads();
    k = ctrl[0];
    const int kw = ctrl[1];
    if (ctrl[2])
      break;
    // ---- trailing update: A(i, j) -= sum_c L(i, p + c) W(j, c), i >= j >= k, lower tiles
    const int m = n - k;
    if (m > 0 && kw > 0) {
      const int tM = (m + 15) >> 4, nt = tM * (tM + 1) / 2;
      for (int t = w.wave; t < nt; t += w.nwaves) {
        int ti = 0, rest = t;
        while (rest > ti) {
          rest -= ti + 1;
          ++ti;
        }
        const int i0 = ti << 4, col = (rest << 4) + li, ai = i0 + li;
        const int colc = col < m ? col : m - 1, aic = ai < m ? ai : m - 1;
        double4_t acc;
        int rows[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = i0 + lk + 4 * r;
          rows[r] = row < m ? row : m - 1;
          const int hi = rows[r] >= colc ? rows[r] : colc, lo = rows[r] >= colc ? colc : rows[r];
          acc[r] = GA(k + hi, k + lo);
        }
        double av[NBP / 4], bv[NBP / 4];
#pragma unroll
        for (int q = 0; q < NBP / 4; ++q) {
          const int c = 4 * q + lk, cc = c < kw ? c : kw - 1;
          const double lv = GA(k + aic, p + cc), wvv = WK(k + colc, cc);
          av[q] = (ai < m && c < kw) ? -lv : 0.0;
          bv[q] = (col < m && c < kw) ? wvv : 0.0;
        }
#pragma unroll
        for (int q = 0; q < NBP / 4; ++q)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = i0 + lk + 4 * r;
          if (row < m && col < m && row >= col)
            GA(k + row, k + col) = acc[r];
        }
      }
    }
    __syncthreads();
  }
  const int info = ctrl[2];
  __syncthreads();
  // subdiag extraction (:393-404); the rows of L were interchanged as the pivots were chosen
  for (int kq = w.tid; kq < n; kq += w.nthr)
    subdiag[kq] = 0.0;
  __syncthreads();
  if (w.tid == 0) { // pairs are rare; a serial pass keeps the pairing unambiguous
    int kq = 0;
    while (kq < n) {
      if (piv[kq] < 0) {
        subdiag[kq] = GA(kq + 1, kq);
        subdiag[kq + 1] = 0.0;
        GA(kq + 1, kq) = 0.0;
        kq += 2;
      } else {
        kq += 1;
      }
    }
  }
  __syncthreads();
  return info;
#undef GA
#undef WK
}

} // namespace gar
